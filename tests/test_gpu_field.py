"""GPU unit tests of the radix-2^29 lazy-limb field arithmetic (bn254_fp29.hpp) against Python big integers: random
values plus the boundary values where carry / range handling can go wrong, single operations and lazy chains."""
import random

import pytest

pytestmark = pytest.mark.gpu

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _values(mod, rnd, n):
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, 2 ** 29 - 1, 2 ** 29, 2 ** 232, 2 ** 232 - 1, 2 ** 253, mod - 2 ** 29, 0x1fffffff * sum(2 ** (29 * i) for i in range(8))]
    return [e % mod for e in edge] + [rnd.randrange(mod) for _ in range(n - len(edge))]


@pytest.mark.parametrize("field,mod", [(0, P), (1, R)])
def test_single_operations_match_big_integers(gsc, field, mod):
    rnd = random.Random(field)
    a = _values(mod, rnd, 512); b = list(reversed(_values(mod, rnd, 512)))
    rnd.shuffle(b)
    ops = {0: lambda x, y: x * y % mod, 1: lambda x, y: (x + y) % mod, 2: lambda x, y: (x - y) % mod, 3: lambda x, y: x * x % mod,
           4: lambda x, y: pow(x, mod - 2, mod), 5: lambda x, y: (x * y - y * x) % mod, 6: lambda x, y: (-x) % mod, 7: lambda x, y: (x - y) * (x + y) % mod}
    for op, f in ops.items():
        got = gsc.debug_field_ops(field, op, a, b)
        assert got == [f(x, y) for x, y in zip(a, b)], op


@pytest.mark.parametrize("field,mod", [(0, P), (1, R)])
def test_operation_chains_stay_exact(gsc, field, mod):
    rnd = random.Random(10 + field)
    a = _values(mod, rnd, 256); b = _values(mod, rnd, 256)[::-1]
    # 16 dependent products / squarings: outputs of one product (in (-p, 2p), possibly negative top limb) feed the next
    got = gsc.debug_field_ops(field, 0, a, b, chain=16)
    assert got == [x * pow(y, 16, mod) % mod for x, y in zip(a, b)]
    got = gsc.debug_field_ops(field, 3, a, b, chain=10)
    assert got == [pow(x, 2 ** 10, mod) for x in a]
    # lazy additions / subtractions (carries only, freeze every 4 steps)
    got = gsc.debug_field_ops(field, 1, a, b, chain=12)
    assert got == [(x + 12 * y) % mod for x, y in zip(a, b)]
    got = gsc.debug_field_ops(field, 2, a, b, chain=12)
    assert got == [(x - 12 * y) % mod for x, y in zip(a, b)]
    # r <- (r - b) * (a + b), 8 times: signed-tight times loose operands every step
    want = []
    for x, y in zip(a, b):
        r = x
        for _ in range(8):
            r = (r - y) * (x + y) % mod
        want.append(r)
    assert gsc.debug_field_ops(field, 7, a, b, chain=8) == want
    # fused r*b - b*a, chained
    want = []
    for x, y in zip(a, b):
        r = x
        for _ in range(6):
            r = (r * y - y * x) % mod
        want.append(r)
    assert gsc.debug_field_ops(field, 5, a, b, chain=6) == want
