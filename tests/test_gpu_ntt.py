"""GPU test of the quotient-polynomial kernels alone (k_ntt.hip) on adversarial vectors.

The kernels keep field elements in lazily-reduced limbs and bound their ranges by analysis; witness-derived
vectors are mostly bits and never reach those bounds, so this test feeds computeH extreme and random vectors directly
(64 independent columns per call) and compares every output element with the oracle's computeH (oracle/groth16.c,
restating gnark backend/groth16/bn254 computeH — SURVEY.md App. D: coset evaluation, pointwise division, interpolation).
The device computes H as (iNTT(c) - negacyclic(a, b)) / 2, which is the same polynomial whenever c = a * b row by row — what
every released proof satisfies — so every column here is consistent: a and b are adversarial, c is their product."""
import random

import numpy as np
import pytest

from conftest import AES, golden_bytes

pytestmark = pytest.mark.gpu

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _be(v):
    return np.frombuffer(int(v).to_bytes(32, "big"), dtype=np.uint8)


def _columns(m, seed):
    """(3, m, 64, 32) uint8: a, b, c as canonical big-endian values; column meaning in the comments."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.integers(0, 256, size=(3, m, 64, 32), dtype=np.uint8)
    x[..., 0] &= 0x1F                                  # < 2^253 < r: uniform-ish canonical values
    top, one, zero = _be(R - 1), _be(1), _be(0)
    x[:2, :, 0] = zero                                 # 0: all zero
    x[:2, :, 1] = top                                  # 1: everything r-1 (largest sums on the DIF sum path)
    x[0, :, 2] = top; x[1, :, 2] = one                 # 2: (r-1) * 1
    x[0, :, 3] = one; x[1, :, 3] = top                 # 3: 1 * (r-1)
    x[:2, 0::2, 4] = top; x[:2, 1::2, 4] = zero        # 4: alternating r-1, 0
    x[:2, : m // 2, 5] = top; x[:2, m // 2:, 5] = zero # 5: a step
    x[:2, :, 6] = zero; x[:2, 0, 6] = top              # 6: a single r-1 at index 0 (constant spectrum)
    x[:2, :, 7] = zero; x[:2, m - 1, 7] = top          # 7: a single r-1 at the last row
    x[0, :, 8] = top                                   # 8: a = r-1 everywhere, b random
    x[1, 1::2, 9] = zero                               # 9: every other b zero
    # c = a*b mod r, row by row (what a satisfied constraint system gives)
    av = [int.from_bytes(v.tobytes(), "big") for v in x[0].reshape(-1, 32)]
    bv = [int.from_bytes(v.tobytes(), "big") for v in x[1].reshape(-1, 32)]
    x[2] = np.frombuffer(b"".join((a * b % R).to_bytes(32, "big") for a, b in zip(av, bv)), dtype=np.uint8).reshape(m, 64, 32)
    return x


def _check(g, oracle, pk, algo, m, seed, check_cols):
    x = _columns(m, seed)
    raw = g.debug_compute_h(algo, x.tobytes(), m)
    n = pk.n
    got = np.frombuffer(raw, dtype=np.uint8).reshape(n, 64, 32)
    lg = n.bit_length() - 1
    idx = np.arange(n)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(lg):
        rev |= ((idx >> b) & 1) << (lg - 1 - b)
    for col in check_cols:
        a, b, c = (np.ascontiguousarray(x[k, :, col]).tobytes() for k in range(3))
        want = np.frombuffer(oracle.compute_h(pk, a, b, c), dtype=np.uint8).reshape(n, 32)
        want_dev = want[rev][:, ::-1]                  # device row k holds coefficient bitrev(k), little-endian
        assert np.array_equal(got[:, col], want_dev), "column %d" % col


def test_compute_h_chacha_domain_extreme_and_random_columns(gsc_chacha, oracle, chacha_oracle):
    cs, pk, vk = chacha_oracle
    _check(gsc_chacha, oracle, pk, gsc_chacha.CHACHA20, cs.n_constraints, 1, range(64))


def test_compute_h_short_vectors(gsc_chacha, oracle, chacha_oracle):
    cs, pk, vk = chacha_oracle
    _check(gsc_chacha, oracle, pk, gsc_chacha.CHACHA20, 777, 2, list(range(12)) + [17, 63])


def test_compute_h_aes_domain(gsc, oracle, aes_keys):
    # 2^17 domain: the strided kernels run 9 stages and 72 KiB LDS tiles
    name = "aes128"; algo = AES[name][0]
    r1cs, pkb, vkb = aes_keys[name]
    assert gsc.init_algorithm(algo, pkb, r1cs)
    pk = oracle.ProvingKey(pkb)
    _check(gsc, oracle, pk, algo, 5000, 3, list(range(12)) + [40])
