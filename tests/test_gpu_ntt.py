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


# ---- the evaluation form (what batch calls run): d_i = A(zeta w^i) B(zeta w^i) on the coset, four transforms instead of six ----
ROOT_2_28 = 0x2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0      # gnark-crypto's 2^28-th root of unity of Fr


def _intt(vals, w_inv, n):
    """coefficients of the polynomial with the given values on 1, w, w^2, ... (plain radix-2, Python integers)"""
    lg = n.bit_length() - 1
    a = [vals[int(format(i, "0%db" % lg)[::-1], 2)] for i in range(n)]
    half = 1
    while half < n:
        step = pow(w_inv, n // (2 * half), R)
        for k in range(0, n, 2 * half):
            t = 1
            for j in range(k, k + half):
                u, v = a[j], a[j + half] * t % R
                a[j], a[j + half] = (u + v) % R, (u - v) % R
                t = t * step % R
        half *= 2
    n_inv = pow(n, R - 2, R)
    return [x * n_inv % R for x in a]


def _check_d(g, oracle, pk, algo, m, seed, check_cols):
    """The device's d against the oracle's computeH through H = (S - D) / 2: D = S - 2 H must be the polynomial whose values on the
    coset zeta * w^i are the device's d_i (after taking out the 2^261 of the kernels' Montgomery domain)."""
    x = _columns(m, seed)
    raw = g.debug_compute_d(algo, x[:2].tobytes(), m)
    n = pk.n
    lg = n.bit_length() - 1
    got = np.frombuffer(raw, dtype=np.uint8).reshape(n, 64, 32)
    zeta = pow(ROOT_2_28, 1 << (27 - lg), R); w = zeta * zeta % R
    assert pow(zeta, n, R) == R - 1
    w_inv, zeta_inv, r261_inv = pow(w, R - 2, R), pow(zeta, R - 2, R), pow(pow(2, 261, R), R - 2, R)
    for col in check_cols:
        a, b, c = (np.ascontiguousarray(x[k, :, col]).tobytes() for k in range(3))
        h = oracle.compute_h(pk, a, b, c)
        H = [int.from_bytes(h[32 * k:32 * k + 32], "big") for k in range(n)]
        cv = [int.from_bytes(c[32 * i:32 * i + 32], "big") for i in range(m)] + [0] * (n - m)
        S = _intt(cv, w_inv, n)
        d = [int.from_bytes(got[i, col].tobytes(), "little") for i in range(n)]
        assert max(d) < R
        E = _intt([v * r261_inv % R for v in d], w_inv, n)             # E_k = zeta^k D_k
        zk = 1
        for k in range(n):
            assert E[k] * zk % R == (S[k] - 2 * H[k]) % R, "column %d coefficient %d" % (col, k)
            zk = zk * zeta_inv % R


def test_compute_d_chacha_domain_extreme_and_random_columns(gsc_chacha, oracle, chacha_oracle):
    cs, pk, vk = chacha_oracle
    _check_d(gsc_chacha, oracle, pk, gsc_chacha.CHACHA20, cs.n_constraints, 1, list(range(10)) + [10, 33, 63])


def test_compute_d_short_vectors(gsc_chacha, oracle, chacha_oracle):
    cs, pk, vk = chacha_oracle
    _check_d(gsc_chacha, oracle, pk, gsc_chacha.CHACHA20, 777, 2, [1, 4, 7, 9, 17])
