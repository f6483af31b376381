"""bench.py's host-side helpers (no GPU): the per-index xoshiro256** statements of SURVEY.md §8(d), the verification sample, the
publicSignals layout of the two verifier families, the CPU-quota probe and the engine environment the timed run uses."""
import os
import sys

from conftest import KAT, ROOT

sys.path.insert(0, ROOT)
import bench


def _xoshiro_ref(index):
    """Scalar restatement: splitmix64 seeding of xoshiro256** (the generators' published reference code), 14 outputs little-endian."""
    M = (1 << 64) - 1
    x = (index + 0x9E3779B97F4A7C15) & M
    s = []
    for _ in range(4):
        x = (x + 0x9E3779B97F4A7C15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        s.append(z ^ (z >> 31))
    rotl = lambda v, k: ((v << k) | (v >> (64 - k))) & M
    out = b""
    for _ in range(14):
        out += ((rotl((s[1] * 5) & M, 7) * 9) & M).to_bytes(8, "little")
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45)
    return out


def test_statements_are_a_function_of_their_index_alone():
    recs = bench.xoshiro_records(300, 5 << 20)
    assert len(recs) == 300 * 112
    for i in (0, 1, 63, 64, 299):
        assert recs[112 * i:112 * (i + 1)] == _xoshiro_ref((5 << 20) + i)
    # a later batch that overlaps the index range reproduces the same statements
    assert bench.xoshiro_records(10, (5 << 20) + 290) == recs[112 * 290:]
    assert bench.synthetic_records(4, 3) == bench.xoshiro_records(4, 3 << 32)
    # AES statements are kept provable: counter + 4 must not wrap (circuits/aesV2/aes128.go:41-53)
    p = bench.provable(recs, "aes128")
    assert all(p[112 * i + 47] < 0x80 for i in range(300)) and bench.provable(recs, "chacha20") == recs


def test_verification_sample_covers_the_edges():
    for n, want in ((8192, 256), (8192, 1024), (64, 256), (1, 256), (3072, 256)):
        idx = bench.sample_indices(n, want, edges=[n // 2])
        assert idx == sorted(set(idx)) and all(0 <= i < n for i in idx)
        assert len(idx) == min(n, want) or (len(idx) >= want and n > want)
        for must in (0, 63, 64, n - 1, n // 2 - 1, n // 2):
            if 0 <= must < n:
                assert must in idx, (n, want, must)
        if n > want:      # spread over the whole batch, not a prefix
            assert max(b - a for a, b in zip(idx, idx[1:])) <= 2 * (n // want) + 64


def test_public_signals_layout():
    rec = KAT["key"] + KAT["nonce"] + KAT["counter"].to_bytes(4, "little") + KAT["input"]
    ct = KAT["ciphertext"]
    # ChaCha20: counter little-endian (libraries/verifier/impl/verifiers.go:68); AES: big-endian (:131)
    assert bench.signals_of("chacha20", rec, ct) == ct + KAT["nonce"] + (3).to_bytes(4, "little") + KAT["input"]
    assert bench.signals_of("aes128", rec, ct) == ct + KAT["nonce"] + (3).to_bytes(4, "big") + KAT["input"]
    assert len(bench.signals_of("aes256", rec, ct)) == 144


def test_usable_cores_is_bounded_by_the_affinity_mask():
    n, how = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and isinstance(how, str) and how


def test_engine_env_of_the_timed_run():
    e = bench.engine_env("chacha20", 8192)
    assert e == {"GSC_MAX_BATCH": "8192", "GSC_Z_TABLE_GB": "140", "GSC_W_TABLE_GB": "56"}
    assert bench.engine_env("mixed", 1024) == {"GSC_MAX_BATCH": "1024"}                    # library defaults: all three algorithms resident
    assert bench.engine_env("chacha20", 4096, share=2)["GSC_Z_TABLE_GB"] == "70"           # two replicas rehearsed on one device split its memory
    assert bench.engine_env("aes128", 1000)["GSC_MAX_BATCH"] == "1024"
