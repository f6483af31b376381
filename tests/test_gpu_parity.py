"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the libprove C-ABI, against the
CPU oracle on the same inputs, against the committed golden vectors, and through size-independent properties
(every proof must verify under the reference's vk.chacha20) at the benchmark's batch size."""
import base64
import hashlib
import json
import os
import random

import pytest

from conftest import KAT, golden_bytes

pytestmark = pytest.mark.gpu


def _params(key, nonce, counter, pt, cipher="chacha20"):
    return {"cipher": cipher, "key": list(key), "nonce": list(nonce), "counter": counter, "input": list(pt)}


def _signals(ct, nonce, counter, pt):
    return ct + nonce + counter.to_bytes(4, "little") + pt          # core_test.go:158-163


def test_native_code_is_loaded(gsc_chacha):
    maps = open("/proc/self/maps").read()
    assert "libprove.so" in maps
    assert "tables=" in gsc_chacha.describe(gsc_chacha.CHACHA20)


def test_kat_pipeline_stages_match_golden_vectors(gsc_chacha):
    g = gsc_chacha
    g.set_deterministic_randomness(0, 0, 0)
    d = g.debug_prove(_params(KAT["key"], KAT["nonce"], KAT["counter"], KAT["input"]))
    g.set_deterministic_randomness(None)
    be = lambda vals: b"".join(v.to_bytes(32, "big") for v in vals)
    assert hashlib.sha256(be(d["W"])).hexdigest() == KAT["sha256_W"]
    assert hashlib.sha256(be(d["A"]) + be(d["B"]) + be(d["C"])).hexdigest() == KAT["sha256_abc"]
    n = len(d["h"]); L = n.bit_length() - 1
    nat = [d["h"][int(format(j, "0%db" % L)[::-1], 2)] for j in range(n)]     # device order is bit-reversed (pk.G1.Z order)
    assert nat[n - 1] == 0
    assert hashlib.sha256(be(nat[: n - 1])).hexdigest() == KAT["sha256_h"]


@pytest.mark.parametrize("rs", list(KAT["proofs"].keys()))
def test_kat_proof_bytes(gsc_chacha, rs):
    g = gsc_chacha
    g.set_deterministic_randomness(rs[0], rs[1], 0)
    out = json.loads(g.prove(_params(KAT["key"], KAT["nonce"], KAT["counter"], KAT["input"])))
    g.set_deterministic_randomness(None)
    assert base64.b64decode(out["publicSignals"]) == KAT["ciphertext"]
    assert base64.b64decode(out["proof"]["proofJson"]).hex() == KAT["proofs"][rs]
    assert list(out.keys()) == ["proof", "publicSignals"]             # field order of OutputParams (prove_impl.go:49-52)


def test_random_inputs_bit_exact_vs_oracle_and_verify(gsc_chacha, oracle, chacha_oracle):
    g = gsc_chacha; cs, pk, vk = chacha_oracle
    rnd = random.Random(2024)
    r, s = rnd.getrandbits(253), rnd.getrandbits(253)
    cases = [(rnd.randbytes(32), rnd.randbytes(12), rnd.getrandbits(32), rnd.randbytes(64)) for _ in range(5)]
    cases.append((bytes(32), bytes(12), 0, bytes(64)))                  # all-zero edge
    cases.append((b"\xff" * 32, b"\xff" * 12, 0xFFFFFFFF, b"\xff" * 64))  # all-ones / max counter edge
    g.set_deterministic_randomness(r, s, 0)
    outs = g.prove_batch([_params(*c) for c in cases])
    g.set_deterministic_randomness(None)
    for c, out in zip(cases, outs):
        proof = base64.b64decode(out["proof"]["proofJson"]); ct = base64.b64decode(out["publicSignals"])
        want, want_ct = oracle.prove(cs, pk, "chacha20", c[0], c[1], c[2], c[3], r, s)
        assert ct == want_ct
        assert proof == want
        assert oracle.verify(vk, "chacha20", proof, _signals(ct, c[1], c[2], c[3]))


def test_csprng_proofs_differ_and_verify_like_TestFullChaCha20(gsc_chacha, oracle, chacha_oracle):
    # libraries/core_test.go:130-172: random inputs, counter = 1, base64 byte fields, verifier must accept
    g = gsc_chacha; _, _, vk = chacha_oracle
    rnd = random.Random(5)
    key, nonce, pt = rnd.randbytes(32), rnd.randbytes(12), rnd.randbytes(64)
    p = {"cipher": "chacha20", "key": base64.b64encode(key).decode(), "nonce": base64.b64encode(nonce).decode(), "counter": 1,
         "input": base64.b64encode(pt).decode()}
    a = json.loads(g.prove(p)); b = json.loads(g.prove(p))
    pa, pb = base64.b64decode(a["proof"]["proofJson"]), base64.b64decode(b["proof"]["proofJson"])
    assert pa != pb and len(pa) == len(pb) == 164                       # fresh (r, s) per proof
    for out, proof in ((a, pa), (b, pb)):
        assert oracle.verify(vk, "chacha20", proof, _signals(base64.b64decode(out["publicSignals"]), nonce, 1, pt))


def test_full_batch_ragged_size_all_verify(gsc_chacha, oracle, chacha_oracle):
    # BASELINE config size per GPU step (1024) plus a ragged tail (not a multiple of the 64-lane solver width):
    # every proof must verify under the reference's vk, and the binary path must agree with the JSON path.
    g = gsc_chacha; cs, pk, vk = chacha_oracle
    n = 1024 + 37
    rnd = random.Random(99)
    recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    r, s = rnd.getrandbits(250), rnd.getrandbits(250)
    g.set_deterministic_randomness(r, s, 0)
    ok, proofs, lens, cts = g.prove_raw(g.CHACHA20, recs, n)
    i = n - 1
    rec = recs[112 * i:112 * (i + 1)]
    single = json.loads(g.prove(_params(rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:])))
    g.set_deterministic_randomness(None)
    assert ok == n and set(lens) == {164}
    assert base64.b64decode(single["proof"]["proofJson"]) == proofs[196 * i:196 * i + 164]
    assert len({proofs[196 * k:196 * k + 164] for k in range(n)}) == n
    check = list(range(0, n, 97)) + [63, 64, 1023, 1024, n - 1]
    for k in check:
        rec = recs[112 * k:112 * (k + 1)]
        key, nonce, ctr, pt = rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:]
        proof, ct = proofs[196 * k:196 * k + 164], cts[64 * k:64 * k + 64]
        assert ct == oracle.chacha20_xor(key, nonce, ctr, pt)
        assert oracle.verify(vk, "chacha20", proof, _signals(ct, nonce, ctr, pt)), k
    # bit-exact against the oracle on a few of them
    for k in (0, 64, n - 1):
        rec = recs[112 * k:112 * (k + 1)]
        want, _ = oracle.prove(cs, pk, "chacha20", rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s)
        assert proofs[196 * k:196 * k + 164] == want


def test_error_reporting_matches_reference_conventions(gsc_chacha):
    g = gsc_chacha
    # TestPanic (core_test.go:120-128): bad cipher name and an array where uint32 is expected -> the decode error object
    bad = json.loads(g.prove({"cipher": "aes-256-ctr1", "key": [0] * 32, "nonce": [0] * 12, "counter": [0, 1], "input": [0] * 64}))
    assert bad["Field"] == "counter" and bad["Value"] == "array" and bad["Struct"] == "InputParams" and "proof" not in bad
    assert json.loads(g.prove({"cipher": "aes-256-ctr1", "key": [], "nonce": [], "counter": 0, "input": []})) == "could not find prover foraes-256-ctr1"
    aes_untouched = g.describe(g.AES_128) == "not initialised"        # other test modules may have initialised it in this process
    if aes_untouched:
        assert json.loads(g.prove({"cipher": "aes-128-ctr", "key": [0] * 16, "nonce": [0] * 12, "counter": 0, "input": [0] * 64})) == \
            "proving params are not initialized for cipher: aes-128-ctr"
    assert json.loads(g.prove({"cipher": "chacha20", "key": [0] * 31, "nonce": [0] * 12, "counter": 0, "input": [0] * 64})) == "key length must be 32: 31"
    assert json.loads(g.prove({"cipher": "chacha20", "key": [0] * 32, "nonce": [0] * 11, "counter": 0, "input": [0] * 64})) == "nonce length must be 12: 11"
    assert json.loads(g.prove({"cipher": "chacha20", "key": [0] * 32, "nonce": [0] * 12, "counter": 0, "input": [0] * 65})) == "plaintext length must be 64: 65"
    assert "Offset" in json.loads(g.prove(b'{"cipher": "chacha20", '))
    assert json.loads(g.prove({"cipher": "chacha20", "key": [0] * 32, "nonce": [0] * 12, "counter": 2 ** 32, "input": [0] * 64}))["Value"] == "number 4294967296"
    # unknown keys are ignored, key matching is case-insensitive (encoding/json)
    ok = json.loads(g.prove({"Cipher": "chacha20", "KEY": [1] * 32, "nonce": [2] * 12, "counter": 5, "input": [3] * 64, "extra": {"x": 1}}))
    assert "proof" in ok
    # InitAlgorithm is idempotent and rejects unknown ids (prove_impl.go:74-76, :113)
    assert g.init_algorithm(g.CHACHA20, b"", b"") is True
    assert g.init_algorithm(7, b"x", b"y") is False
    # garbage key material for a not-yet-initialised algorithm is refused, never crashes
    if aes_untouched:
        assert g.init_algorithm(g.AES_128, b"\x00" * 100, b"\x01" * 100) is False
        assert g.init_algorithm(g.AES_128, golden_bytes("pk.chacha20"), golden_bytes("r1cs.chacha20")) is False   # wrong circuit for the id


def test_concurrent_prove_callers_share_device_batches(gsc_chacha, oracle, chacha_oracle):
    # libraries/core_test.go:38-118 (TestProveVerify) calls Prove from several goroutines at once.  Here 96 threads call the
    # C-ABI concurrently; the library gathers them into a few device batches, so the whole thing takes a fraction of 96
    # sequential single proofs, and every answer still belongs to its own caller.
    import threading
    import time
    g = gsc_chacha; _, _, vk = chacha_oracle
    rnd = random.Random(4242)
    n = 96
    reqs = [_params(rnd.randbytes(32), rnd.randbytes(12), rnd.getrandbits(32), rnd.randbytes(64)) for _ in range(n)]
    outs = [None] * n
    t0 = time.time(); json.loads(g.prove(reqs[0])); single = time.time() - t0

    def work(i):
        outs[i] = json.loads(g.prove(reqs[i]))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.time()
    for t in threads: t.start()
    for t in threads: t.join()
    elapsed = time.time() - t0
    for q, out in zip(reqs, outs):
        proof = base64.b64decode(out["proof"]["proofJson"]); ct = base64.b64decode(out["publicSignals"])
        assert ct == oracle.chacha20_xor(bytes(q["key"]), bytes(q["nonce"]), q["counter"], bytes(q["input"]))
        assert oracle.verify(vk, "chacha20", proof, _signals(ct, bytes(q["nonce"]), q["counter"], bytes(q["input"])))
    # gathered calls must beat 96 sequential ones (a single Prove takes ~2.7 ms since the latency kernels; starting and joining 96 Python threads is a
    # good part of what is left, so the bound is loose)
    assert elapsed < 0.9 * n * single, (elapsed, single)


def test_full_loop_through_both_drop_in_libraries_like_TestFullChaCha20(gsc_chacha):
    # libraries/core_test.go:130-172 end to end: libprove.Prove -> libverify.Verify, nothing but the two C-ABIs
    g = gsc_chacha
    assert g.init_verifier(0, golden_bytes("vk.chacha20"))
    rnd = random.Random(8)
    key, nonce, pt, counter = rnd.randbytes(32), rnd.randbytes(12), rnd.randbytes(64), 1
    out = json.loads(g.prove(_params(key, nonce, counter, pt)))
    signals = base64.b64decode(out["publicSignals"]) + nonce + counter.to_bytes(4, "little") + pt
    assert g.verify({"cipher": "chacha20", "proof": out["proof"]["proofJson"], "publicSignals": base64.b64encode(signals).decode()})
    assert not g.verify({"cipher": "chacha20", "proof": out["proof"]["proofJson"], "publicSignals": base64.b64encode(signals[:-1] + b"\x00").decode()})


_OPTIONS_SCRIPT = r"""
import hashlib, os, random, sys
sys.path.insert(0, sys.argv[1])
import gsc_loader
from bench import golden
g = gsc_loader.load()
algo = int(sys.argv[2])
pk = open(sys.argv[3], "rb").read() if len(sys.argv) > 3 else golden("pk.chacha20")
r1cs = golden(["r1cs.chacha20", "r1cs.aes128", "r1cs.aes256"][algo])
assert g.init_algorithm(algo, pk, r1cs)
assert "lanes=%s " % os.environ.get("GSC_LANES", "2" if algo else "1") in g.describe(algo), g.describe(algo)
assert "devices=%d " % len(os.environ.get("GSC_DEVICES", "0").split(",")) in g.describe(algo), g.describe(algo)
rnd = random.Random(4242)
n = int(os.environ.get("TEST_STATEMENTS", "333" if algo == 0 else "70"))
keylen = 16 if algo == 1 else 32
recs = b"".join(rnd.randbytes(keylen) + bytes(32 - keylen) + rnd.randbytes(12) + rnd.getrandbits(16).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
g.set_deterministic_randomness(rnd.getrandbits(250), rnd.getrandbits(250), rnd.getrandbits(250))
ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
assert ok == n, ok
print("DIGEST", hashlib.sha256(proofs + cts).hexdigest())
print("DESCRIBE", g.describe(algo))
if os.environ.get("TEST_KAT_STAGES"):      # the stage vectors of the App. E statement through whatever witness path this configuration takes
    sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    from conftest import KAT
    g.set_deterministic_randomness(0, 0, 0)
    d = g.debug_prove({"cipher": "chacha20", "key": list(KAT["key"]), "nonce": list(KAT["nonce"]), "counter": KAT["counter"], "input": list(KAT["input"])})
    be = lambda vals: b"".join(v.to_bytes(32, "big") for v in vals)
    print("KAT_W", hashlib.sha256(be(d["W"])).hexdigest()); print("KAT_ABC", hashlib.sha256(be(d["A"]) + be(d["B"]) + be(d["C"])).hexdigest())
"""


def _digest(env_extra, algo=0, pk_path=None):
    import subprocess, sys
    from conftest import ROOT
    # small tables: these child processes share the device with the algorithms the test session already holds (~170 GB)
    env = dict(os.environ, GSC_MAX_BATCH="256", GSC_MIN_SPLIT="64", GSC_WINDOW_Z="6", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="0", GSC_FEW_WIDE="0")
    env.update(env_extra)
    args = [sys.executable, "-c", _OPTIONS_SCRIPT, ROOT, str(algo)] + ([pk_path] if pk_path else [])
    out = subprocess.run(args, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    return [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]


def _child(env_extra, algo=0):
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, GSC_MAX_BATCH="256", GSC_MIN_SPLIT="64", GSC_WINDOW_Z="6", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="0", GSC_FEW_WIDE="0")
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _OPTIONS_SCRIPT, ROOT, str(algo)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    return {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in out.stdout.splitlines() if " " in l}


def test_small_integer_witness_path_is_taken_checked_and_abandoned_when_a_prediction_fails():
    # ChaCha20-V3's witness is small integers: batch calls solve it with the integer kernels on byte planes (csrc/wit_small.hpp) instead of
    # 163 level launches of field arithmetic.  Same bytes as the generic solver (GSC_SMALL_WITNESS=0); the App. E stage hashes sha256(W),
    # sha256(a|b|c) through it (GSC_FEW_PATH=0 sends even a single statement through the batch kernels); and with every constraint row
    # predicted narrow (=2, a test hook) the kernels notice the 672 rows that are not, the chunk is solved again generically — same bytes.
    small = _child({"TEST_KAT_STAGES": "1", "GSC_FEW_PATH": "0"})
    assert "witness=small-integer(162 chained levels, fallbacks 0)" in small["DESCRIBE"], small["DESCRIBE"]
    assert small["KAT_W"] == KAT["sha256_W"] and small["KAT_ABC"] == KAT["sha256_abc"]
    generic = _child({"GSC_SMALL_WITNESS": "0", "GSC_FEW_PATH": "0"})
    assert "witness=generic" in generic["DESCRIBE"] and generic["DIGEST"] == small["DIGEST"]
    wrong = _child({"GSC_SMALL_WITNESS": "2", "GSC_FEW_PATH": "0"})
    assert "witness=small-integer" in wrong["DESCRIBE"] and "fallbacks 0)" not in wrong["DESCRIBE"], wrong["DESCRIBE"]
    assert wrong["DIGEST"] == small["DIGEST"]
    assert _child({})["DIGEST"] == small["DIGEST"]      # (and the default configuration, latency path on)
    # GSC_NTT_PLAIN=0: the first transform kernel takes the planes' Montgomery images (like the generic path) instead of the small integers themselves
    assert _child({"GSC_NTT_PLAIN": "0", "GSC_FEW_PATH": "0"})["DIGEST"] == small["DIGEST"]
    # calls on the latency path: the same kernels (one workgroup walks the levels) or, switched off, the resident lanes-are-terms solver
    few = _child({"TEST_STATEMENTS": "5", "GSC_MAX_BATCH": "64"}); few_res = _child({"TEST_STATEMENTS": "5", "GSC_MAX_BATCH": "64", "GSC_SMALL_WITNESS_FEW": "0"})
    assert few["DIGEST"] == few_res["DIGEST"] == _child({"TEST_STATEMENTS": "5", "GSC_MAX_BATCH": "64", "GSC_FEW_PATH": "0", "GSC_SMALL_WITNESS": "0"})["DIGEST"]


def test_engine_options_do_not_change_the_proofs():
    # Options are read at InitAlgorithm, so each configuration gets its own process (one at a time on the GPU).  GSC_BIT_GROUPS=0
    # predicts nothing: no subset-sum tables, every wire through the windowed kernel; =2 predicts EVERY wire to be a bit, so every
    # group holding a wider value fails its check and its wide scalars take the flat kernel's escape (double-and-add from the
    # scalar itself); GSC_ROW_MARGIN_BITS=-6 gives the narrow wires rows shorter than their values (multiplied out from the row's
    # first entry); the default predicts from a calibration witness.  All must give byte-identical proofs.
    base = _digest({})
    # GSC_DEVICES=0,0: two engine replicas (here both on the one device of the box), every batch split between them — the in-library
    # multi-GPU path of a single FFI host process.
    for extra in ({"GSC_LANES": "2"}, {"GSC_BIT_GROUPS": "0"}, {"GSC_BIT_GROUPS": "2"}, {"GSC_DEVICES": "0,0"}, {"GSC_WINDOW_Z": "11", "GSC_MIN_SPLIT": "512"},
                  {"GSC_WINDOW_Z": "0", "GSC_Z_TABLE_GB": "1", "GSC_LINGER_US": "0"}, {"GSC_SMALL_LANES": "0"},
                  # GSC_QUOTIENT_EVAL=0: the quotient in coefficient form (six transforms, the key's own Z bases) instead of the default evaluation
                  # form (four transforms, the bases V_i and a flat sum over the solver's c rows: k_quot_bases.hip) — the same group element
                  {"GSC_QUOTIENT_EVAL": "0"}, {"GSC_QUOTIENT_EVAL": "0", "GSC_BIT_GROUPS": "0"},
                  # GSC_FUSE_Z_DIGITS=0: the last quotient kernel writes d and a recoding pass makes the digits, instead of writing the digits itself
                  {"GSC_FUSE_Z_DIGITS": "0"}, {"GSC_FUSE_Z_DIGITS": "0", "GSC_WINDOW_Z": "11"},
                  # GSC_OVERLAP_QUOTIENT=0: the quotient kernels before the wire-set MSMs on one stream instead of beside them on the lane's third;
                  # GSC_STREAM_PRIORITIES=0: the lane's streams all at the default priority (shared hardware queues)
                  {"GSC_OVERLAP_QUOTIENT": "0"}, {"GSC_OVERLAP_QUOTIENT": "0", "GSC_LANES": "2"}, {"GSC_OVERLAP_QUOTIENT": "2"}, {"GSC_STREAM_PRIORITIES": "0"}, {"GSC_STREAM_PRIORITIES": "0", "GSC_LANES": "2"}):
        assert _digest(extra) == base, extra


def test_engine_options_do_not_change_the_proofs_aes(aes_keys):
    from conftest import ROOT
    pk_path = os.path.join(ROOT, "build", "keys", "pk.aes128")
    assert os.path.exists(pk_path)
    base = _digest({}, 1, pk_path)
    for extra in ({"GSC_BIT_GROUPS": "0"}, {"GSC_BIT_GROUPS": "2"}, {"GSC_ROW_MARGIN_BITS": "-6"}, {"GSC_WINDOW_W": "9"}, {"GSC_QUOTIENT_EVAL": "0"}, {"GSC_FUSE_Z_DIGITS": "0"},
                  {"GSC_OVERLAP_QUOTIENT": "0"}, {"GSC_STREAM_PRIORITIES": "0"}):
        assert _digest(extra, 1, pk_path) == base, extra


@pytest.mark.parametrize("algo,counts", [(0, ("1", "5")), (1, ("3",))])
def test_latency_path_options_do_not_change_the_proofs(aes_keys, algo, counts):
    # Calls with a handful of statements (GSC_FEW_MAX: 32 ChaCha20, 20 AES) take kernels of their own (resident lanes-are-terms solver with device-wide barriers, flat and
    # windowed MSMs with lanes = bases, the quotient bases as (base, window) rows without a Horner pass, A / B1 sums early on the side
    # stream).  With (r, s, mask) fixed, a handful of statements must give the same bytes as the batch kernels (GSC_FEW_PATH=0
    # GSC_FEW_SOLVER=0), whatever the grid, the quotient layout and the latency rows of the wide wires (GSC_FEW_WIDE).
    from conftest import ROOT
    pk_path = os.path.join(ROOT, "build", "keys", "pk.aes128") if algo else None
    small = {"GSC_MAX_BATCH": "64", "GSC_LANES": "1"}
    for n in counts:
        base = _digest(dict(small, TEST_STATEMENTS=n, GSC_FEW_PATH="0", GSC_FEW_SOLVER="0"), algo, pk_path)
        # quotient layout budgets: ChaCha20 8-bit rows (8.6 GB) and 6-bit ones (2.9 GB); AES 4-bit rows (4.3 GB: the session's own algorithms hold most of the device)
        z, z2 = ("12", "3") if algo == 0 else ("5", "5")
        for extra in ({"GSC_FEW_Z_GB": z, "GSC_FEW_WIDE": "1"}, {"GSC_FEW_Z_GB": "0"}, {"GSC_FEW_Z_GB": z, "GSC_FEW_WGS": "17"},
                      {"GSC_FEW_SOLVER": "0", "GSC_FEW_Z_GB": z2, "GSC_FEW_WIDE": "1"}, {"GSC_FEW_MAX": "2", "GSC_FEW_Z_GB": z2},
                      {"GSC_FEW_Z_GB": "0", "GSC_QUOTIENT_EVAL": "0"}):      # (without the latency layout such calls take the batch form of the quotient: both forms)
            assert _digest(dict(small, TEST_STATEMENTS=n, **extra), algo, pk_path) == base, (n, extra)


def test_resident_solver_gives_up_cleanly_and_the_call_is_solved_again():
    # The resident witness kernel polls its device-wide barriers a bounded number of times.  GSC_FEW_TEST_ABORT (a test hook) makes a
    # barrier unreachable: every workgroup must leave, and the engine must solve the call again with one launch per level — same bytes,
    # one line on stderr.
    import subprocess, sys
    from conftest import ROOT
    # (GSC_SMALL_WITNESS_FEW=0: ChaCha20-V3's latency calls otherwise take the small-integer witness kernels and never launch the resident one)
    env = dict(os.environ, GSC_MAX_BATCH="256", GSC_WINDOW_Z="6", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="0", TEST_STATEMENTS="3", GSC_SMALL_WITNESS_FEW="0")
    want = _digest({"TEST_STATEMENTS": "3"})
    out = subprocess.run([sys.executable, "-c", _OPTIONS_SCRIPT, ROOT, "0"], env=dict(env, GSC_FEW_TEST_ABORT="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0] == want
    assert "solving level by level" in out.stderr


def test_bench_size_batch_every_proof_verifies(gsc_chacha):
    # Full BASELINE size and beyond (4096 statements in one call, CSPRNG randomness): EVERY proof is checked with the drop-in verifier
    # under the reference's vk.chacha20 (three pairings each, on the host cores), plus ciphertext = ChaCha20(key, nonce, counter) XOR input
    # through the identity enc(enc(x)) = x on the same library.
    from concurrent.futures import ThreadPoolExecutor
    g = gsc_chacha
    assert g.init_verifier(0, golden_bytes("vk.chacha20"))
    n = 4096
    rnd = random.Random(2025)
    recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    ok, proofs, lens, cts = g.prove_raw(g.CHACHA20, recs, n)
    assert ok == n and set(lens) == {164}
    assert len({proofs[196 * k:196 * k + 164] for k in range(n)}) == n

    def check(k):
        rec = recs[112 * k:112 * (k + 1)]
        signals = cts[64 * k:64 * k + 64] + rec[32:44] + rec[44:48] + rec[48:]
        return g.verify({"cipher": "chacha20", "proof": base64.b64encode(proofs[196 * k:196 * k + 164]).decode(),
                         "publicSignals": base64.b64encode(signals).decode()})
    with ThreadPoolExecutor(16) as pool:
        res = list(pool.map(check, range(n)))
    assert all(res), [k for k, v in enumerate(res) if not v][:10]
    # a proof does not verify for somebody else's statement
    k = 17
    rec = recs[112 * k:112 * (k + 1)]
    wrong = cts[64 * (k + 1):64 * (k + 2)] + rec[32:44] + rec[44:48] + rec[48:]
    assert not g.verify({"cipher": "chacha20", "proof": base64.b64encode(proofs[196 * k:196 * k + 164]).decode(), "publicSignals": base64.b64encode(wrong).decode()})


def test_a_small_batch_after_a_larger_one_on_the_same_lane(gsc_chacha, oracle, chacha_oracle):
    # Batch buffers are [row][proof] with the batch as the row stride, so a batch of 64 columns finds the rows of an earlier, larger
    # batch where its own rows end: whatever a kernel reads beyond the rows the solver wrote (the zero row behind c that pads the last
    # octet of the evaluation-form quotient's c set) must be set for every batch.  200 statements, then 40, then 3 (latency path), then 40.
    g = gsc_chacha; cs, pk, vk = chacha_oracle
    rnd = random.Random(31337)
    r, s = rnd.getrandbits(250), rnd.getrandbits(250)
    g.set_deterministic_randomness(r, s, 0)
    try:
        for n in (200, 40, 3, 40):
            recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
            ok, proofs, lens, cts = g.prove_raw(g.CHACHA20, recs, n)
            assert ok == n and set(lens) == {164}
            for k in (0, n - 1):
                rec = recs[112 * k:112 * (k + 1)]
                want, want_ct = oracle.prove(cs, pk, "chacha20", rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s)
                assert cts[64 * k:64 * k + 64] == want_ct and proofs[196 * k:196 * k + 164] == want, (n, k)
    finally:
        g.set_deterministic_randomness(None)


def test_repeated_batches_are_bit_identical(gsc_chacha):
    # No atomics, no data-dependent scheduling: with (r, s) fixed, proving the same 2048 statements again — alone or embedded in a
    # larger call that changes chunking and slice counts — must give the same bytes (catches races and uninitialised reads).
    g = gsc_chacha
    rnd = random.Random(31337)
    n = 2048
    recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    g.set_deterministic_randomness(rnd.getrandbits(250), rnd.getrandbits(250), 0)
    try:
        ok1, p1, l1, c1 = g.prove_raw(g.CHACHA20, recs, n)
        ok2, p2, l2, c2 = g.prove_raw(g.CHACHA20, recs, n)
        ok3, p3, l3, c3 = g.prove_raw(g.CHACHA20, recs[:112 * 193], 193)          # a ragged prefix: different batch shape
    finally:
        g.set_deterministic_randomness(None)
    assert ok1 == ok2 == n and ok3 == 193
    assert p1 == p2 and c1 == c2
    assert p3 == p1[:196 * 193] and c3 == c1[:64 * 193]
