"""The engine configuration bench.py TIMES is a tested configuration ("parity gate for every timed configuration";
the reference's own gate is libraries/core_test.go:171).

The rest of the GPU session runs with small tables (conftest.py: GSC_Z_TABLE_GB=24 -> Z digits of c = 14, lanes of 1024 proofs).  The
driver's bench runs bench.engine_env("chacha20", 8192): c = 17 — digits beyond int16, the recoder's wide format of eight int32 per octet
(digits in [-2^16, 2^16 - 1], row index 2^16 - 1) —, 137 GB of Z rows, one lane of 8192 proofs, 128 slices x 15 windows x 128 proof
groups; nothing else in the suite reaches that format.  This test starts a prover
process with exactly that environment, proves 8192 statements in one call with (r, s) fixed, compares eight of them byte for byte
with the CPU oracle (first / last wave and the wave boundaries 63 | 64, 4095 | 4096, 8127 | 8128) and verifies ALL of them with
libverify.so under the reference's vk.chacha20.

The file sorts before every other GPU test: its child processes need ~200 GB of the device, which the session's own algorithms
(~170 GB once the AES tests have run) would not leave."""
import base64
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import ROOT, golden_bytes

pytestmark = pytest.mark.gpu

_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
import torch
free, total = torch.cuda.mem_get_info(0)
assert free > 205e9, "the bench configuration needs ~200 GB of device memory; only %.0f GB are free (run this test before the session loads its own algorithms)" % (free / 1e9)
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
d = g.describe(0)
print("DESCRIBE", d)
assert "window_z=17 " in d and "max_batch=8192 " in d and "lanes=1 " in d and "quotient=evaluation-form+digits" in d, d
n = 8192
recs = bench.xoshiro_records(n, 0x7E57 << 20)
g.set_deterministic_randomness(int(sys.argv[3]), int(sys.argv[4]), 0)
ok, proofs, lens, cts = g.prove_raw(0, recs, n)
assert ok == n and set(lens) == {164}, (ok, set(lens))
name, ms, stmts, cols, nb = g.last_dominant_kernel(0)
assert name.startswith("k_msm_win") and stmts == n and cols == n and nb == 32768, (name, stmts, cols, nb)      # evaluation-form quotient: the n bases V_i
open(sys.argv[2], "wb").write(recs + proofs + cts)
print("CHILD-OK")
"""


def test_the_timed_configuration_proves_8192_statements_bit_exactly(gsc, oracle, chacha_oracle, tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    n = 8192
    r, s = 0x1234567, 0xabcdef0123456789abcdef
    env = {k: v for k, v in os.environ.items() if not k.startswith("GSC_")}
    env.update(bench.engine_env("chacha20", n))                 # exactly what `python bench.py` sets before it loads the library
    env["GSC_ENABLE_TEST_HOOKS"] = "1"                          # fixed (r, s): what byte-level parity is defined on
    out_path = str(tmp_path / "out.bin")
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, out_path, str(r), str(s)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    blob = open(out_path, "rb").read()
    recs, proofs, cts = blob[:112 * n], blob[112 * n:112 * n + 196 * n], blob[112 * n + 196 * n:]
    assert len(cts) == 64 * n and recs == bench.xoshiro_records(n, 0x7E57 << 20)
    # byte for byte against the oracle (pinned by the reference's App. E vectors and vk.chacha20: tests/test_oracle.py)
    cs, pk, vk = chacha_oracle
    for k in (0, 63, 64, 4095, 4096, 8127, 8128, 8191):
        rec = recs[112 * k:112 * (k + 1)]
        want, want_ct = oracle.prove(cs, pk, "chacha20", rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s)
        assert cts[64 * k:64 * k + 64] == want_ct, k
        assert proofs[196 * k:196 * k + 164] == want, k
    # every proof through the product's verifier under the reference's verifying key
    assert gsc.init_verifier(0, golden_bytes("vk.chacha20"))

    def check(k):
        rec = recs[112 * k:112 * (k + 1)]
        return gsc.verify({"cipher": "chacha20", "proof": base64.b64encode(proofs[196 * k:196 * k + 164]).decode(),
                           "publicSignals": base64.b64encode(bench.signals_of("chacha20", rec, cts[64 * k:64 * k + 64])).decode()})
    with ThreadPoolExecutor(min(32, len(os.sched_getaffinity(0)))) as pool:
        res = list(pool.map(check, range(n)))
    assert all(res), [k for k, v in enumerate(res) if not v][:10]
    assert len({proofs[196 * k:196 * k + 164] for k in range(n)}) == n


_CHILD_AES = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
import torch
free, total = torch.cuda.mem_get_info(0)
assert free > 235e9, "the AES bench configuration needs ~225 GB of device memory; only %.0f GB are free" % (free / 1e9)
g = gsc_loader.load()
algo, name = int(sys.argv[5]), sys.argv[6]
assert g.init_algorithm(algo, open(sys.argv[7], "rb").read(), bench.golden("r1cs." + name))
d = g.describe(algo)
print("DESCRIBE", d)
assert "window_z=15 " in d and "max_batch=1024 " in d and "lanes=2 " in d and "quotient=evaluation-form+digits" in d and "witness=generic" in d, d
assert name != "aes128" or "window_w=15 " in d, d
n = 1024
recs = bench.provable(bench.xoshiro_records(n, 0xAE5 << 20), name)
g.set_deterministic_randomness(int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[3]) ^ 0x5555)
ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
assert ok == n and set(lens) == {196}, (ok, set(lens))
open(sys.argv[2], "wb").write(recs + proofs + cts)
print("CHILD-OK")
"""


@pytest.mark.parametrize("name,algo,cipher,keylen", [("aes128", 1, "aes-128-ctr", 16), ("aes256", 2, "aes-256-ctr", 32)])
def test_the_timed_aes_configurations_prove_1024_statements_bit_exactly(gsc, oracle, aes_keys, tmp_path, name, algo, cipher, keylen):
    # the same for `bench.py --workload aes128 / aes256` (BASELINE configs 3 and 5): Z digits of c = 15 (137 GB of rows; wide-wire digits of c = 15 for AES-128),
    # two lanes of 1024 proofs.  Keys: the oracle's Setup for the reference's r1cs.aes* (the reference ships no pk.aes*; parity against gnark is unpinned for
    # AES, tests/test_gpu_aes.py) — three proofs byte for byte against the oracle, all 1024 through libverify.so under the matching vk.
    sys.path.insert(0, ROOT)
    import bench
    n = 1024
    r, s = 0x1234567, 0xabcdef0123456789abcdef
    r1cs, pkb, vkb = aes_keys[name]
    env = {k: v for k, v in os.environ.items() if not k.startswith("GSC_")}
    env.update(bench.engine_env(name, n))
    env["GSC_ENABLE_TEST_HOOKS"] = "1"
    out_path = str(tmp_path / "out.bin")
    p = subprocess.run([sys.executable, "-c", _CHILD_AES, ROOT, out_path, str(r), str(s), str(algo), name, os.path.join(ROOT, "build", "keys", "pk." + name)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    blob = open(out_path, "rb").read()
    recs, proofs, cts = blob[:112 * n], blob[112 * n:112 * n + 196 * n], blob[112 * n + 196 * n:]
    cs, pk = oracle.R1CS(r1cs), oracle.ProvingKey(pkb)
    for k in (0, 63, 1023):
        rec = recs[112 * k:112 * (k + 1)]
        want, want_ct = oracle.prove(cs, pk, cipher, rec[:keylen], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s, r ^ 0x5555)
        assert cts[64 * k:64 * k + 64] == want_ct and proofs[196 * k:196 * k + 196] == want, k
    assert gsc.init_verifier(algo, vkb)
    res = bench.verify_items(gsc, [(cipher, proofs[196 * k:196 * k + 196], bench.signals_of(name, recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])) for k in range(n)],
                             min(32, len(os.sched_getaffinity(0))))
    assert all(res), [k for k, v in enumerate(res) if not v][:10]


_CHILD_MIXED = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
g = gsc_loader.load()
for algo, name in ((0, "chacha20"), (1, "aes128"), (2, "aes256")):
    pk = bench.golden("pk.chacha20") if algo == 0 else open(os.path.join(sys.argv[1], "build", "keys", "pk." + name), "rb").read()
    assert g.init_algorithm(algo, pk, bench.golden("r1cs." + name)), name
    d = g.describe(algo); print("DESCRIBE", name, d)
    assert "max_batch=1024 " in d and "window_z=%d " % (15 if algo == 0 else 13) in d, d      # the library's default table budgets: all three resident
n = 3072
reqs = bench.mixed_json(n, 0x317ED << 20)
g.set_deterministic_randomness(int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[3]) ^ 0x5555)
out = g.prove_batch_bytes(reqs)
assert out.count(b'"proofJson"') == n
open(sys.argv[2], "wb").write(json.dumps({"reqs": json.loads(reqs), "outs": json.loads(out)}).encode())
print("CHILD-OK")
"""


def test_the_timed_mixed_configuration_proves_3x1024_statements_bit_exactly(gsc, oracle, chacha_oracle, aes_keys, tmp_path):
    # `bench.py --workload mixed` (BASELINE config 5): bench.engine_env("mixed", 1024) = the library's default table budgets (ChaCha20 c = 15, AES c = 13)
    # with lanes of 1024, all three algorithms resident, one ProveBatch call of 3 x 1024 statements in JSON (statement i uses cipher i mod 3: bench.mixed_json).
    # Three proofs per cipher byte for byte against the oracle; all 3072 through libverify.so.
    import json
    sys.path.insert(0, ROOT)
    import bench
    r, s = 0x1234567, 0xabcdef0123456789abcdef
    env = {k: v for k, v in os.environ.items() if not k.startswith("GSC_")}
    env.update(bench.engine_env("mixed", 1024))
    env["GSC_ENABLE_TEST_HOOKS"] = "1"
    out_path = str(tmp_path / "mixed.json")
    p = subprocess.run([sys.executable, "-c", _CHILD_MIXED, ROOT, out_path, str(r), str(s)], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    blob = json.load(open(out_path)); reqs, outs = blob["reqs"], blob["outs"]
    n = len(reqs); assert n == 3072 and len(outs) == n
    names = {"chacha20": ("chacha20", 0), "aes-128-ctr": ("aes128", 1), "aes-256-ctr": ("aes256", 2)}
    oracles = {"chacha20": (chacha_oracle[0], chacha_oracle[1])}
    for nm, c in (("aes128", "aes-128-ctr"), ("aes256", "aes-256-ctr")):
        oracles[c] = (oracle.R1CS(aes_keys[nm][0]), oracle.ProvingKey(aes_keys[nm][1]))
        assert gsc.init_verifier(names[c][1], aes_keys[nm][2])
    assert gsc.init_verifier(0, golden_bytes("vk.chacha20"))
    items = []
    for q, o in zip(reqs, outs):
        rec = bytes(32) + base64.b64decode(q["nonce"]) + int(q["counter"]).to_bytes(4, "little") + base64.b64decode(q["input"])
        items.append((q["cipher"], base64.b64decode(o["proof"]["proofJson"]), bench.signals_of(names[q["cipher"]][0], rec, base64.b64decode(o["publicSignals"]))))
    for k in (0, 1, 2, 1536, 1537, 1538, n - 3, n - 2, n - 1):      # three per cipher
        q = reqs[k]; cs, pk = oracles[q["cipher"]]
        want, want_ct = oracle.prove(cs, pk, q["cipher"], base64.b64decode(q["key"]), base64.b64decode(q["nonce"]), q["counter"], base64.b64decode(q["input"]), r, s, r ^ 0x5555)
        assert items[k][1] == want and base64.b64decode(outs[k]["publicSignals"]) == want_ct, k
    res = bench.verify_items(gsc, items, min(32, len(os.sched_getaffinity(0))))
    assert all(res), [k for k, v in enumerate(res) if not v][:10]


def _bench_line(*args):
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + list(args), env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_verifies_its_own_proofs_and_reports_it():
    # bench.py itself, small (a batch of 256 at the session's table budget): the JSON line must carry "verified" and the roofline
    # bookkeeping of the launch it timed; a run whose proofs do not verify exits non-zero (bench.py: "REJECTED").
    line = _bench_line("--steps", "2", "--warmup", "1", "--batch", "256", "--verify", "64")
    assert line["verified"] == 64 and line["n_gpus"] == 1 and line["config"]["batch_per_gpu"] == 256
    rf = line["roofline"]
    assert rf["proofs_per_launch"] == 256 and rf["kernel"].startswith("k_msm_win") and rf["algorithmic_bytes_per_launch"] == 256 * 32768 * 96
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6 and "traffic_source" in rf
    # the binding roofline (VALU issue) is in the line too, recomputable from its own fields: clock measured live by the kernel, instruction
    # count per wave-addition replayed from the committed counter pass
    rv = line["roofline_valu"]
    assert rv["bound"] == "valu-issue" and rv["simds"] >= 64 and 500 < rv["clock_mhz"] < 3500 and rv["instr_per_wave_add"] > 1000 and "instr_source" in rv
    assert rv["wave_adds_per_launch"] == 32768 * rv["windows"] * (256 // 64) and rv["launch_ms"] == rf["launch_ms"]
    cpi = rv["launch_ms"] * 1e-3 * rv["clock_mhz"] * 1e6 * rv["simds"] / (rv["instr_per_wave_add"] * rv["wave_adds_per_launch"])
    assert abs(cpi - rv["cycles_per_wave_instr"]) < 1e-2 * cpi and abs(rv["frac"] - 4.0 / rv["cycles_per_wave_instr"]) < 1e-3
    assert abs(rv["miss_clock_loss_ms"] - (rv["launch_ms"] - rv["valu_only_ms"])) < 1e-2


def test_bench_bookkeeping_of_a_single_statement_call():
    # bytes are counted for the STATEMENTS a launch proved and for the kernel that was timed.  One statement per call:
    # the resident witness kernel of the latency path, SURVEY 8(d)'s witness bytes of ONE proof (not of 64 padded columns).
    one = _bench_line("--steps", "6", "--warmup", "2", "--batch", "1", "--callers", "1", "--verify", "1")["roofline"]      # one caller: concurrent small calls would share a launch
    assert one["kernel"].startswith("k_wit_chain") and one["proofs_per_launch"] == 1 and one["columns_per_launch"] == 64      # ChaCha20-V3: the small-integer witness kernels
    assert one["algorithmic_bytes_per_launch"] == 3012224 and one["traffic"] is None
