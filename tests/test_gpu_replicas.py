"""Small calls reach every engine replica.  The reference's unit of work is one statement per Prove call, from any
number of concurrent FFI threads (libraries/prover/libprove.go:30-47, libraries/core_test.go:44-111).  With GSC_DEVICES=0,0 — two
replicas, here both on the one device of the box — 64 concurrent single Prove callers must be served by BOTH replicas (per-replica
counters in gsc_describe), every proof must verify, and the same holds for concurrent small gsc_prove_raw calls.  A latency-path soak
(random call sizes, sequential and concurrent, every proof verified) covers the resident witness kernel's device-wide barrier."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_CHILD = r"""
import base64, json, os, random, sys, threading
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
from concurrent.futures import ThreadPoolExecutor
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
nrep = len(os.environ["GSC_DEVICES"].split(","))
assert "devices=%d " % nrep in g.describe(0), g.describe(0)
assert g.served(0) == [(0, 0)] * nrep
rnd = random.Random(99)


def statement():
    return {"cipher": "chacha20", "key": list(rnd.randbytes(32)), "nonce": list(rnd.randbytes(12)), "counter": rnd.getrandbits(32), "input": list(rnd.randbytes(64))}


def accepted(q, out):
    out = json.loads(out)
    sig = base64.b64decode(out["publicSignals"]) + bytes(q["nonce"]) + q["counter"].to_bytes(4, "little") + bytes(q["input"])
    return g.verify({"cipher": "chacha20", "proof": out["proof"]["proofJson"], "publicSignals": base64.b64encode(sig).decode()})


# 1. 64 concurrent single-statement Prove callers (JSON in, JSON out, micro-batcher)
qs = [statement() for _ in range(64)]
start = threading.Barrier(64)


def caller(q):
    start.wait()
    return g.prove(q)
with ThreadPoolExecutor(64) as pool:
    outs = list(pool.map(caller, qs))
assert all(accepted(q, o) for q, o in zip(qs, outs))
sv = g.served(0)
print("SERVED after 64 concurrent Prove calls:", sv)
assert sum(s for _, s in sv) == 64 and all(c > 0 for c, _ in sv), sv
# 2. sequential single calls: each goes to the replica that has served less so far (both idle: ties by statements served)
before = g.served(0)
for _ in range(6):
    q = statement(); assert accepted(q, g.prove(q))
after = g.served(0)
assert sum(a[0] - b[0] for a, b in zip(after, before)) == 6
gap = lambda sv: max(s for _, s in sv) - min(s for _, s in sv)
assert gap(after) <= max(gap(before) - 6, 1), (before, after)
# 3. concurrent small binary calls (gsc_prove_raw, 1..32 statements): least-loaded replica each, every proof verified
def raw_call(seed):
    r = random.Random(seed); n = r.choice([1, 2, 5, 17, 32, 33, 64])
    recs = bench.xoshiro_records(n, seed << 20)
    ok, proofs, lens, cts = g.prove_raw(0, recs, n)
    assert ok == n
    return all(bench.verify_items(g, [("chacha20", proofs[196 * k:196 * k + 164], bench.signals_of("chacha20", recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])) for k in range(n)], 4))
with ThreadPoolExecutor(6) as pool:
    assert all(pool.map(raw_call, range(24)))
sv2 = g.served(0)
print("SERVED at the end:", sv2)
lo, hi = min(s for _, s in sv2), max(s for _, s in sv2)
assert lo * 4 >= hi, sv2
print("CHILD-OK")
"""


def _run(devices, extra=None):
    env = dict(os.environ, GSC_DEVICES=devices, GSC_MAX_BATCH="256", GSC_WINDOW_Z="8", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="3")
    env.update(extra or {})
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_concurrent_single_proofs_are_served_by_both_replicas():
    out = _run("0,0")
    assert "SERVED after 64 concurrent Prove calls" in out


_SOAK = r"""
import os, random, sys, threading
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
bad = []


def soak(seed, calls):
    r = random.Random(seed)
    for c in range(calls):
        n = r.choice([1, 1, 1, 2, 3, 5, 8, 13, 21, 32])
        recs = bench.xoshiro_records(n, (seed << 30) + (c << 8))
        ok, proofs, lens, cts = g.prove_raw(0, recs, n)
        res = bench.verify_items(g, [("chacha20", proofs[196 * k:196 * k + 164], bench.signals_of("chacha20", recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])) for k in range(n)], 8)
        if ok != n or not all(res):
            bad.append((seed, c, n, ok, res.count(False)))


soak(1, 60)                                              # sequential
ts = [threading.Thread(target=soak, args=(10 + i, 25)) for i in range(4)]      # four concurrent callers on two lanes: resident launches chained per device
for t in ts: t.start()
for t in ts: t.join()
assert not bad, bad[:5]
name, ms, stmts, cols, nb = g.last_dominant_kernel(0)
assert name.startswith(os.environ["EXPECT_KERNEL"]) and stmts <= 32 and cols == 64, (name, stmts, cols)
print("CHILD-OK")
"""


def test_latency_path_soak_every_proof_verifies():
    # the resident lanes-are-terms witness kernel with its release / acquire device-wide barrier (k_solver.hip few_grid_barrier):
    # a stale wire value at any level gives an unsatisfied system or a proof the verifier rejects
    # (GSC_SMALL_WITNESS_FEW=0: ChaCha20-V3's latency calls otherwise take the small-integer witness kernels, which have no device-wide barrier —
    # the second run soaks those)
    for extra in ({"GSC_SMALL_WITNESS_FEW": "0", "EXPECT_KERNEL": "k_solver_few"}, {"EXPECT_KERNEL": "k_wit_chain"}):
        env = dict(os.environ, GSC_MAX_BATCH="64", GSC_LANES="2", GSC_WINDOW_Z="8", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="9", **extra)
        p = subprocess.run([sys.executable, "-c", _SOAK, ROOT], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


_FOUR = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import bench, gsc_loader
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
assert "devices=4 " in g.describe(0), g.describe(0)
n = 4 * 64 + 37
recs = bench.xoshiro_records(n, 0x4444 << 20)
ok, proofs, lens, cts = g.prove_raw(0, recs, n)
assert ok == n and set(lens) == {164}
res = bench.verify_items(g, [("chacha20", proofs[196 * k:196 * k + 164], bench.signals_of("chacha20", recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])) for k in range(n)], 16)
assert all(res), [k for k, v in enumerate(res) if not v][:10]
assert len({proofs[196 * k:196 * k + 164] for k in range(n)}) == n
sv = g.served(0)
print("SERVED", sv)
# contiguous shares of ceil(293 / 4) -> 128 statements (whole 64-column batches): 128 + 128 + 37, the fourth replica gets nothing (dispatch.hpp plan_shares)
assert sv == [(1, 128), (1, 128), (1, 37), (0, 0)], sv
# ... and a call of one batch goes whole to the replica that has served least
ok, proofs, lens, cts = g.prove_raw(0, recs[:112 * 64], 64)
assert ok == 64 and g.served(0)[3] == (1, 64), g.served(0)
print("CHILD-OK")
"""


def test_a_call_of_four_batches_and_a_ragged_tail_is_shared_out_over_four_replicas():
    # the in-library multi-GPU path with more than two replicas (all four on the one device of the box): per-device
    # tables and streams, hipSetDevice discipline of the share threads, every result slot written exactly once.
    env = dict(os.environ, GSC_DEVICES="0,0,0,0", GSC_MAX_BATCH="256", GSC_WINDOW_Z="8", GSC_W_TABLE_GB="8", GSC_FEW_Z_GB="3")
    p = subprocess.run([sys.executable, "-c", _FOUR, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "CHILD-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
