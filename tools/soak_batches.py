"""Soak of the batch path at the bench configuration: rounds of gsc_prove_raw calls of random sizes (64 .. 8192 statements, ragged ones included),
one or two concurrent callers, a sample of every call's proofs checked with libverify.so (all of them for calls up to 256 statements).
usage: soak_batches.py [seed] [rounds]    (environment as bench.py sets it: bench.engine_env)"""
import os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
for k, v in bench.engine_env("chacha20", 8192).items():
    os.environ.setdefault(k, v)
import gsc_loader
g = gsc_loader.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
print(g.describe(0), flush=True)
rnd = random.Random(seed)
sizes = [64, 65, 100, 127, 128, 200, 256, 333, 512, 777, 1024, 2048, 3000, 4096, 5000, 8191, 8192]
bad = 0; total = 0; checked = 0; t0 = time.time()


def one(call):
    n = rnd.choice(sizes)
    recs = bench.xoshiro_records(n, (seed << 40) + (call << 20))
    ok, proofs, lens, cts = g.prove_raw(0, recs, n)
    idx = list(range(n)) if n <= 256 else bench.sample_indices(n, 256)
    res = bench.verify_items(g, [("chacha20", proofs[196 * k:196 * k + 164], bench.signals_of("chacha20", recs[112 * k:112 * (k + 1)], cts[64 * k:64 * k + 64])) for k in idx], 16)
    return n, ok, len(idx), res.count(False)


for r in range(rounds):
    with ThreadPoolExecutor(1 + r % 2) as pool:
        for n, ok, c, rej in pool.map(one, range(4 * r, 4 * r + 4)):
            total += n; checked += c; bad += rej + (n - ok)
    print("round %d: %d statements so far, %d checked, %d bad, %.0f s" % (r, total, checked, bad, time.time() - t0), flush=True)
print("SOAK", "FAILED" if bad else "OK", total, checked)
sys.exit(1 if bad else 0)
