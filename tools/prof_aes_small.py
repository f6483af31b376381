"""Latency of small AES-128 calls (keys from the product's Setup); GSC_* options from the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
r1cs = bench.golden("r1cs.aes128")
pk, vk = g.setup(r1cs)
assert g.init_algorithm(1, pk, r1cs)
print(g.describe(1), flush=True)
for n in [int(a) for a in sys.argv[1:]] or [1, 1, 4, 16, 32, 33]:
    recs = bench.provable(bench.synthetic_records(n, n), "aes128")
    best = 1e9
    for _ in range(4):
        t = time.time(); ok, *_ = g.prove_raw(1, recs, n); dt = time.time() - t
        assert ok == n, ok
        best = min(best, dt)
    print("n = %3d  %.2f ms  %s" % (n, best * 1e3, {k: round(v, 2) for k, v in g.last_stage_ms(1).items()}), flush=True)
