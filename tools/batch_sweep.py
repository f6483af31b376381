"""Throughput vs batch size for ChaCha20-V3 on one GPU (timing only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
print(g.describe(0), flush=True)
for n in (256, 512, 1024, 2048, 4096):
    if n > int(os.environ.get("GSC_MAX_BATCH", "1024")): break
    best = 0
    for rep in range(3):
        recs = bench.synthetic_records(n, rep)
        t = time.time(); ok, *_ = g.prove_raw(0, recs, n); dt = time.time() - t
        best = max(best, n / dt)
    print("batch %5d: %.0f proofs/s  stages %s" % (n, best, {k: round(v, 1) for k, v in g.last_stage_ms(0).items()}), flush=True)
