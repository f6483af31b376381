"""Two batches of N ChaCha proofs (one warm, one measured) for rocprofv3 runs; window from GSC_WINDOW_Z."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
os.environ.setdefault("GSC_MAX_BATCH", str(n))
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
for i in range(2):
    ok, *_ = g.prove_raw(0, bench.synthetic_records(n, i), n)
    print(ok, g.last_stage_ms(0), g.last_msm_z_kernel(0), flush=True)
