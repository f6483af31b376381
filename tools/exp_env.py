"""Z-kernel time for the environment this process was started with (one process per setting; same box, back to back).
usage: GSC_WIN_SLICE=64 python tools/exp_env.py 8192 4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
n = int(sys.argv[1]); reps = int(sys.argv[2])
os.environ.setdefault("GSC_MAX_BATCH", str(n)); os.environ.setdefault("GSC_Z_TABLE_GB", "72")
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
recs = bench.synthetic_records(n, 1)
g.prove_raw(0, recs, n)
ms = []
for r in range(reps):
    g.prove_raw(0, recs, n); ms.append(g.last_msm_z_kernel(0)[0])
print({k: v for k, v in os.environ.items() if k.startswith("GSC_") and k not in ("GSC_MAX_BATCH", "GSC_Z_TABLE_GB")}, "z-kernel ms: min %.2f median %.2f" % (min(ms), sorted(ms)[len(ms) // 2]), g.last_stage_ms(0), flush=True)
