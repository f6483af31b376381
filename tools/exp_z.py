"""Interleaved experiment rounds in ONE process (same thermal state): GSC_MSM_EXP values cycled per batch.
usage: GSC_WINDOW_Z=16 python tools/exp_z.py 8192 4 0 1 2   -> 4 rounds of exp=0, exp=1, exp=2"""
import os, sys, time
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
n = int(sys.argv[1]); rounds = int(sys.argv[2]); exps = sys.argv[3:]
os.environ.setdefault("GSC_MAX_BATCH", str(n))
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
print(g.describe(0), flush=True)
recs = bench.synthetic_records(n, 1)
g.prove_raw(0, recs, n)
res = {e: [] for e in exps}
for r in range(rounds):
    for e in exps:
        os.environ["GSC_MSM_EXP"] = e
        g.prove_raw(0, recs, n)
        res[e].append(g.last_msm_z_kernel(0)[0])
for e in exps:
    v = sorted(res[e]); print("exp=%s z-kernel ms: min %.2f median %.2f  all %s" % (e, v[0], v[len(v) // 2], " ".join("%.1f" % x for x in res[e])), flush=True)
