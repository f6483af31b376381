"""Device memory left after each InitAlgorithm of the all-resident configuration (bench.engine_env("mixed", 1024)): what the third algorithm's
table budget is derived from (engine_tables.hip init_key).  Run on the GPU box; needs build/keys (tests' aes_keys fixture or bench.py --workload aes128 once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for k, v in bench.engine_env("mixed", 1024).items():
    os.environ.setdefault(k, v)
os.environ.setdefault("GSC_TRACE_HOST", "1")
import gsc_loader, torch
g = gsc_loader.load()
def free():
    f, t = torch.cuda.mem_get_info(0); return f / 2**30, t / 2**30
print("start: free %.1f GiB of %.1f" % free())
for algo, name in ((0, "chacha20"), (1, "aes128"), (2, "aes256")):
    p = os.path.join("build", "keys", "pk." + name)
    pk = bench.golden("pk.chacha20") if algo == 0 else open(p, "rb").read()
    assert g.init_algorithm(algo, pk, bench.golden("r1cs." + name)), name
    print(name, "free %.1f GiB" % free()[0], "|", g.describe(algo)[:160])
