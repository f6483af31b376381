import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
os.environ.update(bench.engine_env("chacha20", 8192)); os.environ["GSC_TRACE_HOST"] = "1"
import gsc_loader
g = gsc_loader.load()
for rep in range(1):
    t = time.time(); assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")); print("InitAlgorithm chacha20 at bench.engine_env: %.2f s" % (time.time() - t), flush=True)
print(g.describe(0))
