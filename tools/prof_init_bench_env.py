"""InitAlgorithm of ChaCha20-V3 at exactly bench.py's engine settings, with the host-side breakdown (GSC_TRACE_HOST) on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
os.environ.update(bench.engine_env("chacha20", 8192)); os.environ["GSC_TRACE_HOST"] = "1"
import gsc_loader
g = gsc_loader.load()
t = time.time(); assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")); print("InitAlgorithm chacha20 at bench.engine_env: %.2f s" % (time.time() - t), flush=True)
print(g.describe(0))
