import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsc_loader, bench
g = gsc_loader.load()
t = time.time(); assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")); print("InitAlgorithm chacha20 %.2f s" % (time.time() - t), flush=True)
r1cs = bench.golden("r1cs.aes128"); pk, vk = g.setup(r1cs)
t = time.time(); assert g.init_algorithm(1, pk, r1cs); print("InitAlgorithm aes128 %.2f s" % (time.time() - t), flush=True)
print(g.describe(0)); print(g.describe(1))
