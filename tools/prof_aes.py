"""Two batches of N AES-128 proofs (one warm, one measured) for rocprofv3 runs; keys from the product's Setup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
os.environ.setdefault("GSC_MAX_BATCH", str(n))
r1cs = bench.golden("r1cs.aes128")
pk, vk = g.setup(r1cs)
assert g.init_algorithm(1, pk, r1cs)
print(g.describe(1), flush=True)
for i in range(2):
    ok, *_ = g.prove_raw(1, bench.provable(bench.synthetic_records(n, i), "aes128"), n)
    print(ok, g.last_stage_ms(1), g.last_msm_z_kernel(1), flush=True)
