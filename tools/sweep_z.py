"""One process = one Z-table digit width (GSC_WINDOW_Z): KAT proof check, then timing of full batches.
usage: GSC_WINDOW_Z=16 python tools/sweep_z.py [batch] [reps]"""
import base64, json, os, sys, time
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ.setdefault("GSC_MAX_BATCH", str(n))
t = time.time()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
print("init %.2fs" % (time.time() - t), g.describe(0), flush=True)
PT = "a3f7e592aeda1507a7f51b35812dfc50a263d5a6d2df625e563b02e49c08bf30d0e7483f5b13ff079532224ee8fbc31ab1899b18e453d36d9793a8355eb0dee9"
params = {"cipher": "chacha20", "key": [2] * 32, "nonce": [3] * 12, "counter": 3, "input": list(bytes.fromhex(PT))}
KAT1 = "ea49984df7447a7faa42e256b4ba77b18c134f87a9f8938bcc7722f9524b81f0a81727420993af92a92d8e28650e858ff01fbded7396dd3a41033abac4d97d5c0097d13efc1142d0730950c26c4c55037bb1dc96b9e3422eae0729ef36af113fd8fa21e5ff66d3c144a6d75436c9b87866463d76e98b68310f3bce6130d699a6000000004000000000000000000000000000000000000000000000000000000000000000"
g.set_deterministic_randomness(0x1234567, 0xabcdef0123456789abcdef, 0)
res = json.loads(g.prove(params))
ok = base64.b64decode(res["proof"]["proofJson"]).hex() == KAT1
print("KAT proof bytes:", ok, flush=True)
g.set_deterministic_randomness(None)
if not ok:
    sys.exit(1)
for i in range(reps):
    t = time.time()
    okn, *_ = g.prove_raw(0, bench.synthetic_records(n, i), n)
    dt = time.time() - t
    print("batch %d ok=%d %.3fs -> %.1f proofs/s" % (n, okn, dt, n / dt), g.last_stage_ms(0), "z-kernel ms/batch/nbases:", g.last_msm_z_kernel(0), flush=True)
