"""A few single-statement Prove calls (batch of 64 lanes on the device) for rocprofv3 kernel traces of the latency path."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
p = {"cipher": "chacha20", "key": [2] * 32, "nonce": [3] * 12, "counter": 3, "input": [7] * 64}
for i in range(6):
    t = time.time(); out = json.loads(g.prove(p)); dt = time.time() - t
    print("Prove %.2f ms" % (dt * 1e3), "proof" in out, g.last_stage_ms(0), flush=True)
