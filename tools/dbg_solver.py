import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
n = 64
for i in range(2):
    ok, *_ = g.prove_raw(0, bench.synthetic_records(n, i), n)
print("GSC_DBG=%s witness_ms=%.3f" % (os.environ.get("GSC_DBG", "0"), g.last_stage_ms(0)["witness"]))
