import os, sys, lzma, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
from oracle import oracle as O
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r1cs = lzma.open(os.path.join(ROOT, "tests", "golden", "r1cs.aes128.xz")).read()
pkb, vkb = O.setup(O.R1CS(r1cs), bytes([1] * 32))
assert g.init_algorithm(1, pkb, r1cs); print(g.describe(1), flush=True)
rnd = random.Random(1)
for rep in range(2):
    recs = b"".join(rnd.randbytes(32) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    ok, *_ = g.prove_raw(1, recs, n)
    print(ok, g.last_stage_ms(1), flush=True)
