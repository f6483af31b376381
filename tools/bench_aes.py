"""AES-128/256-V2 throughput on one GPU with oracle-made test keys (the reference ships no AES proving key)."""
import os, sys, time, lzma, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
from oracle import oracle as O
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for algo, name, cipher, kl in ((1, "aes128", "aes-128-ctr", 16), (2, "aes256", "aes-256-ctr", 32)):
    r1cs = lzma.open(os.path.join(ROOT, "tests", "golden", "r1cs.%s.xz" % name)).read()
    pkb, vkb = O.setup(O.R1CS(r1cs), bytes([algo] * 32))
    t = time.time(); assert g.init_algorithm(algo, pkb, r1cs); print(name, "init %.1fs" % (time.time() - t), g.describe(algo), flush=True)
    vk = O.VerifyingKey(vkb)
    rnd = random.Random(algo)
    for rep in range(2):
        recs = b"".join(rnd.randbytes(32) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
        t = time.time(); ok, proofs, lens, cts = g.prove_raw(algo, recs, n); dt = time.time() - t
        k = n - 1; rec = recs[112 * k:112 * (k + 1)]
        v = O.verify(vk, cipher, proofs[196 * k:196 * k + 196], cts[64 * k:64 * k + 64] + rec[32:44] + rec[44:48][::-1] + rec[48:])
        print(name, "batch %d ok=%d %.3fs -> %.1f proofs/s verifies=%s" % (n, ok, dt, n / dt, v), g.last_stage_ms(algo), flush=True)
