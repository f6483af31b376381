"""Latency of one ProveBatch call against the number of statements (the latency kernels serve up to 8, the batch kernels the rest)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
rnd = random.Random(1)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 9, 16, 32, 64]:
    recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    best = 1e9
    for _ in range(5):
        t = time.time(); ok, proofs, lens, cts = g.prove_raw(0, recs, n); dt = time.time() - t
        assert ok == n
        best = min(best, dt)
    print("n = %3d  %.2f ms  %s" % (n, best * 1e3, {k: round(v, 2) for k, v in g.last_stage_ms(0).items()}), flush=True)
