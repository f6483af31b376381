"""Soak of the latency path: thousands of small ProveBatch calls of random sizes, every proof checked with the drop-in verifier
(ChaCha20 under the reference's vk; AES-128 under the vk of the product's Setup).  Prints one summary line per cipher."""
import base64, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
g = gsc_loader.load()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 600


def soak(algo, cipher, keylen, big_endian_counter, sizes, ncalls):
    t0 = time.time(); total = 0; bad = 0
    pool = ThreadPoolExecutor(16)
    for c in range(ncalls):
        n = rnd.choice(sizes)
        recs = b"".join(rnd.randbytes(keylen) + bytes(32 - keylen) + rnd.randbytes(12) + rnd.getrandbits(30).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
        ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
        assert ok == n, (c, n, ok)

        def check(k):
            rec = recs[112 * k:112 * (k + 1)]
            ctr = int.from_bytes(rec[44:48], "little").to_bytes(4, "big" if big_endian_counter else "little")
            sig = cts[64 * k:64 * k + 64] + rec[32:44] + ctr + rec[48:]
            return g.verify({"cipher": cipher, "proof": base64.b64encode(proofs[196 * k:196 * k + lens[k]]).decode(), "publicSignals": base64.b64encode(sig).decode()})
        res = list(pool.map(check, range(n)))
        bad += res.count(False); total += n
    print("%s: %d calls, %d proofs, %d rejected, %.1f s" % (cipher, ncalls, total, bad, time.time() - t0), flush=True)
    return bad


assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
r1cs = bench.golden("r1cs.aes128"); pk, vk = g.setup(r1cs)
assert g.init_algorithm(1, pk, r1cs) and g.init_verifier(1, vk)
jobs = [(0, "chacha20", 32, False, [1, 1, 1, 2, 3, 5, 8, 13, 21, 32, 33], calls), (1, "aes-128-ctr", 16, True, [1, 1, 2, 3, 7, 20, 21], max(calls // 6, 10))]
if len(sys.argv) > 3 and sys.argv[3] == "concurrent":      # both ciphers at once, two callers each: the resident kernels of different algorithms and lanes take turns
    import threading
    out = []
    rnds = [random.Random(1000 + i) for i in range(4)]

    def run(j, r):
        global rnd
        out.append(soak(*j))
    ts = [threading.Thread(target=run, args=(jobs[i % 2], rnds[i])) for i in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    bad = sum(out)
else:
    bad = sum(soak(*j) for j in jobs)
sys.exit(1 if bad else 0)
