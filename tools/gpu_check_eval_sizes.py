"""Evaluation-form quotient at the batch sizes between the latency path and a full wave: every proof verified with libverify.so."""
import base64, json, lzma, os, random, sys, threading, time
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
for k, v in (("GSC_Z_TABLE_GB", "24"), ("GSC_W_TABLE_GB", "8"), ("GSC_FEW_Z_GB", "9")):
    os.environ.setdefault(k, v)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
g = gsc_loader.load()
G = os.path.join(ROOT, "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "aes128"
algo, cipher, keylen = {"chacha20": (0, "chacha20", 32), "aes128": (1, "aes-128-ctr", 16), "aes256": (2, "aes-256-ctr", 32)}[name]
cs = lzma.open(os.path.join(G, "r1cs.%s.xz" % name)).read()
if algo:
    pk = open(os.path.join(ROOT, "build", "keys", "pk." + name), "rb").read(); vk = open(os.path.join(ROOT, "build", "keys", "vk." + name), "rb").read()
else:
    pk = open(os.path.join(G, "pk.chacha20"), "rb").read(); vk = open(os.path.join(G, "vk.chacha20"), "rb").read()
assert g.init_algorithm(algo, pk, cs)
assert g.init_verifier(algo, vk)
print(g.describe(algo), flush=True)
rnd = random.Random(11)
def verify(rec, proof, ct):
    ctr = rec[44:48] if algo == 0 else rec[44:48][::-1]
    sig = ct + rec[32:44] + ctr + rec[48:112]
    return g.verify({"cipher": cipher, "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()})
plen = 164 if algo == 0 else 196
bad_total = 0
for n in [int(a) for a in sys.argv[2:]] or (1, 5, 21, 30, 40, 64, 65, 130):
    recs = b"".join(rnd.randbytes(keylen) + bytes(32 - keylen) + rnd.randbytes(12) + rnd.getrandbits(30).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    ok, proofs, lens, cts = g.prove_raw(algo, recs, n)
    bad = [k for k in range(n) if not verify(recs[112 * k:112 * k + 112], proofs[196 * k:196 * k + plen], cts[64 * k:64 * k + 64])]
    print("n=%d ok=%d rejected=%s" % (n, ok, bad[:12]), flush=True)
    bad_total += len(bad)
# concurrent single callers (the micro-batcher forms device batches of whatever is queued)
N = 60
reqs = [{"cipher": cipher, "key": list(rnd.randbytes(keylen)), "nonce": list(rnd.randbytes(12)), "counter": rnd.getrandbits(30), "input": list(rnd.randbytes(64))} for _ in range(N)]
outs = [None] * N
def one(i): outs[i] = json.loads(g.prove(reqs[i]))
for rep in range(3):
    th = [threading.Thread(target=one, args=(i,)) for i in range(N)]
    for t in th: t.start()
    for t in th: t.join()
    bad = []
    for i, (q, o) in enumerate(zip(reqs, outs)):
        proof = base64.b64decode(o["proof"]["proofJson"]); ct = base64.b64decode(o["publicSignals"])
        ctr = q["counter"].to_bytes(4, "little" if algo == 0 else "big")
        sig = ct + bytes(q["nonce"]) + ctr + bytes(q["input"])
        if not g.verify({"cipher": cipher, "proof": base64.b64encode(proof).decode(), "publicSignals": base64.b64encode(sig).decode()}): bad.append(i)
    print("concurrent round %d: rejected %s" % (rep, bad), flush=True)
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
