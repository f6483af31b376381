"""Mixed ChaCha20 / AES-128 / AES-256 batch through ProveBatch (JSON in, JSON out), all three algorithms resident on one GPU with the
default table budgets — BASELINE.json configs[4] ("mixed ChaCha/AES batch to stress scheduler").  Statement i uses cipher i mod 3."""
import base64, json, lzma, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
from oracle import oracle as O
g = gsc_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
G = os.path.join(ROOT, "tests", "golden")
ALGOS = [(0, "chacha20", 32), (1, "aes-128-ctr", 16), (2, "aes-256-ctr", 32)]
vks = {}
for algo, cipher, kl in ALGOS:
    name = ["chacha20", "aes128", "aes256"][algo]
    r1cs = lzma.open(os.path.join(G, "r1cs.%s.xz" % name)).read()
    if algo == 0:
        pk, vk = open(os.path.join(G, "pk.chacha20"), "rb").read(), open(os.path.join(G, "vk.chacha20"), "rb").read()
    else:
        pk, vk = O.setup(O.R1CS(r1cs), bytes([algo] * 32))
    t = time.time(); assert g.init_algorithm(algo, pk, r1cs); print(cipher, "init %.1fs" % (time.time() - t), g.describe(algo), flush=True)
    vks[cipher] = O.VerifyingKey(vk)
rnd = random.Random(5)
for rep in range(3):
    params = []
    for i in range(n):
        algo, cipher, kl = ALGOS[i % 3]
        params.append({"cipher": cipher, "key": base64.b64encode(rnd.randbytes(kl)).decode(), "nonce": base64.b64encode(rnd.randbytes(12)).decode(),
                       "counter": rnd.getrandbits(30), "input": base64.b64encode(rnd.randbytes(64)).decode()})
    t = time.time(); outs = g.prove_batch(params); dt = time.time() - t
    good = sum(1 for o in outs if isinstance(o, dict) and "proof" in o)
    checks = []
    for i in (0, 1, 2, n - 3, n - 2, n - 1):
        p, o = params[i], outs[i]
        ctr = p["counter"].to_bytes(4, "little" if p["cipher"] == "chacha20" else "big")
        sig = base64.b64decode(o["publicSignals"]) + base64.b64decode(p["nonce"]) + ctr + base64.b64decode(p["input"])
        checks.append(O.verify(vks[p["cipher"]], p["cipher"], base64.b64decode(o["proof"]["proofJson"]), sig))
    print("mixed batch %d (JSON): ok=%d %.3fs -> %.1f proofs/s, sample verifies=%s" % (n, good, dt, n / dt, all(checks)), flush=True)
