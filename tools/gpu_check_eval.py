"""Evaluation-form quotient on a GPU box: a batch (evaluation form) against a single Prove (coefficient form, latency path) and the oracle."""
import base64, json, lzma, os, random, sys, time
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
from oracle import oracle as O
g = gsc_loader.load()
G = os.path.join(ROOT, "tests", "golden")
algo = sys.argv[1] if len(sys.argv) > 1 else "chacha20"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
assert algo == "chacha20"
pk = open(os.path.join(G, "pk.chacha20"), "rb").read()
cs = lzma.open(os.path.join(G, "r1cs.chacha20.xz")).read()
t = time.time()
ok = g.init_algorithm(0, pk, cs)
print("InitAlgorithm:", ok, "%.2fs" % (time.time() - t), flush=True)
if not ok:
    sys.exit(1)
print(g.describe(0), flush=True)
ocs, opk = O.R1CS(cs), O.ProvingKey(pk)
rnd = random.Random(7)
recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
r, s = rnd.getrandbits(250), rnd.getrandbits(250)
g.set_deterministic_randomness(r, s, 0)
okn, proofs, lens, cts = g.prove_raw(0, recs, n)
print("batch ok", okn, "of", n, g.last_stage_ms(0), flush=True)
bad = 0
for k in (0, 1, 63, 64, n - 1):
    rec = recs[112 * k:112 * (k + 1)]
    want, _ = O.prove(ocs, opk, "chacha20", rec[:32], rec[32:44], int.from_bytes(rec[44:48], "little"), rec[48:], r, s)
    same = proofs[196 * k:196 * k + 164] == want
    p = {"cipher": "chacha20", "key": list(rec[:32]), "nonce": list(rec[32:44]), "counter": int.from_bytes(rec[44:48], "little"), "input": list(rec[48:])}
    single = base64.b64decode(json.loads(g.prove(p))["proof"]["proofJson"])
    print("proof", k, "batch == oracle:", same, " single == oracle:", single == want, flush=True)
    if not same:
        bad += 1
        print(" batch ", proofs[196 * k:196 * k + 164].hex()); print(" oracle", want.hex())
g.set_deterministic_randomness(None)
for m in (1024, 1024):
    recs = b"".join(rnd.randbytes(44) + rnd.getrandbits(32).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(m))
    t = time.time(); okn, proofs, lens, cts = g.prove_raw(0, recs, m); dt = time.time() - t
    print("batch %d: ok=%d %.3fs -> %.1f proofs/s" % (m, okn, dt, m / dt), g.last_stage_ms(0), flush=True)
sys.exit(1 if bad else 0)
