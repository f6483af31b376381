"""AES-V2 end to end on a GPU box: keys from the oracle's Setup (no reference AES pk ships), device proof vs oracle proof."""
import base64, json, lzma, os, sys, time, random
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader
from oracle import oracle as O
g = gsc_loader.load()
G = os.path.join(ROOT, "tests", "golden")
for algo, name, cipher, kl in ((1, "aes128", "aes-128-ctr", 16), (2, "aes256", "aes-256-ctr", 32)):
    r1cs = lzma.open(os.path.join(G, "r1cs.%s.xz" % name)).read()
    ocs = O.R1CS(r1cs)
    t = time.time(); pkb, vkb = O.setup(ocs, bytes([algo] * 32)); print(name, "oracle setup %.1fs pk=%d B" % (time.time() - t, len(pkb)), flush=True)
    t = time.time(); ok = g.init_algorithm(algo, pkb, r1cs); print("InitAlgorithm", ok, "%.1fs" % (time.time() - t), g.describe(algo), flush=True)
    if not ok: sys.exit(1)
    opk, ovk = O.ProvingKey(pkb), O.VerifyingKey(vkb)
    rnd = random.Random(algo)
    key, nonce, pt, ctr = rnd.randbytes(kl), rnd.randbytes(12), rnd.randbytes(64), rnd.getrandbits(31)
    r, s, mask = rnd.getrandbits(250), rnd.getrandbits(250), rnd.getrandbits(250)
    g.set_deterministic_randomness(r, s, mask)
    t = time.time(); out = json.loads(g.prove({"cipher": cipher, "key": list(key), "nonce": list(nonce), "counter": ctr, "input": list(pt)})); dt = time.time() - t
    g.set_deterministic_randomness(None)
    if "proof" not in out: print("FAILED:", out); sys.exit(1)
    proof = base64.b64decode(out["proof"]["proofJson"]); ct = base64.b64decode(out["publicSignals"])
    want, want_ct = O.prove(ocs, opk, cipher, key, nonce, ctr, pt, r, s, mask)
    sig = ct + nonce + ctr.to_bytes(4, "big") + pt
    print(name, "prove %.2fs len=%d ct_ok=%s bit_exact=%s verifies=%s" % (dt, len(proof), ct == want_ct, proof == want, O.verify(ovk, cipher, proof, sig)), g.last_stage_ms(algo), flush=True)
    if proof != want: print(proof.hex()); print(want.hex())
    n = 256
    recs = b"".join(rnd.randbytes(32) + rnd.randbytes(12) + rnd.getrandbits(31).to_bytes(4, "little") + rnd.randbytes(64) for _ in range(n))
    t = time.time(); okn, proofs, lens, cts = g.prove_raw(algo, recs, n); dt = time.time() - t
    k = n - 1; rec = recs[112 * k:112 * (k + 1)]
    sig = cts[64 * k:64 * k + 64] + rec[32:44] + rec[44:48][::-1] + rec[48:]
    print(name, "batch %d ok=%d %.2fs -> %.1f proofs/s, last verifies=%s" % (n, okn, dt, n / dt, O.verify(ovk, cipher, proofs[196 * k:196 * k + 196], sig)), g.last_stage_ms(algo), flush=True)
    # unprovable statement: counter too close to 2^32 (circuits/aesV2/aes128.go:41-53)
    bad = json.loads(g.prove({"cipher": cipher, "key": list(key), "nonce": list(nonce), "counter": 0xFFFFFFFE, "input": list(pt)}))
    print(name, "counter overflow ->", bad)
