"""Sustained throughput of the reference's entry point — one statement per `Prove` call (libprove.go:30-47) — under C concurrent callers
(closed loop: every caller issues its next call when the previous one returns; libraries/core_test.go:44-111 is the pattern).  The
library's micro-batcher gathers concurrent callers into device batches.  Usage: prove_callers.py [seconds per point] [callers ...]; GSC_TOOL_CIPHER=aes128 for AES-128-V2 (keys from the product's Setup)."""
import base64, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsc_loader, bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
points = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]
g = gsc_loader.load()
name = os.environ.get("GSC_TOOL_CIPHER", "chacha20")
algo, cipher, klen = {"chacha20": (0, "chacha20", 32), "aes128": (1, "aes-128-ctr", 16)}[name]
if name == "chacha20":
    assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20")) and g.init_verifier(0, bench.golden("vk.chacha20"))
else:
    r1cs = bench.golden("r1cs." + name); pk, vk = g.setup(r1cs)
    assert g.init_algorithm(algo, pk, r1cs) and g.init_verifier(algo, vk)
print(g.describe(algo), flush=True)
recs = bench.provable(bench.xoshiro_records(1024, 0xCA11 << 20), name)
def request(k):
    r = recs[112 * k:112 * (k + 1)]
    return json.dumps({"cipher": cipher, "key": base64.b64encode(r[:klen]).decode(), "nonce": base64.b64encode(r[32:44]).decode(),
                       "counter": int.from_bytes(r[44:48], "little"), "input": base64.b64encode(r[48:]).decode()}).encode()
reqs = [request(k) for k in range(1024)]
for _ in range(3): assert b'"proof"' in g.prove(reqs[0])
for C in points:
    stop = time.time() + secs; counts = [0] * C; lat = [0.0] * C; last = [None] * C
    def work(i):
        k = i
        while time.time() < stop:
            t = time.time(); out = g.prove(reqs[k % 1024]); lat[i] += time.time() - t
            if b'"proof"' not in out: raise SystemExit("Prove failed: %r" % out[:200])
            counts[i] += 1; last[i] = (k % 1024, out); k += C
    th = [threading.Thread(target=work, args=(i,)) for i in range(C)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    el = time.time() - t0; n = sum(counts)
    ok = 0
    for i in range(min(C, 16)):      # the last answer of up to 16 callers through the product's verifier
        k, out = last[i]; o = json.loads(out); r = recs[112 * k:112 * (k + 1)]; ct = base64.b64decode(o["publicSignals"])
        ok += bool(g.verify({"cipher": cipher, "proof": o["proof"]["proofJson"], "publicSignals": base64.b64encode(bench.signals_of(name, r, ct)).decode()}))
    print("callers %4d  %8.1f proofs/s  mean latency %7.2f ms  (%d calls in %.2f s; verified %d/%d)" % (C, n / el, 1e3 * sum(lat) / max(n, 1), n, el, ok, min(C, 16)), flush=True)
    if ok != min(C, 16): raise SystemExit("REJECTED")
