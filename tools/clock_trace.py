"""Shader clock over one timed step of the default bench configuration (test hooks): a resident one-wave sampler beside the call.
Prints the mean clock inside each stage window of the step (witness | quotient | msm), located with the engine's own stage times.
usage: python tools/clock_trace.py [batch]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GSC_ENABLE_TEST_HOOKS"] = "1"
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for k, v in bench.engine_env("chacha20", B).items():
    os.environ.setdefault(k, v)
import gsc_loader
g = gsc_loader.load()
assert g.init_algorithm(0, bench.golden("pk.chacha20"), bench.golden("r1cs.chacha20"))
recs = bench.xoshiro_records(B, 1 << 20)
for _ in range(2):
    assert g.prove_raw(0, recs, B)[0] == B
samples = []
N, IV = 1200, 500                                   # 0.6 s of samples, 0.5 ms apart
th = threading.Thread(target=lambda: samples.extend(g.debug_clock_trace(N, IV)))
th.start(); time.sleep(0.05)
t0 = time.time(); assert g.prove_raw(0, recs, B)[0] == B; t1 = time.time()
th.join()
st = g.last_stage_ms(0)
print("step %.1f ms, stages %s" % (1e3 * (t1 - t0), {k: round(v, 1) for k, v in st.items()}))
busy = [(t, mhz) for t, mhz in samples]
# the call starts ~50 ms into the trace; find the first sample after which the clock departs from idle by looking at the stage lengths from the END of the call
mhz = [m for _, m in busy]
print("clock MHz: min %.0f max %.0f" % (min(mhz), max(mhz)))
step = 20
for i in range(0, len(busy), step):
    seg = busy[i:i + step]
    print("t=%6.1f ms  %7.1f MHz" % (1e3 * seg[0][0], sum(m for _, m in seg) / len(seg)))
