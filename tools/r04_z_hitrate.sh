#!/bin/bash
# Round 4: what would a higher L2 hit rate of the Z kernel buy?  The same launch with every gather confined to the first 2^bits entries of
# its row (a test hook: WRONG sums, timing only): same instructions, same digit stream, the hit rate of rows of 2^bits entries.  Prints launch ms and live clock.
set -o pipefail
export PYTHONUNBUFFERED=1 GSC_ENABLE_TEST_HOOKS=1
O=gpurun_out/r04z; mkdir -p $O
for b in 0 15 14 13 12 10 4; do
  GSC_Z_EXP_ENTRY_BITS=$b python bench.py --steps 4 --warmup 1 --no-cpu-baseline --verify 0 > $O/bench_bits$b.json 2> $O/bench_bits$b.err || { tail -3 $O/bench_bits$b.err; exit 1; }
  python - $b $O/bench_bits$b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); rv = d.get("roofline_valu", {})
print("entry bits %2s (0 = all 2^16): Z kernel %7.2f ms  clock %6.1f MHz  cycles/instr %.3f  step %7.2f ms" % (sys.argv[1], d["roofline"]["launch_ms"], rv.get("clock_mhz", 0), rv.get("cycles_per_wave_instr", 0), d["ms_per_step"]))
PY
done
