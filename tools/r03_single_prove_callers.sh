#!/bin/bash
# Concurrent single-statement callers (the reference's unit of work: one Prove per FFI call) against the micro-batcher: proofs/s and ms per call.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03callers; mkdir -p $O
for c in 1 4 16 64 128; do
  s=$((c * 40)); [ $s -lt 200 ] && s=200
  timeout -k 10 300 python bench.py --batch 1 --callers $c --steps $s --warmup $((c * 2)) --no-cpu-baseline --verify 16 > $O/b1_c$c.json 2> $O/b1_c$c.err || { echo "callers $c failed"; tail -5 $O/b1_c$c.err; exit 1; }
  python - $O/b1_c$c.json $c <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("callers %4s  %9.1f proofs/s  %7.3f ms per step  verified %s  kernel %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["verified"], d["roofline"]["kernel"][:40]), flush=True)
PY
done
