#!/bin/bash
# Round 4: the plain-integer first transform stages: parity, then a same-box A/B of the headline line.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "small_integer or kat or timed_configuration or ragged or repeated or engine_options_do_not_change_the_proofs or a_small_batch" > $O/pytest_sel.txt 2>&1; rc=$?; tail -6 $O/pytest_sel.txt; [ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do GSC_NTT_PLAIN=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_plain$v.json 2> $O/bench_plain$v.err && python - $v $O/bench_plain$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); print("GSC_NTT_PLAIN=%s: %.1f proofs/s, %.2f ms per step, stages %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_last_step"].items()}))
PY
done
