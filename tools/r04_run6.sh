#!/bin/bash
# Round 4: the chain kernel with LDS forwarding: witness / latency / bench-configuration tests, then latency lines.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "small_integer or soak or kat or secrets or resident or latency_path or bench_bookkeeping or a_small_batch or timed_configuration or ragged or repeated" > $O/pytest_sel.txt 2>&1; rc=$?; tail -6 $O/pytest_sel.txt; [ $rc -eq 0 ] || exit $rc
python bench.py --batch 1 --callers 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_b1_c1.json 2> $O/bench_b1_c1.err && cut -c1-150 $O/bench_b1_c1.json
python bench.py --batch 64 --callers 2 --steps 48 --warmup 6 --no-cpu-baseline > $O/bench_b64.json 2> $O/bench_b64.err && cut -c1-150 $O/bench_b64.json
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cut -c1-150 $O/bench_chacha20.json
python - <<'PY'
import json
for f in ("bench_b1_c1", "bench_b64", "bench_chacha20"):
    d = json.load(open("gpurun_out/r04g/%s.json" % f)); print(f, d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms_last_step"].items()})
PY
SECS=3 CALLERS="1 64" bash tools/r03_prove_callers_c.sh > $O/callers.txt 2>&1; tail -2 $O/callers.txt
