#!/bin/bash
# Round 4: compile-time window offsets in the last transform kernel's digit writer: bench-configuration parity, kernel time under rocprofv3, headline line.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04i; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "timed_configuration or kat or small_integer or compute_h or compute_d or aes" > $O/pytest_sel.txt 2>&1; rc=$?; tail -5 $O/pytest_sel.txt; [ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok"
rm -rf $O/stats/*kernel_trace.csv 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
grep -E "k_ntt|k_msm_win<bn254::Fp29f, true>" $O/kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
python bench.py --steps 10 --warmup 3 > $O/bench_chacha20.json 2> $O/bench_chacha20.err && python - <<'PY'
import json
d = json.load(open("gpurun_out/r04i/bench_chacha20.json")); print(d["value"], d["ms_per_step"], d["stage_ms_last_step"], d["roofline_valu"]["frac"], d["roofline_valu"]["clock_mhz"])
PY
