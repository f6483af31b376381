#!/bin/bash
# Round 4: full GPU suite on the final tree, then latency and headline lines.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04f; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; rc=$?; tail -12 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit $rc
python bench.py --batch 1 --callers 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_b1_c1.json 2> $O/bench_b1_c1.err && cut -c1-150 $O/bench_b1_c1.json
python bench.py --workload aes128 --batch 1 --callers 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_aes128_b1_c1.json 2> $O/bench_aes128_b1_c1.err && cut -c1-150 $O/bench_aes128_b1_c1.json
python bench.py --steps 10 --warmup 3 > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cut -c1-150 $O/bench_chacha20.json
SECS=3 CALLERS="1 64" bash tools/r03_prove_callers_c.sh > $O/callers.txt 2>&1; tail -3 $O/callers.txt
