#!/bin/bash
# Round 4, first GPU run of the small-integer witness path: the GPU suite, then bench lines.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; rc=$?; tail -15 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 3 > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench ok" && cat $O/bench_chacha20.json &&
GSC_SMALL_WITNESS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_chacha20_generic.json 2> $O/bench_chacha20_generic.err && cat $O/bench_chacha20_generic.json &&
for b in 64 256 1024; do python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && cat $O/bench_chacha20_b$b.json; done
