#!/bin/bash
# Round 4: selected GPU tests (arguments: pytest -k expression), then the default bench line.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "$1" > $O/pytest_sel.txt 2>&1; rc=$?; tail -12 $O/pytest_sel.txt; [ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 3 > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cat $O/bench_chacha20.json
