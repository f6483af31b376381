#!/bin/bash
# Round 4: full GPU suite, then the latency / mid-size / many-caller measurements of the small-integer witness path.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; rc=$?; tail -12 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit $rc
for b in 1 16 32; do python bench.py --batch $b --callers 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${b}_c1.json 2> $O/bench_b${b}_c1.err && cut -c1-160 $O/bench_b${b}_c1.json; done
GSC_SMALL_WITNESS_FEW=0 python bench.py --batch 1 --callers 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b1_c1_resident.json 2> $O/bench_b1_c1_resident.err && cut -c1-160 $O/bench_b1_c1_resident.json
python bench.py --batch 64 --callers 6 --steps 48 --warmup 6 --no-cpu-baseline > $O/bench_b64_c6.json 2> $O/bench_b64_c6.err && cut -c1-160 $O/bench_b64_c6.json
GSC_ENABLE_TEST_HOOKS=1 GSC_WIN_SLICE=512 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_slice512.json 2> $O/bench_slice512.err && cut -c1-160 $O/bench_slice512.json
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_slice256.json 2> $O/bench_slice256.err && cut -c1-160 $O/bench_slice256.json
SECS=3 CALLERS="1 8 64 256" bash tools/r03_prove_callers_c.sh > $O/callers.txt 2>&1; tail -8 $O/callers.txt
