#!/bin/bash
# Round 4, final tree: full GPU suite, every BASELINE config's bench line, the latency / caller measurements.  Outputs: gpurun_out/r04final/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04final; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; rc=$?; tail -4 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cut -c1-140 $O/bench_chacha20.json
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err && cut -c1-140 $O/bench_$w.json; done
python bench.py --library-defaults --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_library_defaults.json 2> $O/bench_chacha20_library_defaults.err && cut -c1-140 $O/bench_chacha20_library_defaults.json
for b in 64 256 512 1024; do python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && cut -c1-140 $O/bench_chacha20_b$b.json; done
python bench.py --batch 1 --callers 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_chacha20_b1.json 2> $O/bench_chacha20_b1.err && cut -c1-140 $O/bench_chacha20_b1.json
python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_forcedist.json 2> $O/bench_chacha20_forcedist.err && cut -c1-140 $O/bench_chacha20_forcedist.json
python bench.py --gpus 2 --in-library --devices 0,0 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_inlibrary_2x4096.json 2> $O/bench_chacha20_inlibrary_2x4096.err && cut -c1-140 $O/bench_chacha20_inlibrary_2x4096.json
SECS=3 CALLERS="1 8 64 256 512" bash tools/r03_prove_callers_c.sh > $O/callers.txt 2>&1; tail -5 $O/callers.txt
