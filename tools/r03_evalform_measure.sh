#!/bin/bash
# Measurement batch after the evaluation-form quotient (run through gpurun): bench lines of the BASELINE configs and the rocprofv3
# kernel summary of the default bench command.  Outputs under gpurun_out/r03e/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench chacha20 ok" &&
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err && echo "bench $w ok"; done &&
for b in 64 256 1024; do python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && echo "bench b$b ok"; done &&
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok" &&
GSC_QUOTIENT_EVAL=0 python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_aes128_coeff.json 2> $O/bench_aes128_coeff.err && echo "aes128 coefficient form ok"
rm -rf $O/stats/*kernel_trace.csv 2>/dev/null
ls $O $O/stats
