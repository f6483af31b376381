#!/bin/bash
# Second measurement batch of the evaluation-form tree (run through gpurun): GPU test suite, rocprofv3 kernel summaries of the default
# bench command and of the AES-128 one, final bench lines.  Outputs under gpurun_out/r03g/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench chacha20 ok" &&
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok" &&
rocprofv3 --kernel-trace --stats -d $O/stats_aes -o run --output-format csv -- python3 bench.py --workload aes128 --steps 4 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_aes_bench.json 2> $O/stats_aes.err && echo "stats aes ok" &&
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err && echo "bench $w ok"; done
rm -rf $O/stats/*kernel_trace.csv $O/stats_aes/*kernel_trace.csv 2>/dev/null
ls $O
