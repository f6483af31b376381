#!/bin/bash
# Round 4, final tree: default bench line (with the CPU baseline), its rocprofv3 kernel summary, AES-128 / mixed / 1024 lines again (stage accounting changed).  Outputs: gpurun_out/r04final2/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04final2; mkdir -p $O
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cut -c1-140 $O/bench_chacha20.json
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err && cut -c1-140 $O/bench_$w.json; done
python bench.py --batch 1024 --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_chacha20_b1024.json 2> $O/bench_chacha20_b1024.err && cut -c1-140 $O/bench_chacha20_b1024.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok"
rm -rf $O/stats/*kernel_trace.csv 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-150
