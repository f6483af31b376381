#!/bin/bash
# rocprofv3 kernel summary of the default bench command (round 4).  Output: gpurun_out/r04s/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok"
rm -rf $O/stats/*kernel_trace.csv 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -30 $O/kernel_stats.csv | cut -c1-170
