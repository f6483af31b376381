#!/bin/bash
# rocprofv3 kernel summaries of the AES-128 and mixed bench commands (round 3).  Outputs under gpurun_out/r03aes/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03aes; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/aes128 -o run --output-format csv -- python3 bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/aes128_bench.json 2> $O/aes128.err && echo "aes128 ok" &&
rocprofv3 --kernel-trace --stats -d $O/mixed -o run --output-format csv -- python3 bench.py --workload mixed --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/mixed_bench.json 2> $O/mixed.err && echo "mixed ok"
rm -rf $O/*/*kernel_trace.csv 2>/dev/null
ls $O $O/aes128
