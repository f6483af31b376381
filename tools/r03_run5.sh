#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $O/trace_b64 -o run --output-format csv -- python3 bench.py --batch 64 --callers 6 --steps 40 --warmup 6 --no-cpu-baseline --verify 0 > $O/trace_b64.json 2> $O/trace_b64.err && python tools/trace_concurrency.py $(ls $O/trace_b64/*kernel_trace.csv | head -1) | tee $O/trace_b64_concurrency.txt
GSC_TRACE_HOST=1 python bench.py --batch 64 --callers 6 --steps 12 --warmup 6 --no-cpu-baseline --verify 0 2>&1 >/dev/null | grep -E "prove_chunk|gsc_prove_raw" | tail -12
rm -f $O/trace_b64/*kernel_trace.csv
