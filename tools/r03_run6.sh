#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()}, d['verified'])" $1; }
for pr in 1 0; do for sl in 2 4; do for b in 64 256; do
  GSC_EXP_SIDE_PRIORITY=$pr GSC_SMALL_LANES=$sl python bench.py --batch $b --callers 6 --steps 30 --warmup 6 --no-cpu-baseline > $O/pr${pr}_sl${sl}_b$b.json 2> $O/pr${pr}_sl${sl}_b$b.err && show $O/pr${pr}_sl${sl}_b$b.json
done; done; done
for pr in 1 0; do GSC_EXP_SIDE_PRIORITY=$pr python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pr${pr}_8192.json 2> $O/pr${pr}_8192.err && show $O/pr${pr}_8192.json; done
GSC_EXP_SIDE_PRIORITY=1 rocprofv3 --kernel-trace -d $O/trace_b64p -o run --output-format csv -- python3 bench.py --batch 64 --callers 6 --steps 40 --warmup 6 --no-cpu-baseline --verify 0 > $O/trace_b64p.json 2> $O/trace_b64p.err && python tools/trace_concurrency.py $(ls $O/trace_b64p/*kernel_trace.csv | head -1) | tee $O/trace_b64p_concurrency.txt
rm -f $O/trace_b64p/*kernel_trace.csv
