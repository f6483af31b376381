#!/bin/bash
# Round-3 fourth GPU pass: hardware queues x small lanes for mid-size batches; in-library two-replica rehearsal.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()}, d['verified'])" $1; }
for q in 4 8 16; do for sl in 2 4; do for b in 64 256; do
  GPU_MAX_HW_QUEUES=$q GSC_SMALL_LANES=$sl python bench.py --batch $b --callers 6 --steps 30 --warmup 6 --no-cpu-baseline > $O/q${q}_sl${sl}_b$b.json 2> $O/q${q}_sl${sl}_b$b.err && show $O/q${q}_sl${sl}_b$b.json
done; done; done
python bench.py --gpus 2 --in-library --devices 0,0 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline > $O/inlib_2x4096.json 2> $O/inlib_2x4096.err && show $O/inlib_2x4096.json
python bench.py --gpus 2 --in-library --devices 0,0 --batch 64 --callers 4 --steps 30 --warmup 6 --no-cpu-baseline > $O/inlib_2x64.json 2> $O/inlib_2x64.err && show $O/inlib_2x64.json
