#!/bin/bash
# Is the mid-size-batch bound on the host (HIP launch path shared by the caller threads of ONE process) or on the device?  Same total
# concurrency as one process with six callers, but as three processes with two callers each (own HIP runtime, own tables: small ones).
set -o pipefail
export PYTHONUNBUFFERED=1 GSC_WINDOW_Z=12 GSC_FEW_Z_GB=0
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()})" $1; }
python bench.py --batch 64 --callers 6 --steps 60 --warmup 6 --no-cpu-baseline --verify 0 > $O/one_proc.json 2> $O/one_proc.err && show $O/one_proc.json
for i in 1 2 3; do python bench.py --batch 64 --callers 2 --steps 60 --warmup 6 --no-cpu-baseline --verify 0 > $O/three_proc_$i.json 2> $O/three_proc_$i.err & done; wait
for i in 1 2 3; do show $O/three_proc_$i.json; done
