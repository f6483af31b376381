#!/bin/bash
# Round-3 third GPU pass (same box for every A/B): box reference, heavy chain on/off for AES-128 and the mixed batch, batch sweep with 2 and 4 callers.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()}, d['verified'])" $1; }
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/ref_chacha20.json 2> $O/ref_chacha20.err && show $O/ref_chacha20.json


for nc in 2 4; do for b in 64 128 256 512 1024; do python bench.py --batch $b --callers $nc --steps 24 --warmup 4 --no-cpu-baseline > $O/sweep_b${b}_c$nc.json 2> $O/sweep_b${b}_c$nc.err && show $O/sweep_b${b}_c$nc.json; done; done
