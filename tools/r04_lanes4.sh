#!/bin/bash
# AES-128: one full lane + a small one against two full lanes (round 4).  Output: gpurun_out/r04ln4/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ln4; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs python bench.py "$@" --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json "$tag ($envs $*)" || { echo "$tag failed"; tail -2 $O/$tag.err; }; }
for rep in 1 2; do
for cfgs in "GSC_LANES=2 GSC_SMALL_LANES=0" "GSC_LANES=1 GSC_SMALL_LANES=0" "GSC_LANES=1 GSC_SMALL_LANES=1 GSC_SMALL_LANE_CAP=512" "GSC_LANES=1 GSC_SMALL_LANES=2 GSC_SMALL_LANE_CAP=256"; do
  t=$(echo $cfgs | tr -d ' =_A-Z')
  run aes1024_$t $cfgs --workload aes128 --steps 5 --warmup 1
  run aes256x3_$t $cfgs --workload aes128 --batch 256 --callers 3 --steps 8 --warmup 2
  run aes64x4_$t $cfgs --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2
done; done
