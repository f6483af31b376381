#!/bin/bash
# Which priority level for which of a lane's three streams (GSC_STREAM_PRIORITIES codes, test hook: digits main / side / third, 1 high 2 normal 3 low; 0 = plain streams).  Output: gpurun_out/r04pr/
set -o pipefail
export PYTHONUNBUFFERED=1 GSC_ENABLE_TEST_HOOKS=1
O=gpurun_out/r04pr; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in (d.get('stage_ms_last_step') or {}).items()})"; }
run() { tag=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs python bench.py "$@" --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json "$tag ($envs $*)" || { echo "$tag failed"; tail -2 $O/$tag.err; }; }
for rep in 1 2; do for c in ${CODES:-0 213 212 223 211}; do
  run aesb1_$c GSC_STREAM_PRIORITIES=$c --workload aes128 --batch 1 --callers 1 --steps 40 --warmup 5
  run chab1_$c GSC_STREAM_PRIORITIES=$c --batch 1 --callers 1 --steps 40 --warmup 5
  run aes64x4_$c GSC_STREAM_PRIORITIES=$c --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2
  run aes1024_$c GSC_STREAM_PRIORITIES=$c --workload aes128 --steps 5 --warmup 1
  run b64_$c GSC_STREAM_PRIORITIES=$c --batch 64 --steps 24 --warmup 4
  run b256_$c GSC_STREAM_PRIORITIES=$c --batch 256 --steps 24 --warmup 4
  run b1024_$c GSC_STREAM_PRIORITIES=$c GSC_MAX_BATCH=8192 --batch 1024 --steps 24 --warmup 4
done; done
