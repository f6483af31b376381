#!/bin/bash
# Lane counts again, now that lanes no longer share hardware queues (round 4).  Output: gpurun_out/r04ln/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ln; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs python bench.py "$@" --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json "$tag ($envs $*)" || { echo "$tag failed"; tail -2 $O/$tag.err; }; }
for rep in 1 2; do
run b8192_l1 GSC_LANES=1 --steps 6 --warmup 2
run b8192_l2 GSC_LANES=2 --steps 6 --warmup 2
run b8192_l2c3 GSC_LANES=2 --steps 6 --warmup 2 --callers 3
run b1024_l1 GSC_LANES=1 --batch 1024 --steps 24 --warmup 4
run b1024_l2 GSC_LANES=2 --batch 1024 --steps 24 --warmup 4
run b1024_l2c3 GSC_LANES=2 --batch 1024 --steps 24 --warmup 4 --callers 3
run b64c6_s2 GSC_SMALL_LANES=2 --batch 64 --callers 6 --steps 24 --warmup 4
run b64c6_s4 GSC_SMALL_LANES=4 --batch 64 --callers 6 --steps 24 --warmup 4
run b256c4_s2 GSC_SMALL_LANES=2 --batch 256 --callers 4 --steps 24 --warmup 4
run b256c4_s4 GSC_SMALL_LANES=4 --batch 256 --callers 4 --steps 24 --warmup 4
done
