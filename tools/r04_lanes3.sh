#!/bin/bash
# Small lanes of 1024 statements (ChaCha20-V3), small lanes for AES-V2 (round 4).  Output: gpurun_out/r04ln3/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ln3; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs python bench.py "$@" --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json "$tag ($envs $*)" || { echo "$tag failed"; tail -2 $O/$tag.err; }; }
for rep in 1 2; do
run b1024_max8192 GSC_MAX_BATCH=8192 --batch 1024 --steps 24 --warmup 4
run b1024 X=1 --batch 1024 --steps 24 --warmup 4
run b512 X=1 --batch 512 --steps 24 --warmup 4
run b256 X=1 --batch 256 --steps 24 --warmup 4
run b64 X=1 --batch 64 --steps 24 --warmup 4
run aes_b64c4_s0 GSC_SMALL_LANES=0 --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2
run aes_b64c4_s2 GSC_SMALL_LANES=2 --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2
run aes_b64c6_s0 GSC_SMALL_LANES=0 --workload aes128 --batch 64 --callers 6 --steps 12 --warmup 2
run aes_b64c6_s4 GSC_SMALL_LANES=4 --workload aes128 --batch 64 --callers 6 --steps 12 --warmup 2
run aes_b256c3_s0 GSC_SMALL_LANES=0 --workload aes128 --batch 256 --callers 3 --steps 8 --warmup 2
run aes_b256c3_s2 GSC_SMALL_LANES=2 --workload aes128 --batch 256 --callers 3 --steps 8 --warmup 2
done
