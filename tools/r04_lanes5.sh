#!/bin/bash
# ChaCha20-V3 mid-size calls: one small lane against two (round 4).  Output: gpurun_out/r04ln5/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ln5; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs python bench.py "$@" --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err && line $O/$tag.json "$tag ($envs $*)" || { echo "$tag failed"; tail -2 $O/$tag.err; }; }
for rep in 1 2; do for s in 1 2; do
  run b64c2_s$s GSC_SMALL_LANES=$s --batch 64 --steps 24 --warmup 4
  run b64c6_s$s GSC_SMALL_LANES=$s --batch 64 --callers 6 --steps 24 --warmup 4
  run b256c2_s$s GSC_SMALL_LANES=$s --batch 256 --steps 24 --warmup 4
  run b256c4_s$s GSC_SMALL_LANES=$s --batch 256 --callers 4 --steps 24 --warmup 4
  run b1024c2_s$s GSC_SMALL_LANES=$s --batch 1024 --steps 24 --warmup 4
  run b1024c3_s$s GSC_SMALL_LANES=$s --batch 1024 --callers 3 --steps 24 --warmup 4
done; done
SECS=2 CALLERS="8 64 256" GSC_SMALL_LANES=1 bash tools/r03_prove_callers_c.sh 2>&1 | grep "proofs/s" | sed "s/^/small=1 /"
SECS=2 CALLERS="8 64 256" GSC_SMALL_LANES=2 bash tools/r03_prove_callers_c.sh 2>&1 | grep "proofs/s" | sed "s/^/small=2 /"
