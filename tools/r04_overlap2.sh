#!/bin/bash
# GSC_OVERLAP_QUOTIENT off / on for mid-size calls, with the lanes' streams in separate hardware queues (round 4).  Output: gpurun_out/r04ov2/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ov2; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for ov in 0 1; do
  for b in 64 256 512; do GSC_OVERLAP_QUOTIENT=$ov python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline --verify 0 > $O/b${b}_ov$ov.json 2> $O/b${b}_ov$ov.err && line $O/b${b}_ov$ov.json "rep$rep chacha b$b overlap=$ov"; done
  GSC_OVERLAP_QUOTIENT=$ov python bench.py --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2 --no-cpu-baseline --verify 0 > $O/aes128b64_ov$ov.json 2> $O/aes128b64_ov$ov.err && line $O/aes128b64_ov$ov.json "rep$rep aes128 b64 callers=4 overlap=$ov"
  GSC_OVERLAP_QUOTIENT=$ov python bench.py --workload aes128 --batch 256 --callers 2 --steps 8 --warmup 2 --no-cpu-baseline --verify 0 > $O/aes128b256_ov$ov.json 2> $O/aes128b256_ov$ov.err && line $O/aes128b256_ov$ov.json "rep$rep aes128 b256 overlap=$ov"
done; done
