#!/bin/bash
# Quotient kernels beside the wire-set MSMs (GSC_OVERLAP_QUOTIENT): parity, then off / on at every batch size (round 4).  Output: gpurun_out/r04ov/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ov; mkdir -p $O
python -m pytest tests/test_gpu_00_bench_config.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; [ $rc -eq 0 ] || exit $rc
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d.get('stage_ms_last_step'))"; }
for ov in 0 1; do
  for b in 64 256 1024; do GSC_OVERLAP_QUOTIENT=$ov python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline --verify 0 > $O/b${b}_ov$ov.json 2> $O/b${b}_ov$ov.err && line $O/b${b}_ov$ov.json "chacha b$b overlap=$ov"; done
  GSC_OVERLAP_QUOTIENT=$ov python bench.py --steps 8 --warmup 2 --no-cpu-baseline --verify 0 > $O/b8192_ov$ov.json 2> $O/b8192_ov$ov.err && line $O/b8192_ov$ov.json "chacha b8192 overlap=$ov"
  GSC_OVERLAP_QUOTIENT=$ov python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/aes128_ov$ov.json 2> $O/aes128_ov$ov.err && line $O/aes128_ov$ov.json "aes128 overlap=$ov"
done
