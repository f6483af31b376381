#!/bin/bash
# AES-V2 parity tests, then the AES bench lines (round 4).  Output: gpurun_out/r04ac/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04ac; mkdir -p $O
python -m pytest tests/test_gpu_00_bench_config.py tests/test_gpu_aes.py tests/test_gpu_baseline_configs.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; [ $rc -eq 0 ] || exit $rc
for w in ${WORKLOADS:-aes128 aes256}; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err && python3 -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('verified'), d.get('stage_ms_last_step'))"; done
