#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['stage_ms_last_step'].items()}, d['verified'])" $1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aes.py -m gpu -x -q > $O/pytest_gpu_5.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_5.txt
timeout -k 10 200 python bench.py --batch 1 --callers 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/lat_b1.json 2> $O/lat_b1.err && show $O/lat_b1.json
for rs in 1 0; do for b in 64 256 1024; do
  GSC_RES_SOLVER=$rs timeout -k 10 200 python bench.py --batch $b --callers 4 --steps 30 --warmup 6 --no-cpu-baseline > $O/rd${rs}_b$b.json 2> $O/rd${rs}_b$b.err && show $O/rd${rs}_b$b.json
done; done
for rs in 1 0; do GSC_RES_SOLVER=$rs timeout -k 10 200 python bench.py --batch 64 --callers 1 --steps 30 --warmup 6 --no-cpu-baseline > $O/rd${rs}_b64_c1.json 2> $O/rd${rs}_b64_c1.err && show $O/rd${rs}_b64_c1.json; done
for rs in 1 0; do GSC_RES_SOLVER=$rs timeout -k 10 300 python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline > $O/rd${rs}_aes128.json 2> $O/rd${rs}_aes128.err && show $O/rd${rs}_aes128.json; done
