#!/bin/bash
# Resident batch solver (k_solver_res): the test suite (resident by default up to 2048 columns), then on / off on the same box.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()}, d['verified'])" $1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu_4.txt 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu_4.txt
for rs in 1 0; do for b in 64 256 1024; do
  GSC_RES_SOLVER=$rs timeout -k 10 200 python bench.py --batch $b --callers 4 --steps 30 --warmup 6 --no-cpu-baseline > $O/rs${rs}_b$b.json 2> $O/rs${rs}_b$b.err && show $O/rs${rs}_b$b.json
done; done
for rs in 1 0; do GSC_RES_SOLVER=$rs timeout -k 10 200 python bench.py --batch 64 --callers 1 --steps 30 --warmup 6 --no-cpu-baseline > $O/rs${rs}_b64_c1.json 2> $O/rs${rs}_b64_c1.err && show $O/rs${rs}_b64_c1.json; done
for rs in 1 0; do GSC_RES_SOLVER=$rs timeout -k 10 300 python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline > $O/rs${rs}_aes128.json 2> $O/rs${rs}_aes128.err && show $O/rs${rs}_aes128.json; done
GSC_RES_SOLVER_MAX=8192 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rs_8192.json 2> $O/rs_8192.err && show $O/rs_8192.json
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rs_8192_off.json 2> $O/rs_8192_off.err && show $O/rs_8192_off.json
