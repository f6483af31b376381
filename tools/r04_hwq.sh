#!/bin/bash
# Does the number of hardware queues HIP spreads the lanes' streams over matter?  MODES = list of GSC_STREAM_PRIORITIES:GPU_MAX_HW_QUEUES (round 4)  Output: gpurun_out/r04hq/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04hq; mkdir -p $O
line() { python3 -c "import json; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for m in ${MODES:-0:4 1:4 1:8}; do
  export GSC_STREAM_PRIORITIES=${m%%:*} GPU_MAX_HW_QUEUES=${m##*:}; q=p${m%%:*}q${m##*:}
  for b in 64 1024; do python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline --verify 0 > $O/b${b}_q$q.json 2> $O/b${b}_q$q.err && line $O/b${b}_q$q.json "rep$rep chacha b$b queues=$q"; done
  python bench.py --batch 64 --callers 6 --steps 24 --warmup 4 --no-cpu-baseline --verify 0 > $O/b64c6_q$q.json 2> $O/b64c6_q$q.err && line $O/b64c6_q$q.json "rep$rep chacha b64 callers=6 queues=$q"
  python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/aes128_q$q.json 2> $O/aes128_q$q.err && line $O/aes128_q$q.json "rep$rep aes128 queues=$q"
  python bench.py --workload aes128 --batch 64 --callers 4 --steps 12 --warmup 2 --no-cpu-baseline --verify 0 > $O/aes128b64_q$q.json 2> $O/aes128b64_q$q.err && line $O/aes128b64_q$q.json "rep$rep aes128 b64 callers=4 queues=$q"
  SECS=2 CALLERS="8 64 256" bash tools/r03_prove_callers_c.sh 2>&1 | grep -i "callers\|proofs/s" | sed "s/^/rep$rep queues=$q /" | tail -4
done
done
