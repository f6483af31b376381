#!/bin/bash
# AES-128 at batch 1024: how the call is cut over lanes (round 4).  Output: gpurun_out/r04al/
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04al; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err; echo "== $tag ($*)"; python3 -c "import json; d=json.load(open('$O/$tag.json')); print(d['value'], d['ms_per_step'], d.get('stage_ms_last_step'))"; }
run default GSC_NOP=1
run lanes1 GSC_LANES=1
run nosplit GSC_ENABLE_TEST_HOOKS=1 GSC_MIN_SPLIT=100000
tag=lanes1_callers1; GSC_LANES=1 python bench.py --workload aes128 --callers 1 --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/$tag.json 2> $O/$tag.err; echo "== $tag"; python3 -c "import json; d=json.load(open('$O/$tag.json')); print(d['value'], d['ms_per_step'], d.get('stage_ms_last_step'))"
