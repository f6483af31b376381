#!/bin/bash
# hipGraph replay of mid-size calls: the test suite, then graphs on / off at the same box.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_last_step'].items()}, d['verified'], d['roofline']['kernel'][:24])" $1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu_3.txt 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu_3.txt
for gr in 1 0; do for b in 64 128 256 512; do
  GSC_GRAPHS=$gr python bench.py --batch $b --callers 6 --steps 30 --warmup 6 --no-cpu-baseline > $O/gr${gr}_b$b.json 2> $O/gr${gr}_b$b.err && show $O/gr${gr}_b$b.json
done; done
for gr in 1 0; do GSC_GRAPHS=$gr python bench.py --batch 64 --callers 1 --steps 30 --warmup 6 --no-cpu-baseline > $O/gr${gr}_b64_c1.json 2> $O/gr${gr}_b64_c1.err && show $O/gr${gr}_b64_c1.json; done
