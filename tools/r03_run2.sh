#!/bin/bash
# Round-3 second GPU pass: test suite on the final NTT code + small lanes + heavy chain; batch sweep; one or two full lanes at 8192.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu_2.txt 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu_2.txt
for b in 64 256 1024; do python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && echo "b$b: $(python -c "import json;d=json.load(open('$O/bench_chacha20_b$b.json'));print(d['value'], d['ms_per_step'], d['stage_ms_last_step'], d['verified'])")"; done
for l in 1 2; do GSC_LANES=$l python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_lanes$l.json 2> $O/bench_chacha20_lanes$l.err && echo "lanes$l: $(python -c "import json;d=json.load(open('$O/bench_chacha20_lanes$l.json'));print(d['value'], d['ms_per_step'], d['stage_ms_last_step'], d['verified'])")"; done
ls $O
