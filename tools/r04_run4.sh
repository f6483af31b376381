#!/bin/bash
# Round 4: parity of the final tree on the witness / latency tests, latency lines, the other BASELINE configs, then the PMC passes.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "small_integer or soak or kat or secrets or resident or latency_path or bench_bookkeeping or a_small_batch" > $O/pytest_sel.txt 2>&1; rc=$?; tail -6 $O/pytest_sel.txt; [ $rc -eq 0 ] || exit $rc
python bench.py --batch 1 --callers 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_b1_c1.json 2> $O/bench_b1_c1.err && cut -c1-150 $O/bench_b1_c1.json
python bench.py --batch 64 --callers 2 --steps 48 --warmup 6 --no-cpu-baseline > $O/bench_b64.json 2> $O/bench_b64.err && cut -c1-150 $O/bench_b64.json
python bench.py --steps 10 --warmup 3 > $O/bench_chacha20.json 2> $O/bench_chacha20.err && cut -c1-150 $O/bench_chacha20.json
python bench.py --library-defaults --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_library_defaults.json 2> $O/bench_chacha20_library_defaults.err && cut -c1-150 $O/bench_chacha20_library_defaults.json
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err && cut -c1-150 $O/bench_$w.json; done
bash tools/r04_pmc.sh
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok"
rm -rf $O/stats/*kernel_trace.csv 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
