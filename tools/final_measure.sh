#!/bin/bash
# Round-2 measurement batch (run through gpurun): bench lines for every BASELINE config, the rocprofv3 summary of the default
# bench command, PMC passes for the dominant kernel, the GPU test suite.  Outputs under gpurun_out/r02/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench chacha20 ok" &&
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err && echo "bench $w ok"; done &&
for b in 1 64 1024; do python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && echo "bench b$b ok"; done &&
python bench.py --force-dist --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_chacha20_forcedist.json 2> $O/bench_chacha20_forcedist.err && echo "force-dist ok" &&
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write_bench.json 2> $O/pmc_write.err && echo "write ok" &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $O/pmc_tcc -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_tcc_bench.json 2> $O/pmc_tcc.err && echo "tcc ok" &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_sq_bench.json 2> $O/pmc_sq.err && echo "sq ok" &&
./build/ubench_gather 64 16 24 > $O/ubench_gather.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
rm -rf $O/stats/*kernel_trace.csv $O/pmc_*/*kernel_trace.csv 2>/dev/null    # (big; the summaries are what is kept)
ls $O
