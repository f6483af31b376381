#!/bin/bash
# Round-3 measurement batch (run through gpurun): bench lines for every BASELINE config, the rocprofv3 summary of the default bench
# command, PMC passes for the dominant kernel and the quotient kernels, the GPU test suite.  Outputs under gpurun_out/r03f/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench chacha20 ok" &&
for w in aes128 aes256 mixed; do python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err && echo "bench $w ok"; done &&
for b in 1 16 32 64 256 1024; do python bench.py --batch $b --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && echo "bench b$b ok"; done &&
python bench.py --workload aes128 --batch 1 --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_aes128_b1.json 2> $O/bench_aes128_b1.err && echo "aes b1 ok" &&
python bench.py --force-dist --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_chacha20_forcedist.json 2> $O/bench_chacha20_forcedist.err && echo "force-dist ok" &&
python bench.py --gpus 2 --in-library --devices 0,0 --batch 4096 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_chacha20_inlibrary_2x4096_rehearsal.json 2> $O/inlib1.err && echo "in-library 2x4096 ok" &&
python bench.py --gpus 2 --in-library --devices 0,0 --batch 64 --callers 4 --steps 30 --warmup 6 --no-cpu-baseline > $O/bench_chacha20_inlibrary_2x64_rehearsal.json 2> $O/inlib2.err && echo "in-library 2x64 ok" &&
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify 0 > $O/stats_bench.json 2> $O/stats.err && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_write_bench.json 2> $O/pmc_write.err && echo "write ok" &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $O/pmc_tcc -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_tcc_bench.json 2> $O/pmc_tcc.err && echo "tcc ok" &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $O/pmc_sq_bench.json 2> $O/pmc_sq.err && echo "sq ok" &&
python tools/make_msm_z_pmc.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) $(ls $O/pmc_write/*counter_collection.csv | head -1) $(ls $O/pmc_tcc/*counter_collection.csv | head -1) $O/msm_z_pmc.json 17 &&
python tools/pmc_summary.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) > $O/pmc_sq_counters.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
rm -rf $O/stats/*kernel_trace.csv $O/pmc_*/*kernel_trace.csv 2>/dev/null    # (big; the summaries are what is kept)
ls $O $O/stats
