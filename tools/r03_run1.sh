#!/bin/bash
# Round-3 first GPU pass: the GPU test suite, one bench line, SQ / LDS counters of the quotient kernels.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu_1.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_1.txt
python bench.py --steps 5 --warmup 2 > $O/bench_chacha20_1.json 2> $O/bench_chacha20_1.err && echo "bench ok" && cat $O/bench_chacha20_1.json &&
GSC_Z_TABLE_GB=140 GSC_MAX_BATCH=8192 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o run --output-format csv -- python3 tools/prof_z.py 8192 > $O/pmc_sq.out 2> $O/pmc_sq.err && echo "sq ok" &&
python tools/pmc_summary.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) > $O/pmc_sq_counters.txt; cat $O/pmc_sq_counters.txt | grep -E "k_ntt|k_msm_win"
rm -rf $O/pmc_*/*kernel_trace.csv 2>/dev/null
ls $O
