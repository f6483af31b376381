#!/bin/bash
# Round 4: PMC passes of the default bench command (run through gpurun; counters in their own runs, --kernel-trace only).  Outputs under gpurun_out/r04p/.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o run --output-format csv -- $B > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o run --output-format csv -- $B > $O/pmc_write_bench.json 2> $O/pmc_write.err && echo "write ok" &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $O/pmc_tcc -o run --output-format csv -- $B > $O/pmc_tcc_bench.json 2> $O/pmc_tcc.err && echo "tcc ok" &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o run --output-format csv -- $B > $O/pmc_sq_bench.json 2> $O/pmc_sq.err && echo "sq ok" &&
python tools/make_msm_z_pmc.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) $(ls $O/pmc_write/*counter_collection.csv | head -1) $(ls $O/pmc_tcc/*counter_collection.csv | head -1) $O/msm_z_pmc.json 17 8192 32768 evaluation-form+digits &&
python tools/pmc_summary.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) > $O/pmc_sq_counters.txt &&
python tools/make_valu_per_add.py $O/pmc_sq_counters.txt $O/valu_per_add.json 8192 32768 15
rm -rf $O/pmc_*/*kernel_trace.csv 2>/dev/null
ls $O; grep -E "k_ntt|k_msm_win|k_wit" $O/pmc_sq_counters.txt | cut -c1-330
