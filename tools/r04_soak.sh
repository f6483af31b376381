#!/bin/bash
# Round 4, final tree: soaks (every checked proof through libverify.so): random-size batches at the bench configuration, latency-path calls, AES sizes.
set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r04soak; mkdir -p $O
python tools/soak_batches.py 11 30 > $O/soak_batches.txt 2>&1; rc=$?; tail -3 $O/soak_batches.txt; [ $rc -eq 0 ] || exit $rc
python tools/soak_latency.py 5 300 > $O/soak_latency.txt 2>&1; rc=$?; tail -3 $O/soak_latency.txt; [ $rc -eq 0 ] || exit $rc
GSC_SMALL_WITNESS_FEW=0 python tools/soak_latency.py 6 150 > $O/soak_latency_resident.txt 2>&1; rc=$?; tail -3 $O/soak_latency_resident.txt; [ $rc -eq 0 ] || exit $rc
