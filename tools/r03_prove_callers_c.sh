#!/bin/bash
# Closed-loop single-Prove callers from a plain-C FFI host (integration/ffi_harness.c, pthreads): proofs/s and latency against the number of callers.
set -o pipefail
mkdir -p build gpurun_out/callers
gcc -O2 -Wall -o build/ffi_harness integration/ffi_harness.c -ldl -lpthread || exit 1
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
open("build/pk.chacha20", "wb").write(bench.golden("pk.chacha20")); open("build/r1cs.chacha20", "wb").write(bench.golden("r1cs.chacha20"))
PY
timeout -k 10 600 ./build/ffi_harness gnark-symmetric-crypto_amd/libprove.so callers build/pk.chacha20 build/r1cs.chacha20 ${SECS:-4} ${CALLERS:-1 2 4 8 16 32 64 128 256 512} 2>&1 | tee gpurun_out/callers/c_harness.txt
