#!/bin/sh
# The reference's own CPU number for bench.py's metric: run on a box that has Go >= 1.22, the module cache of
# the reference's go.mod and — because libraries/core_test.go:20-28 reads them at package init — the AES proving keys
# (generate them first: `go run keygen.go`, keygen.go:384,423).  Prints ns/op per Prove; proofs/s = GOMAXPROCS-independent
# 1e9 / (ns/op) for the single-proof path the benchmark measures (core_test.go:262-290).
#   usage: REFERENCE=/path/to/gnark-symmetric-crypto sh integration/cpu_baseline_go.sh
set -e
cd "${REFERENCE:?set REFERENCE to the reference checkout}/libraries"
echo "host: $(nproc) hardware threads, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
go version
go test -run xxx -bench 'Benchmark_ProveChacha|Benchmark_ProveAES128|Benchmark_ProveAES256' -benchtime 20x -cpu "$(nproc)" | tee /dev/stderr | \
  awk '/^Benchmark/ { printf "%s: %.2f proofs/s (one Prove at a time, gnark multithreaded over %s cores)\n", $1, 1e9 / $3, ENVIRON["NPROC"] ? ENVIRON["NPROC"] : "all" }'
