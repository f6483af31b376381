// Calibration of rocprofv3's FETCH_SIZE for THIS workload's access pattern: every lane gathers one random, 64-byte-aligned
// 64-byte entry (four 16-byte loads) from a table far larger than the 256 MiB Infinity Cache — what k_msm does with the
// digit tables.  The number of gathered bytes is known exactly, so FETCH_SIZE / (gathers * 64 B) is the correction factor
// to apply to the counter for this pattern (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your
// own access pattern").  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o build/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void k_gather(const uint4* table, size_t entries, int iters, uint32_t* sink) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint64_t x = tid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    uint4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint4* e = table + (x % entries) * 4;
        const uint4 a = e[0], b = e[1], c = e[2], d = e[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.z + c.w + d.x;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;     // never true in practice; keeps the loads alive
}

int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? atoi(argv[1]) : 32;
    const int iters = argc > 2 ? atoi(argv[2]) : 16;
    const size_t threads = (size_t)1 << (argc > 3 ? atoi(argv[3]) : 24);
    const size_t bytes = gb << 30, entries = bytes / 64;
    uint4* t; uint32_t* sink;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(t, 1, bytes); (void)hipMemset(sink, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_gather<<<dim3((unsigned)(threads / 256)), dim3(256)>>>(t, entries, 1, sink);     // warm-up (page tables)
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_gather<<<dim3((unsigned)(threads / 256)), dim3(256)>>>(t, entries, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gathered = (double)threads * iters * 64.0;
    printf("table %zu GiB, %zu threads x %d gathers of 64 B = %.3f GB gathered, %.3f ms -> %.1f GB/s (64-byte accounting)\n",
           gb, threads, iters, gathered / 1e9, ms, gathered / 1e9 / (ms * 1e-3));
    printf("expected FETCH_SIZE of the timed launch if requests are tallied at 64 B: %.0f KB\n", gathered / 1024.0);
    return 0;
}
