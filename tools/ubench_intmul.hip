// Micro-benchmark: issue rate of the integer / fp64 multiply instructions that bound
// BN254 modular arithmetic on gfx950.  Not part of the product; evidence for DESIGN.md.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_intmul.hip -o gpurun_out/ubench_intmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // independent chains per lane

__global__ void k_mad64(uint32_t* out, uint32_t seed) {
    uint64_t acc[UNROLL];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}

__global__ void k_mullo(uint32_t* out, uint32_t seed) {
    uint32_t acc[UNROLL];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mulhi(uint32_t* out, uint32_t seed) {
    uint32_t acc[UNROLL];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad24(uint32_t* out, uint32_t seed) {
    uint32_t acc[UNROLL];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add32(uint32_t* out, uint32_t seed) {
    uint32_t acc[UNROLL];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_addc(uint32_t* out, uint32_t seed) {
    uint32_t acc[UNROLL];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_dfma(uint32_t* out, uint32_t seed) {
    double acc[UNROLL];
    double a = 1.0 + 1e-9 * (seed + threadIdx.x), b = 1e-7 * blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

__global__ void k_ffma(uint32_t* out, uint32_t seed) {
    float acc[UNROLL];
    float a = 1.0f + 1e-6f * (seed + threadIdx.x), b = 1e-7f * blockIdx.x;
    for (int i = 0; i < UNROLL; i++) acc[i] = i + a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

template <typename K>
static double run(const char* name, K kern, uint32_t* d_out, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double wave_instrs = (double)blocks * 4 * ITERS * UNROLL;       // wave-instructions issued
    double per_simd = wave_instrs / 1024.0;                           // per SIMD
    double cycles = ms * 1e-3 * 2.4e9;                                // at nominal 2.4 GHz
    printf("%-14s waves/SIMD=%d  %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4GHz)  %7.2f Tops/s lane-ops\n",
           name, waves_per_simd, ms, cycles / per_simd, wave_instrs * 64 / (ms * 1e-3) / 1e12);
    return ms;
}

int main() {
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 2, 4}) {
        run("v_add_u32", k_add32, d_out, w);
        run("v_addc_co_u32", k_addc, d_out, w);
        run("v_mad_u32_u24", k_mad24, d_out, w);
        run("v_mul_lo_u32", k_mullo, d_out, w);
        run("v_mul_hi_u32", k_mulhi, d_out, w);
        run("v_mad_u64_u32", k_mad64, d_out, w);
        run("v_fma_f64", k_dfma, d_out, w);
        run("v_fma_f32", k_ffma, d_out, w);
    }
    hipFree(d_out);
    return 0;
}
