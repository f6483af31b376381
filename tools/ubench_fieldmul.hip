// Micro-benchmark: one 254-bit Montgomery product on gfx950, two ways (Next #4).  Not part of the product; evidence for
// DESIGN.md §5.
//   int29 : the product's own Fp arithmetic (bn254_fp29.hpp): nine 29-bit limbs, v_mad_u64_u32 columns, R = 2^261
//   f64x52: five 52-bit limbs held in doubles; a 52x52-bit product is split exactly into its high and low 52 bits by two
//           round-toward-zero v_fma_f64 (hi = rz(a*b + 2^104), lo = rz(a*b + (2^104 + 2^52 - hi))), the halves are summed as
//           64-bit integers (v_lshl_add_u64), reduction is word-serial Montgomery with q = low 52 bits of t * p'; R = 2^260.
// Each lane runs a dependent chain x <- x * y; both results are checked on the host against plain big-number arithmetic.
// Build: hipcc -O3 --offload-arch=gfx950 -I gnark-symmetric-crypto_amd/csrc tools/ubench_fieldmul.hip -o build/ubench_fieldmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "bn254_fp29.hpp"

using namespace bn254;
constexpr int CHAIN = 512;

__global__ void k_int29(const fe* a, const fe* b, fe* out, int chain) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    fe9 x = Fp29::unpack(a[i]); const fe9 y = Fp29::unpack(b[i]);
    for (int c = 0; c < chain; c++) x = Fp29::mul(x, y);
    out[i] = Fp29::pack(Fp29::freeze(x));
}

struct F52 { double l[5]; };
struct P52 { double p[5]; double pinv; };       // modulus limbs and -p^-1 mod 2^52 as doubles
__device__ __forceinline__ double fma_hw(double a, double b, double c) { double r; asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ double sub_hw(double a, double b) { double r; asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ F52 mul52(const F52& a, const F52& b, const P52& m) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52, T52 = 0x1p52;
    const uint64_t BH = 0x4670000000000000ull, BL = 0x4330000000000000ull;      // bit patterns of 2^104 and 2^52 (sums of them wrap: unsigned arithmetic)
    const uint64_t MASK = (1ull << 52) - 1;
    uint64_t S[11];
#pragma unroll
    for (int k = 0; k < 11; k++) S[k] = 0;
    auto mac = [&](double x, double y, int k) {      // S[k] += lo(x*y), S[k+1] += hi(x*y): 2 fma, 1 add, 2 integer adds (+ constants folded below)
        const double hi = fma_hw(x, y, C1), lo = fma_hw(x, y, sub_hw(C2, hi));
        S[k + 1] += (uint64_t)__double_as_longlong(hi); S[k] += (uint64_t)__double_as_longlong(lo);
    };
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) mac(a.l[i], b.l[j], i + j);
        // remove the exponent patterns column i has collected so far: it is complete after this round's q*p products are added,
        // but q needs its true low 52 bits now
        const uint64_t n_lo = (uint64_t)(2 * i + 1), n_hi = (uint64_t)(2 * i);      // (#lo, #hi) patterns added to S[i] by a*b and earlier q*p rounds
        const uint64_t t = (S[i] - n_lo * BL - n_hi * BH) & MASK;
        const double td = sub_hw(__longlong_as_double((long long)(t | BL)), T52);
        const double qhi = fma_hw(td, m.pinv, C1), qlo = fma_hw(td, m.pinv, sub_hw(C2, qhi));
        const double q = sub_hw(qlo, T52);
#pragma unroll
        for (int j = 0; j < 5; j++) mac(q, m.p[j], i + j);
        // column i is now a multiple of 2^52 (plus its patterns): carry it into column i + 1
        const uint64_t full = S[i] - (n_lo + 1) * BL - n_hi * BH;      // a true non-negative integer below 2^63
        S[i + 1] += full >> 52;
        S[i] = 0;
    }
    // columns 5..9 hold the result; patterns collected: column k got lo patterns from pairs (i,j), i+j = k, and hi patterns from i+j = k-1, twice (a*b and q*p)
    F52 r;
#pragma unroll
    for (int k = 5; k < 10; k++) {
        const uint64_t n_lo = (uint64_t)(2 * (9 - k)), n_hi = (uint64_t)(2 * (10 - k));      // pairs with i + j = k: 9 - k (k >= 4); with i + j = k - 1: 10 - k
        const uint64_t full = S[k] - n_lo * BL - n_hi * BH;
        r.l[k - 5] = sub_hw(__longlong_as_double((long long)((full & MASK) | BL)), T52);
        S[k + 1] += full >> 52;
    }
    return r;
}
__global__ void k_f64x52(const double* a, const double* b, double* out, P52 m, int chain) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");      // FP64 rounding: toward zero
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F52 x, y;
    for (int k = 0; k < 5; k++) { x.l[k] = a[5 * i + k]; y.l[k] = b[5 * i + k]; }
    for (int c = 0; c < chain; c++) x = mul52(x, y, m);
    for (int k = 0; k < 5; k++) out[5 * i + k] = x.l[k];
}

// ---- host-side big numbers (slow, obviously correct) ----
typedef unsigned __int128 u128;
struct Big { uint64_t w[5]; };      // < 2^320
static const uint64_t Pm[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static bool geq(const uint64_t* a, const uint64_t* b) { for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return true; }
static void sub(uint64_t* a, const uint64_t* b) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void addmod(uint64_t* a, const uint64_t* b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; a[i] = (uint64_t)c; c >>= 64; } if (c || geq(a, Pm)) sub(a, Pm); }
static void mulmod(uint64_t* r, const uint64_t* a, const uint64_t* b) {      // double-and-add
    uint64_t acc[4] = {0, 0, 0, 0};
    for (int i = 255; i >= 0; i--) { uint64_t t[4]; memcpy(t, acc, 32); addmod(acc, t); if ((b[i / 64] >> (i % 64)) & 1) addmod(acc, a); }
    memcpy(r, acc, 32);
}
static void halve(uint64_t* a) { u128 c = 0; if (a[0] & 1) { for (int i = 0; i < 4; i++) { c += (u128)a[i] + Pm[i]; a[i] = (uint64_t)c; c >>= 64; } } uint64_t top = (uint64_t)c; for (int i = 0; i < 4; i++) a[i] = (a[i] >> 1) | ((i < 3 ? a[i + 1] : top) << 63); }

template <class F> static float timeit(F launch, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < reps; r++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
    const size_t n = 256 * 4 * 256;      // 4 workgroups of 256 threads per CU: four waves per SIMD
    std::vector<uint64_t> ha(4 * n), hb(4 * n);
    uint64_t x = 0x9E3779B97F4A7C15ull; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 4; k++) { ha[4 * i + k] = rnd(); hb[4 * i + k] = rnd(); if (k == 3) { ha[4 * i + k] &= 0x0fffffffffffffffull; hb[4 * i + k] &= 0x0fffffffffffffffull; } }
    // int29 inputs are 8 x u32 words (same bytes); f64 inputs: 52-bit limbs as doubles
    std::vector<double> da(5 * n), db(5 * n);
    auto limbs52 = [](const uint64_t* w, double* out) { for (int k = 0; k < 5; k++) { const int bit = 52 * k, wi = bit / 64, sh = bit % 64; u128 v = w[wi]; if (wi + 1 < 4) v |= (u128)w[wi + 1] << 64; out[k] = (double)(uint64_t)((v >> sh) & ((1ull << 52) - 1)); } };
    for (size_t i = 0; i < n; i++) { limbs52(&ha[4 * i], &da[5 * i]); limbs52(&hb[4 * i], &db[5 * i]); }
    P52 m; limbs52(Pm, m.p);
    { uint64_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - Pm[0] * inv; m.pinv = (double)((0 - inv) & ((1ull << 52) - 1)); }
    fe *d_a, *d_b, *d_o; double *d_da, *d_db, *d_do;
    (void)hipMalloc(&d_a, 32 * n); (void)hipMalloc(&d_b, 32 * n); (void)hipMalloc(&d_o, 32 * n); (void)hipMalloc(&d_da, 40 * n); (void)hipMalloc(&d_db, 40 * n); (void)hipMalloc(&d_do, 40 * n);
    (void)hipMemcpy(d_a, ha.data(), 32 * n, hipMemcpyHostToDevice); (void)hipMemcpy(d_b, hb.data(), 32 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_da, da.data(), 40 * n, hipMemcpyHostToDevice); (void)hipMemcpy(d_db, db.data(), 40 * n, hipMemcpyHostToDevice);
    // ---- correctness, chain = 1: int29 gives a*b/2^261, f64x52 gives a*b/2^260 (mod p, any representative below 2^260)
    hipLaunchKernelGGL(k_int29, dim3(n / 256), dim3(256), 0, 0, d_a, d_b, d_o, 1);
    hipLaunchKernelGGL(k_f64x52, dim3(n / 256), dim3(256), 0, 0, d_da, d_db, d_do, m, 1);
    std::vector<uint64_t> ho(4 * n); std::vector<double> hdo(5 * n);
    (void)hipMemcpy(ho.data(), d_o, 32 * n, hipMemcpyDeviceToHost); (void)hipMemcpy(hdo.data(), d_do, 40 * n, hipMemcpyDeviceToHost);
    int bad29 = 0, bad52 = 0;
    for (size_t i = 0; i < 64; i++) {
        uint64_t a[4], b[4], want[4]; memcpy(a, &ha[4 * i], 32); memcpy(b, &hb[4 * i], 32);
        while (geq(a, Pm)) sub(a, Pm); while (geq(b, Pm)) sub(b, Pm);
        mulmod(want, a, b);
        uint64_t w260[4]; memcpy(w260, want, 32); for (int k = 0; k < 260; k++) halve(w260);
        uint64_t w261[4]; memcpy(w261, w260, 32); halve(w261);
        if (memcmp(w261, &ho[4 * i], 32)) bad29++;
        // f64 result: sum limbs * 2^(52k), reduce mod p
        u128 acc = 0; uint64_t r[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5; k++) { const uint64_t v = (uint64_t)hdo[5 * i + k]; const int bit = 52 * k, wi = bit / 64, sh = bit % 64; acc = (u128)v << sh; u128 c = 0; for (int q = wi; q < 5; q++) { c += (u128)r[q] + (uint64_t)(acc & 0xFFFFFFFFFFFFFFFFull); r[q] = (uint64_t)c; c >>= 64; acc >>= 64; } }
        uint64_t rr[4] = {r[0], r[1], r[2], r[3]};
        if (r[4]) bad52++; else { while (geq(rr, Pm)) sub(rr, Pm); if (memcmp(rr, w260, 32)) bad52++; }
    }
    printf("correctness on 64 random products: int29 mismatches %d, f64x52 mismatches %d\n", bad29, bad52);
    // ---- timing: dependent chains, four waves per SIMD on every CU
    const float t29 = timeit([&] { hipLaunchKernelGGL(k_int29, dim3(n / 256), dim3(256), 0, 0, d_a, d_b, d_o, CHAIN); }, 5);
    const float t52 = timeit([&] { hipLaunchKernelGGL(k_f64x52, dim3(n / 256), dim3(256), 0, 0, d_da, d_db, d_do, m, CHAIN); }, 5);
    const double prods = (double)n * CHAIN;
    printf("int29  (9 x 29-bit limbs, v_mad_u64_u32): %8.3f ms  %.3e products/s  %.1f cycles/product/SIMD at 2.4 GHz\n", t29, prods / (t29 * 1e-3), t29 * 1e-3 * 2.4e9 / (prods / 64 / 1024));
    printf("f64x52 (5 x 52-bit limbs, v_fma_f64)    : %8.3f ms  %.3e products/s  %.1f cycles/product/SIMD at 2.4 GHz\n", t52, prods / (t52 * 1e-3), t52 * 1e-3 * 2.4e9 / (prods / 64 / 1024));
    printf("ratio f64x52 / int29 = %.2f (adopt only below 0.85)\n", t52 / t29);
    return 0;
}
