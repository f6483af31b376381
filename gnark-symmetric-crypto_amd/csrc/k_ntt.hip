// Quotient polynomial H = (A*B - C) / (X^n - 1) over BN254 Fr, batched over proofs.
//
// Replaces computeH inside groth16.Prove (reference libraries/prover/impl/provers.go:148,216; gnark
// backend/groth16/bn254.computeH + gnark-crypto fr/fft — SURVEY.md §8(a) a7, mathematics App. D):
//   3 inverse NTTs (DIF, natural -> bit-reversed), coset scaling, 3 forward NTTs (DIT, bit-reversed -> natural),
//   pointwise (a*b - c)/(g^n - 1), 1 inverse coset NTT (DIF) whose bit-reversed output order is exactly the
//   order pk.G1.Z is stored in.
//
// Data stay in the solver's [index][proof] layout for the whole pipeline (no transposes).  n = 2^L is split
// as 2^Lhi x 2^Llo: "strided" kernels own the stages that couple the top Lhi index bits (groups of 2^Lhi
// elements at stride 2^Llo), "contiguous" kernels the low Llo bits.  Because DIF ends where DIT begins, the
// seven transforms take four kernels:
//   K1 strided DIF head (a,b,c) | K2 contiguous DIF tail + coset scale + DIT head (a,b,c)
//   K3 strided DIT tail (a,b,c) + pointwise + strided DIF head (h) | K4 contiguous DIF tail + final scale (h).
// Every kernel stages a tile of P proofs x 2^Lhi (or 2^Llo) elements in LDS, limb-major, one radix-2 stage
// per barrier; global accesses are P*32 = 128-byte segments.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

using F = Fr29;            // radix-2^29 lazy-limb scalar field (bn254_fp29.hpp): 227-instruction products, carry-free add/sub
constexpr int P = 4;       // proofs per workgroup tile

// Ranges inside the transforms (Fr29: R' = 2^261 ~ 169 r, a product a*w with w < r lands in (-|a|/169, |a|/169 + r)):
//   * limbs: everything written to a tile has |limb| < 2^30 (one lazy add/sub of tight values); an operand is carried
//     (norm(), 24 instructions) only where it would otherwise meet a second lazy addition;
//   * values: a DIT stage adds a fresh product to u, so |value| grows by <= 1.1 r per stage; a DIF stage doubles the
//     sum path.  Instead of a comparison chain per butterfly, reduce_top() is applied once per DIF run of >= 4 stages
//     and once before a kernel stores its tile: it estimates q = floor(value / r) from the top limb (float multiply)
//     and subtracts the tabulated q*r, leaving a value in (-1.001 r, 2.001 r).
constexpr float INV_TOP_R = 1.0f / 3171407.0f;     // top limb of r is 3171406
__device__ __forceinline__ fe9 reduce_top(const fe9& x, const int32_t* qr) {
    const fe9 t = F::norm(x);
    int q = (int)floorf((float)t.l[8] * INV_TOP_R);
    q = q < -NTT_QMAX ? -NTT_QMAX : (q > NTT_QMAX ? NTT_QMAX : q);
    const int4* e = reinterpret_cast<const int4*>(qr + 12 * (q + NTT_QMAX));
    const int4 a = e[0], b = e[1], c = e[2];
    fe9 r;
    r.l[0] = t.l[0] - a.x; r.l[1] = t.l[1] - a.y; r.l[2] = t.l[2] - a.z; r.l[3] = t.l[3] - a.w;
    r.l[4] = t.l[4] - b.x; r.l[5] = t.l[5] - b.y; r.l[6] = t.l[6] - b.z; r.l[7] = t.l[7] - b.w; r.l[8] = t.l[8] - c.x;
    return r;      // signed-tight
}
// memory image between kernels: non-negative, tight, < 2^256 (not necessarily < r)
__device__ __forceinline__ void store_lazy(fe* p, const fe9& x, const int32_t* qr) {
    fe9 t = F::norm(reduce_top(x, qr));
    const bool lo = t.l[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] += lo ? F::PK(1, i) : 0;     // + 2r: (-1.001 r, 2.001 r) -> [0, 2.001 r)
    store_fe(p, F::pack(F::norm(t)));
}

// twiddles are stored already split into limbs: 12 int32 per entry (9 used; 48-byte stride keeps the three 16-byte loads aligned)
__device__ __forceinline__ fe9 load_tw(const int32_t* tw, uint32_t ex) {
    const int4* q = reinterpret_cast<const int4*>(tw + 12 * (size_t)ex);
    const int4 a = q[0], b = q[1], c = q[2];
    fe9 r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    return r;
}

struct Tile {
    int32_t* lds; uint32_t plane;   // plane = elements * P (words per limb plane)
    __device__ __forceinline__ fe9 get(uint32_t e, uint32_t q) const {
        fe9 r; const uint32_t o = e * P + q;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = lds[i * plane + o];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t e, uint32_t q, const fe9& v) const {
        const uint32_t o = e * P + q;
#pragma unroll
        for (int i = 0; i < 9; i++) lds[i * plane + o] = v.l[i];
    }
};

// one DIF stage on the tile: pairs (e1, e1 + he); twiddle exponent = (gidx(e1) mod hg) << s.  Operands in the tile are tight or
// signed-tight (previous sums are carried before they are written); REDUCE additionally pulls the sum path back to (-r, 2r).
template <bool STRIDED, bool REDUCE>
__device__ __forceinline__ void dif_stage(const Tile& t, uint32_t bf, uint32_t q, uint32_t he, uint32_t hg, int s, int Llo, uint32_t tile_id, const int32_t* tw, const int32_t* qr) {
    const uint32_t e1 = 2 * bf - (bf & (he - 1)), e2 = e1 + he;
    const uint32_t gi = STRIDED ? ((e1 << Llo) + tile_id) : ((tile_id << Llo) + e1);
    const uint32_t ex = (gi & (hg - 1)) << s;
    const fe9 u = t.get(e1, q), v = t.get(e2, q);
    const fe9 dif = F::sub(u, v), sum = F::add(u, v);
    t.put(e1, q, REDUCE ? reduce_top(sum, qr) : F::norm(sum));
    t.put(e2, q, ex ? F::mul(dif, load_tw(tw, ex)) : (REDUCE ? reduce_top(dif, qr) : F::norm(dif)));
}
// a run of DIF stages s0 .. s1-1 with one barrier per stage; the stage that completes four doublings reduces the sum path
template <bool STRIDED>
__device__ __forceinline__ void dif_run(const Tile& t, uint32_t bf, uint32_t q, int s0, int s1, int done, int L, int Llo, uint32_t tile_id, const NttPlan& pl) {
    for (int s = s0; s < s1; s++) {
        const uint32_t hg = 1u << (L - 1 - s);
        const uint32_t he = STRIDED ? hg >> Llo : hg;
        if (((s - s0 + done) & 3) == 3) dif_stage<STRIDED, true>(t, bf, q, he, hg, s, Llo, tile_id, pl.tw_inv, pl.qr);
        else dif_stage<STRIDED, false>(t, bf, q, he, hg, s, Llo, tile_id, pl.tw_inv, pl.qr);
        __syncthreads();
    }
}
// one DIT stage: half = 2^s; twiddle exponent = (gidx(e1) mod 2^s) << (L-1-s).  u is carried, v (|limb| < 2^30) goes straight into
// the product, so both results are again single lazy sums of tight values.
template <bool STRIDED>
__device__ __forceinline__ void dit_stage(const Tile& t, uint32_t bf, uint32_t q, uint32_t he, int s, int L, int Llo, uint32_t tile_id, const int32_t* tw) {
    const uint32_t e1 = 2 * bf - (bf & (he - 1)), e2 = e1 + he;
    const uint32_t gi = STRIDED ? ((e1 << Llo) + tile_id) : ((tile_id << Llo) + e1);
    const uint32_t ex = (gi & ((1u << s) - 1)) << (L - 1 - s);
    const fe9 u = F::norm(t.get(e1, q));
    fe9 v = t.get(e2, q);
    v = ex ? F::mul(v, load_tw(tw, ex)) : F::norm(v);
    t.put(e1, q, F::add(u, v)); t.put(e2, q, F::sub(u, v));
}

// K1: strided DIF head.  grid (2^Llo, batch/P, nvec); block (2^(Lhi-1) * P).  Input: the solver's a/b/c rows (canonical values of
// the 2^256 Montgomery domain); they are used as they are — every stage is linear, and K2's scale table folds in the
// change of domain (2^256 -> 2^261).
__global__ void k_ntt_dif_strided(NttPlan pl, fe* v0, fe* v1, fe* v2, size_t m, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t G = 1u << Lhi;
    fe* vec = blockIdx.z == 0 ? v0 : blockIdx.z == 1 ? v1 : v2;
    const uint32_t g = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, bf = threadIdx.x / P;
    Tile t{smem, G * P};
    for (uint32_t e = bf; e < G; e += G / 2) {
        const size_t idx = ((size_t)e << Llo) + g;
        t.put(e, q, idx < m ? F::load(vec + idx * batch + q0 + q) : F::zero());
    }
    __syncthreads();
    dif_run<true>(t, bf, q, 0, Lhi, 0, L, Llo, g, pl);
    for (uint32_t e = bf; e < G; e += G / 2) {
        const size_t idx = ((size_t)e << Llo) + g;
        store_lazy(vec + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
}

// K2: contiguous DIF tail, coset scale, contiguous DIT head.  grid (2^Lhi, batch/P, nvec); block (2^(Llo-1) * P)
__global__ void k_ntt_mid_contig(NttPlan pl, fe* v0, fe* v1, fe* v2, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t Cn = 1u << Llo;
    fe* vec = blockIdx.z == 0 ? v0 : blockIdx.z == 1 ? v1 : v2;
    const uint32_t b = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, bf = threadIdx.x / P;
    Tile t{smem, Cn * P};
    for (uint32_t e = bf; e < Cn; e += Cn / 2) {
        const size_t idx = ((size_t)b << Llo) + e;
        t.put(e, q, F::load(vec + idx * batch + q0 + q));
    }
    __syncthreads();
    dif_run<false>(t, bf, q, Lhi, L, 0, L, Llo, b, pl);
    for (uint32_t e = bf; e < Cn; e += Cn / 2) {
        const size_t idx = ((size_t)b << Llo) + e;
        t.put(e, q, F::mul(t.get(e, q), F::load(pl.scale_mid + idx)));
    }
    __syncthreads();
    for (int s = 0; s < Llo; s++) {
        dit_stage<false>(t, bf, q, 1u << s, s, L, Llo, b, pl.tw_fwd);
        __syncthreads();
    }
    for (uint32_t e = bf; e < Cn; e += Cn / 2) {
        const size_t idx = ((size_t)b << Llo) + e;
        store_lazy(vec + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
}

// K3: strided DIT tail for a, b, c; h = (a*b - c) * den_inv; strided DIF head for h (written over a).
__global__ void k_ntt_pointwise_strided(NttPlan pl, fe* va, const fe* vb, const fe* vc, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t G = 1u << Lhi;
    const uint32_t g = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, bf = threadIdx.x / P;
    Tile t{smem, G * P};
    fe9 r1[3], r2[3];
    for (int k = 0; k < 3; k++) {
        const fe* vec = k == 0 ? va : k == 1 ? vb : vc;
        for (uint32_t e = bf; e < G; e += G / 2) {
            const size_t idx = ((size_t)e << Llo) + g;
            t.put(e, q, F::load(vec + idx * batch + q0 + q));
        }
        __syncthreads();
        for (int s = Llo; s < L - 1; s++) {
            dit_stage<true>(t, bf, q, 1u << (s - Llo), s, L, Llo, g, pl.tw_fwd);
            __syncthreads();
        }
        {   // last DIT stage (s = L-1): pairs (bf, bf + G/2), results stay in registers
            const uint32_t e1 = bf, e2 = bf + G / 2;
            const uint32_t ex = ((e1 << Llo) + g) & ((1u << (L - 1)) - 1);
            const fe9 u = F::norm(t.get(e1, q));
            fe9 v = t.get(e2, q);
            v = ex ? F::mul(v, load_tw(pl.tw_fwd, ex)) : F::norm(v);
            const fe9 a1 = F::norm(F::add(u, v)), a2 = F::norm(F::sub(u, v));      // tight; |value| <= 2^256/r + 8 * 1.1 < 15 r
            if (k == 0) { r1[0] = a1; r2[0] = a2; } else if (k == 1) { r1[1] = a1; r2[1] = a2; } else { r1[2] = a1; r2[2] = a2; }
        }
        __syncthreads();
    }
    const fe9 den = F::load(pl.den_inv);
    const fe9 h1 = F::mul(F::sub(F::mul(r1[0], r1[1]), r1[2]), den);     // a*b in (-1.4r, 2.4r), minus c: |.| < 18 r, signed-tight: fine as a product operand
    const fe9 h2 = F::mul(F::sub(F::mul(r2[0], r2[1]), r2[2]), den);
    {   // first DIF stage (s = 0): same pairs; twiddle exponent = gidx(e1) mod n/2
        const uint32_t e1 = bf, e2 = bf + G / 2;
        const uint32_t ex = ((e1 << Llo) + g) & ((1u << (L - 1)) - 1);
        const fe9 dif = F::sub(h1, h2);
        t.put(e1, q, F::norm(F::add(h1, h2)));
        t.put(e2, q, ex ? F::mul(dif, load_tw(pl.tw_inv, ex)) : F::norm(dif));
    }
    __syncthreads();
    dif_run<true>(t, bf, q, 1, Lhi, 1, L, Llo, g, pl);
    for (uint32_t e = bf; e < G; e += G / 2) {
        const size_t idx = ((size_t)e << Llo) + g;
        store_lazy(va + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
}

// K4: contiguous DIF tail on h, then scale by n^-1 g^-j and leave Montgomery form (canonical output in [0, r)).
__global__ void k_ntt_final_contig(NttPlan pl, fe* vh, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t Cn = 1u << Llo;
    const uint32_t b = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, bf = threadIdx.x / P;
    Tile t{smem, Cn * P};
    for (uint32_t e = bf; e < Cn; e += Cn / 2) {
        const size_t idx = ((size_t)b << Llo) + e;
        t.put(e, q, F::load(vh + idx * batch + q0 + q));
    }
    __syncthreads();
    dif_run<false>(t, bf, q, Lhi, L, 0, L, Llo, b, pl);
    for (uint32_t e = bf; e < Cn; e += Cn / 2) {
        const size_t idx = ((size_t)b << Llo) + e;
        F::store(vh + idx * batch + q0 + q, F::mul(t.get(e, q), F::load(pl.scale_out + idx)));
    }
}

}  // namespace

void launch_compute_h(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s) {
    const int L = p.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const unsigned G = 1u << Lhi, Cn = 1u << Llo;
    const unsigned pb = (unsigned)(batch / P);
    const size_t lds_s = (size_t)G * P * 36, lds_c = (size_t)Cn * P * 36;      // nine 32-bit limb planes per element
    if (lds_s > 65536) {       // 2^17 domains (AES-V2): 72 KiB tiles need the opt-in LDS limit (a CU has 160 KiB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_dif_strided), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pointwise_strided), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    }
    hipLaunchKernelGGL(k_ntt_dif_strided, dim3(Cn, pb, 3), dim3(G / 2 * P), lds_s, s, p, a, b, c, m, batch);
    hipLaunchKernelGGL(k_ntt_mid_contig, dim3(G, pb, 3), dim3(Cn / 2 * P), lds_c, s, p, a, b, c, batch);
    hipLaunchKernelGGL(k_ntt_pointwise_strided, dim3(Cn, pb, 1), dim3(G / 2 * P), lds_s, s, p, a, b, c, batch);
    hipLaunchKernelGGL(k_ntt_final_contig, dim3(G, pb, 1), dim3(Cn / 2 * P), lds_c, s, p, a, batch);
}

}  // namespace gsc
