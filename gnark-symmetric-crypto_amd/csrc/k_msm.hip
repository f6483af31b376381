// Slice reductions of the MSMs, wire classification (calibration) and proof assembly.
//
// The gather-accumulate kernels themselves are in k_msm_win.hip; this file holds what follows them in groth16.Prove
// (reference libraries/prover/impl/provers.go:148,216; SURVEY.md §8(a) a9 msmReduceChunk, a12 assembly, App. D).
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {


// |s| <= (r-1)/2 after sign normalisation; returns true when the point must be negated
__device__ __forceinline__ bool sign_normalise(fe& s) {
    // (r-1)/2
    bool gt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const uint32_t lo = FrParams::mod(i) - (i == 0 ? 1u : 0u);
        const uint32_t hi = i < 7 ? FrParams::mod(i + 1) : 0u;
        const uint32_t h = (lo >> 1) | (hi << 31);
        if (!decided && s.l[i] != h) { gt = s.l[i] > h; decided = true; }
    }
    if (gt) {
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)FrParams::mod(i) - s.l[i] - br; s.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
    }
    return gt;
}

__device__ __forceinline__ fe9 shfl_xor_e(const fe9& v, int m) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor(v.l[i], m);
    return r;
}
__device__ __forceinline__ fe9x2 shfl_xor_e(const fe9x2& v, int m) { return fe9x2{shfl_xor_e(v.a0, m), shfl_xor_e(v.a1, m)}; }

// one wave sums up to 64 slices of one proof: lanes = slices, butterfly over __shfl_xor
template <class F>
__global__ __launch_bounds__(64) void k_msm_reduce(const fe* partial, size_t nslices, size_t batch, fe* out) {
    using C = Curve9<F>;
    const size_t p = blockIdx.x, grp = blockIdx.y;
    const size_t slice = grp * 64 + threadIdx.x;
    Xyzz9<F> v = slice < nslices ? C::load_xyzz(partial + (slice * batch + p) * (4 * F::WORDS)) : C::infinity();
    for (int m = 32; m >= 1; m >>= 1) {
        Xyzz9<F> o;
        o.x = shfl_xor_e(v.x, m); o.y = shfl_xor_e(v.y, m); o.zz = shfl_xor_e(v.zz, m); o.zzz = shfl_xor_e(v.zzz, m);
        o.inf = __shfl_xor((int)v.inf, m) != 0;
        v = C::add(v, o);
    }
    if (threadIdx.x == 0) C::store_xyzz(out + (grp * batch + p) * (4 * F::WORDS), v);
}

// the same sum with lanes = proofs: every lane adds up to MSM_REDUCE_FANIN slice partials of ITS proof one after the other.  No lane
// idles (the butterfly above keeps 64 lanes busy for 63 useful additions out of 384), loads are coalesced; used whenever the
// batch is large enough to fill the chip this way (launch_msm_reduce picks).
template <class F>
__global__ __launch_bounds__(64) void k_msm_reduce_seq(const fe* partial, size_t nslices, size_t batch, fe* out) {
    using C = Curve9<F>;
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, grp = blockIdx.y;
    const size_t s0 = grp * MSM_REDUCE_FANIN, s1 = s0 + MSM_REDUCE_FANIN < nslices ? s0 + MSM_REDUCE_FANIN : nslices;
    Xyzz9<F> v = C::load_xyzz(partial + (s0 * batch + p) * (4 * F::WORDS));
    for (size_t sl = s0 + 1; sl < s1; sl++) v = C::add(v, C::load_xyzz(partial + (sl * batch + p) * (4 * F::WORDS)));
    C::store_xyzz(out + (grp * batch + p) * (4 * F::WORDS), v);
}

// ---- calibration: what do the wires of this circuit look like? (engine.hip calibrate) ----
__global__ __launch_bounds__(64) void k_classify_wires(const fe* W, size_t n_wires, size_t batch, const uint32_t* status, uint8_t* cls) {
    const size_t w = blockIdx.x;                 // one wave per wire, lanes = proofs (strided over the batch)
    if (w >= n_wires) return;
    int worst = 0;
    for (size_t p = threadIdx.x; p < batch; p += 64) {
        if (status[p] != 0xFFFFFFFFu) continue;  // a statement the solver rejected says nothing
        fe s = load_fe(W + w * batch + p);
        uint32_t z = 0, o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { z |= s.l[i]; o |= s.l[i] ^ FrParams::one(i); }
        if (z == 0 || o == 0) continue;          // 0 or 1
        s = Fr::from_mont(s);
        (void)sign_normalise(s);
        int top = 0;
#pragma unroll
        for (int i = 7; i >= 0; i--) if (!top && s.l[i]) top = 32 * i + 32 - __clz(s.l[i]);
                                                 // (a value that is not 0/1 but has magnitude 1 is -1: class 1)
        worst = top > worst ? top : worst;
    }
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(worst, m); worst = o > worst ? o : worst; }
    if (threadIdx.x == 0) cls[w] = (uint8_t)worst;
}

// ---- proof assembly ----
__device__ __forceinline__ void store_canon(uint8_t* dst, const fe9& mont) {      // canonical integer, 8 little-endian words
    const fe c = Fp29::pack(Fp29::from_mont(mont));
    uint32_t* q = reinterpret_cast<uint32_t*>(dst);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = c.l[i];
}

// role 0: Ar = sumA, t0 = s * Ar.  role 1: Bs1 = sumB1, t1 = r * Bs1.   tmp[role][proof]
__global__ __launch_bounds__(64) void k_fin_scalarmul(const G1Xyzz* sumA, const G1Xyzz* sumB1, const uint8_t* rs, size_t batch,
                                                       G1Xyzz* tmp, uint8_t* out, uint8_t* flags) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    const int role = blockIdx.y;
    if (p >= batch) return;
    const fe* src = reinterpret_cast<const fe*>(role == 0 ? sumA : sumB1) + 4 * p;
    Xyzz9<Fp29f> P = G1x::load_xyzz(src);
    const uint32_t* sc = reinterpret_cast<const uint32_t*>(rs + 64 * p) + (role == 0 ? 8 : 0);   // role 0 uses s, role 1 uses r
    Xyzz9<Fp29f> acc = G1x::infinity();
    if (!P.inf) {
        Aff9<Fp29f> A = G1x::to_aff(P);
        if (role == 0) { store_canon(out + 256 * p, A.x); store_canon(out + 256 * p + 32, A.y); }
        for (int i = 253; i >= 0; i--) {
            acc = G1x::dbl(acc);
            if ((sc[i >> 5] >> (i & 31)) & 1u) acc = G1x::madd<true>(acc, A);
        }
    } else if (role == 0) {
        atomicOr(reinterpret_cast<unsigned int*>(flags) + (p >> 2), 1u << (8 * (p & 3)));
    }
    G1x::store_xyzz(reinterpret_cast<fe*>(tmp) + ((size_t)role * batch + p) * 4, acc);
}
__global__ __launch_bounds__(64) void k_fin_combine(const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ, const G1Xyzz* tmp,
                                                     size_t batch, uint8_t* out, uint8_t* flags) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= batch) return;
    const int role = blockIdx.y;
    uint32_t fl = 0;
    if (role == 0) {
        const fe* K = reinterpret_cast<const fe*>(sumK); const fe* Z = reinterpret_cast<const fe*>(sumZ); const fe* T = reinterpret_cast<const fe*>(tmp);
        Xyzz9<Fp29f> v = G1x::add(G1x::add(G1x::load_xyzz(K + 4 * p), G1x::load_xyzz(Z + 4 * p)), G1x::add(G1x::load_xyzz(T + 4 * p), G1x::load_xyzz(T + 4 * (batch + p))));
        if (v.inf) fl |= 4;
        else { Aff9<Fp29f> A = G1x::to_aff(v); store_canon(out + 256 * p + 192, A.x); store_canon(out + 256 * p + 224, A.y); }
    } else {
        Xyzz9<Fp2x> v = G2x::load_xyzz(reinterpret_cast<const fe*>(sumB2) + 8 * p);
        if (v.inf) fl |= 2;
        else {
            Aff9<Fp2x> A = G2x::to_aff(v);
            store_canon(out + 256 * p + 64, A.x.a0); store_canon(out + 256 * p + 96, A.x.a1);
            store_canon(out + 256 * p + 128, A.y.a0); store_canon(out + 256 * p + 160, A.y.a1);
        }
    }
    if (fl) atomicOr(reinterpret_cast<unsigned int*>(flags) + (p >> 2), fl << (8 * (p & 3)));
}

__global__ void k_points_to_affine_be(const G1Xyzz* points, size_t batch, uint8_t* out, uint8_t* flags, uint32_t bit) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= batch) return;
    Xyzz9<Fp29f> v = G1x::load_xyzz(reinterpret_cast<const fe*>(points) + 4 * p);
    uint8_t* o = out + 64 * p;
    if (v.inf) { for (int i = 0; i < 64; i++) o[i] = 0; atomicOr(reinterpret_cast<unsigned int*>(flags) + (p >> 2), bit << (8 * (p & 3))); return; }
    Aff9<Fp29f> A = G1x::to_aff(v);
    const fe x = Fp29::pack(Fp29::from_mont(A.x)), y = Fp29::pack(Fp29::from_mont(A.y));
    for (int i = 0; i < 8; i++) {
        const uint32_t xw = x.l[7 - i], yw = y.l[7 - i];
        o[4 * i] = (uint8_t)(xw >> 24); o[4 * i + 1] = (uint8_t)(xw >> 16); o[4 * i + 2] = (uint8_t)(xw >> 8); o[4 * i + 3] = (uint8_t)xw;
        o[32 + 4 * i] = (uint8_t)(yw >> 24); o[32 + 4 * i + 1] = (uint8_t)(yw >> 16); o[32 + 4 * i + 2] = (uint8_t)(yw >> 8); o[32 + 4 * i + 3] = (uint8_t)yw;
    }
}
// 384-bit big-endian integer mod r by Horner over bytes (gnark-crypto fr.Hash reduces the 48 xmd bytes the same way)
__global__ void k_challenge_from_hash(const uint8_t* h48, fe* commit, size_t batch) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= batch) return;
    const uint8_t* h = h48 + 48 * p;
    fe acc = Fr::zero();
    const fe b256 = Fr::from_u32(256);
    for (int i = 0; i < 48; i++) acc = Fr::add(Fr::mul(acc, b256), Fr::from_u32(h[i]));
    store_fe(commit + p, acc);
}

}  // namespace

// returns the number of partial sums left per proof (1 = done)
template <class F>
static size_t launch_msm_reduce(const fe* partial, size_t nslices, size_t batch, fe* out, hipStream_t s) {
    const size_t groups = msm_reduce_groups(nslices, batch);
    if (msm_reduce_by_proof(nslices, batch)) hipLaunchKernelGGL(k_msm_reduce_seq<F>, dim3((unsigned)(batch / 64), (unsigned)groups), dim3(64), 0, s, partial, nslices, batch, out);
    else hipLaunchKernelGGL(k_msm_reduce<F>, dim3((unsigned)batch, (unsigned)groups), dim3(64), 0, s, partial, nslices, batch, out);
    return groups;
}
size_t launch_msm_reduce_g1(const G1Xyzz* partial, size_t nslices, size_t batch, G1Xyzz* out, hipStream_t s) {
    return launch_msm_reduce<Fp29f>(reinterpret_cast<const fe*>(partial), nslices, batch, reinterpret_cast<fe*>(out), s);
}
size_t launch_msm_reduce_g2(const G2Xyzz* partial, size_t nslices, size_t batch, G2Xyzz* out, hipStream_t s) {
    return launch_msm_reduce<Fp2x>(reinterpret_cast<const fe*>(partial), nslices, batch, reinterpret_cast<fe*>(out), s);
}
void launch_classify_wires(const fe* W, size_t n_wires, size_t batch, const uint32_t* status, uint8_t* cls, hipStream_t s) {
    if (n_wires) hipLaunchKernelGGL(k_classify_wires, dim3((unsigned)n_wires), dim3(64), 0, s, W, n_wires, batch, status, cls);
}
void launch_points_to_affine_be(const G1Xyzz* points, size_t batch, uint8_t* out, uint8_t* flags, uint32_t bit, hipStream_t s) {
    hipLaunchKernelGGL(k_points_to_affine_be, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, points, batch, out, flags, bit);
}
void launch_challenge_from_hash(const uint8_t* h48, fe* commit, size_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_challenge_from_hash, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, h48, commit, batch);
}
void launch_fin_scalarmul(const G1Xyzz* sumA, const G1Xyzz* sumB1, const uint8_t* rs, size_t batch, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s) {
    hipLaunchKernelGGL(k_fin_scalarmul, dim3((unsigned)((batch + 63) / 64), 2), dim3(64), 0, s, sumA, sumB1, rs, batch, tmp, out, flags);
}
void launch_fin_combine(const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ, const G1Xyzz* tmp, size_t batch, uint8_t* out, uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_fin_combine, dim3((unsigned)((batch + 63) / 64), 2), dim3(64), 0, s, sumB2, sumK, sumZ, tmp, batch, out, flags);
}

}  // namespace gsc
