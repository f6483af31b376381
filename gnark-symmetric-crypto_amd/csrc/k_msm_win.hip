// Fixed-base multi-scalar multiplication with ONE table per base and one accumulator per window.
//
// Replaces (*G1Jac).MultiExp / (*G2Jac).MultiExp of groth16.Prove (reference libraries/prover/impl/provers.go:148,216;
// SURVEY.md §8(a) a9/a10, algebra App. D) for a batch of independent proofs that share the proving key.
//
//   T[k][d-1] = d * P_k,  d = 1 .. D = 2^(c-1)                      (InitAlgorithm, k_build_base_table below)
//   s_k = sum_j e_kj 2^(c j),  e_kj in [-D, D-1]                     (k_recode: signed digits, once per batch)
//   S_j = sum_k sign(e_kj) T[k][|e_kj|-1]                            (k_msm_win: one wave per (slice of bases, window j, 64 proofs))
//   sum_k s_k P_k = sum_j 2^(c j) S_j                                (k_msm_horner: c doublings + one addition per window)
//
// The same number of additions as a table per (base, window), with nwin times less HBM, which is spent on wider digits
// instead (fewer windows = fewer additions).  Lanes of a wave are 64 proofs working on the same base, so digit loads are
// coalesced and the table gathers of a wave fall into one row of D * 64 bytes; all waves of a slice (every window, every
// group of proofs) read the same rows and are placed on ONE XCD so that they share them in its L2.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// |s| <= (r-1)/2 after sign normalisation; returns true when the point must be negated
__device__ __forceinline__ bool sign_normalise(fe& s) {
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

// ---- signed-digit recoding -------------------------------------------------------------------------------------------------
// One thread per (octet of bases, proof): eight scalars -> for every window one 16-byte word of eight int16 digits.
// digits[(j * noct + o) * batch + p] = {e_{8o,j}, ..., e_{8o+7,j}} of proof p.  MONT: the scalars are Montgomery residues of wire
// values (sign-normalised first: wires are mostly tiny or -tiny); otherwise canonical integers below r (the quotient h).
// The digit of a negated scalar is negated here, so the MSM kernel only ever sees sign(e) and |e|; the split threshold moves by
// one for negated scalars so that every stored digit lies in [-D, D-1] (int16 also for c = 16).
template <bool MONT>
__global__ __launch_bounds__(64) void k_recode(MsmRecodeArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x, o = blockIdx.y;
    const size_t noct = (a.nbases + 7) / 8;
    fe s[8]; uint32_t neg = 0, carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t k = 8 * o + i;
        if (k < a.nbases) {                                         // wave-uniform
            const size_t row = a.rows ? uni(a.rows[k]) : k;
            s[i] = load_fe(a.scalars + row * a.batch + p);
            if (MONT) { s[i] = Fr::from_mont(s[i]); if (sign_normalise(s[i])) neg |= 1u << i; }
        } else s[i] = fe{};
    }
    const uint32_t c = (uint32_t)a.c, cmask = (1u << c) - 1, D = 1u << (c - 1);
    for (int j = 0; j < a.nwin; j++) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t raw = (s[i].l[0] & cmask) + ((carry >> i) & 1u);
#pragma unroll
            for (int q = 0; q < 7; q++) s[i].l[q] = __builtin_amdgcn_alignbit(s[i].l[q + 1], s[i].l[q], c);
            s[i].l[7] >>= c;
            const bool ng = (neg >> i) & 1u;
            int32_t d = (int32_t)raw;
            if (raw >= D + (ng ? 1u : 0u)) { d -= (int32_t)(1u << c); carry |= 1u << i; } else carry &= ~(1u << i);
            if (ng) d = -d;
            w[i >> 1] |= ((uint32_t)d & 0xFFFFu) << (16 * (i & 1));
        }
        a.digits[((size_t)j * noct + o) * a.batch + p] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- gather-accumulate -------------------------------------------------------------------------------------------------------
template <class F> struct RawAff { fe w[2 * F::WORDS]; };
template <class F> __device__ __forceinline__ RawAff<F> load_raw(const fe* p) {
    RawAff<F> r;
#pragma unroll
    for (int i = 0; i < 2 * F::WORDS; i++) r.w[i] = load_fe(p + i);
    return r;
}
__device__ __forceinline__ Aff9<Fp29f> unpack_aff(const RawAff<Fp29f>& r, bool negate) {
    Aff9<Fp29f> e{Fp29::unpack(r.w[0]), Fp29::unpack(r.w[1])};
    if (negate) e.y = Fp29::neg(e.y);                      // signed-tight: fine as a product operand
    return e;
}
__device__ __forceinline__ Aff9<Fp2x> unpack_aff(const RawAff<Fp2x>& r, bool negate) {
    Aff9<Fp2x> e{fe9x2{Fp29::unpack(r.w[0]), Fp29::unpack(r.w[1])}, fe9x2{Fp29::unpack(r.w[2]), Fp29::unpack(r.w[3])}};
    if (negate) e.y = Fp2x::neg(e.y);
    return e;
}

// Bases [k0, k1) of window j for one proof (lane).  EXACT = false is the hot path (no degenerate-case tests inside madd).
// Software pipelining: the digits of the NEXT octet of bases and the table entry of the NEXT base are requested before the
// current mixed addition (~2 300 instructions) starts, so neither the coalesced digit stream nor the 64-byte random gathers
// are on the critical path.
template <class F, bool EXACT>
__device__ __forceinline__ Xyzz9<F> accumulate_window(const MsmWinArgs& a, size_t k0, size_t k1, uint32_t j, size_t p) {
    using C = Curve9<F>;
    const size_t noct = (a.nbases + 7) / 8, D = (size_t)1 << (a.c - 1);
    const fe* table = reinterpret_cast<const fe*>(a.table);
    const uint4* dig = a.digits + ((size_t)j * noct + k0 / 8) * a.batch + p;
    Xyzz9<F> acc = C::infinity();
    RawAff<F> pend = {}; int32_t dp = 0;                    // table entry fetched for the previous base, its digit (0: none)
    uint4 cur = *dig;
    for (size_t kk = k0; kk < k1; kk += 8) {
        dig += a.batch;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (kk + 8 < k1) nxt = *dig;
        const uint64_t lo = (uint64_t)cur.x | ((uint64_t)cur.y << 32), hi = (uint64_t)cur.z | ((uint64_t)cur.w << 32);
        const uint32_t lim = k1 - kk < 8 ? (uint32_t)(k1 - kk) : 8u;
#pragma unroll 1
        for (uint32_t i = 0; i < lim; i++) {
            const int32_t d = (int32_t)(int16_t)(uint16_t)(((i & 4) ? hi : lo) >> (16 * (i & 3)));
            // The gather is unconditional (a zero digit fetches entry 0 and drops it): a load inside a branch would have to be
            // waited for at the join, i.e. before the addition it is meant to overlap with.
            const int32_t mag = d < 0 ? -d : d;
            const RawAff<F> e = load_raw<F>(table + ((kk + i) * D + (size_t)(mag && !a.exp_same_entry ? mag - 1 : 0)) * (2 * F::WORDS));
            if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
            pend = e; dp = d;
        }
        cur = nxt;
    }
    if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
    return acc;
}

// grid: nslices * nwin * (batch / 64) workgroups of one wave.  partial[(slice * nwin + j) * batch + p]
template <class F>
__global__ __launch_bounds__(64, F::WORDS == 1 ? 3 : 1) void k_msm_win(MsmWinArgs a) {      // G1: three waves per SIMD (<= 168 VGPRs)
    using C = Curve9<F>;
    // XCD-aware order: workgroups go round-robin over the 8 XCDs by linear id and each XCD has its own L2.  Every wave of a slice
    // (all windows, all groups of proofs) gathers from the same table rows, so a slice is placed on ONE XCD (consecutive ids there).
    const size_t G = a.batch / 64, GW = G * (size_t)a.nwin, L = blockIdx.x, S8 = a.nslices & ~(size_t)7;
    size_t slice, rem;
    if (a.placement == 1 && (G & 3) == 0 && a.nslices >= 2) {
        // Two slices at a time, each shared by FOUR XCDs that take a quarter of its groups of proofs: the four gather from the same
        // rows at the same time, so a row comes from HBM once and from the Infinity Cache three times.
        const size_t S2 = a.nslices & ~(size_t)1, Q = GW / 4;
        if (L < S2 * GW) { const size_t xcd = L & 7, i = L >> 3; slice = (i / Q) * 2 + (xcd >> 2); rem = (xcd & 3) * Q + i % Q; }
        else { slice = L / GW; rem = L % GW; }
    } else if (L < S8 * GW) { const size_t xcd = L & 7, i = L >> 3; slice = (i / GW) * 8 + xcd; rem = i % GW; }
    else { slice = L / GW; rem = L % GW; }
    const uint32_t j = (uint32_t)(rem / G);
    const size_t p = (rem % G) * 64 + threadIdx.x;
    const size_t k0 = slice * a.per < a.nbases ? slice * a.per : a.nbases, k1 = k0 + a.per < a.nbases ? k0 + a.per : a.nbases;
    Xyzz9<F> acc = C::infinity();
    if (k0 < k1) {
        acc = accumulate_window<F, false>(a, k0, k1, j, p);
        // A degenerate step (accumulator == +-entry) zeroes ZZ for good; it cannot be told from a genuine point at infinity
        // without the exact tests, so the (very rare) lane is recomputed with them.
        if (!acc.inf && F::is_zero(acc.zz)) acc = accumulate_window<F, true>(a, k0, k1, j, p);
    }
    C::store_xyzz(reinterpret_cast<fe*>(a.partial) + ((slice * a.nwin + j) * a.batch + p) * (4 * F::WORDS), acc);
}

// out[p] = sum_j 2^(c j) S[j][p]: Horner from the top window, lanes = proofs.  254 doublings per proof whatever the width.
template <class F>
__global__ __launch_bounds__(64) void k_msm_horner(const fe* S, int nwin, int c, size_t batch, fe* out) {
    using C = Curve9<F>;
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    Xyzz9<F> r = C::load_xyzz(S + ((size_t)(nwin - 1) * batch + p) * (4 * F::WORDS));
    for (int j = nwin - 2; j >= 0; j--) {
#pragma unroll 1
        for (int q = 0; q < c; q++) r = C::dbl(r);
        r = C::add(r, C::load_xyzz(S + ((size_t)j * batch + p) * (4 * F::WORDS)));
    }
    C::store_xyzz(out + p * (4 * F::WORDS), r);
}

// ---- InitAlgorithm: T[k][d-1] = d * P_k ----------------------------------------------------------------------------------------
// Montgomery value of the 8 x 32-bit domain (R = 2^256, what the decompression kernels produce) -> radix-2^29 domain (R' = 2^261)
__device__ __forceinline__ fe9 to_fp29(const fe& old_mont) { return Fp29::to_mont(Fp29::unpack(Fp::from_mont(old_mont))); }
__device__ __forceinline__ Aff9<Fp29f> base_to_fp29(const Aff<Fp>* b) { return Aff9<Fp29f>{to_fp29(b->x), to_fp29(b->y)}; }
__device__ __forceinline__ Aff9<Fp2x> base_to_fp29(const Aff<Fp2>* b) {
    return Aff9<Fp2x>{fe9x2{to_fp29(b->x.a0), to_fp29(b->x.a1)}, fe9x2{to_fp29(b->y.a0), to_fp29(b->y.a1)}};
}

// One thread per segment of `seg` consecutive multiples of one base: the first multiple by double-and-add, the rest by mixed
// additions, kept in XYZZ in `scratch` with the running product of the ZZZ parked in the table slots; one inversion per
// segment turns them into affine entries (Montgomery batch inversion).  Threads [t0, t0 + nthreads) of the (base, segment)
// grid are processed by one launch; scratch holds nthreads * seg points.
template <class F, class OldF>
__global__ __launch_bounds__(64) void k_build_base_table(const Aff<OldF>* bases, size_t t0, size_t nthreads, int c, uint32_t seg, fe* table, fe* scratch) {
    using C = Curve9<F>;
    using E = typename F::E;
    constexpr int CW = F::WORDS;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    const size_t D = (size_t)1 << (c - 1), segs = D / seg, id = t0 + t, k = id / segs, q = id % segs;
    const Aff9<F> P = base_to_fp29(bases + k);
    const uint32_t d0 = (uint32_t)(q * seg + 1);
    Xyzz9<F> Ed = C::infinity();
    for (int b = 31 - __clz(d0); b >= 0; b--) { Ed = C::dbl(Ed); if ((d0 >> b) & 1u) Ed = C::template madd<true>(Ed, P); }
    fe* out = table + (k * D + (d0 - 1)) * (2 * CW);
    fe* sc = scratch + t * seg * (4 * CW);
    E prefix = F::one();
    for (uint32_t d = 0; d < seg; d++) {
        if (d) Ed = C::template madd<true>(Ed, P);
        C::store_xyzz(sc + d * (4 * CW), Ed);
        F::store(out + d * (2 * CW), prefix);                       // product of the ZZZ before this entry
        prefix = F::mul(prefix, Ed.zzz);
    }
    E inv = F::inv(prefix);
    for (uint32_t d = seg; d-- > 0;) {
        Ed = C::load_xyzz(sc + d * (4 * CW));
        const E pre = F::load(out + d * (2 * CW));
        const E izzz = F::mul(inv, pre);                             // 1 / ZZZ_d
        inv = F::mul(inv, Ed.zzz);
        const E izz = F::mul(F::sqr(Ed.zz), F::sqr(izzz));          // 1/ZZ = ZZ^2 / ZZZ^2
        Aff9<F> e; e.x = F::mul(Ed.x, izz); e.y = F::mul(Ed.y, izzz);
        C::store_aff(out + d * (2 * CW), e);
    }
}

}  // namespace

void launch_msm_recode(const MsmRecodeArgs& a, hipStream_t s) {
    if (!a.nbases) return;
    const dim3 grid((unsigned)(a.batch / 64), (unsigned)((a.nbases + 7) / 8));
    if (a.mont) hipLaunchKernelGGL(k_recode<true>, grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_recode<false>, grid, dim3(64), 0, s, a);
}
void launch_msm_win_g1(const MsmWinArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_win<Fp29f>, dim3((unsigned)(a.nslices * a.nwin * (a.batch / 64))), dim3(64), 0, s, a);
}
void launch_msm_win_g2(const MsmWinArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_win<Fp2x>, dim3((unsigned)(a.nslices * a.nwin * (a.batch / 64))), dim3(64), 0, s, a);
}
void launch_msm_horner_g1(const G1Xyzz* S, int nwin, int c, size_t batch, G1Xyzz* out, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_horner<Fp29f>, dim3((unsigned)(batch / 64)), dim3(64), 0, s, reinterpret_cast<const fe*>(S), nwin, c, batch, reinterpret_cast<fe*>(out));
}
void launch_msm_horner_g2(const G2Xyzz* S, int nwin, int c, size_t batch, G2Xyzz* out, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_horner<Fp2x>, dim3((unsigned)(batch / 64)), dim3(64), 0, s, reinterpret_cast<const fe*>(S), nwin, c, batch, reinterpret_cast<fe*>(out));
}
void launch_build_base_table_g1(const G1Aff* bases, size_t t0, size_t nthreads, int c, uint32_t seg, G1Aff* table, G1Xyzz* scratch, hipStream_t s) {
    if (nthreads) hipLaunchKernelGGL((k_build_base_table<Fp29f, Fp>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s,
                                     reinterpret_cast<const Aff<Fp>*>(bases), t0, nthreads, c, seg, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch));
}
void launch_build_base_table_g2(const G2Aff* bases, size_t t0, size_t nthreads, int c, uint32_t seg, G2Aff* table, G2Xyzz* scratch, hipStream_t s) {
    if (nthreads) hipLaunchKernelGGL((k_build_base_table<Fp2x, Fp2>), dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, s,
                                     reinterpret_cast<const Aff<Fp2>*>(bases), t0, nthreads, c, seg, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch));
}

}  // namespace gsc
