// Fixed-base multi-scalar multiplication for a batch of independent proofs that share the proving key.
//
// Replaces (*G1Jac).MultiExp / (*G2Jac).MultiExp of groth16.Prove (reference libraries/prover/impl/provers.go:148,216;
// SURVEY.md §8(a) a9/a10, algebra App. D).  Bases are fixed for the life of the process, so every base has ONE table row
// T_k[d-1] = d * P_k in HBM and an MSM is a gather-accumulate; lanes of a wave are 64 proofs working on the same base, so scalar
// digits are read coalesced and the gathers of a wave fall into one row.
//
// Full-width scalars (the quotient h over pk.G1.Z; the few wide wires of a wire set) — "windowed":
//   s_k = sum_j e_kj 2^(c j),  e_kj in [-D, D-1], D = 2^(c-1)          (k_recode: signed digits, once per batch)
//   S_j = sum_k sign(e_kj) T_k[|e_kj|-1]                                (k_msm_win: one wave per (slice of bases, window j, 64 proofs))
//   sum_k s_k P_k = sum_j 2^(c j) S_j                                   (k_msm_horner: c doublings + one addition per window)
// The same number of additions as a table per (base, window) with nwin times less HBM, which is spent on wider digits instead
// (fewer windows = fewer additions).  All waves of a slice (every window, every group of proofs) read the same rows and are
// placed on ONE XCD so that they share them in its L2.
//
// Small scalars (nearly every wire of these circuits: bits, values in {-1, 0, 1}, bytes) — "flat":
//   one signed 16-bit value per (base, proof), one accumulator, one wave per (slice of bases, 64 proofs)     (k_recode_flat, k_msm_flat)
//   eight ternary wires at a time through a table of their signed subset sums; rows only as long as the wire was seen to need.
// Both are predictions made at InitAlgorithm and checked for every wave of proofs: a scalar that does not fit is multiplied out
// by double-and-add, so results never depend on the prediction.
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

// ---- signed-digit recoding (windowed sets) --------------------------------------------------------------------------------------
// One thread per (octet of bases, proof): eight scalars -> for every window one 16-byte word of eight int16 digits.
// digits[(j * noct + o) * batch + p] = {e_{8o,j}, ..., e_{8o+7,j}} of proof p.  MONT: the scalars are Montgomery residues of wire
// values (sign-normalised first: wires are mostly tiny or -tiny); otherwise canonical integers below r (the quotient h).
// The digit of a negated scalar is negated here, so the MSM kernel only ever sees sign(e) and |e|; the split threshold moves by
// one for negated scalars so that every stored digit lies in [-D, D-1] (int16 also for c = 16).  WIDE (c = 17: digits of up to 17 bits + sign):
// eight int32 per octet, two 16-byte words side by side — digits[2 * index] and digits[2 * index + 1].
template <bool MONT, bool WIDE>
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
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
            if (WIDE) w[i] = (uint32_t)d; else w[i >> 1] |= ((uint32_t)d & 0xFFFFu) << (16 * (i & 1));
        }
        const size_t at = ((size_t)j * noct + o) * a.batch + p;
        if (WIDE) { a.digits[2 * at] = make_uint4(w[0], w[1], w[2], w[3]); a.digits[2 * at + 1] = make_uint4(w[4], w[5], w[6], w[7]); }
        else a.digits[at] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- recoding of a flat set ------------------------------------------------------------------------------------------------------
// One thread per (octet of bases, proof) -> ONE 16-byte word of eight int16 values, digits[o * batch + p].  Three kinds of octets:
//  * bit groups (o < nbit / 8): the eight wires are predicted to carry scalars in {-1, 0, 1}; they form a balanced-ternary number
//    v = sum t_i 3^i and ONE addition of the tabulated signed subset sum replaces up to eight.  Every wave checks the prediction
//    (all eight scalars ternary for all 64 proofs; group_ok[o]: the group's subset sums could be tabulated) and says so in
//    gok[o][wave]: 1 = slot 0 carries v; 0 = ordinary values, the eight bases are walked one by one;
//  * ordinary octets: slot i = the sign-normalised scalar of base 8o + i if it fits 15 bits, else MSM_FLAT_ESCAPE (the MSM kernel
//    then multiplies that scalar out by double-and-add: a wrong prediction costs time, never correctness);
//  * window octets (octwin[o] >= 0): slots = the signed c-bit digits octwin[o] .. octwin[o] + 7 of the ONE scalar rows[8o]: a wide
//    wire whose (base, window) pairs were laid out as bases of their own, with the points 2^(c j) P (tiny wide sets: no Horner pass).
// FEW = false: one wave per (64 proofs, octet), o wave-uniform, gok per (octet, wave).  FEW = true (calls with a handful of statements): lanes are
// octets of ONE proof (blockIdx.y), gok per (octet, proof): gok[o * MSM_FEW_PROOFS + p].
// A byte-plane entry (0, 1, -1) as the Montgomery image the generic solver would have stored
__device__ __forceinline__ fe fe_of_plane(int t) {
    const fe one = Fr::one(), mone = Fr::neg(one);
    fe r;
#pragma unroll
    for (int q = 0; q < 8; q++) r.l[q] = t > 0 ? one.l[q] : (t < 0 ? mone.l[q] : 0u);
    return r;
}
template <bool FEW>
__device__ __forceinline__ void recode_flat_octet(const MsmFlatRecodeArgs& a, size_t o, size_t p) {
    auto U = [](uint32_t v) { return FEW ? v : uni(v); };
    const int32_t win0 = a.octwin ? (int32_t)U((uint32_t)a.octwin[o]) : -1;
    // scalar row `row` of proof p: from the byte plane when the small-integer witness path left it there (tv: its entry), else the 32-byte element
    const int8_t* pp = a.plane ? a.plane + (p >> 6) * a.plane_stride * 64 + (p & 63) : nullptr;
    auto load_scalar = [&](uint32_t row, int& tv) {
        tv = (int)WS_PLANE_WIDE;
        if (pp && row < a.plane_rows) tv = (int)pp[(size_t)row * 64];
        return tv == (int)WS_PLANE_WIDE ? load_fe(a.scalars + (size_t)row * a.batch + p) : fe_of_plane(tv);
    };
    uint32_t w[4] = {0, 0, 0, 0};
    if (win0 >= 0) {
        int tv;
        fe s = load_scalar(U(a.rows[8 * o]), tv);
        if (a.mont) s = Fr::from_mont(s);
        const bool ng = sign_normalise(s);
        const uint32_t c = (uint32_t)a.c, cmask = (1u << c) - 1, D = 1u << (c - 1);
        uint32_t carry = 0;
        for (int32_t j = 0; j < win0 + 8; j++) {                    // digits below win0 are recomputed for their carry only
            uint32_t raw = (s.l[0] & cmask) + carry;
#pragma unroll
            for (int q = 0; q < 7; q++) s.l[q] = __builtin_amdgcn_alignbit(s.l[q + 1], s.l[q], c);
            s.l[7] >>= c;
            int32_t d = (int32_t)raw;
            if (raw >= D + (ng ? 1u : 0u)) { d -= (int32_t)(1u << c); carry = 1; } else carry = 0;
            if (ng) d = -d;
            if (j >= win0) { const int i = j - win0; w[i >> 1] |= ((uint32_t)d & 0xFFFFu) << (16 * (i & 1)); }
        }
        a.digits[o * a.batch + p] = make_uint4(w[0], w[1], w[2], w[3]);
        return;
    }
    if (pp && a.mont) {
        // Byte-plane rows: the values are already the small integers the recoding is after.  (Rows are wave-uniform, so is "all eight in the plane".)
        int tv[8]; bool all_plane = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const size_t k = 8 * o + i;
            const uint32_t row = k < a.nbases ? U(a.rows[k]) : 0xFFFFFFFFu;
            tv[i] = row < a.plane_rows ? (int)pp[(size_t)row * 64] : (int)WS_PLANE_WIDE;
            if (k >= a.nbases) tv[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) all_plane = all_plane && tv[i] != (int)WS_PLANE_WIDE;
        if (FEW ? all_plane : (bool)__all(all_plane)) {
            if (o < a.nbit / 8) {
                int32_t v = 0, w3 = 1;
#pragma unroll
                for (int i = 0; i < 8; i++) { v += tv[i] * w3; w3 *= 3; }
                const bool all_ok = U(a.group_ok[o]) != 0;
                if (FEW) a.gok[o * MSM_FEW_PROOFS + p] = all_ok ? 1 : 0;
                else if (threadIdx.x == 0) a.gok[o * (a.batch / 64) + blockIdx.x] = all_ok ? 1 : 0;
                if (all_ok) { a.digits[o * a.batch + p] = make_uint4((uint32_t)v & 0xFFFFu, 0, 0, 0); return; }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) w[i >> 1] |= ((uint32_t)tv[i] & 0xFFFFu) << (16 * (i & 1));
            a.digits[o * a.batch + p] = make_uint4(w[0], w[1], w[2], w[3]);
            return;
        }
    }
    fe s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t k = 8 * o + i;
        int tv;
        s[i] = k < a.nbases ? load_scalar(U(a.rows[k]), tv) : fe{};
    }
    if (o < a.nbit / 8) {
        const fe minus_one = Fr::neg(Fr::one());
        int32_t v = 0, w3 = 1; bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t z = 0, e1 = 0, m1 = 0;   // == 0, == 1, == -1 (Montgomery images)
#pragma unroll
            for (int q = 0; q < 8; q++) { z |= s[i].l[q]; e1 |= s[i].l[q] ^ FrParams::one(q); m1 |= s[i].l[q] ^ minus_one.l[q]; }
            ok = ok && (z == 0 || e1 == 0 || m1 == 0);
            v += e1 == 0 ? w3 : (m1 == 0 ? -w3 : 0);
            w3 *= 3;
        }
        const bool all_ok = (FEW ? ok : (bool)__all(ok)) && U(a.group_ok[o]) != 0;
        if (FEW) a.gok[o * MSM_FEW_PROOFS + p] = all_ok ? 1 : 0;
        else if (threadIdx.x == 0) a.gok[o * (a.batch / 64) + blockIdx.x] = all_ok ? 1 : 0;
        if (all_ok) { a.digits[o * a.batch + p] = make_uint4((uint32_t)v & 0xFFFFu, 0, 0, 0); return; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[i] = Fr::from_mont(s[i]);
        const bool ng = sign_normalise(s[i]);
        uint32_t hi = s[i].l[0] >> 15;
#pragma unroll
        for (int q = 1; q < 8; q++) hi |= s[i].l[q];
        const int32_t d = hi ? MSM_FLAT_ESCAPE : (ng ? -(int32_t)s[i].l[0] : (int32_t)s[i].l[0]);
        w[i >> 1] |= ((uint32_t)d & 0xFFFFu) << (16 * (i & 1));
    }
    a.digits[o * a.batch + p] = make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ __launch_bounds__(64) void k_recode_flat(MsmFlatRecodeArgs a) { recode_flat_octet<false>(a, blockIdx.y, (size_t)blockIdx.x * 64 + threadIdx.x); }
__global__ __launch_bounds__(64) void k_recode_flat_few(MsmFlatRecodeArgs a) {
    const size_t o = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (o < (a.nbases + 7) / 8) recode_flat_octet<true>(a, o, blockIdx.y);
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
// are on the critical path.  Uniform rows: entry d - 1 of base k is table[k * D + d - 1].
// digit i (wave-uniform i) of an octet: eight int16 in one 16-byte word, or (WIDE) eight int32 in two
template <bool WIDE> __device__ __forceinline__ int32_t octet_digit(const uint4& w0, const uint4& w1, uint32_t i) {
    if (WIDE) {
        const uint4& w = (i & 4) ? w1 : w0;
        return (int32_t)((i & 2) ? ((i & 1) ? w.w : w.z) : ((i & 1) ? w.y : w.x));
    }
    const uint64_t lo = (uint64_t)w0.x | ((uint64_t)w0.y << 32), hi = (uint64_t)w0.z | ((uint64_t)w0.w << 32);
    return (int32_t)(int16_t)(uint16_t)(((i & 4) ? hi : lo) >> (16 * (i & 3)));
}
template <class F, bool EXACT, bool WIDE>
__device__ __forceinline__ Xyzz9<F> accumulate_window(const MsmWinArgs& a, size_t k0, size_t k1, uint32_t j, size_t p) {
    using C = Curve9<F>;
    constexpr size_t DW = WIDE ? 2 : 1;                     // 16-byte words per (octet, proof)
    const size_t noct = (a.nbases + 7) / 8, D = (size_t)1 << (a.c - 1);
    const fe* table = reinterpret_cast<const fe*>(a.table);
    const uint4* dig = a.digits + (((size_t)j * noct + k0 / 8) * a.batch + p) * DW;
    Xyzz9<F> acc = C::infinity();
    RawAff<F> pend = {}; int32_t dp = 0;                    // table entry fetched for the previous base, its digit (0: none)
    uint4 cur = dig[0], cur1 = WIDE ? dig[1] : make_uint4(0, 0, 0, 0);
    for (size_t kk = k0; kk < k1; kk += 8) {
        dig += a.batch * DW;
        uint4 nxt = make_uint4(0, 0, 0, 0), nxt1 = make_uint4(0, 0, 0, 0);
        if (kk + 8 < k1) { nxt = dig[0]; if (WIDE) nxt1 = dig[1]; }
        const uint32_t lim = k1 - kk < 8 ? (uint32_t)(k1 - kk) : 8u;
#pragma unroll 1
        for (uint32_t i = 0; i < lim; i++) {
            const int32_t d = octet_digit<WIDE>(cur, cur1, i);
            // The gather is unconditional (a zero digit fetches entry 0 and drops it): a load inside a branch would have to be
            // waited for at the join, i.e. before the addition it is meant to overlap with.
            const int32_t mag = d < 0 ? -d : d;
            const size_t ent = a.exp_entry_mask ? (size_t)((uint32_t)(mag ? mag - 1 : 0) & a.exp_entry_mask) : (size_t)(mag ? mag - 1 : 0);      // (exp_entry_mask: timing experiment only)
            const RawAff<F> e = load_raw<F>(table + ((kk + i) * D + ent) * (2 * F::WORDS));
            if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
            pend = e; dp = d;
        }
        cur = nxt; cur1 = nxt1;
    }
    if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
    return acc;
}

// grid: nslices * nwin * (batch / 64) workgroups of one wave.  partial[(slice * nwin + j) * batch + p]
template <class F, bool WIDE>
__global__ __launch_bounds__(64, F::WORDS == 1 ? 3 : 1) void k_msm_win(MsmWinArgs a) {      // G1: three waves per SIMD (<= 168 VGPRs); measured in round 3: four (128 VGPRs, 34 spilled) 272.3 ms, two 272.9 ms — the same
    using C = Curve9<F>;
    // XCD-aware order: workgroups go round-robin over the 8 XCDs by linear id and each XCD has its own L2.  Every wave of a slice
    // (all windows, all groups of proofs) gathers from the same table rows, so a slice is placed on ONE XCD (consecutive ids there).
    const size_t G = a.batch / 64, GW = G * (size_t)a.nwin, L = blockIdx.x, S8 = a.nslices & ~(size_t)7;
    // clock stamps: eight waves spread over the launch (at 1/16, 3/16, ... of the grid), the k-th shifted by k workgroups so that each lands on another
    // XCD (workgroups go round-robin over the XCDs, which are clocked separately), each record {100 MHz clock, shader clock} at their start and end
    const uint32_t eighth = gridDim.x / 8, k8 = eighth ? blockIdx.x / eighth : 8u;
    const bool stamp = a.clk && eighth >= 16 && threadIdx.x == 0 && k8 < 8 && blockIdx.x % eighth == eighth / 2 + k8;
    unsigned long long* const clk = a.clk + (stamp ? 4 * k8 : 0);
    if (stamp) { clk[0] = wall_clock64(); clk[1] = clock64(); }
    size_t slice, rem;
    if (L < S8 * GW) { const size_t xcd = L & 7, i = L >> 3; slice = (i / GW) * 8 + xcd; rem = i % GW; }
    else { slice = L / GW; rem = L % GW; }
    // consecutive workgroups of a slice: same window, consecutive groups of proofs
    const uint32_t j = (uint32_t)(rem / G);
    const size_t p = (rem % G) * 64 + threadIdx.x;
    const size_t k0 = slice * a.per < a.nbases ? slice * a.per : a.nbases, k1 = k0 + a.per < a.nbases ? k0 + a.per : a.nbases;
    Xyzz9<F> acc = C::infinity();
    if (k0 < k1) {
        acc = accumulate_window<F, false, WIDE>(a, k0, k1, j, p);
        // A degenerate step (accumulator == +-entry) zeroes ZZ for good; it cannot be told from a genuine point at infinity
        // without the exact tests, so the (very rare) lane is recomputed with them.
        if (!acc.inf && F::is_zero(acc.zz)) acc = accumulate_window<F, true, WIDE>(a, k0, k1, j, p);
    }
    C::store_xyzz(reinterpret_cast<fe*>(a.partial) + ((slice * a.nwin + j) * a.batch + p) * (4 * F::WORDS), acc);
    if (stamp) { clk[2] = wall_clock64(); clk[3] = clock64(); }
}

// The latency path: a handful of proofs (a single Prove call).  With lanes = proofs a wave would do 64 additions per useful one, so
// here lanes are BASES: every lane fetches the entry of its own (base, window) for ONE proof, lanes add up 64-base chunks of the
// slice independently and a __shfl_xor butterfly of six exact additions folds the wave.  Same partial sums, ~10x less work for
// one proof; the crossover with the batch kernel is around ten proofs.  grid: (nslices * nwin, nproofs).
__device__ __forceinline__ fe9 shfl_xor_e(const fe9& v, int m) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor(v.l[i], m);
    return r;
}
__device__ __forceinline__ fe9x2 shfl_xor_e(const fe9x2& v, int m) { return fe9x2{shfl_xor_e(v.a0, m), shfl_xor_e(v.a1, m)}; }
template <class F>
__global__ __launch_bounds__(64) void k_msm_win_few(MsmWinArgs a) {
    using C = Curve9<F>;
    const size_t slice = blockIdx.x / (uint32_t)a.nwin, p = blockIdx.y, noct = (a.nbases + 7) / 8, D = (size_t)1 << (a.c - 1);
    const uint32_t j = blockIdx.x % (uint32_t)a.nwin, lane = threadIdx.x;
    const size_t k0 = slice * a.per < a.nbases ? slice * a.per : a.nbases, k1 = k0 + a.per < a.nbases ? k0 + a.per : a.nbases;
    const fe* table = reinterpret_cast<const fe*>(a.table);
    Xyzz9<F> acc = C::infinity();
    for (size_t kb = k0; kb < k1; kb += 64) {
        const size_t k = kb + lane;
        if (k < k1) {
            const size_t at = ((size_t)j * noct + k / 8) * a.batch + p;
            const uint32_t s = (uint32_t)(k & 7);
            int32_t d;
            if (a.c > 16) { const uint4 w = a.digits[2 * at + (s >> 2)]; d = (int32_t)((s & 2) ? ((s & 1) ? w.w : w.z) : ((s & 1) ? w.y : w.x)); }      // wide digits (k_recode)
            else { const uint4 w = a.digits[at]; const uint32_t word = s < 2 ? w.x : s < 4 ? w.y : s < 6 ? w.z : w.w; d = (int32_t)(int16_t)(uint16_t)(word >> (16 * (s & 1))); }
            if (d) {
                const int32_t mag = d < 0 ? -d : d;
                const Aff9<F> e = unpack_aff(load_raw<F>(table + (k * D + (size_t)(mag - 1)) * (2 * F::WORDS)), d < 0);
                Xyzz9<F> x = C::from_aff(Aff9<F>{e.x, F::norm(e.y)});
                acc = C::add(acc, x);
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        Xyzz9<F> o;
        o.x = shfl_xor_e(acc.x, m); o.y = shfl_xor_e(acc.y, m); o.zz = shfl_xor_e(acc.zz, m); o.zzz = shfl_xor_e(acc.zzz, m);
        o.inf = __shfl_xor((int)acc.inf, m) != 0;
        acc = C::add(acc, o);
    }
    if (lane == 0) C::store_xyzz(reinterpret_cast<fe*>(a.partial) + ((slice * a.nwin + j) * a.batch + p) * (4 * F::WORDS), acc);
}

// ---- flat sets ----------------------------------------------------------------------------------------------------------------------
// Bases [k0, k1) of a flat set for one proof (lane): see k_recode_flat for the three kinds of octets.  Rows have a length of their
// own (rowlen[k] multiples of P_k at entry rowoff[k]); a value beyond its row, or MSM_FLAT_ESCAPE, is multiplied out by
// double-and-add from the row's first entry.
template <class F, bool EXACT>
__device__ __forceinline__ Xyzz9<F> accumulate_flat(const MsmFlatArgs& a, size_t k0, size_t k1, size_t g, size_t p) {
    using C = Curve9<F>;
    const size_t G = a.batch / 64, o0 = k0 / 8, o1 = (k1 + 7) / 8, nbit8 = a.nbit / 8;
    const fe* table = reinterpret_cast<const fe*>(a.table);
    const fe* sub = reinterpret_cast<const fe*>(a.sub);
    const uint32_t lane = threadIdx.x;
    uint64_t okmask = 0;                                    // bit i: octet o0 + i is a group that passed k_recode_flat's check for this wave
    if (o0 < nbit8) { const size_t o = o0 + lane; const uint32_t f = (o < o1 && o < nbit8) ? a.gok[o * G + g] : 0u; okmask = __ballot(f != 0); }
    const uint4* dig = a.digits + o0 * a.batch + p;
    Xyzz9<F> acc = C::infinity();
    RawAff<F> pend = {}; int32_t dp = 0;
    uint4 cur = *dig;
    // row geometry of an octet's bases: lane i < 8 holds (offset, length) of base 8 o + i; fetched one octet ahead
    uint32_t roff_lo = 0, roff_hi = 0, rlen = 0;
    auto load_rows = [&](size_t o) {
        const size_t k = 8 * o + (lane & 7);
        const uint64_t off = k < a.nbases ? a.rowoff[k] : 0; roff_lo = (uint32_t)off; roff_hi = (uint32_t)(off >> 32); rlen = k < a.nbases ? a.rowlen[k] : 0u;
    };
    load_rows(o0);
    for (size_t o = o0; o < o1; o++) {
        dig += a.batch;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        const uint32_t c_lo = roff_lo, c_hi = roff_hi, c_len = rlen;
        if (o + 1 < o1) { nxt = *dig; load_rows(o + 1); }
        const bool grp = o < nbit8 && ((okmask >> (o - o0)) & 1);                                   // wave-uniform
        const uint64_t lo = (uint64_t)cur.x | ((uint64_t)cur.y << 32), hi = (uint64_t)cur.z | ((uint64_t)cur.w << 32);
        const uint32_t lim = grp ? 1u : (k1 - 8 * o < 8 ? (uint32_t)(k1 - 8 * o) : 8u);
#pragma unroll 1
        for (uint32_t i = 0; i < lim; i++) {
            int32_t d = (int32_t)(int16_t)(uint16_t)(((i & 4) ? hi : lo) >> (16 * (i & 3)));
            int32_t mag = d < 0 ? -d : d;
            const fe* src;
            if (grp) src = sub + (o * MSM_GROUP_ENTRIES + (size_t)(mag ? mag - 1 : 0)) * (2 * F::WORDS);
            else {
                const uint64_t off = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c_lo, (int)i) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c_hi, (int)i) << 32);
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)c_len, (int)i);
                const bool beyond = (uint32_t)mag > len;                                            // MSM_FLAT_ESCAPE has magnitude 32768 > any flat row
                if (__any(beyond)) {                                                                // wrong prediction: rare, slow, exact
                    if (beyond) {
                        fe sc = fe{}; bool sneg = d < 0;
                        if (d == MSM_FLAT_ESCAPE) { sc = Fr::from_mont(load_fe(a.scalars + (size_t)a.rows[8 * o + i] * a.batch + p)); sneg = sign_normalise(sc); }
                        else sc.l[0] = (uint32_t)mag;
                        const Aff9<F> P1 = unpack_aff(load_raw<F>(table + off * (2 * F::WORDS)), sneg);
                        Xyzz9<F> Q = C::infinity();
                        for (int b = 253; b >= 0; b--) {
                            if (!Q.inf) Q = C::dbl(Q);
                            uint32_t word = sc.l[0];
#pragma unroll
                            for (int q = 1; q < 8; q++) word = (b >> 5) == q ? sc.l[q] : word;
                            if ((word >> (b & 31)) & 1u) Q = C::template madd<true>(Q, P1);
                        }
                        acc = C::add(acc, Q);
                        d = 0; mag = 0;
                    }
                }
                src = table + (off + (size_t)(mag ? mag - 1 : 0)) * (2 * F::WORDS);
            }
            const RawAff<F> e = load_raw<F>(src);
            if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
            pend = e; dp = d;
        }
        cur = nxt;
    }
    if (dp) acc = C::template madd<EXACT>(acc, unpack_aff(pend, dp < 0));
    return acc;
}

// grid: nslices * (batch / 64) workgroups of one wave.  partial[slice * batch + p]
template <class F>
__global__ __launch_bounds__(64, F::WORDS == 1 ? 2 : 1) void k_msm_flat(MsmFlatArgs a) {
    using C = Curve9<F>;
    const size_t G = a.batch / 64, L = blockIdx.x, S8 = a.nslices & ~(size_t)7;
    size_t slice, g;                                        // all groups of proofs of a slice on one XCD (see k_msm_win)
    if (L < S8 * G) { const size_t xcd = L & 7, i = L >> 3; slice = (i / G) * 8 + xcd; g = i % G; }
    else { slice = L / G; g = L % G; }
    const size_t p = g * 64 + threadIdx.x;
    const size_t k0 = slice * a.per < a.nbases ? slice * a.per : a.nbases, k1 = k0 + a.per < a.nbases ? k0 + a.per : a.nbases;
    Xyzz9<F> acc = C::infinity();
    if (k0 < k1) {
        acc = accumulate_flat<F, false>(a, k0, k1, g, p);
        // (see k_msm_win)  The repair runs with ALL lanes of the wave: accumulate_flat uses __ballot / readlane, which need them.
        const bool bad = !acc.inf && F::is_zero(acc.zz);
        if (__any(bad)) { const Xyzz9<F> fix = accumulate_flat<F, true>(a, k0, k1, g, p); if (bad) acc = fix; }
    }
    C::store_xyzz(reinterpret_cast<fe*>(a.partial) + (slice * a.batch + p) * (4 * F::WORDS), acc);
}

// The latency path of a flat set (see k_msm_win_few): lanes are OCTETS of bases for ONE proof — a bit group is one exact addition,
// an ordinary or window octet up to eight — and a butterfly folds the wave.  partial[blockIdx.x * batch + p]; gok per (octet, proof)
// from k_recode_flat_few.  grid: (ceil(octets / 64), nproofs).
template <class F>
__global__ __launch_bounds__(64) void k_msm_flat_few(MsmFlatArgs a) {
    using C = Curve9<F>;
    const size_t p = blockIdx.y, noct = (a.nbases + 7) / 8, nbit8 = a.nbit / 8;
    const size_t o = (size_t)blockIdx.x * 64 + threadIdx.x;
    const fe* table = reinterpret_cast<const fe*>(a.table);
    const fe* sub = reinterpret_cast<const fe*>(a.sub);
    Xyzz9<F> acc = C::infinity();
    if (o < noct) {
        const uint4 w = a.digits[o * a.batch + p];
        const bool grp = o < nbit8 && a.gok[o * MSM_FEW_PROOFS + p] != 0;
        const uint64_t lo = (uint64_t)w.x | ((uint64_t)w.y << 32), hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
        const uint32_t lim = grp ? 1u : (a.nbases - 8 * o < 8 ? (uint32_t)(a.nbases - 8 * o) : 8u);
#pragma unroll 1
        for (uint32_t i = 0; i < lim; i++) {
            int32_t d = (int32_t)(int16_t)(uint16_t)(((i & 4) ? hi : lo) >> (16 * (i & 3)));
            int32_t mag = d < 0 ? -d : d;
            const fe* src;
            if (grp) src = sub + (o * MSM_GROUP_ENTRIES + (size_t)(mag ? mag - 1 : 0)) * (2 * F::WORDS);
            else {
                const uint64_t off = a.rowoff[8 * o + i];
                if ((uint32_t)mag > a.rowlen[8 * o + i]) {                 // wrong prediction (MSM_FLAT_ESCAPE has magnitude 32768 > any flat row): rare, slow, exact
                    fe sc = fe{}; bool sneg = d < 0;
                    if (d == MSM_FLAT_ESCAPE) { sc = Fr::from_mont(load_fe(a.scalars + (size_t)a.rows[8 * o + i] * a.batch + p)); sneg = sign_normalise(sc); }
                    else sc.l[0] = (uint32_t)mag;
                    const Aff9<F> P1 = unpack_aff(load_raw<F>(table + off * (2 * F::WORDS)), sneg);
                    Xyzz9<F> Q = C::infinity();
                    for (int b = 253; b >= 0; b--) {
                        if (!Q.inf) Q = C::dbl(Q);
                        uint32_t word = sc.l[0];
#pragma unroll
                        for (int q = 1; q < 8; q++) word = (b >> 5) == q ? sc.l[q] : word;
                        if ((word >> (b & 31)) & 1u) Q = C::template madd<true>(Q, P1);
                    }
                    acc = C::add(acc, Q);
                    d = 0; mag = 0;
                }
                src = table + (off + (size_t)(mag ? mag - 1 : 0)) * (2 * F::WORDS);
            }
            const RawAff<F> e = load_raw<F>(src);
            if (d) acc = C::template madd<true>(acc, unpack_aff(e, d < 0));
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        Xyzz9<F> q;
        q.x = shfl_xor_e(acc.x, m); q.y = shfl_xor_e(acc.y, m); q.zz = shfl_xor_e(acc.zz, m); q.zzz = shfl_xor_e(acc.zzz, m);
        q.inf = __shfl_xor((int)acc.inf, m) != 0;
        acc = C::add(acc, q);
    }
    if (threadIdx.x == 0) C::store_xyzz(reinterpret_cast<fe*>(a.partial) + ((size_t)blockIdx.x * a.batch + p) * (4 * F::WORDS), acc);
}

// out[p] = sum_j 2^(c j) S[j][p] (+ addend[p]): Horner from the top window, lanes = proofs.  254 doublings per proof whatever the width:
// a serial chain of ~1.2 ms (G1) that does not shrink with the batch, so the independent chains of several sets run as ONE launch
// (blockIdx.y = job).
template <class F>
__global__ __launch_bounds__(64) void k_msm_horner(MsmHornerJobs jobs, size_t batch) {
    using C = Curve9<F>;
    const MsmHornerJob jb = jobs.job[blockIdx.y];
    const fe* S = reinterpret_cast<const fe*>(jb.S); const fe* addend = reinterpret_cast<const fe*>(jb.addend);
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    Xyzz9<F> r = C::load_xyzz(S + ((size_t)(jb.nwin - 1) * batch + p) * (4 * F::WORDS));
    for (int j = jb.nwin - 2; j >= 0; j--) {
#pragma unroll 1
        for (int q = 0; q < jb.c; q++) r = C::dbl(r);
        r = C::add(r, C::load_xyzz(S + ((size_t)j * batch + p) * (4 * F::WORDS)));
    }
    if (addend) r = C::add(r, C::load_xyzz(addend + p * (4 * F::WORDS)));
    C::store_xyzz(reinterpret_cast<fe*>(jb.out) + p * (4 * F::WORDS), r);
}

// Montgomery value of the 8 x 32-bit domain (R = 2^256, what the decompression kernels produce) -> radix-2^29 domain (R' = 2^261)
__device__ __forceinline__ fe9 to_fp29(const fe& old_mont) { return Fp29::to_mont(Fp29::unpack(Fp::from_mont(old_mont))); }
__device__ __forceinline__ Aff9<Fp29f> base_to_fp29(const Aff<Fp>* b) { return Aff9<Fp29f>{to_fp29(b->x), to_fp29(b->y)}; }
__device__ __forceinline__ Aff9<Fp2x> base_to_fp29(const Aff<Fp2>* b) {
    return Aff9<Fp2x>{fe9x2{to_fp29(b->x.a0), to_fp29(b->x.a1)}, fe9x2{to_fp29(b->y.a0), to_fp29(b->y.a1)}};
}

// and back
__device__ __forceinline__ fe from_fp29_fe(const fe9& m) { return Fp::to_mont(Fp29::pack(Fp29::from_mont(m))); }
__device__ __forceinline__ Aff<Fp> from_fp29(const Aff9<Fp29f>& a) { return Aff<Fp>{from_fp29_fe(a.x), from_fp29_fe(a.y)}; }
__device__ __forceinline__ Aff<Fp2> from_fp29(const Aff9<Fp2x>& a) { return Aff<Fp2>{fe2{from_fp29_fe(a.x.a0), from_fp29_fe(a.x.a1)}, fe2{from_fp29_fe(a.y.a0), from_fp29_fe(a.y.a1)}}; }

// out[i] = 2^shift[i] * in[src[i]] (affine, 8 x 32-bit Montgomery images like the decompression kernels' output): the points of the
// (base, window) pairs of a wide wire laid out as bases of their own.
template <class F, class OldF>
__global__ __launch_bounds__(64) void k_shift_bases(const Aff<OldF>* in, const uint32_t* src, const uint32_t* shift, size_t n, Aff<OldF>* out) {
    using C = Curve9<F>;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz9<F> P = C::from_aff(base_to_fp29(in + src[i]));
    for (uint32_t q = 0; q < shift[i]; q++) P = C::dbl(P);
    const Aff9<F> A = C::to_aff(P);
    out[i] = from_fp29(A);
}

// ---- InitAlgorithm: T[k][d-1] = d * P_k ----------------------------------------------------------------------------------------
// One thread per segment (MsmRowSeg) of up to `cap` consecutive multiples of one base: the first multiple by double-and-add, the
// rest by mixed additions, kept in XYZZ in `scratch` with the running product of the ZZZ parked in the table slots; one inversion
// per segment turns them into affine entries (Montgomery batch inversion).  scratch holds nsegs * cap points.
template <class F, class OldF>
__global__ __launch_bounds__(64) void k_build_rows(const Aff<OldF>* bases, const MsmRowSeg* segs, size_t nsegs, uint32_t cap, fe* table, fe* scratch) {
    using C = Curve9<F>;
    using E = typename F::E;
    constexpr int CW = F::WORDS;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nsegs) return;
    const MsmRowSeg sg = segs[t];
    const Aff9<F> P = base_to_fp29(bases + sg.base);
    Xyzz9<F> Ed = C::infinity();
    for (int b = 31 - __clz(sg.first); b >= 0; b--) { Ed = C::dbl(Ed); if ((sg.first >> b) & 1u) Ed = C::template madd<true>(Ed, P); }
    fe* out = table + sg.entry * (2 * CW);
    fe* sc = scratch + t * cap * (4 * CW);
    E prefix = F::one();
    for (uint32_t d = 0; d < sg.count; d++) {
        if (d) Ed = C::template madd<true>(Ed, P);
        C::store_xyzz(sc + d * (4 * CW), Ed);
        F::store(out + d * (2 * CW), prefix);                       // product of the ZZZ before this entry
        prefix = F::mul(prefix, Ed.zzz);
    }
    E inv = F::inv(prefix);
    for (uint32_t d = sg.count; d-- > 0;) {
        Ed = C::load_xyzz(sc + d * (4 * CW));
        const E pre = F::load(out + d * (2 * CW));
        const E izzz = F::mul(inv, pre);                             // 1 / ZZZ_d
        inv = F::mul(inv, Ed.zzz);
        const E izz = F::mul(F::sqr(Ed.zz), F::sqr(izzz));          // 1/ZZ = ZZ^2 / ZZZ^2
        Aff9<F> e; e.x = F::mul(Ed.x, izz); e.y = F::mul(Ed.y, izzz);
        C::store_aff(out + d * (2 * CW), e);
    }
}

// ---- Groth16 Setup: many independent multiples of ONE fixed point (the group generator) ----------------------------------------
// out[i] = s_i * G from the window rows table[j][d-1] = d * 2^(c j) * G (built like any other rows from the bases 2^(c j) G).
// Lanes are scalars.  Output: affine coordinates as canonical integers (8 little-endian 32-bit words per Fp element; G2: a0 then
// a1), inf[i] = 1 for the point at infinity (zero scalar).
__device__ __forceinline__ void store_canon_words(fe* dst, const fe9& mont) { store_fe(dst, Fp29::pack(Fp29::from_mont(mont))); }
__device__ __forceinline__ void store_aff_canon(fe* dst, const Aff9<Fp29f>& a) { store_canon_words(dst, a.x); store_canon_words(dst + 1, a.y); }
__device__ __forceinline__ void store_aff_canon(fe* dst, const Aff9<Fp2x>& a) {
    store_canon_words(dst, a.x.a0); store_canon_words(dst + 1, a.x.a1); store_canon_words(dst + 2, a.y.a0); store_canon_words(dst + 3, a.y.a1);
}
template <class F>
__global__ __launch_bounds__(64) void k_fixed_mul(const fe* table, int c, int nwin, const fe* scalars, size_t n, fe* out, uint8_t* inf) {
    using C = Curve9<F>;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe s = load_fe(scalars + i);
    const uint32_t cmask = (1u << c) - 1, D = 1u << (c - 1);
    uint32_t carry = 0;
    Xyzz9<F> acc = C::infinity();
    for (int j = 0; j < nwin; j++) {
        uint32_t raw = (s.l[0] & cmask) + carry;
#pragma unroll
        for (int q = 0; q < 7; q++) s.l[q] = __builtin_amdgcn_alignbit(s.l[q + 1], s.l[q], c);
        s.l[7] >>= c;
        int32_t d = (int32_t)raw;
        if (raw >= D) { d -= (int32_t)(1u << c); carry = 1; } else carry = 0;
        if (d) {
            const int32_t mag = d < 0 ? -d : d;
            acc = C::template madd<true>(acc, unpack_aff(load_raw<F>(table + ((size_t)j * D + (size_t)(mag - 1)) * (2 * F::WORDS)), d < 0));
        }
    }
    fe* o = out + i * (2 * F::WORDS);
    if (acc.inf) { inf[i] = 1; for (int q = 0; q < 2 * F::WORDS; q++) store_fe(o + q, fe{}); }
    else { inf[i] = 0; store_aff_canon(o, C::to_aff(acc)); }
}

}  // namespace

void launch_fixed_mul_g1(const G1Aff* table, int c, int nwin, const fe* scalars, size_t n, fe* out, uint8_t* inf, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fixed_mul<Fp29f>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const fe*>(table), c, nwin, scalars, n, out, inf);
}
void launch_fixed_mul_g2(const G2Aff* table, int c, int nwin, const fe* scalars, size_t n, fe* out, uint8_t* inf, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fixed_mul<Fp2x>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const fe*>(table), c, nwin, scalars, n, out, inf);
}
void launch_msm_recode(const MsmRecodeArgs& a, hipStream_t s) {
    if (!a.nbases) return;
    const dim3 grid((unsigned)(a.batch / 64), (unsigned)((a.nbases + 7) / 8));
    if (a.c > 16) { if (a.mont) hipLaunchKernelGGL((k_recode<true, true>), grid, dim3(64), 0, s, a); else hipLaunchKernelGGL((k_recode<false, true>), grid, dim3(64), 0, s, a); }
    else if (a.mont) hipLaunchKernelGGL((k_recode<true, false>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_recode<false, false>), grid, dim3(64), 0, s, a);
}
void launch_msm_recode_flat_few(const MsmFlatRecodeArgs& a, size_t nproofs, hipStream_t s) {
    if (a.nbases) hipLaunchKernelGGL(k_recode_flat_few, dim3((unsigned)(((a.nbases + 7) / 8 + 63) / 64), (unsigned)nproofs), dim3(64), 0, s, a);
}
void launch_msm_flat_few_g1(const MsmFlatArgs& a, size_t nproofs, hipStream_t s) { hipLaunchKernelGGL(k_msm_flat_few<Fp29f>, dim3((unsigned)a.nslices, (unsigned)nproofs), dim3(64), 0, s, a); }
void launch_msm_flat_few_g2(const MsmFlatArgs& a, size_t nproofs, hipStream_t s) { hipLaunchKernelGGL(k_msm_flat_few<Fp2x>, dim3((unsigned)a.nslices, (unsigned)nproofs), dim3(64), 0, s, a); }
void launch_msm_recode_flat(const MsmFlatRecodeArgs& a, hipStream_t s) {
    if (a.nbases) hipLaunchKernelGGL(k_recode_flat, dim3((unsigned)(a.batch / 64), (unsigned)((a.nbases + 7) / 8)), dim3(64), 0, s, a);
}
void launch_msm_win_g1(const MsmWinArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.nslices * a.nwin * (a.batch / 64)));
    if (a.c > 16) hipLaunchKernelGGL((k_msm_win<Fp29f, true>), grid, dim3(64), 0, s, a); else hipLaunchKernelGGL((k_msm_win<Fp29f, false>), grid, dim3(64), 0, s, a);
}
void launch_msm_win_g2(const MsmWinArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.nslices * a.nwin * (a.batch / 64)));
    if (a.c > 16) hipLaunchKernelGGL((k_msm_win<Fp2x, true>), grid, dim3(64), 0, s, a); else hipLaunchKernelGGL((k_msm_win<Fp2x, false>), grid, dim3(64), 0, s, a);
}
void launch_msm_win_few_g1(const MsmWinArgs& a, size_t nproofs, hipStream_t s) { hipLaunchKernelGGL(k_msm_win_few<Fp29f>, dim3((unsigned)(a.nslices * a.nwin), (unsigned)nproofs), dim3(64), 0, s, a); }
void launch_msm_win_few_g2(const MsmWinArgs& a, size_t nproofs, hipStream_t s) { hipLaunchKernelGGL(k_msm_win_few<Fp2x>, dim3((unsigned)(a.nslices * a.nwin), (unsigned)nproofs), dim3(64), 0, s, a); }
void launch_msm_flat_g1(const MsmFlatArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_msm_flat<Fp29f>, dim3((unsigned)(a.nslices * (a.batch / 64))), dim3(64), 0, s, a); }
void launch_msm_flat_g2(const MsmFlatArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_msm_flat<Fp2x>, dim3((unsigned)(a.nslices * (a.batch / 64))), dim3(64), 0, s, a); }
void launch_msm_horner_g1(const MsmHornerJobs& jobs, size_t batch, hipStream_t s) {
    if (jobs.n) hipLaunchKernelGGL(k_msm_horner<Fp29f>, dim3((unsigned)(batch / 64), (unsigned)jobs.n), dim3(64), 0, s, jobs, batch);
}
void launch_msm_horner_g2(const MsmHornerJobs& jobs, size_t batch, hipStream_t s) {
    if (jobs.n) hipLaunchKernelGGL(k_msm_horner<Fp2x>, dim3((unsigned)(batch / 64), (unsigned)jobs.n), dim3(64), 0, s, jobs, batch);
}
void launch_build_rows_g1(const G1Aff* bases, const MsmRowSeg* segs, size_t nsegs, uint32_t cap, G1Aff* table, G1Xyzz* scratch, hipStream_t s) {
    if (nsegs) hipLaunchKernelGGL((k_build_rows<Fp29f, Fp>), dim3((unsigned)((nsegs + 63) / 64)), dim3(64), 0, s,
                                  reinterpret_cast<const Aff<Fp>*>(bases), segs, nsegs, cap, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch));
}
void launch_build_rows_g2(const G2Aff* bases, const MsmRowSeg* segs, size_t nsegs, uint32_t cap, G2Aff* table, G2Xyzz* scratch, hipStream_t s) {
    if (nsegs) hipLaunchKernelGGL((k_build_rows<Fp2x, Fp2>), dim3((unsigned)((nsegs + 63) / 64)), dim3(64), 0, s,
                                  reinterpret_cast<const Aff<Fp2>*>(bases), segs, nsegs, cap, reinterpret_cast<fe*>(table), reinterpret_cast<fe*>(scratch));
}
void launch_shift_bases_g1(const G1Aff* in, const uint32_t* src, const uint32_t* shift, size_t n, G1Aff* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL((k_shift_bases<Fp29f, Fp>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const Aff<Fp>*>(in), src, shift, n, reinterpret_cast<Aff<Fp>*>(out));
}
void launch_shift_bases_g2(const G2Aff* in, const uint32_t* src, const uint32_t* shift, size_t n, G2Aff* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL((k_shift_bases<Fp2x, Fp2>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const Aff<Fp2>*>(in), src, shift, n, reinterpret_cast<Aff<Fp2>*>(out));
}

}  // namespace gsc
