// Quotient polynomial H = (A*B - C) / (X^n - 1) over BN254 Fr, batched over proofs.
//
// Replaces computeH inside groth16.Prove (reference libraries/prover/impl/provers.go:148,216; gnark
// backend/groth16/bn254.computeH + gnark-crypto fr/fft — SURVEY.md §8(a) a7, mathematics App. D).  gnark evaluates A, B, C on a
// coset, divides pointwise and interpolates back: seven transforms.  For a SATISFIED constraint system (c_i = a_i b_i on the
// domain: what the solver has just checked — a proof is only released when it holds) six are enough and C is never multiplied:
//   A B = P_lo + X^n P_hi,  H = P_hi  (deg A, B < n),      A B mod (X^n - 1) = P_lo + P_hi = S,      A B mod (X^n + 1) = P_lo - P_hi = D
//   S = iNTT(c)  (c_i = a_i b_i are the values of A B on the n-th roots of unity),   D = the negacyclic product: values of A and B on
//   zeta * (roots of unity), zeta^n = -1, multiplied pointwise and interpolated back;   H = (S - D) / 2.
// The same polynomial, hence the same canonical coefficients, as gnark's (A B - C) / (X^n - 1).
//   2 inverse NTTs (a, b: DIF, natural -> bit-reversed), scaling by zeta^j, 2 forward NTTs (DIT, bit-reversed -> natural),
//   pointwise a*b, 1 inverse NTT + scaling by zeta^-j (D), 1 inverse NTT (c -> S); the bit-reversed output order of a DIF
//   transform is exactly the order pk.G1.Z is stored in.
//
// Data stay in the solver's [index][proof] layout for the whole pipeline (no transposes).  n = 2^L is split
// as 2^Lhi x 2^Llo: "strided" kernels own the stages that couple the top Lhi index bits (groups of 2^Lhi
// elements at stride 2^Llo), "contiguous" kernels the low Llo bits.  Because DIF ends where DIT begins, the
// six transforms take four kernels:
//   K1 strided DIF head (a,b,c) | K2 contiguous DIF tail + zeta scale + DIT head (a,b)
//   K3 strided DIT tail (a,b) + pointwise + strided DIF head (d) | K4 contiguous DIF tails of c and d, h = (S - D) / 2.
// Every kernel stages a tile of P proofs x 2^Lhi (or 2^Llo) elements in LDS, limb-major, two radix-2 stages
// per barrier; global accesses are P*32 = 128-byte segments.
//
// Evaluation form (k_quot_bases.hip; what batch calls run): only the transforms of a and b — K1, K2, K3 — and K3 ends at d = a b on the zeta-coset, whose
// signed digits it writes for the Z sum.  The step is power-limited: what buys time in these kernels is fewer instructions, so
//   * K1 takes byte planes (the small-integer witness path) as plain integers: the first two stages of a ternary tile are twiddles scaled by integers in [-4, 4];
//   * K2 runs the last two inverse stages, the scaling and the first two forward stages of a thread's four consecutive elements in registers (the
//     pairs whose twiddle is 1 for every thread cost no product);
//   * K3 runs its last two forward stages in registers straight into the pointwise product, canonicalises with three conditional subtractions and takes
//     the digits of the bench configuration's width (c = 17) at compile-time bit offsets.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

using F = Fr29;            // radix-2^29 lazy-limb scalar field (bn254_fp29.hpp): 227-instruction products, carry-free add/sub
constexpr int P = 4;       // proofs per workgroup tile (measured on 8192 proofs, round 3: P = 2 -> 125.5 ms, P = 4 -> 120.3 ms, P = 8 -> 128.9 ms for the quotient stage)

// Ranges inside the transforms (Fr29: R' = 2^261 ~ 169 r, a product a*w with w < r lands in (-|a|/169, |a|/169 + r)):
//   * limbs: everything written to a tile has |limb| < 2^30 (one lazy add/sub of tight values); an operand is carried
//     (norm(), 24 instructions) only where it would otherwise meet a second lazy addition;
//   * values: a DIT stage adds a fresh product to u, so |value| grows by <= 1.1 r per stage; a DIF stage doubles the
//     sum path.  Instead of a comparison chain per butterfly, reduce_top() is applied once per DIF run of >= 4 stages
//     and once before a kernel stores its tile: it estimates q = floor(value / r) from the top limb (float multiply)
//     and subtracts the tabulated q*r, leaving a value in (-1.001 r, 2.001 r).
// Global accesses of these kernels are plain.  Non-temporal loads / stores (every vector passes through each kernel exactly once) were tried in
// round 3: quotient stage 122 -> 125 ms per 8192 proofs, and one run of the 2^17-domain parity test failed with them — not pursued, not used.
__device__ __forceinline__ fe ld_stream(const fe* p) { return load_fe(p); }
__device__ __forceinline__ void st_stream(fe* p, const fe& v) { store_fe(p, v); }
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
// memory image between kernels: non-negative, tight, < 2^256 (not necessarily < r).  REDUCED: x is known to lie in (-1.001 r, 2.001 r) already (signed-tight limbs: the
// outputs of a reducing butterfly round), so the estimate-and-subtract step is skipped.
template <bool REDUCED = false>
__device__ __forceinline__ void store_lazy(fe* p, const fe9& x, const int32_t* qr) {
    fe9 t = REDUCED ? F::norm(x) : F::norm(reduce_top(x, qr));
    const bool lo = t.l[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; i++) t.l[i] += lo ? F::PK(1, i) : 0;     // + 2r: (-1.001 r, 2.001 r) -> [0, 2.001 r)
    st_stream(p, F::pack(F::norm(t)));
}

// twiddles are stored already split into limbs: 12 int32 per entry (9 used; 48-byte stride keeps the three 16-byte loads aligned)
__device__ __forceinline__ fe9 load_tw(const int32_t* tw, uint32_t ex) {
    const int4* q = reinterpret_cast<const int4*>(tw + 12 * (size_t)ex);
    const int4 a = q[0], b = q[1], c = q[2];
    fe9 r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
    return r;
}

// LDS tile: nine limb planes of (elements x P) words.  ds_read_b32 / ds_write_b32 serve a wave in two groups of 32 lanes = eight
// consecutive threads-of-a-proof-quad, i.e. eight elements x P proofs, on 32 banks of 4 bytes: a group is conflict-free when its
// eight elements fall into eight different 4-word slots (slot = element mod 8).  The butterfly stages address elements at strides
// 1, 2, 4, 8, ... — eight lanes then differ in element bits {2,3,4}, {0,3,4}, {0,1,4}, {0,1,2} (two stages per round trip) or
// {1,2,3}, {0,2,3}, {0,1,3} (single stages), which all land in ONE or TWO slots of a plain layout (4- and 8-way conflicts: 1.6
// extra LDS cycles per instruction measured in round 2).  Elements are therefore stored at the swizzled position
//   e' = e ^ (bit3(e) ? 7 : 0) ^ (bit4(e) ? 5 : 0)
// a bijection that only changes the low three bits: as a GF(2) map of the element bits its columns are 001, 010, 100 (bits 0-2),
// 111 (bit 3), 101 (bit 4), and every one of the seven bit triples above is linearly independent — each access pattern spreads over
// all eight slots.  Contiguous runs (bits {0,1,2}: the global loads / stores) stay conflict-free.
__device__ __forceinline__ uint32_t swz(uint32_t e) { return e ^ (((e >> 3) & 1u) * 7u) ^ (((e >> 4) & 1u) * 5u); }
struct Tile {
    int32_t* lds; uint32_t plane;   // plane = elements * P (words per limb plane)
    __device__ __forceinline__ fe9 get(uint32_t e, uint32_t q) const {
        fe9 r; const uint32_t o = swz(e) * P + q;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = lds[i * plane + o];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t e, uint32_t q, const fe9& v) const {
        const uint32_t o = swz(e) * P + q;
#pragma unroll
        for (int i = 0; i < 9; i++) lds[i * plane + o] = v.l[i];
    }
};

// ---- stage machinery ----------------------------------------------------------------------------------------------------
// A workgroup has (tile elements / 4) * P threads; thread (u4, q) owns proof q of the tile.  Two radix-2 stages are done per LDS
// round trip ("round4": four elements in registers, two stages, four butterflies), a left-over single stage is done as two
// butterflies per thread.  Twiddle exponents are those of the plain radix-2 stages, so results are bit-identical to them.
template <bool STRIDED> __device__ __forceinline__ uint32_t gidx(uint32_t e, int Llo, uint32_t tile_id) { return STRIDED ? ((e << Llo) + tile_id) : ((tile_id << Llo) + e); }
// d * w (a product contracts: whatever the range of d, the result is tight and in (-r, 2r))
template <bool REDUCE> __device__ __forceinline__ fe9 mulw(const fe9& d, const int32_t* tw, uint32_t ex, const int32_t* qr) {
    // No per-lane test for the twiddle 1 (exponent 0: entry 0 of the table is the Montgomery image of 1): a branch around every product
    // cuts a round into basic blocks for nothing (measured: same time, 20 % more code, 18 more registers); the stages whose twiddles are
    // ALL 1 — the last DIF stage, the first DIT stage — are skipped wave-uniformly instead (dif_round4 / dit_round4).
    return F::mul(d, load_tw(tw, ex));
}
template <bool REDUCE> __device__ __forceinline__ fe9 carry(const fe9& x, const int32_t* qr) { return REDUCE ? reduce_top(x, qr) : F::norm(x); }

// DIF.  Tile convention: entries are tight or signed-tight (sums are carried before they are written).
// one stage s: pairs (e1, e1 + he); twiddle exponent = (gidx(e1) mod hg) << s, hg = 2^(L-1-s)
template <bool STRIDED, bool REDUCE>
__device__ __forceinline__ void dif_stage(const Tile& t, uint32_t bf, uint32_t q, uint32_t he, int s, int L, int Llo, uint32_t tile_id, const NttPlan& pl) {
    const uint32_t e1 = 2 * bf - (bf & (he - 1)), e2 = e1 + he, hg = 1u << (L - 1 - s);
    const uint32_t ex = (gidx<STRIDED>(e1, Llo, tile_id) & (hg - 1)) << s;
    const fe9 u = t.get(e1, q), v = t.get(e2, q);
    t.put(e1, q, carry<REDUCE>(F::add(u, v), pl.qr));
    if (s + 1 == L) t.put(e2, q, carry<REDUCE>(F::sub(u, v), pl.qr));      // wave-uniform: the last stage, every twiddle is 1
    else t.put(e2, q, mulw<REDUCE>(F::sub(u, v), pl.tw_inv, ex, pl.qr));
}
// stages s and s+1 on elements e0 + {0, 1, 2, 3} * he2 (he2 = tile half of stage s+1): stage s pairs (e0,e2), (e1,e3), stage s+1 (e0,e1), (e2,e3)
template <bool STRIDED, bool REDUCE>
__device__ __forceinline__ void dif_round4(const Tile& t, uint32_t u4, uint32_t q, uint32_t he2, int s, int L, int Llo, uint32_t tile_id, const NttPlan& pl) {
    const uint32_t e0 = 4 * u4 - 3 * (u4 & (he2 - 1)), e1 = e0 + he2, e2 = e0 + 2 * he2, e3 = e0 + 3 * he2;
    const uint32_t hg1 = 1u << (L - 1 - s), hg2 = hg1 >> 1;
    const uint32_t g0 = gidx<STRIDED>(e0, Llo, tile_id), g1 = gidx<STRIDED>(e1, Llo, tile_id), g2 = gidx<STRIDED>(e2, Llo, tile_id);
    const uint32_t exA = (g0 & (hg1 - 1)) << s, exB = (g1 & (hg1 - 1)) << s, exC = (g0 & (hg2 - 1)) << (s + 1), exD = (g2 & (hg2 - 1)) << (s + 1);
    const fe9 x0 = t.get(e0, q), x1 = t.get(e1, q), x2 = t.get(e2, q), x3 = t.get(e3, q);
    // tile entries are (signed-)tight: the two inner sums stay un-carried (|limb| < 2^30) — their sum (< 2^31) is carried once below,
    // their difference (< 2^30) is a legal product operand as it is
    const fe9 a0 = F::add(x0, x2), a1 = F::add(x1, x3);
    const fe9 a2 = mulw<false>(F::sub(x0, x2), pl.tw_inv, exA, pl.qr), a3 = mulw<false>(F::sub(x1, x3), pl.tw_inv, exB, pl.qr);
    t.put(e0, q, carry<REDUCE>(F::add(a0, a1), pl.qr));
    t.put(e2, q, carry<REDUCE>(F::add(a2, a3), pl.qr));
    if (s + 2 == L) {      // wave-uniform: stage L-1 is the last one, every twiddle is 1
        t.put(e1, q, carry<REDUCE>(F::sub(a0, a1), pl.qr));
        t.put(e3, q, carry<REDUCE>(F::sub(a2, a3), pl.qr));
    } else {
        const fe9 r1 = mulw<REDUCE>(F::sub(a0, a1), pl.tw_inv, exC, pl.qr), r3 = mulw<REDUCE>(F::sub(a2, a3), pl.tw_inv, exD, pl.qr);
        t.put(e1, q, r1); t.put(e3, q, r3);
    }
}
// DIF stages s0 .. s1-1 on a tile of E elements.  `d` = doublings of the sum path since its last range reduction (on entry: d0 <= 1);
// a round that would leave more than four of them reduces instead (values stay below 32 r, products need < 111 r).
template <bool STRIDED>
// Returns true when the last round reduced every entry it wrote (values in (-1.001 r, 2.001 r)): the caller's store can then skip its own range reduction.
__device__ __forceinline__ bool dif_run(const Tile& t, uint32_t u4, uint32_t q, int s0, int s1, int d0, int L, int Llo, uint32_t tile_id, const NttPlan& pl, uint32_t E) {
    int s = s0, d = d0; bool last_reduced = false;
    auto half_of = [&](int st) { const uint32_t hg = 1u << (L - 1 - st); return STRIDED ? hg >> Llo : hg; };
    if ((s1 - s0) & 1) {
        dif_stage<STRIDED, false>(t, u4, q, half_of(s), s, L, Llo, tile_id, pl);
        dif_stage<STRIDED, false>(t, u4 + E / 4, q, half_of(s), s, L, Llo, tile_id, pl);
        __syncthreads();
        s++; d++;
    }
    for (; s < s1; s += 2) {
        if (d >= 1) { dif_round4<STRIDED, true>(t, u4, q, half_of(s + 1), s, L, Llo, tile_id, pl); d = 0; last_reduced = true; }
        else { dif_round4<STRIDED, false>(t, u4, q, half_of(s + 1), s, L, Llo, tile_id, pl); d += 2; last_reduced = false; }
        __syncthreads();
    }
    return last_reduced;
}

// DIT.  Tile convention: entries may carry one lazy addition (|limb| < 2^30); u is carried when it is read, v goes straight into
// the product.  Values grow by at most 1.1 r per stage (a fresh product is added), so no range reduction inside a run.
// one stage s: half = 2^s; twiddle exponent = (gidx(e1) mod 2^s) << (L-1-s)
template <bool STRIDED>
__device__ __forceinline__ void dit_stage(const Tile& t, uint32_t bf, uint32_t q, uint32_t he, int s, int L, int Llo, uint32_t tile_id, const NttPlan& pl) {
    const uint32_t e1 = 2 * bf - (bf & (he - 1)), e2 = e1 + he;
    const uint32_t ex = (gidx<STRIDED>(e1, Llo, tile_id) & ((1u << s) - 1)) << (L - 1 - s);
    const fe9 u = F::norm(t.get(e1, q));
    const fe9 v = s == 0 ? F::norm(t.get(e2, q)) : mulw<false>(t.get(e2, q), pl.tw_fwd, ex, pl.qr);      // wave-uniform: stage 0, every twiddle is 1
    t.put(e1, q, F::add(u, v)); t.put(e2, q, F::sub(u, v));
}
// stages s and s+1 on elements e0 + {0, 1, 2, 3} * he1 (he1 = tile half of stage s): stage s pairs (e0,e1), (e2,e3), stage s+1 (e0,e2), (e1,e3)
template <bool STRIDED>
__device__ __forceinline__ void dit_round4(const Tile& t, uint32_t u4, uint32_t q, uint32_t he1, int s, int L, int Llo, uint32_t tile_id, const NttPlan& pl) {
    const uint32_t e0 = 4 * u4 - 3 * (u4 & (he1 - 1)), e1 = e0 + he1, e2 = e0 + 2 * he1, e3 = e0 + 3 * he1;
    const uint32_t g0 = gidx<STRIDED>(e0, Llo, tile_id), g1 = gidx<STRIDED>(e1, Llo, tile_id), g2 = gidx<STRIDED>(e2, Llo, tile_id);
    const uint32_t m1 = (1u << s) - 1, m2 = (2u << s) - 1;
    const uint32_t exA = (g0 & m1) << (L - 1 - s), exB = (g2 & m1) << (L - 1 - s), exC = (g0 & m2) << (L - 2 - s), exD = (g1 & m2) << (L - 2 - s);
    const fe9 n0 = F::norm(t.get(e0, q)), n2 = F::norm(t.get(e2, q));
    fe9 v1, v3;
    if (s == 0) { v1 = F::norm(t.get(e1, q)); v3 = F::norm(t.get(e3, q)); }      // wave-uniform: stage 0 is the first one, every twiddle is 1
    else { v1 = mulw<false>(t.get(e1, q), pl.tw_fwd, exA, pl.qr); v3 = mulw<false>(t.get(e3, q), pl.tw_fwd, exB, pl.qr); }
    const fe9 a0 = F::norm(F::add(n0, v1)), a1 = F::sub(n0, v1);      // a1: tight - tight is signed-tight as it is (|limb| < 2^29)
    const fe9 w2 = mulw<false>(F::add(n2, v3), pl.tw_fwd, exC, pl.qr), w3 = mulw<false>(F::sub(n2, v3), pl.tw_fwd, exD, pl.qr);
    t.put(e0, q, F::add(a0, w2)); t.put(e2, q, F::sub(a0, w2));
    t.put(e1, q, F::add(a1, w3)); t.put(e3, q, F::sub(a1, w3));
}
// SINGLE_LAST: an odd number of stages runs its left-over single stage at the end instead of at the start
template <bool STRIDED, bool SINGLE_LAST = false>
__device__ __forceinline__ void dit_run(const Tile& t, uint32_t u4, uint32_t q, int s0, int s1, int L, int Llo, uint32_t tile_id, const NttPlan& pl, uint32_t E) {
    int s = s0;
    auto half_of = [&](int st) { return STRIDED ? 1u << (st - Llo) : 1u << st; };
    if (SINGLE_LAST) {
        for (; s + 1 < s1; s += 2) { dit_round4<STRIDED>(t, u4, q, half_of(s), s, L, Llo, tile_id, pl); __syncthreads(); }
        if (s < s1) {
            dit_stage<STRIDED>(t, u4, q, half_of(s), s, L, Llo, tile_id, pl);
            dit_stage<STRIDED>(t, u4 + E / 4, q, half_of(s), s, L, Llo, tile_id, pl);
            __syncthreads();
        }
        return;
    }
    if ((s1 - s0) & 1) {
        dit_stage<STRIDED>(t, u4, q, half_of(s), s, L, Llo, tile_id, pl);
        dit_stage<STRIDED>(t, u4 + E / 4, q, half_of(s), s, L, Llo, tile_id, pl);
        __syncthreads();
        s++;
    }
    for (; s < s1; s += 2) { dit_round4<STRIDED>(t, u4, q, half_of(s), s, L, Llo, tile_id, pl); __syncthreads(); }
}

// An entry of a byte plane (0, 1, -1) as the limbs of its 2^256 Montgomery image (what the generic solver stores for that value)
__device__ __forceinline__ fe9 narrow9(int t) {
    fe one, mone;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = FrParams::one(i);
    { uint64_t br = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)FrParams::mod(i) - one.l[i] - br; mone.l[i] = (uint32_t)d; br = (d >> 32) & 1; } }
    const fe9 p1 = F::unpack(one), m1 = F::unpack(mone);
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = t > 0 ? p1.l[i] : (t < 0 ? m1.l[i] : 0);
    return r;
}

// a small integer as limbs; w * k for a canonical twiddle w and |k| <= 4 (|limb| < 2^31: norm() takes it from there)
__device__ __forceinline__ fe9 small9(int k) { fe9 r = F::zero(); r.l[0] = k; return r; }
__device__ __forceinline__ fe9 scale9(const fe9& w, int k) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = w.l[i] * k;
    return r;
}

struct ClkStamp {      // diagnostics: the shader clock a kernel runs at = (shader-clock ticks) / (100 MHz ticks) over the life of one workgroup in the middle of the grid
    unsigned long long* p;
    __device__ __forceinline__ ClkStamp(unsigned long long* clk, int slot) : p(nullptr) {
        if (clk && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0) { p = clk + 4 * slot; p[0] = wall_clock64(); p[1] = clock64(); }
    }
    __device__ __forceinline__ void end() const { if (p) { p[2] = wall_clock64(); p[3] = clock64(); } }
};

// K1: strided DIF head.  grid (2^Llo, batch/P, nvec); block (2^(Lhi-2) * P).  Input: the solver's a/b/c rows (canonical values of
// the 2^256 Montgomery domain); they are used as they are — every stage is linear, and K2's scale table folds in the
// change of domain (2^256 -> 2^261).
__global__ __launch_bounds__(512) void k_ntt_dif_strided(NttPlan pl, fe* v0, fe* v1, fe* v2, size_t m, size_t batch, NttNarrow nr) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t G = 1u << Lhi;
    fe* vec = blockIdx.z == 0 ? v0 : blockIdx.z == 1 ? v1 : v2;
    const uint32_t g = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, u4 = threadIdx.x / P;
    const ClkStamp cs(pl.clk, 0);
    Tile t{smem, G * P};
    const int8_t* plane = blockIdx.z == 0 ? nr.plane[0] : blockIdx.z == 1 ? nr.plane[1] : nr.plane[2];
    if (plane && nr.plain) {
        // The plane's small integers themselves, not their Montgomery images (the next kernel's scale table makes up for it).  This thread's four
        // elements u4 + k G/4 are exactly the ones its first two stages couple, so those run in registers: on ternary inputs (every row of `a`, all but
        // the 336 add32 rows of `b`) three of the four products are twiddles scaled by integers in [-4, 4].
        const int8_t* pp = plane + ((q0 + q) >> 6) * nr.crows * 64 + ((q0 + q) & 63);
        const uint32_t he2 = G / 4, e0 = u4, e1 = u4 + he2, e2 = u4 + 2 * he2, e3 = u4 + 3 * he2;
        const size_t i0 = ((size_t)e0 << Llo) + g, i1 = ((size_t)e1 << Llo) + g, i2 = ((size_t)e2 << Llo) + g, i3 = ((size_t)e3 << Llo) + g;
        const int t0 = i0 < m ? (int)pp[i0 * 64] : 0, t1 = i1 < m ? (int)pp[i1 * 64] : 0, t2 = i2 < m ? (int)pp[i2 * 64] : 0, t3 = i3 < m ? (int)pp[i3 * 64] : 0;
        const uint32_t hg1 = 1u << (L - 1), hg2 = hg1 >> 1;
        const uint32_t g0 = (uint32_t)i0, g1 = (uint32_t)i1, g2 = (uint32_t)i2;
        const uint32_t exA = g0 & (hg1 - 1), exB = g1 & (hg1 - 1), exC = (g0 & (hg2 - 1)) << 1, exD = (g2 & (hg2 - 1)) << 1;
        const bool small = t0 != (int)WS_PLANE_WIDE && t1 != (int)WS_PLANE_WIDE && t2 != (int)WS_PLANE_WIDE && t3 != (int)WS_PLANE_WIDE;
        if (__all(small)) {
            const fe9 wA = load_tw(pl.tw_inv_plain, exA), wB = load_tw(pl.tw_inv_plain, exB), wC = load_tw(pl.tw_inv_plain, exC);
            const int a0 = t0 + t2, a1 = t1 + t3;
            const fe9 A2 = scale9(wA, t0 - t2), A3 = scale9(wB, t1 - t3);
            t.put(e0, q, small9(a0 + a1));
            t.put(e2, q, F::norm(F::add(A2, A3)));
            t.put(e1, q, F::norm(scale9(wC, a0 - a1)));
            t.put(e3, q, F::mul(F::norm(F::sub(A2, A3)), load_tw(pl.tw_inv, exD)));
        } else {
            fe9 c32 = F::zero(); c32.l[0] = 32;      // x * 32 / 2^261 = x / 2^256: a 32-byte row out of the solver's Montgomery domain
            auto val = [&](int tv, size_t idx) { return tv == (int)WS_PLANE_WIDE ? F::mul(F::unpack(ld_stream(vec + idx * batch + q0 + q)), c32) : small9(tv); };
            const fe9 x0 = val(t0, i0), x1 = val(t1, i1), x2 = val(t2, i2), x3 = val(t3, i3);
            const fe9 a0 = F::add(x0, x2), a1 = F::add(x1, x3);
            const fe9 a2 = mulw<false>(F::sub(x0, x2), pl.tw_inv, exA, pl.qr), a3 = mulw<false>(F::sub(x1, x3), pl.tw_inv, exB, pl.qr);
            t.put(e0, q, F::norm(F::add(a0, a1)));
            t.put(e2, q, F::norm(F::add(a2, a3)));
            t.put(e1, q, mulw<false>(F::sub(a0, a1), pl.tw_inv, exC, pl.qr));
            t.put(e3, q, mulw<false>(F::sub(a2, a3), pl.tw_inv, exD, pl.qr));
        }
        __syncthreads();
        const bool red = dif_run<true>(t, u4, q, 2, Lhi, 2, L, Llo, g, pl, G);      // (wave-uniform: it depends on the stage count only)
        for (uint32_t e = u4; e < G; e += G / 4) {
            const size_t idx = ((size_t)e << Llo) + g;
            if (red) store_lazy<true>(vec + idx * batch + q0 + q, t.get(e, q), pl.qr); else store_lazy(vec + idx * batch + q0 + q, t.get(e, q), pl.qr);
        }
        cs.end();
        return;
    }
    if (plane) {      // (wave-uniform) the small-integer witness path left this vector as a byte plane; the few wide rows are 32-byte elements in `vec`
        const int8_t* pp = plane + ((q0 + q) >> 6) * nr.crows * 64 + ((q0 + q) & 63);
        for (uint32_t e = u4; e < G; e += G / 4) {
            const size_t idx = ((size_t)e << Llo) + g;
            fe9 x = F::zero();
            if (idx < m) {
                const int tv = (int)pp[idx * 64];
                x = tv == (int)WS_PLANE_WIDE ? F::unpack(ld_stream(vec + idx * batch + q0 + q)) : narrow9(tv);
            }
            t.put(e, q, x);
        }
    } else {
        for (uint32_t e = u4; e < G; e += G / 4) {
            const size_t idx = ((size_t)e << Llo) + g;
            t.put(e, q, idx < m ? F::unpack(ld_stream(vec + idx * batch + q0 + q)) : F::zero());
        }
    }
    __syncthreads();
    const bool red = dif_run<true>(t, u4, q, 0, Lhi, 0, L, Llo, g, pl, G);
    for (uint32_t e = u4; e < G; e += G / 4) {
        const size_t idx = ((size_t)e << Llo) + g;
        if (red) store_lazy<true>(vec + idx * batch + q0 + q, t.get(e, q), pl.qr); else store_lazy(vec + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
    cs.end();
}

// K2: contiguous DIF tail, scale by zeta^j / n (and into the 2^261 domain), contiguous DIT head.  grid (2^Lhi, batch/P, 2: a, b); block (2^(Llo-2) * P)
__global__ __launch_bounds__(256) void k_ntt_mid_contig(NttPlan pl, fe* v0, fe* v1, fe* v2, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t Cn = 1u << Llo;
    fe* vec = blockIdx.z == 0 ? v0 : blockIdx.z == 1 ? v1 : v2;
    const uint32_t b = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, u4 = threadIdx.x / P;
    const ClkStamp cs(pl.clk, 1);
    Tile t{smem, Cn * P};
    for (uint32_t e = u4; e < Cn; e += Cn / 4) {
        const size_t idx = ((size_t)b << Llo) + e;
        t.put(e, q, F::unpack(ld_stream(vec + idx * batch + q0 + q)));
    }
    __syncthreads();
    dif_run<false>(t, u4, q, Lhi, L - 2, 0, L, Llo, b, pl, Cn);      // all but the last two inverse stages
    {
        // The last two inverse stages, the scaling and the first two forward stages couple the SAME four consecutive elements 4 u4 .. 4 u4 + 3: they run in
        // registers — two LDS round trips, two barriers and the range reductions between them are gone, and the twiddles that are 1 for every thread
        // (exponent 0: the even pairs of stage L-2 and of stage 1) cost no product.  Same values mod r as the stage-by-stage form.
        const uint32_t e0 = 4 * u4;
        const size_t idx0 = ((size_t)b << Llo) + e0;
        const uint32_t exQ = 1u << (L - 2);                              // the primitive fourth root's exponent: the odd pairs of stage L-2 (inverse) and of stage 1 (forward)
        const fe9 x0 = t.get(e0, q), x1 = t.get(e0 + 1, q), x2 = t.get(e0 + 2, q), x3 = t.get(e0 + 3, q);
        const fe9 a0 = F::add(x0, x2), a1 = F::add(x1, x3);
        const fe9 a2 = F::sub(x0, x2), a3 = mulw<false>(F::sub(x1, x3), pl.tw_inv, exQ, pl.qr);
        // (products contract whatever the range: at most four unreduced doublings here, < 32 r against the 111 r a product takes)
        const fe9 z0 = F::mul(F::norm(F::add(a0, a1)), F::load(pl.scale_mid + idx0)), z1 = F::mul(F::norm(F::sub(a0, a1)), F::load(pl.scale_mid + idx0 + 1));
        const fe9 z2 = F::mul(F::norm(F::add(a2, a3)), F::load(pl.scale_mid + idx0 + 2)), z3 = F::mul(F::norm(F::sub(a2, a3)), F::load(pl.scale_mid + idx0 + 3));
        const fe9 c0 = F::norm(F::add(z0, z1)), c1 = F::sub(z0, z1);     // forward stage 0: pairs (e0, e0 + 1), (e0 + 2, e0 + 3), every twiddle 1
        const fe9 w2 = F::norm(F::add(z2, z3)), w3 = mulw<false>(F::sub(z2, z3), pl.tw_fwd, exQ, pl.qr);      // forward stage 1: pairs (e0, e0 + 2) twiddle 1, (e0 + 1, e0 + 3) the fourth root
        t.put(e0, q, F::add(c0, w2)); t.put(e0 + 2, q, F::sub(c0, w2));
        t.put(e0 + 1, q, F::add(c1, w3)); t.put(e0 + 3, q, F::sub(c1, w3));
    }
    __syncthreads();
    dit_run<false, true>(t, u4, q, 2, Llo, L, Llo, b, pl, Cn);
    for (uint32_t e = u4; e < Cn; e += Cn / 4) {
        const size_t idx = ((size_t)b << Llo) + e;
        store_lazy(vec + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
    cs.end();
}

// K3: strided DIT tail for a and b; d = a*b on the zeta-coset; strided DIF head for d (written over a).  The last DIT stage and the
// first DIF stage pair the same elements (e, e + G/2), so they stay in registers: each thread keeps two such pairs and folds
// a, then b into one running value per element.
// EVAL (evaluation-form quotient, k_quot_bases.hip): the kernel stops at d — d_i * 2^261 mod r as a canonical integer, natural order,
// written over a — and the MSM takes it from there with the bases V_i; the inverse transform of d and the one of c are never run.
// EVAL == 2: d is not written either: the thread recodes its four values into the signed c-bit digits of the windowed MSM and writes those
// (kernels.hpp QuotDigits; the recoding of k_msm_win.hip k_recode for canonical scalars): its elements u4 + {0, 1, 2, 3} * G/4 are the
// bases 4 m .. 4 m + 3, m = g * G/4 + u4, i.e. one half of octet m / 2 — one 16-byte word of four int32 digits per window when c > 16, half
// a word of int16 digits otherwise.
template <int EVAL>
__global__ __launch_bounds__(512) void k_ntt_pointwise_strided(NttPlan pl, fe* va, const fe* vb, size_t batch, QuotDigits qd) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t G = 1u << Lhi;
    const uint32_t g = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, u4 = threadIdx.x / P;
    const ClkStamp cs(pl.clk, 2);
    Tile t{smem, G * P};
    fe9 lo0, hi0, lo1, hi1;        // running values at elements (u4 + j*G/4) and (u4 + j*G/4 + G/2), j = 0, 1
    for (int k = 0; k < 2; k++) {
        const fe* vec = k == 0 ? va : vb;
        for (uint32_t e = u4; e < G; e += G / 4) {
            const size_t idx = ((size_t)e << Llo) + g;
            t.put(e, q, F::unpack(ld_stream(vec + idx * batch + q0 + q)));
        }
        __syncthreads();
        dit_run<true>(t, u4, q, Llo, L - 2, L, Llo, g, pl, G);
        {
            // The last two forward stages (s = L-2, L-1) couple this thread's own four elements u4 + {0, 1, 2, 3} G/4: in registers, straight into the
            // pointwise product (stage L-2 pairs (e0, e1), (e2, e3); stage L-1 pairs (e0, e2), (e1, e3); twiddle exponents of the plain radix-2 stages).
            const uint32_t e0 = u4, e1 = u4 + G / 4, e2 = u4 + G / 2, e3 = u4 + 3 * (G / 4);
            const uint32_t g0 = (e0 << Llo) + g, g1 = (e1 << Llo) + g, g2 = (e2 << Llo) + g;
            const uint32_t m1 = (1u << (L - 2)) - 1, m2 = (1u << (L - 1)) - 1;
            const uint32_t exA = (g0 & m1) << 1, exB = (g2 & m1) << 1, exC = g0 & m2, exD = g1 & m2;
            const fe9 n0 = F::norm(t.get(e0, q)), n2 = F::norm(t.get(e2, q));
            const fe9 v1 = mulw<false>(t.get(e1, q), pl.tw_fwd, exA, pl.qr), v3 = mulw<false>(t.get(e3, q), pl.tw_fwd, exB, pl.qr);
            const fe9 a0 = F::norm(F::add(n0, v1)), a1 = F::sub(n0, v1);
            const fe9 w2 = mulw<false>(F::add(n2, v3), pl.tw_fwd, exC, pl.qr), w3 = mulw<false>(F::sub(n2, v3), pl.tw_fwd, exD, pl.qr);
            // tight; |value| <= 2^256/r + 9 * 1.1 < 16 r
            const fe9 x0 = F::norm(F::add(a0, w2)), x2 = F::norm(F::sub(a0, w2)), x1 = F::norm(F::add(a1, w3)), x3 = F::norm(F::sub(a1, w3));
            if (k == 0) { lo0 = x0; hi0 = x2; lo1 = x1; hi1 = x3; }
            else { lo0 = F::mul(lo0, x0); hi0 = F::mul(hi0, x2); lo1 = F::mul(lo1, x1); hi1 = F::mul(hi1, x3); }      // a*b in (-1.6r, 2.6r)
        }
        __syncthreads();
    }
    if (EVAL == 1) {
        auto put_d = [&](uint32_t e, const fe9& v) { st_stream(va + (((size_t)e << Llo) + g) * batch + q0 + q, F::pack(F::freeze_near(v))); };      // a*b in (-1.4 r, 2.4 r)
        put_d(u4, lo0); put_d(u4 + G / 2, hi0); put_d(u4 + G / 4, lo1); put_d(u4 + G / 4 + G / 2, hi1);
        return;
    }
    if (EVAL == 2) {
        fe s[4] = {F::pack(F::freeze_near(lo0)), F::pack(F::freeze_near(lo1)), F::pack(F::freeze_near(hi0)), F::pack(F::freeze_near(hi1))};      // elements u4 + {0, 1, 2, 3} * G/4; a*b in (-1.4 r, 2.4 r)
        const size_t mq = (size_t)g * (G / 4) + u4, o = mq >> 1, half = mq & 1, noct = ((size_t)1 << L) / 8, p = q0 + q;
        const uint32_t c = (uint32_t)qd.c, cmask = (1u << c) - 1, D = 1u << (c - 1);
        uint32_t carry = 0;
        if (c == 17 && qd.nwin == 15) {      // the bench configuration's digit width: window j sits at a compile-time bit offset — one funnel shift instead of shifting the whole scalar down every window
#pragma unroll
            for (int j = 0; j < 15; j++) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int bit = 17 * j, wi = bit >> 5, sh = bit & 31;
                    const uint32_t bits = wi + 1 < 8 ? __builtin_amdgcn_alignbit(s[i].l[wi + 1], s[i].l[wi], sh) : s[i].l[wi] >> sh;
                    const uint32_t raw = (bits & 0x1FFFFu) + ((carry >> i) & 1u);
                    int32_t dg = (int32_t)raw;
                    if (raw >= 0x10000u) { dg -= (int32_t)0x20000; carry |= 1u << i; } else carry &= ~(1u << i);
                    w[i] = (uint32_t)dg;
                }
                const size_t at = ((size_t)j * noct + o) * batch + p;
                qd.digits[2 * at + half] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            cs.end();
            return;
        }
        for (int j = 0; j < qd.nwin; j++) {
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t raw = (s[i].l[0] & cmask) + ((carry >> i) & 1u);
#pragma unroll
                for (int k = 0; k < 7; k++) s[i].l[k] = __builtin_amdgcn_alignbit(s[i].l[k + 1], s[i].l[k], c);
                s[i].l[7] >>= c;
                int32_t dg = (int32_t)raw;
                if (raw >= D) { dg -= (int32_t)(1u << c); carry |= 1u << i; } else carry &= ~(1u << i);
                w[i] = (uint32_t)dg;
            }
            const size_t at = ((size_t)j * noct + o) * batch + p;
            if (c > 16) qd.digits[2 * at + half] = make_uint4(w[0], w[1], w[2], w[3]);
            else reinterpret_cast<uint2*>(qd.digits + at)[half] = make_uint2((w[0] & 0xFFFFu) | (w[1] << 16), (w[2] & 0xFFFFu) | (w[3] << 16));
        }
        cs.end();
        return;
    }
    auto first_dif = [&](uint32_t e1, const fe9& lo, const fe9& hi) {       // first DIF stage (s = 0) on d: same pairs; twiddle exponent = gidx(e1) mod n/2
        const uint32_t e2 = e1 + G / 2;
        const uint32_t ex = ((e1 << Llo) + g) & ((1u << (L - 1)) - 1);
        t.put(e1, q, F::norm(F::add(lo, hi)));
        t.put(e2, q, mulw<false>(F::sub(lo, hi), pl.tw_inv, ex, pl.qr));
    };
    first_dif(u4, lo0, hi0);
    first_dif(u4 + G / 4, lo1, hi1);
    __syncthreads();
    dif_run<true>(t, u4, q, 1, Lhi, 1, L, Llo, g, pl, G);
    for (uint32_t e = u4; e < G; e += G / 4) {
        const size_t idx = ((size_t)e << Llo) + g;
        store_lazy(va + idx * batch + q0 + q, t.get(e, q), pl.qr);
    }
}

// K4: contiguous DIF tails of c (-> n S, still in the solver's 2^256 domain) and of d (-> n zeta^j D_j, 2^261 domain), then
// h = S/2 - D/2 with ONE reduction and out of Montgomery form: c * (16 / n) - d * (zeta^-j / 2n) (canonical output in [0, r),
// written over a).  A thread keeps its four S values in registers while the tile is reused for d.
__global__ __launch_bounds__(256) void k_ntt_final_contig(NttPlan pl, fe* vh, const fe* vc, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int L = pl.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    const uint32_t Cn = 1u << Llo;
    const uint32_t b = blockIdx.x; const size_t q0 = (size_t)blockIdx.y * P;
    const uint32_t q = threadIdx.x % P, u4 = threadIdx.x / P;
    Tile t{smem, Cn * P};
    auto run_tail = [&](const fe* vec) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t e = u4 + k * (Cn / 4); const size_t idx = ((size_t)b << Llo) + e;
            t.put(e, q, F::unpack(ld_stream(vec + idx * batch + q0 + q)));
        }
        __syncthreads();
        dif_run<false>(t, u4, q, Lhi, L, 0, L, Llo, b, pl, Cn);
    };
    run_tail(vc);
    fe9 s0 = t.get(u4, q), s1 = t.get(u4 + Cn / 4, q), s2 = t.get(u4 + 2 * (Cn / 4), q), s3 = t.get(u4 + 3 * (Cn / 4), q);
    __syncthreads();
    run_tail(vh);
    const fe9 kc = F::load(pl.half_c);
    auto finish = [&](uint32_t e, const fe9& sv) {
        const size_t idx = ((size_t)b << Llo) + e;
        st_stream(vh + idx * batch + q0 + q, F::pack(F::freeze(F::fmms(sv, kc, t.get(e, q), F::load(pl.scale_out + idx)))));
    };
    finish(u4, s0); finish(u4 + Cn / 4, s1); finish(u4 + 2 * (Cn / 4), s2); finish(u4 + 3 * (Cn / 4), s3);
}

}  // namespace

namespace {
hipError_t launch_quotient(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s, size_t ncols, int eval, const QuotDigits& qd, const NttNarrow* narrow);
}
hipError_t launch_compute_h(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s, size_t ncols, const NttNarrow* narrow) { return launch_quotient(p, a, b, c, m, batch, s, ncols, 0, QuotDigits{nullptr, 0, 0}, narrow); }
hipError_t launch_compute_d(const NttPlan& p, fe* a, fe* b, size_t m, size_t batch, hipStream_t s, size_t ncols, const NttNarrow* narrow) { return launch_quotient(p, a, b, nullptr, m, batch, s, ncols, 1, QuotDigits{nullptr, 0, 0}, narrow); }
hipError_t launch_compute_d_digits(const NttPlan& p, fe* a, fe* b, size_t m, size_t batch, const QuotDigits& qd, hipStream_t s, const NttNarrow* narrow) {
    if (!qd.digits || qd.c < 4 || qd.c > MSM_MAX_WINDOW || qd.nwin != msm_windows(qd.c)) return hipErrorInvalidValue;
    return launch_quotient(p, a, b, nullptr, m, batch, s, 0, 2, qd, narrow);
}
namespace {
hipError_t launch_quotient(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s, size_t ncols, int eval, const QuotDigits& qd, const NttNarrow* narrow) {
    const int L = p.L, Lhi = (L + 1) / 2, Llo = L - Lhi;
    if (L < NTT_MIN_LOG2 || L > NTT_MAX_LOG2 || batch % P) return hipErrorInvalidValue;      // block sizes / launch bounds below assume this range
    const unsigned G = 1u << Lhi, Cn = 1u << Llo;
    if (ncols > batch) return hipErrorInvalidValue;
    const unsigned pb = (unsigned)((ncols ? (ncols + P - 1) / P * P : batch) / P);      // groups of P columns that are transformed
    const size_t lds_s = (size_t)G * P * 36, lds_c = (size_t)Cn * P * 36;      // nine 32-bit limb planes per element
    hipError_t e = hipSuccess;
    auto opt_in = [&](const void* f, size_t lds) { if (e == hipSuccess && lds > 65536) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); };
    // 2^17 domains (AES-V2): 72 KiB tiles need the opt-in LDS limit (a CU has 160 KiB)
    opt_in(reinterpret_cast<const void*>(k_ntt_dif_strided), lds_s);
    opt_in(eval == 2 ? reinterpret_cast<const void*>(k_ntt_pointwise_strided<2>) : eval ? reinterpret_cast<const void*>(k_ntt_pointwise_strided<1>) : reinterpret_cast<const void*>(k_ntt_pointwise_strided<0>), lds_s);
    opt_in(reinterpret_cast<const void*>(k_ntt_mid_contig), lds_c); opt_in(reinterpret_cast<const void*>(k_ntt_final_contig), lds_c);
    if (e != hipSuccess) return e;
    NttNarrow nr = narrow ? *narrow : NttNarrow{{nullptr, nullptr, nullptr}, 0, 0};
    // plain-integer inputs: evaluation form only (the coefficient form's last kernel prices c in the solver's domain), both vectors as planes, tables present
    nr.plain = nr.plain && eval && nr.plane[0] && nr.plane[1] && p.tw_inv_plain && p.scale_mid_plain && Lhi >= 3 ? 1 : 0;
    NttPlan p2 = p; if (nr.plain) p2.scale_mid = p.scale_mid_plain;
    hipLaunchKernelGGL(k_ntt_dif_strided, dim3(Cn, pb, eval ? 2 : 3), dim3(G / 4 * P), lds_s, s, p, a, b, c, m, batch, nr);
    hipLaunchKernelGGL(k_ntt_mid_contig, dim3(G, pb, 2), dim3(Cn / 4 * P), lds_c, s, p2, a, b, c, batch);
    if (eval == 2) { hipLaunchKernelGGL(k_ntt_pointwise_strided<2>, dim3(Cn, pb, 1), dim3(G / 4 * P), lds_s, s, p, a, b, batch, qd); return hipGetLastError(); }
    if (eval) { hipLaunchKernelGGL(k_ntt_pointwise_strided<1>, dim3(Cn, pb, 1), dim3(G / 4 * P), lds_s, s, p, a, b, batch, qd); return hipGetLastError(); }
    hipLaunchKernelGGL(k_ntt_pointwise_strided<0>, dim3(Cn, pb, 1), dim3(G / 4 * P), lds_s, s, p, a, b, batch, qd);
    hipLaunchKernelGGL(k_ntt_final_contig, dim3(G, pb, 1), dim3(Cn / 4 * P), lds_c, s, p, a, c, batch);
    return hipGetLastError();
}
}  // namespace

}  // namespace gsc
