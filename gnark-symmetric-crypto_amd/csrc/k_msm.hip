// Fixed-base multi-scalar multiplication over HBM-resident digit tables, and proof assembly.
//
// Replaces the five MultiExp calls of groth16.Prove (reference libraries/prover/impl/provers.go:148,216;
// gnark-crypto (*G1Jac).MultiExp / (*G2Jac).MultiExp — SURVEY.md §8(a) a8-a12, algebra App. D).
//
// Every base of a proving key is fixed for the life of the process, and an MI355X has 288 GB of HBM, so
// InitAlgorithm precomputes T[k][j][d] = d * 2^(c j) * P_k for all signed c-bit digits d (k_init.hip).
// An MSM is then a pure gather-accumulate: sum_k sum_j +-T[k][j][|d_kj|] — no buckets, no sorting, no
// atomics, no inter-thread hazards.  Lanes of a wave are 64 proofs working on the same base k, so the
// scalar loads are coalesced (2 KiB per wave) and the table gathers of a wave fall into one 2^(c-1)*64 B row.
// Partial sums per (slice of bases, proof) are reduced with wavefront __shfl_xor butterflies.
#include "kernels.hpp"
#include "bn254_fp29.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

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

// raw memory image of one affine table entry (2 * WORDS field elements), kept packed until it is consumed
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

// One slice of bases for one proof (lane).  EXACT = false is the hot path (no degenerate-case tests inside madd).
// Software pipelining: the table entry for the NEXT non-zero digit and the scalar of the NEXT base are requested before the
// current mixed addition (~2 500 instructions) starts, so the 64-byte random HBM gathers are never on the critical path.
//
// Groups (bases [nwide, nwide + nbit)): every wire of these circuits is tiny after sign normalisation — ChaCha20-V3: 62 % bits,
// 38 % values in {-1, 0, 1} — and lanes are different proofs, so a wave pays one mixed addition per base as soon as a single
// proof has a non-zero value.  Eight such bases are taken together instead: the eight scalars become a balanced-ternary number
// and ONE addition of the tabulated signed subset sum replaces up to eight.  The grouping is a prediction made at
// InitAlgorithm; here every wave checks it (all 64 proofs, all eight scalars in {-1, 0, 1}) and otherwise walks the eight
// bases through the digit tables like any other base, so results never depend on the prediction.
template <class F, bool EXACT, bool BITS>
__device__ __forceinline__ Xyzz9<F> accumulate_slice(const MsmArgs& a, size_t k0, size_t k1, size_t p) {
    using C = Curve9<F>;
    const fe* sub = reinterpret_cast<const fe*>(a.sub);
    const size_t bit0 = BITS ? a.nwide : 0, bit1 = BITS ? a.nwide + a.nbit : 0;
    fe minus_one;                                   // -1 in the representation of the scalars
    {
        fe one1 = fe{}; one1.l[0] = 1;
        minus_one = Fr::neg(a.scalars_mont ? Fr::one() : one1);
    }
    Xyzz9<F> acc = C::infinity();
    RawAff<F> pend; bool have = false, pend_neg = false;
    auto scalar_of = [&](size_t k) { const size_t row = a.rows ? uni(a.rows[k]) : k; return load_fe(a.scalars + row * a.batch + p); };
    size_t k = k0, single_until = 0;        // bases below single_until are walked one by one even inside the bit-group region
    fe s_next = fe{}; bool have_next = false;
    while (k < k1) {
        if (BITS && k >= bit0 && k < bit1 && k >= single_until) {      // wave-uniform: k is a multiple of 8 here
            int32_t v = 0; bool ok = true;
#pragma unroll 1
            for (int h = 0; h < 8; h += 4) {        // four scalars in flight at a time: eight would cost a wave of occupancy
                fe s4[4];
#pragma unroll
                for (int b = 0; b < 4; b++) s4[b] = scalar_of(k + h + b);
                int32_t w3 = h ? 81 : 1;            // 3^(h+b)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t z = 0, o = 0, m = 0;   // == 0, == 1, == -1 (Montgomery images when the scalars are)
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t one = a.scalars_mont ? FrParams::one(i) : (i == 0 ? 1u : 0u);
                        z |= s4[b].l[i]; o |= s4[b].l[i] ^ one; m |= s4[b].l[i] ^ minus_one.l[i];
                    }
                    ok = ok && (z == 0 || o == 0 || m == 0);
                    v += o == 0 ? w3 : (m == 0 ? -w3 : 0);
                    w3 *= 3;
                }
            }
            const size_t grp = (k - bit0) >> 3;
            if (__all(ok) && uni(a.group_ok[grp])) {
                if (v) {
                    const uint32_t idx = (uint32_t)(v < 0 ? -v : v) - 1;
                    const RawAff<F> nxt = load_raw<F>(sub + (grp * MSM_GROUP_ENTRIES + idx) * (2 * F::WORDS));
                    if (have) acc = C::template madd<EXACT>(acc, unpack_aff(pend, pend_neg));
                    pend = nxt; pend_neg = v < 0; have = true;
                }
                k += 8;
                continue;
            }
            single_until = k + 8; have_next = false;
        }
        fe s = have_next ? s_next : scalar_of(k);
        have_next = k + 1 < k1 && (!BITS || k + 1 < bit0 || k + 1 >= bit1 || k + 1 < single_until);
        if (have_next) s_next = scalar_of(k + 1);
        if (a.scalars_mont) s = Fr::from_mont(s);
        const bool neg = sign_normalise(s);
        // digit table of this base (wave-uniform): the wide-digit table for the bases predicted to carry full-width scalars
        const bool wide = BITS && k < a.nwide;
        const uint32_t c = wide ? (uint32_t)a.c2 : (uint32_t)a.c, nwin = wide ? (uint32_t)a.nwin2 : (uint32_t)a.nwin, D = 1u << (c - 1);
        const fe* table = reinterpret_cast<const fe*>(wide ? a.table2 : a.table);
        // number of windows this lane needs: highest set bit / c + 1 (+1 for a possible carry)
        int top = -1;
#pragma unroll
        for (int i = 7; i >= 0; i--) if (top < 0 && s.l[i]) top = 32 * i + 31 - __clz(s.l[i]);
        uint32_t need = top < 0 ? 0u : (uint32_t)top / c + 2u;
        if (need > nwin) need = nwin;
        uint32_t carry = 0;
        const uint32_t cmask = (1u << c) - 1;
        for (uint32_t j = 0; j < need; j++) {
            uint32_t raw = (s.l[0] & cmask) + carry;        // the scalar is shifted down one digit per step: no variable limb indexing
#pragma unroll
            for (int i = 0; i < 7; i++) s.l[i] = __builtin_amdgcn_alignbit(s.l[i + 1], s.l[i], c);
            s.l[7] >>= c;
            bool dneg = false;
            if (raw > D) { raw = (1u << c) - raw; dneg = true; carry = 1; } else carry = 0;
            if (raw) {
                const RawAff<F> nxt = load_raw<F>(table + (((size_t)k * nwin + j) * D + (raw - 1)) * (2 * F::WORDS));
                if (have) acc = C::template madd<EXACT>(acc, unpack_aff(pend, pend_neg));
                pend = nxt; pend_neg = dneg != neg; have = true;
            }
        }
        k++;
    }
    if (have) acc = C::template madd<EXACT>(acc, unpack_aff(pend, pend_neg));
    return acc;
}

// BITS = false is the instantiation for sets without bit groups (the Z tables: the dominant launch keeps its register budget)
template <class F, bool BITS>
__global__ __launch_bounds__(64) void k_msm(MsmArgs a) {
    using C = Curve9<F>;
    // XCD-aware order: workgroups go round-robin over the 8 XCDs by linear id, and each XCD has its own L2.  All proof groups of
    // a slice read the same table rows, so they are placed on ONE XCD (consecutive ids there), not spread over all eight.
    size_t slice = blockIdx.y, grp = blockIdx.x;
    {
        const size_t G = gridDim.x, L = (size_t)blockIdx.y * G + blockIdx.x, S8 = (size_t)gridDim.y & ~(size_t)7;
        if (L < S8 * G) { const size_t xcd = L & 7, i = L >> 3; slice = (i / G) * 8 + xcd; grp = i % G; }
    }
    const size_t p = grp * 64 + threadIdx.x;
    const size_t per = ((a.nbases + a.nslices - 1) / a.nslices + 7) & ~(size_t)7;      // bit groups never straddle slices
    const size_t k0 = slice * per < a.nbases ? slice * per : a.nbases, k1 = k0 + per < a.nbases ? k0 + per : a.nbases;
    Xyzz9<F> acc = accumulate_slice<F, false, BITS>(a, k0, k1, p);
    // A degenerate step (accumulator == +-entry) zeroes ZZ for good; it cannot be told from a genuine point at infinity
    // without the exact tests, so the (very rare) lane is recomputed with them.
    if (!acc.inf && F::is_zero(acc.zz)) acc = accumulate_slice<F, true, BITS>(a, k0, k1, p);
    C::store_xyzz(reinterpret_cast<fe*>(a.partial) + (slice * a.batch + p) * (4 * F::WORDS), acc);
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

void launch_msm_g1(const MsmArgs& a, hipStream_t s) {
    if (a.nbit) hipLaunchKernelGGL((k_msm<Fp29f, true>), dim3((unsigned)(a.batch / 64), (unsigned)a.nslices), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_msm<Fp29f, false>), dim3((unsigned)(a.batch / 64), (unsigned)a.nslices), dim3(64), 0, s, a);
}
void launch_msm_g2(const MsmArgs& a, hipStream_t s) {
    if (a.nbit) hipLaunchKernelGGL((k_msm<Fp2x, true>), dim3((unsigned)(a.batch / 64), (unsigned)a.nslices), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_msm<Fp2x, false>), dim3((unsigned)(a.batch / 64), (unsigned)a.nslices), dim3(64), 0, s, a);
}
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
void launch_finalize(const G1Xyzz* sumA, const G1Xyzz* sumB1, const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ,
                     const uint8_t* rs, size_t batch, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s) {
    const unsigned nb = (unsigned)((batch + 63) / 64);
    hipLaunchKernelGGL(k_fin_scalarmul, dim3(nb, 2), dim3(64), 0, s, sumA, sumB1, rs, batch, tmp, out, flags);
    hipLaunchKernelGGL(k_fin_combine, dim3(nb, 2), dim3(64), 0, s, sumB2, sumK, sumZ, tmp, batch, out, flags);
}

}  // namespace gsc
