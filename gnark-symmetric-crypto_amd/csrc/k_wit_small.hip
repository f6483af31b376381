// The small-integer witness path on the GPU (layout and rationale: wit_small.hpp).
//
// Replaces cs.Solve inside groth16.Prove (reference libraries/prover/impl/provers.go:148; gnark constraint/bn254 solver — SURVEY.md
// §8(a) a5) and the witness assignment of provers.go:106-142 for circuits whose witness is small integers (ChaCha20-V3).
// Lanes of a wave are 64 proofs; everything a wave touches for a wire or a constraint row is ONE 64-byte segment.
#include "kernels.hpp"
#include "wit_small.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Descriptors are wave-uniform; a wave fetches eight of them (128 words) with two coalesced loads — lane t of `lo` holds word t of items
// 0..3, of `hi` items 4..7 — and reads fields with v_readlane (compile-time lane numbers), so that the order below is what runs:
// ONE round of 48 byte loads for the eight items, then the arithmetic.  (With plain indexing the compiler re-reads the descriptors
// item by item and the items' loads become eight dependent rounds.)
struct Bundle { uint32_t lo, hi; };
__device__ __forceinline__ Bundle bundle_fetch(const uint32_t* __restrict__ items, uint32_t i, uint32_t lane) {
    const uint32_t* d = items + (size_t)WS_TINY_WORDS * i;
    return Bundle{d[lane], d[64 + lane]};
}
template <int K, int F> __device__ __forceinline__ uint32_t bfield(const Bundle& b) { return (uint32_t)__builtin_amdgcn_readlane((int)(K < 4 ? b.lo : b.hi), 16 * (K & 3) + F); }
struct TinyVals { int v[6]; };
template <int K> __device__ __forceinline__ TinyVals tiny_load(const Bundle& b, const int8_t* __restrict__ Wg, uint32_t lane) {
    TinyVals t;
    t.v[0] = (int)Wg[(size_t)bfield<K, 2>(b) * 64 + lane]; t.v[1] = (int)Wg[(size_t)bfield<K, 3>(b) * 64 + lane];
    t.v[2] = (int)Wg[(size_t)bfield<K, 4>(b) * 64 + lane]; t.v[3] = (int)Wg[(size_t)bfield<K, 5>(b) * 64 + lane];
    t.v[4] = (int)Wg[(size_t)bfield<K, 6>(b) * 64 + lane]; t.v[5] = (int)Wg[(size_t)bfield<K, 7>(b) * 64 + lane];
    return t;
}
// L = c0 v0 + c1 v1, R, O alike (|c| < 2^28, v in {-1, 0, 1})
template <int K> __device__ __forceinline__ void tiny_sums(const Bundle& b, const TinyVals& t, int& L, int& R, int& O) {
    L = (int)bfield<K, 8>(b) * t.v[0] + (int)bfield<K, 9>(b) * t.v[1];
    R = (int)bfield<K, 10>(b) * t.v[2] + (int)bfield<K, 11>(b) * t.v[3];
    O = (int)bfield<K, 12>(b) * t.v[4] + (int)bfield<K, 13>(b) * t.v[5];
}
template <int K> __device__ __forceinline__ void chain_item(const Bundle& b, const TinyVals& t, int8_t* __restrict__ Wg, uint32_t lane, bool& bad) {
    int L, R, O; tiny_sums<K>(b, t, L, R, O);
    long long w = (long long)L * R - O;
    if (bfield<K, 0>(b) & WS_F_NEG) w = -w;
    bad |= (unsigned long long)(w + 1) > 2ull;                         // the solved wire was predicted to stay in {-1, 0, 1}
    Wg[(size_t)bfield<K, 1>(b) * 64 + lane] = (int8_t)w;               // (padding items write 0 to the scratch row)
}

// the terms of one part of an nBits sum: lane t < 32 holds term t
struct PartTerms { uint32_t tw; long long tc; };
__device__ __forceinline__ PartTerms part_fetch(const uint32_t* __restrict__ twire, const long long* __restrict__ tcoef, uint32_t tt, uint32_t lane) { return PartTerms{twire[tt + (lane & 31)], tcoef[tt + (lane & 31)]}; }
struct PartVals { int v[WS_PART_CHUNKS * WS_CHUNK]; };
__device__ __forceinline__ PartVals part_load(const PartTerms& t, uint32_t nch, const int8_t* __restrict__ Wg, uint32_t lane) {
    PartVals r;
#pragma unroll
    for (uint32_t c = 0; c < WS_PART_CHUNKS; c++) {
        if (c < nch) {
#pragma unroll
            for (uint32_t k = 0; k < WS_CHUNK; k++) r.v[c * WS_CHUNK + k] = (int)Wg[(size_t)(uint32_t)__builtin_amdgcn_readlane((int)t.tw, c * WS_CHUNK + k) * 64 + lane];
        } else {
#pragma unroll
            for (uint32_t k = 0; k < WS_CHUNK; k++) r.v[c * WS_CHUNK + k] = 0;
        }
    }
    return r;
}
__device__ __forceinline__ long long part_sum(const PartTerms& t, const PartVals& pv, uint32_t nch) {
    const int clo = (int)(uint32_t)t.tc, chi = (int)(t.tc >> 32);
    long long acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < WS_PART_CHUNKS * WS_CHUNK; k++) {      // (terms beyond the part: coefficient 0 or value 0)
        const long long cf = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(chi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(clo, k));
        acc += (k < nch * WS_CHUNK ? cf : 0) * (long long)pv.v[k];
    }
    return acc;
}
struct BundleVals { TinyVals v0, v1, v2, v3, v4, v5, v6, v7; };
__device__ __forceinline__ BundleVals bundle_load(const Bundle& b, const int8_t* __restrict__ Wg, uint32_t lane) {
    return BundleVals{tiny_load<0>(b, Wg, lane), tiny_load<1>(b, Wg, lane), tiny_load<2>(b, Wg, lane), tiny_load<3>(b, Wg, lane), tiny_load<4>(b, Wg, lane), tiny_load<5>(b, Wg, lane), tiny_load<6>(b, Wg, lane), tiny_load<7>(b, Wg, lane)};
}
__device__ __forceinline__ void bundle_finish(const Bundle& b, const BundleVals& v, int8_t* __restrict__ Wg, uint32_t lane, bool& bad) {
    chain_item<0>(b, v.v0, Wg, lane, bad); chain_item<1>(b, v.v1, Wg, lane, bad); chain_item<2>(b, v.v2, Wg, lane, bad); chain_item<3>(b, v.v3, Wg, lane, bad);
    chain_item<4>(b, v.v4, Wg, lane, bad); chain_item<5>(b, v.v5, Wg, lane, bad); chain_item<6>(b, v.v6, Wg, lane, bad); chain_item<7>(b, v.v7, Wg, lane, bad);
}

// One workgroup per group of 64 proofs walks every level that produces wires; NW waves share a level's items.
// A level costs its dependent memory round trips: wire loads, then the stores' acknowledgement before the barrier.  The descriptors of a
// wave's FIRST bundle and part of the next level (static data) are therefore fetched while the current level computes — for ChaCha20-V3 a
// wave never has more than one of each per level — so that nothing but the wire loads stands between a barrier and the arithmetic.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_wit_chain(const uint32_t* __restrict__ tiny, const uint32_t* __restrict__ parts, const uint32_t* __restrict__ bits, const uint32_t* __restrict__ twire,
                                                       const long long* __restrict__ tcoef, const uint32_t* __restrict__ levels, uint32_t nlevels, int8_t* __restrict__ W8, size_t rows_per_group, uint32_t* __restrict__ flag) {
    extern __shared__ long long s_sum[];                     // [slot][lane]: the partial sums of the level's nBits inputs
    const uint32_t lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
    int8_t* __restrict__ Wg = W8 + (size_t)blockIdx.x * rows_per_group * 64;
    bool bad = false;
    struct Pref { Bundle b; PartTerms t; };
    auto prefetch = [&](uint32_t l) {
        Pref p{Bundle{0u, 0u}, PartTerms{0u, 0}};
        if (l < nlevels) {
            const uint32_t i = levels[6 * l] + wave * WS_IB, ip = levels[6 * l + 2] + wave;
            if (i < levels[6 * l + 1]) p.b = bundle_fetch(tiny, i, lane);
            if (ip < levels[6 * l + 3]) p.t = part_fetch(twire, tcoef, parts[4 * ip + 1], lane);
        }
        return p;
    };
    Pref nx = prefetch(0);
    for (uint32_t l = 0; l < nlevels; l++) {
        const uint32_t t0 = levels[6 * l], t1 = levels[6 * l + 1], p0 = levels[6 * l + 2], p1 = levels[6 * l + 3], b0 = levels[6 * l + 4], b1 = levels[6 * l + 5];
        const Pref cur = nx;
        // this wave's first bundle (products: out = +-(L R - O), eight items per load round) and first part (an nBits sum of at most 32 terms):
        // ONE round of wire loads; the next level's descriptors are requested behind them
        const uint32_t i0 = t0 + wave * WS_IB, ip0 = p0 + wave;
        const bool has_b = i0 < t1, has_p = ip0 < p1;
        if (has_b) {
            const BundleVals bv = bundle_load(cur.b, Wg, lane);
            if (!has_p) nx = prefetch(l + 1);
            __builtin_amdgcn_sched_barrier(0);
            bundle_finish(cur.b, bv, Wg, lane, bad);
        }
        if (has_p) {
            const uint32_t pslot = parts[4 * ip0], pnch = parts[4 * ip0 + 2];
            const PartVals pv = part_load(cur.t, pnch, Wg, lane);
            nx = prefetch(l + 1);
            __builtin_amdgcn_sched_barrier(0);
            s_sum[pslot * 64 + lane] = part_sum(cur.t, pv, pnch);
        }
        if (!has_b && !has_p) nx = prefetch(l + 1);
        // (levels wider than one bundle / part per wave: the rest, fetched in place)
        for (uint32_t i = i0 + NW * WS_IB; i < t1; i += NW * WS_IB) {
            const Bundle b = bundle_fetch(tiny, i, lane);
            const BundleVals v = bundle_load(b, Wg, lane);
            __builtin_amdgcn_sched_barrier(0);
            bundle_finish(b, v, Wg, lane, bad);
        }
        for (uint32_t i = ip0 + NW; i < p1; i += NW) {
            const uint32_t slot = parts[4 * i], nch = parts[4 * i + 2];
            const PartTerms t = part_fetch(twire, tcoef, parts[4 * i + 1], lane);
            const PartVals v = part_load(t, nch, Wg, lane);
            __builtin_amdgcn_sched_barrier(0);
            s_sum[slot * 64 + lane] = part_sum(t, v, nch);
        }
        if (b0 != b1) {                                          // (the same for every wave of the workgroup)
            __syncthreads();
            for (uint32_t i = b0 + wave; i < b1; i += NW) {
                const uint32_t out = bits[4 * i], sl = bits[4 * i + 1], bb = bits[4 * i + 2];
                const uint32_t slot0 = sl & 0xFFFFu, np = sl >> 16, sh = bb & 0xFFu, nb = bb >> 8;
                long long s = 0;
                for (uint32_t k = 0; k < np; k++) s += s_sum[(slot0 + k) * 64 + lane];
                bad |= s < 0;                                    // gnark's hint takes the bits of the canonical residue: r + s, not ours
                for (uint32_t b = 0; b < nb; b++) Wg[(size_t)(out + b) * 64 + lane] = (int8_t)((s >> (sh + b)) & 1);
            }
        }
        __syncthreads();                                         // the level's wires are visible to every wave of the workgroup
    }
    if (bad) atomicOr(flag, 1u);
}

// int64 -> Montgomery element of the solver's 2^256 domain
__device__ __forceinline__ fe fe_from_i64(long long v) {
    const unsigned long long m = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    fe x = Fr::zero(); x.l[0] = (uint32_t)m; x.l[1] = (uint32_t)(m >> 32);
    x = Fr::to_mont(x);
    return v < 0 ? Fr::neg(x) : x;
}
__device__ __forceinline__ void store_row(uint32_t wide, int8_t* __restrict__ plane, fe* __restrict__ mat, size_t crows, size_t batch, uint32_t cidx, long long v, uint32_t lane, bool& bad) {
    const size_t g = blockIdx.x;
    if (wide) store_fe(mat + (size_t)cidx * batch + g * 64 + lane, fe_from_i64(v));
    else { bad |= (unsigned long long)(v + 1) > 2ull; plane[(g * crows + cidx) * 64 + lane] = (int8_t)v; }
}

template <int K> __device__ __forceinline__ void row_item(const Bundle& b, const TinyVals& t, const WitRowsArgs& a, uint32_t lane, bool& bad, uint32_t& failed) {
    const uint32_t fl = bfield<K, 0>(b), cidx = bfield<K, 1>(b);
    if (!(fl & WS_F_ITEM)) return;                                     // padding (wave-uniform)
    int L, R, O; tiny_sums<K>(b, t, L, R, O);
    if ((long long)L * R != (long long)O) failed = failed < cidx ? failed : cidx;
    store_row((fl >> WS_CLS_SHIFT_A) & 1u, a.A8, a.A, a.crows, a.batch, cidx, L, lane, bad);
    store_row((fl >> WS_CLS_SHIFT_B) & 1u, a.B8, a.B, a.crows, a.batch, cidx, R, lane, bad);
    store_row((fl >> WS_CLS_SHIFT_C) & 1u, a.C8, a.C, a.crows, a.batch, cidx, O, lane, bad);
}

// a = L(w), b = R(w), c = O(w) for every constraint, a b == c checked.  grid (proof groups, chunks of tiny constraints + general
// constraints); one wave per workgroup.
__global__ __launch_bounds__(64) void k_wit_rows(WitRowsArgs a) {
    const uint32_t lane = threadIdx.x;
    const int8_t* __restrict__ Wg = a.W8 + (size_t)blockIdx.x * a.rows_per_group * 64;
    const uint32_t* __restrict__ rtiny = a.rtiny; const uint32_t* __restrict__ rgen = a.rgen;      // (rgen / tw / tc: read with vector loads — the general rows are few)
    const uint32_t* __restrict__ tw = a.rtwire; const long long* __restrict__ tc = a.rtcoef;
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    bool bad = false; uint32_t failed = 0xFFFFFFFFu;
    if (blockIdx.y < a.n_tiny_chunks) {
        const uint32_t i0 = blockIdx.y * a.tiny_per_chunk, i1 = i0 + a.tiny_per_chunk < a.n_rtiny ? i0 + a.tiny_per_chunk : a.n_rtiny;
        for (uint32_t i = i0; i < i1; i += WS_IB) {
            const Bundle b = bundle_fetch(rtiny, i, lane);
            const TinyVals v0 = tiny_load<0>(b, Wg, lane), v1 = tiny_load<1>(b, Wg, lane), v2 = tiny_load<2>(b, Wg, lane), v3 = tiny_load<3>(b, Wg, lane);
            const TinyVals v4 = tiny_load<4>(b, Wg, lane), v5 = tiny_load<5>(b, Wg, lane), v6 = tiny_load<6>(b, Wg, lane), v7 = tiny_load<7>(b, Wg, lane);
            __builtin_amdgcn_sched_barrier(0);
            row_item<0>(b, v0, a, lane, bad, failed); row_item<1>(b, v1, a, lane, bad, failed); row_item<2>(b, v2, a, lane, bad, failed); row_item<3>(b, v3, a, lane, bad, failed);
            row_item<4>(b, v4, a, lane, bad, failed); row_item<5>(b, v5, a, lane, bad, failed); row_item<6>(b, v6, a, lane, bad, failed); row_item<7>(b, v7, a, lane, bad, failed);
        }
    } else {
        const uint32_t* __restrict__ d = rgen + 8 * (size_t)(blockIdx.y - a.n_tiny_chunks);
        const uint32_t fl = d[0], cidx = d[1];
        uint32_t tt = d[2];
        long long val[3] = {0, 0, 0};
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const uint32_t nch = d[3 + s];
            long long acc = 0;
#pragma unroll 1
            for (uint32_t c = 0; c < nch; c++, tt += WS_CHUNK) {
                int v[WS_CHUNK];
#pragma unroll
                for (uint32_t k = 0; k < WS_CHUNK; k++) v[k] = (int)Wg[(size_t)tw[tt + k] * 64 + lane];
#pragma unroll
                for (uint32_t k = 0; k < WS_CHUNK; k++) acc += tc[tt + k] * (long long)v[k];
            }
            val[s] = acc;
        }
        if (val[0] * val[1] != val[2]) failed = cidx;
        store_row((fl >> WS_CLS_SHIFT_A) & 1u, a.A8, a.A, a.crows, a.batch, cidx, val[0], lane, bad);
        store_row((fl >> WS_CLS_SHIFT_B) & 1u, a.B8, a.B, a.crows, a.batch, cidx, val[1], lane, bad);
        store_row((fl >> WS_CLS_SHIFT_C) & 1u, a.C8, a.C, a.crows, a.batch, cidx, val[2], lane, bad);
    }
    if (failed != 0xFFFFFFFFu) atomicMin(a.status + p, failed);       // status: 0xFFFFFFFF = satisfied, else a failing constraint
    if (bad) atomicOr(a.flag, 1u);
}

// The byte planes as 32-byte Montgomery elements (debug dumps; consumers that have no byte-plane form).  rows: row indices to
// expand (nullptr: all of 0 .. nrows-1); cls (optional): only rows whose class is 0 (the others already hold their element).
__global__ __launch_bounds__(256) void k_wit_expand(const int8_t* __restrict__ plane, size_t rows_per_group, size_t nrows, const uint8_t* __restrict__ cls, fe* __restrict__ mat, size_t batch) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrows * batch) return;
    const size_t row = idx / batch, p = idx % batch;
    if (cls && cls[row]) return;
    const int t = (int)plane[((p >> 6) * rows_per_group + row) * 64 + (p & 63)];
    const fe one = Fr::one();
    fe v = Fr::zero();
    if (t == 1) v = one; else if (t == -1) v = Fr::neg(one);
    store_fe(mat + row * batch + p, v);
}

__global__ __launch_bounds__(256) void k_wit_mark_wide(int8_t* __restrict__ plane, size_t rows_per_group, size_t nrows, const uint8_t* __restrict__ cls, size_t batch) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrows * batch) return;
    const size_t row = idx / batch, p = idx % batch;
    if (cls[row]) plane[((p >> 6) * rows_per_group + row) * 64 + (p & 63)] = WS_PLANE_WIDE;
}

// The input wires (written as 32-byte elements by k_assign_*) into the byte plane: rows 0 .. nrows-1; anything but 0, 1, -1 raises the flag
__global__ __launch_bounds__(256) void k_wit_narrow(const fe* __restrict__ W, size_t batch, size_t nrows, int8_t* __restrict__ W8, size_t rows_per_group, uint32_t* __restrict__ flag) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrows * batch) return;
    const size_t row = idx / batch, p = idx % batch;
    const fe v = load_fe(W + row * batch + p), one = Fr::one();
    int t = 0;
    if (Fr::eq(v, one)) t = 1; else if (Fr::eq(v, Fr::neg(one))) t = -1; else if (!Fr::is_zero(v)) atomicOr(flag, 1u);
    W8[((p >> 6) * rows_per_group + row) * 64 + (p & 63)] = (int8_t)t;
}

// InitAlgorithm: the R1CS coefficients as integers where |c| < 2^62 (ok = 1), sign-normalised
__global__ void k_coeff_small(const fe* __restrict__ coeff, size_t n, long long* __restrict__ out, uint8_t* __restrict__ ok) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe c = Fr::from_mont(load_fe(coeff + i)), m = Fr::from_mont(Fr::neg(load_fe(coeff + i)));
    auto small = [](const fe& x) { return (x.l[2] | x.l[3] | x.l[4] | x.l[5] | x.l[6] | x.l[7]) == 0 && (x.l[1] >> 30) == 0; };
    if (small(c)) { out[i] = (long long)((unsigned long long)c.l[0] | ((unsigned long long)c.l[1] << 32)); ok[i] = 1; }
    else if (small(m)) { out[i] = -(long long)((unsigned long long)m.l[0] | ((unsigned long long)m.l[1] << 32)); ok[i] = 1; }
    else { out[i] = 0; ok[i] = 0; }
}

}  // namespace

void launch_wit_chain(const WitChainArgs& a, size_t groups, size_t lds_bytes, hipStream_t s) {
    if (!groups || !a.nlevels) return;
    hipLaunchKernelGGL((k_wit_chain<16>), dim3((unsigned)groups), dim3(64 * 16), lds_bytes, s, a.tiny, a.parts, a.bits, a.twire, a.tcoef, a.levels, a.nlevels, a.W8, a.rows_per_group, a.flag);
}
void launch_wit_rows(const WitRowsArgs& a, size_t groups, hipStream_t s) {
    const unsigned ny = a.n_tiny_chunks + a.n_rgen;
    if (!groups || !ny) return;
    hipLaunchKernelGGL(k_wit_rows, dim3((unsigned)groups, ny), dim3(64), 0, s, a);
}
void launch_wit_expand(const int8_t* plane, size_t rows_per_group, size_t nrows, const uint8_t* cls, fe* mat, size_t batch, hipStream_t s) {
    const size_t n = nrows * batch;
    if (!n) return;
    hipLaunchKernelGGL(k_wit_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, plane, rows_per_group, nrows, cls, mat, batch);
}
void launch_wit_mark_wide(int8_t* plane, size_t rows_per_group, size_t nrows, const uint8_t* cls, size_t batch, hipStream_t s) {
    const size_t n = nrows * batch;
    if (!n) return;
    hipLaunchKernelGGL(k_wit_mark_wide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, plane, rows_per_group, nrows, cls, batch);
}
void launch_wit_narrow(const fe* W, size_t batch, size_t nrows, int8_t* W8, size_t rows_per_group, uint32_t* flag, hipStream_t s) {
    const size_t n = nrows * batch;
    if (!n) return;
    hipLaunchKernelGGL(k_wit_narrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, batch, nrows, W8, rows_per_group, flag);
}
void launch_coeff_small(const fe* coeff, size_t n, long long* out, uint8_t* ok, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_coeff_small, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, coeff, n, out, ok);
}

}  // namespace gsc
