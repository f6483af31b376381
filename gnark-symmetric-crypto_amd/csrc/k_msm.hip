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

// the butterfly for calls with a handful of statements: only the columns of the first `npr` proofs of every row of `stride` columns
// (a row = a window; one row for a flat set) — the other columns of the 64-wide batch hold nothing anybody reads
template <class F>
__global__ __launch_bounds__(64) void k_msm_reduce_few(const fe* partial, size_t nslices, size_t cols, fe* out, uint32_t npr, uint32_t stride) {
    using C = Curve9<F>;
    const size_t p = (size_t)(blockIdx.x / npr) * stride + blockIdx.x % npr, grp = blockIdx.y;
    const size_t slice = grp * 64 + threadIdx.x;
    Xyzz9<F> v = slice < nslices ? C::load_xyzz(partial + (slice * cols + p) * (4 * F::WORDS)) : C::infinity();
    for (int m = 32; m >= 1; m >>= 1) {
        Xyzz9<F> o;
        o.x = shfl_xor_e(v.x, m); o.y = shfl_xor_e(v.y, m); o.zz = shfl_xor_e(v.zz, m); o.zzz = shfl_xor_e(v.zzz, m);
        o.inf = __shfl_xor((int)v.inf, m) != 0;
        v = C::add(v, o);
    }
    if (threadIdx.x == 0) C::store_xyzz(out + (grp * cols + p) * (4 * F::WORDS), v);
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
// The same for calls with a handful of statements: one WAVE per (statement, role), and the scalar split in two halves by the curve's
// endomorphism (glv.hpp: k = k1 + k2 lambda, k A = k1 A + k2 phi(A), phi(x, y) = (beta x, y)).  Each half is cut into 26 five-bit
// chunks; lanes 0..25 own the chunks of k1, lanes 32..57 those of k2: all lanes walk the ONE doubling chain 2^i A together (125
// doublings, nothing else on it), a lane keeps 2^(5q) A when the chain passes it (the k2 lanes apply phi: one product), multiplies it
// by its chunk (5 doublings, <= 5 additions) and a butterfly adds the pieces up.  125 doublings + 16 additions on the critical
// path instead of 254 + ~127 (and no field inversion: the chain starts from the XYZZ sum, a third wave turns Ar into the affine
// form the proof carries): the longest serial chain of a single Prove, 2.3 -> 0.68 ms.
__global__ __launch_bounds__(64) void k_fin_scalarmul_few(const G1Xyzz* sumA, const G1Xyzz* sumB1, const GlvSplit* glv, size_t batch,
                                                           G1Xyzz* tmp, uint8_t* out, uint8_t* flags) {
    using F = Fp29f;
    const size_t p = blockIdx.x;
    const int role = blockIdx.y;
    const uint32_t lane = threadIdx.x;
    const fe* src = reinterpret_cast<const fe*>(role == 1 ? sumB1 : sumA) + 4 * p;
    const Xyzz9<F> P = G1x::load_xyzz(src);
    if (role == 2) {                                         // the proof's Ar in affine form: a field inversion (0.28 ms) that the chains need not wait for
        if (lane == 0) {
            if (!P.inf) { const Aff9<F> A = G1x::to_aff(P); store_canon(out + 256 * p, A.x); store_canon(out + 256 * p + 32, A.y); }
            else atomicOr(reinterpret_cast<unsigned int*>(flags) + (p >> 2), 1u << (8 * (p & 3)));
        }
        return;
    }
    const GlvSplit* g = glv + 2 * p + role;                  // role 0: s, role 1: r
    Xyzz9<F> acc = G1x::infinity();
    if (!P.inf) {                                            // (wave-uniform)
        const bool second = lane >= 32;
        const uint32_t q = lane & 31;
        Xyzz9<F> R = P, mine = R;
#pragma unroll 1
        for (uint32_t i = 0; i < 125; i++) {
            R = G1x::dbl(R);
            const bool take = 5 * q == i + 1;
            mine.x = take ? R.x : mine.x; mine.y = take ? R.y : mine.y; mine.zz = take ? R.zz : mine.zz; mine.zzz = take ? R.zzz : mine.zzz;
        }
        {
            fe bw = {};                                      // beta, canonical
            bw.l[0] = 0x77fffffeu; bw.l[1] = 0x57634731u; bw.l[2] = 0xacdb5c4fu; bw.l[3] = 0xd4f263f1u; bw.l[4] = 0xa0d48bacu; bw.l[5] = 0x59e26bceu;
            const fe9 bx = F::mul(mine.x, Fp29::to_mont(Fp29::unpack(bw)));
            mine.x = second ? bx : mine.x;
        }
        const uint32_t* kw = second ? g->k2 : g->k1;
        if ((g->neg >> (second ? 1 : 0)) & 1u) mine.y = F::norm(F::neg(mine.y));
        uint32_t chunk = 0;
        if (q < 26) {
            const uint32_t o = 5 * q, w = o >> 5, sh = o & 31;
            chunk = kw[w] >> sh;
            if (sh > 27) chunk |= kw[w + 1] << (32 - sh);    // (w + 1 <= 4)
            chunk &= 31u;
        }
#pragma unroll 1
        for (int b = 4; b >= 0; b--) {
            acc = G1x::dbl(acc);
            const Xyzz9<F> t = G1x::add(acc, mine);
            if ((chunk >> b) & 1u) acc = t;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            Xyzz9<F> o;
            o.x = shfl_xor_e(acc.x, m); o.y = shfl_xor_e(acc.y, m); o.zz = shfl_xor_e(acc.zz, m); o.zzz = shfl_xor_e(acc.zzz, m);
            o.inf = __shfl_xor((int)acc.inf, m) != 0;
            acc = G1x::add(acc, o);
        }
    }
    if (lane == 0) G1x::store_xyzz(reinterpret_cast<fe*>(tmp) + ((size_t)role * batch + p) * 4, acc);
}
__global__ __launch_bounds__(64) void k_fin_combine(const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ, const G1Xyzz* sumC, const G1Xyzz* tmp,
                                                     size_t batch, uint8_t* out, uint8_t* flags) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= batch) return;
    const int role = blockIdx.y;
    uint32_t fl = 0;
    if (role == 0) {
        const fe* K = reinterpret_cast<const fe*>(sumK); const fe* Z = reinterpret_cast<const fe*>(sumZ); const fe* T = reinterpret_cast<const fe*>(tmp);
        Xyzz9<Fp29f> v = G1x::add(G1x::add(G1x::load_xyzz(K + 4 * p), G1x::load_xyzz(Z + 4 * p)), G1x::add(G1x::load_xyzz(T + 4 * p), G1x::load_xyzz(T + 4 * (batch + p))));
        if (sumC) v = G1x::add(v, G1x::load_xyzz(reinterpret_cast<const fe*>(sumC) + 4 * p));
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
// ---- commitment challenge on the device (AES-V2): hash_to_field(D) with RFC 9380 expand_message_xmd(SHA-256) ----
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ void sha256_block(uint32_t st[8], const uint32_t blk[16]) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
__device__ __forceinline__ void sha256_iv(uint32_t st[8]) {
    const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (int i = 0; i < 8; i++) st[i] = iv[i];
}
// tail block of the three hashes: [first 32 bytes] | idx | "bsb22-commitment" | 0x10 | 0x80 | zeros | bit length — as big-endian words.
// (b_0's last block instead starts with l_i_b_str = 00 30 00: built separately below.)
__device__ __forceinline__ void xmd_tail(uint32_t blk[16], const uint32_t first[8], uint32_t idx) {
    // "bsb22-commitment" = 62 73 62 32 | 32 2d 63 6f | 6d 6d 69 74 | 6d 65 6e 74
    for (int i = 0; i < 8; i++) blk[i] = first[i];
    blk[8] = (idx << 24) | 0x627362u; blk[9] = 0x32322d63u; blk[10] = 0x6f6d6d69u; blk[11] = 0x746d656eu; blk[12] = 0x74108000u;
    blk[13] = 0; blk[14] = 0; blk[15] = 50 * 8;
}
// cpts: batch x 64 bytes (uncompressed big-endian X | Y of the commitment, as gnark's Marshal() writes it) -> commit[proof] =
// hash_to_field(cpts[proof], DST "bsb22-commitment") in Montgomery form: 48 xmd bytes as a big-endian integer mod r
// (gnark-crypto fr.Hash).  Replaces a device -> host -> device round trip per batch.
__global__ void k_challenge_from_point(const uint8_t* cpts, fe* commit, size_t batch) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= batch) return;
    const uint32_t* m = reinterpret_cast<const uint32_t*>(cpts + 64 * p);
    uint32_t st[8], blk[16], b0[8], b1[8], b2[8];
    // b_0 = H(Z_pad (64 zero bytes) | msg (64) | 00 30 00 | DST | 10): 148 bytes, three blocks
    sha256_iv(st);
    for (int i = 0; i < 16; i++) blk[i] = 0;
    sha256_block(st, blk);
    for (int i = 0; i < 16; i++) blk[i] = __builtin_bswap32(m[i]);
    sha256_block(st, blk);
    blk[0] = 0x00300062u; blk[1] = 0x73623232u; blk[2] = 0x2d636f6du; blk[3] = 0x6d69746du; blk[4] = 0x656e7410u; blk[5] = 0x80000000u;
    for (int i = 6; i < 15; i++) blk[i] = 0;
    blk[15] = 148 * 8;
    sha256_block(st, blk);
    for (int i = 0; i < 8; i++) b0[i] = st[i];
    // b_1 = H(b_0 | 01 | DST | 10), b_2 = H((b_0 xor b_1) | 02 | DST | 10)
    sha256_iv(st); xmd_tail(blk, b0, 1); sha256_block(st, blk);
    for (int i = 0; i < 8; i++) b1[i] = st[i];
    uint32_t x[8]; for (int i = 0; i < 8; i++) x[i] = b0[i] ^ b1[i];
    sha256_iv(st); xmd_tail(blk, x, 2); sha256_block(st, blk);
    for (int i = 0; i < 8; i++) b2[i] = st[i];
    // 48 bytes = b_1 | first 16 bytes of b_2, big-endian integer mod r by Horner over 32-bit words
    fe acc = Fr::zero();
    const fe w32 = Fr::mul(Fr::from_u32(65536), Fr::from_u32(65536));
    for (int i = 0; i < 12; i++) acc = Fr::add(Fr::mul(acc, w32), Fr::from_u32(i < 8 ? b1[i] : b2[i - 8]));
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
size_t launch_msm_reduce_few_g1(const G1Xyzz* partial, size_t nslices, size_t cols, size_t stride, size_t npr, G1Xyzz* out, hipStream_t s) {
    const size_t groups = (nslices + 63) / 64;
    hipLaunchKernelGGL(k_msm_reduce_few<Fp29f>, dim3((unsigned)(cols / stride * npr), (unsigned)groups), dim3(64), 0, s, reinterpret_cast<const fe*>(partial), nslices, cols, reinterpret_cast<fe*>(out), (uint32_t)npr, (uint32_t)stride);
    return groups;
}
size_t launch_msm_reduce_few_g2(const G2Xyzz* partial, size_t nslices, size_t cols, size_t stride, size_t npr, G2Xyzz* out, hipStream_t s) {
    const size_t groups = (nslices + 63) / 64;
    hipLaunchKernelGGL(k_msm_reduce_few<Fp2x>, dim3((unsigned)(cols / stride * npr), (unsigned)groups), dim3(64), 0, s, reinterpret_cast<const fe*>(partial), nslices, cols, reinterpret_cast<fe*>(out), (uint32_t)npr, (uint32_t)stride);
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
void launch_challenge_from_point(const uint8_t* cpts, fe* commit, size_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_challenge_from_point, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, cpts, commit, batch);
}
void launch_fin_scalarmul(const G1Xyzz* sumA, const G1Xyzz* sumB1, const uint8_t* rs, size_t batch, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s) {
    hipLaunchKernelGGL(k_fin_scalarmul, dim3((unsigned)((batch + 63) / 64), 2), dim3(64), 0, s, sumA, sumB1, rs, batch, tmp, out, flags);
}
void launch_fin_scalarmul_few(const G1Xyzz* sumA, const G1Xyzz* sumB1, const GlvSplit* glv, size_t batch, size_t nproofs, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s) {
    hipLaunchKernelGGL(k_fin_scalarmul_few, dim3((unsigned)nproofs, 3), dim3(64), 0, s, sumA, sumB1, glv, batch, tmp, out, flags);      // roles: s * Ar, r * Bs1, Ar -> affine
}
void launch_fin_combine(const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ, const G1Xyzz* sumC, const G1Xyzz* tmp, size_t batch, uint8_t* out, uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_fin_combine, dim3((unsigned)((batch + 63) / 64), 2), dim3(64), 0, s, sumB2, sumK, sumZ, sumC, tmp, batch, out, flags);
}

}  // namespace gsc
