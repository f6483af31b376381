// GLV split of a BN254 scalar for the latency path's two scalar multiplications (k_fin_scalarmul_few, k_msm.hip).
//
// G1 of BN254 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) with beta^3 = 1 in Fp, lambda^3 = 1 in Fr, so
// k P = k1 P + k2 phi(P) for k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^128: half the doublings.  The short basis
// {(a1, b1), (a2, b2)} of the lattice {(a, b): a + b lambda = 0 mod r} comes from the extended Euclid run on (r, lambda)
// (Gallant-Lambert-Vanstone 2001; what gnark-crypto's ecc.PrecomputeLattice does); c_i = floor(k g_i / 2^384) with
// g_1 = floor(2^384 b2 / r), g_2 = floor(2^384 (-b1) / r) approximates the rounding of (k, 0) onto the lattice (off by at most one
// per coefficient: one more bit of magnitude).  tests/test_glv.py recomputes the constants from r alone and checks the split
// against big-integer arithmetic through gsc_debug_glv_split.
//   lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd        beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
//   a1 = b2 = 9931322734385697763      b1 = -147946756881789319000765030803803410728      a2 = 147946756881789319010696353538189108491
#pragma once
#include <cstdint>

namespace gsc {

struct GlvSplit { uint32_t k1[5], k2[5]; uint32_t neg; };      // magnitudes (little-endian words, < 2^130); neg bit 0: k1 < 0, bit 1: k2 < 0

namespace glv_detail {
typedef unsigned __int128 u128;
constexpr uint64_t A1 = 0x89d3256894d213e3ull;                                            // = b2
constexpr uint64_t A2[2] = {0x0be4e1541221250bull, 0x6f4d8248eeb859fdull};
constexpr uint64_t B1_ABS[2] = {0x8211bbeb7d4f1128ull, 0x6f4d8248eeb859fcull};
constexpr uint64_t G1[4] = {0x8fa7d32d2fafba64ull, 0x6eb9c714773a6ef2ull, 0xd91d232ec7e0b3d7ull, 0x2ull};
constexpr uint64_t G2[5] = {0x869375169b9bdffaull, 0xa5e38cfb5eaa26d9ull, 0x7a7bd9d4391eb18dull, 0x4ccef014a773d2cfull, 0x2ull};
// out[na + nb] = a[na] * b[nb]  (64-bit limbs, little-endian)
inline void mul(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out) {
    for (int i = 0; i < na + nb; i++) out[i] = 0;
    for (int i = 0; i < na; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < nb; j++) { const u128 t = (u128)a[i] * b[j] + out[i + j] + carry; out[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64); }
        out[i + nb] = carry;
    }
}
// acc (256-bit two's complement) +-= a[na] * b[nb]
inline void mul_acc(uint64_t* acc, const uint64_t* a, int na, const uint64_t* b, int nb, bool add) {
    uint64_t full[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mul(a, na, b, nb, full);
    if (add) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)acc[i] + full[i]; acc[i] = (uint64_t)c; c >>= 64; } }
    else { uint64_t br = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)acc[i] - full[i] - br; acc[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
}
inline bool to_magnitude(uint64_t* v) {      // 256-bit two's complement -> magnitude; returns the sign
    if (!(v[3] >> 63)) return false;
    u128 c = 1; for (int i = 0; i < 4; i++) { c += (u128)(~v[i]); v[i] = (uint64_t)c; c >>= 64; }
    return true;
}
}  // namespace glv_detail

// k: canonical scalar < r, eight little-endian 32-bit words.  false if a magnitude does not fit 130 bits (cannot happen for k < r).
inline bool glv_split(const uint32_t k_words[8], GlvSplit& out) {
    using namespace glv_detail;
    uint64_t k[4];
    for (int i = 0; i < 4; i++) k[i] = (uint64_t)k_words[2 * i] | ((uint64_t)k_words[2 * i + 1] << 32);
    uint64_t p1[8], p2[9];
    mul(k, 4, G1, 4, p1); mul(k, 4, G2, 5, p2);
    const uint64_t c1[2] = {p1[6], p1[7]}, c2[3] = {p2[6], p2[7], p2[8]};     // >> 384
    uint64_t k1[4] = {k[0], k[1], k[2], k[3]}, k2[4] = {0, 0, 0, 0};
    mul_acc(k1, c1, 2, &A1, 1, false); mul_acc(k1, c2, 3, A2, 2, false);        // k1 = k - c1 a1 - c2 a2
    mul_acc(k2, c1, 2, B1_ABS, 2, true); mul_acc(k2, c2, 3, &A1, 1, false);     // k2 = -c1 b1 - c2 b2 = c1 |b1| - c2 a1
    out.neg = (to_magnitude(k1) ? 1u : 0u) | (to_magnitude(k2) ? 2u : 0u);
    if (k1[3] || k2[3] || (k1[2] >> 2) || (k2[2] >> 2)) return false;
    for (int i = 0; i < 5; i++) { out.k1[i] = (uint32_t)(k1[i / 2] >> (32 * (i & 1))); out.k2[i] = (uint32_t)(k2[i / 2] >> (32 * (i & 1))); }
    return true;
}

}  // namespace gsc
