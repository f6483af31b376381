// 4 x 64-bit Montgomery arithmetic over the BN254 base and scalar fields for HOST code: the CPU-side verifier (verifier.cpp,
// like the reference's libraries/verifier) and the scalar side of Groth16 Setup (setup.cpp; the reference calls groth16.Setup at
// keygen.go:345,384,423).  The prover's hot path does no field arithmetic on the host.
#pragma once
#include <cstdint>
#include <cstring>

namespace gsc { namespace hostf {

using u64 = uint64_t;
using u128 = unsigned __int128;

struct U256 { u64 w[4]; };
constexpr bool geq(const U256& a, const U256& b) { for (int i = 3; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; } return true; }
constexpr u64 sub_into(U256& a, const U256& b) { u64 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; a.w[i] = (u64)d; br = (u64)(d >> 64) & 1; } return br; }
constexpr u64 add_into(U256& a, const U256& b) { u64 c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)a.w[i] + b.w[i] + c; a.w[i] = (u64)s; c = (u64)(s >> 64); } return c; }
// 2^bits mod m by repeated doubling (compile time: the Montgomery constants below are constexpr, so no thread ever sees them half-built)
constexpr U256 pow2_mod(int bits, const U256& m) {
    U256 t{{1, 0, 0, 0}};
    for (int i = 0; i < bits; i++) {
        const u64 top = t.w[3] >> 63;
        for (int k = 3; k > 0; k--) t.w[k] = (t.w[k] << 1) | (t.w[k - 1] >> 63);
        t.w[0] <<= 1;
        if (top || geq(t, m)) sub_into(t, m);
    }
    return t;
}
constexpr u64 neg_inv64(u64 m0) { u64 inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - m0 * inv; return 0 - inv; }

// prime field with Montgomery representation; Tag selects the modulus
template <int Tag>
struct Fe {
    U256 v;   // Montgomery form
    static constexpr U256 MOD = Tag == 0 ? U256{{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}}     // p
                                         : U256{{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};    // r
    static constexpr U256 R1 = pow2_mod(256, MOD), R2 = pow2_mod(512, MOD);      // 2^256, 2^512 mod MOD
    static constexpr u64 NINV = neg_inv64(MOD.w[0]);                                // -MOD^-1 mod 2^64
    static void init() {}      // (kept for callers: the constants are compile-time values — nothing to initialise, nothing to race on)
    static Fe zero() { return Fe{{{0, 0, 0, 0}}}; }
    static Fe one() { return Fe{R1}; }
    static Fe from_u64(u64 x) { Fe a{{{x, 0, 0, 0}}}; return a * Fe{R2}; }
    bool is_zero() const { return (v.w[0] | v.w[1] | v.w[2] | v.w[3]) == 0; }
    bool operator==(const Fe& o) const { return !memcmp(v.w, o.v.w, 32); }
    Fe operator+(const Fe& o) const { Fe r = *this; add_into(r.v, o.v); if (geq(r.v, MOD)) sub_into(r.v, MOD); return r; }
    Fe operator-(const Fe& o) const { Fe r = *this; if (sub_into(r.v, o.v)) add_into(r.v, MOD); return r; }
    Fe neg() const { return is_zero() ? *this : Fe{MOD} - *this; }
    Fe operator*(const Fe& o) const {
        u64 t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)v.w[j] * o.v.w[i] + t[j]; t[j] = (u64)c; c >>= 64; }
            c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
            const u64 m = t[0] * NINV;
            c = (u128)m * MOD.w[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * MOD.w[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
            c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
        }
        Fe r{{{t[0], t[1], t[2], t[3]}}};
        if (t[4] || geq(r.v, MOD)) sub_into(r.v, MOD);
        return r;
    }
    Fe sq() const { return *this * *this; }
    Fe pow(const u64* e, int limbs) const {
        Fe acc = one();
        for (int i = limbs * 64 - 1; i >= 0; i--) { acc = acc.sq(); if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this; }
        return acc;
    }
    Fe inv() const { U256 e = MOD; e.w[0] -= 2; return pow(e.w, 4); }
    U256 canon() const { Fe o{{{1, 0, 0, 0}}}; return (*this * o).v; }
    static bool from_be(const uint8_t* b, Fe& out) {
        U256 c; for (int i = 0; i < 4; i++) { u64 x = 0; for (int k = 0; k < 8; k++) x = (x << 8) | b[(3 - i) * 8 + k]; c.w[i] = x; }
        if (geq(c, MOD)) return false;
        out = Fe{c} * Fe{R2}; return true;
    }
    void to_be(uint8_t* b) const { U256 c = canon(); for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[(3 - i) * 8 + k] = (uint8_t)(c.w[i] >> (56 - 8 * k)); }
    bool lex_large() const {     // canonical value > (MOD-1)/2
        U256 c = canon(), h = MOD; h.w[0] -= 1;
        for (int i = 0; i < 4; i++) h.w[i] = (h.w[i] >> 1) | (i < 3 ? h.w[i + 1] << 63 : 0);
        for (int i = 3; i >= 0; i--) if (c.w[i] != h.w[i]) return c.w[i] > h.w[i];
        return false;
    }
};
using Fp = Fe<0>;
using Fr = Fe<1>;

}}  // namespace gsc::hostf
