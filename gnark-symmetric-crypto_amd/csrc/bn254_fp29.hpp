// BN254 base-field arithmetic in radix 2^29 ("unsaturated limbs") for gfx950, and the G1/G2 group law on top of it.
//
// Why a second representation: on gfx950 a 32x32->64 multiply-add (v_mad_u64_u32 / v_mad_i64_i32) costs the same as ONE
// carry-propagating add (tools/ubench_intmul.hip).  With saturated 32-bit limbs every partial product needs a carry
// instruction (353 VALU instructions per Montgomery product even with hand-written mad+addc pairs); with nine 29-bit
// limbs a whole column of partial products accumulates in a 64-bit register without any carry, so the product is
// 164 multiply-adds + ~60 shifts/masks = 227 instructions in plain C, and additions/subtractions are nine independent
// 32-bit adds with the carries deferred ("lazy").  The MSM kernels are VALU-issue-bound, so instruction
// count is what matters.
//
// Representation: value = sum l[i] * 2^(29 i), i = 0..8, limbs are SIGNED 32-bit integers.
//   tight  (T1)  : limbs 0..7 in [0, 2^29), limb 8 small signed   — products, loads, norm() results
//   signed (T1s) : |limb| < 2^29                                   — difference of two tight values
//   loose  (T2)  : |limb| < 2^30                                   — sum of two tight values, tight +- signed, 2*tight
// mul(a, b) is exact as long as 9*max|a_i|*max|b_j| + 2^61.2 < 2^63:  T1/T1s x T1/T1s and T2 x T1/T1s are fine, T2 x T2 is
// NOT (normalise one side first).  Montgomery radix is R' = 2^261 (not gnark's 2^256: this domain is private to the device
// code; everything crossing an interface is converted).  Because R' = 128 p, products contract: for |a*b| < 2^514 the
// result lies in (-p, 2p), so no conditional subtraction is ever needed between operations; freeze() produces the
// canonical representative only where a comparison or an output needs it.
//
// Replaces gnark-crypto v0.14.0 ecc/bn254/fp + the G1/G2 formulas behind (*G1Jac).MultiExp / (*G2Jac).MultiExp
// (reference go.mod:9; call sites libraries/prover/impl/provers.go:148,216).
#pragma once
#include "bn254_dev.hpp"

namespace bn254 {

struct fe9 { int32_t l[9]; };

struct Fp29Q {     // base field p
    static constexpr uint32_t NINV = 75916169u;      // -p^-1 mod 2^29
    DEVFN static constexpr int32_t ONE(int i) { constexpr int32_t v[9] = {360500257, 337389400, 408039635, 21759001, 178483129, 490881230, 299191303, 86689704, 903222}; return v[i]; }   // 2^261 mod p
    DEVFN static constexpr int32_t R2(int i) { constexpr int32_t v[9] = {94088208, 219480995, 25171640, 279645352, 40052281, 46143135, 379321683, 294034764, 2757030}; return v[i]; }       // 2^522 mod p
    // limb i of 2^s * p, s = 0..4 (each a tight representation)
    DEVFN static constexpr int32_t PK(int s, int i) {
        constexpr int32_t v[5][9] = {
            {410844487, 17064118, 477274959, 47522512, 361093496, 47923392, 10936641, 240920116, 3171406},
            {284818062, 34128237, 417679006, 95045025, 185316080, 95846785, 21873282, 481840232, 6342812},
            {32765212, 68256475, 298487100, 190090051, 370632160, 191693570, 43746564, 426809552, 12685625},
            {65530424, 136512950, 60103288, 380180103, 204393408, 383387141, 87493128, 316748192, 25371251},
            {131060848, 273025900, 120206576, 223489294, 408786817, 229903370, 174986257, 96625472, 50742503}};
        return v[s][i];
    }
    DEVFN static constexpr int32_t HALF(int i) { constexpr int32_t v[9] = {205422243, 276967515, 238637479, 23761256, 180546748, 292397152, 5468320, 120460058, 1585703}; return v[i]; }   // (p-1)/2
    DEVFN static constexpr uint32_t INVE(int i) { constexpr uint32_t v[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}; return v[i]; }   // p-2
};
struct Fr29Q {     // scalar field r
    static constexpr uint32_t NINV = 268435455u;     // -r^-1 mod 2^29
    DEVFN static constexpr int32_t ONE(int i) { constexpr int32_t v[9] = {268435287, 514263732, 86771339, 391139145, 178784091, 490881230, 299191303, 86689704, 903222}; return v[i]; }
    DEVFN static constexpr int32_t R2(int i) { constexpr int32_t v[9] = {95853524, 102173274, 34397646, 498479371, 240439551, 486036963, 471195907, 131109217, 656714}; return v[i]; }
    DEVFN static constexpr int32_t PK(int s, int i) {
        constexpr int32_t v[5][9] = {
            {268435457, 521120927, 240919632, 131109107, 361091715, 47923392, 10936641, 240920116, 3171406},
            {2, 505370943, 481839265, 262218214, 185312518, 95846785, 21873282, 481840232, 6342812},
            {4, 473870974, 426807619, 524436429, 370625036, 191693570, 43746564, 426809552, 12685625},
            {8, 410871036, 316744327, 512001947, 204379161, 383387141, 87493128, 316748192, 25371251},
            {16, 284871160, 96617743, 487132983, 408758323, 229903370, 174986257, 96625472, 50742503}};
        return v[s][i];
    }
    DEVFN static constexpr int32_t HALF(int i) { constexpr int32_t v[9] = {402653184, 260560463, 388895272, 333990009, 180545857, 292397152, 5468320, 120460058, 1585703}; return v[i]; }
    DEVFN static constexpr uint32_t INVE(int i) { constexpr uint32_t v[8] = {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}; return v[i]; }   // r-2
    // 2^266 mod r = R'^2 / 2^256: Montgomery-multiplying a value of the 2^256 domain by this constant lands in the 2^261 domain
    DEVFN static constexpr int32_t FROM_R256(int i) { constexpr int32_t v[9] = {268430039, 492061940, 71535269, 62181526, 323781850, 244503300, 348886451, 68918589, 360451}; return v[i]; }
};

template <class Q>
struct Field29 {
    using E = fe9;
    static constexpr int32_t MASK = (1 << 29) - 1;
    static constexpr uint32_t NINV = Q::NINV;
    DEVFN static constexpr int32_t P(int i) { return Q::PK(0, i); }
    DEVFN static constexpr int32_t ONE(int i) { return Q::ONE(i); }
    DEVFN static constexpr int32_t R2(int i) { return Q::R2(i); }
    DEVFN static constexpr int32_t PK(int s, int i) { return Q::PK(s, i); }
    DEVFN static constexpr int32_t HALF(int i) { return Q::HALF(i); }

    DEVFN static E zero() { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = 0;
        return r; }
    DEVFN static E one() { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = ONE(i);
        return r; }
    // ---- lazy limb-wise operations (no carries) ----
    DEVFN static E add(const E& a, const E& b) { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
        return r; }
    DEVFN static E sub(const E& a, const E& b) { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i];
        return r; }
    DEVFN static E neg(const E& a) { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = -a.l[i];
        return r; }
    DEVFN static E dbl(const E& a) { return add(a, a); }
    // carry propagation: any |limb| < 2^31 -> tight
    DEVFN static E norm(const E& a) {
        E r; int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const int32_t t = a.l[i] + c; c = t >> 29; r.l[i] = t & MASK; }
        r.l[8] = a.l[8] + c;
        return r;
    }
    // Montgomery product a*b/2^261 mod p, result tight and in (-p, 2p).  Column-wise: nothing overflows, nothing carries.
    DEVFN static E mul(const E& a, const E& b) {
        int32_t m[9]; E r; int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P(k - i);
            m[k] = (int32_t)(((uint32_t)acc * NINV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * P(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // a^2 / 2^261: cross products taken once against the doubled operand (45 multiply-adds instead of 81).  a signed-tight.
    DEVFN static E sqr(const E& a) {
        int32_t m[9], a2[9]; E r; int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) a2[i] = 2 * a.l[i];
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) acc += (int64_t)a.l[i] * a2[k - i];
            if (k % 2 == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P(k - i);
            m[k] = (int32_t)(((uint32_t)acc * NINV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * P(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; i++) acc += (int64_t)a.l[i] * a2[k - i];
            if (k % 2 == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // (a*b - c*d) / 2^261 with ONE reduction: both products accumulate in the same columns (18 terms of < 2^58 plus the
    // reduction terms stay below 2^63).  All four operands signed-tight.  Result tight, in (-1.6p, 2.6p).
    DEVFN static E fmms(const E& a, const E& b, const E& c, const E& d) {
        int32_t m[9]; E r; int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; acc -= (int64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P(k - i);
            m[k] = (int32_t)(((uint32_t)acc * NINV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * P(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) { acc += (int64_t)a.l[i] * b.l[k - i]; acc -= (int64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // canonical representative in [0, p), tight.  Input: value in (-8p, 24p), |limb| < 2^30.
    DEVFN static E freeze(const E& a) {
        E x;
#pragma unroll
        for (int i = 0; i < 9; i++) x.l[i] = a.l[i] + PK(3, i);
        x = norm(x);                                   // (0, 32p)
#pragma unroll
        for (int s = 4; s >= 0; s--) {                  // subtract 16p, 8p, 4p, 2p, p when possible
            E y;
#pragma unroll
            for (int i = 0; i < 9; i++) y.l[i] = x.l[i] - PK(s, i);
            y = norm(y);
            const bool ge = y.l[8] >= 0;
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] = ge ? y.l[i] : x.l[i];
        }
        return x;
    }
    // the same for a value known to lie in (-2p, 6p) (e.g. a product of operands below 15p: (-1.4p, 2.4p)): three conditional subtractions instead of five
    DEVFN static E freeze_near(const E& a) {
        E x;
#pragma unroll
        for (int i = 0; i < 9; i++) x.l[i] = a.l[i] + PK(1, i);
        x = norm(x);                                   // (0, 8p)
#pragma unroll
        for (int s = 2; s >= 0; s--) {                  // subtract 4p, 2p, p when possible
            E y;
#pragma unroll
            for (int i = 0; i < 9; i++) y.l[i] = x.l[i] - PK(s, i);
            y = norm(y);
            const bool ge = y.l[8] >= 0;
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] = ge ? y.l[i] : x.l[i];
        }
        return x;
    }
    DEVFN static bool is_zero_frozen(const E& f) { int32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) o |= f.l[i];
        return o == 0; }
    DEVFN static bool is_zero(const E& a) { return is_zero_frozen(freeze(a)); }       // a in (-8p, 24p)
    DEVFN static bool eq(const E& a, const E& b) { return is_zero(sub(a, b)); }        // a - b in (-8p, 24p)
    // ---- conversions ----
    // 8 x 32-bit little-endian words (non-negative integer < 2^256) <-> limbs
    DEVFN static E unpack(const fe& w) {
        E r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
            uint64_t v = w.l[wi];
            if (wi + 1 < 8) v |= (uint64_t)w.l[wi + 1] << 32;
            r.l[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)MASK);
        }
        return r;     // limb 8 holds bits 232..255 (24 bits)
    }
    // requires tight, non-negative, < 2^256 (e.g. a freeze() result)
    DEVFN static fe pack(const E& a) {
        fe w;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int bit = 32 * j, li = bit / 29, sh = bit - 29 * li;     // word j starts inside limb li at offset sh
            uint64_t v = (uint64_t)(uint32_t)a.l[li] >> sh;
            int have = 29 - sh;
            if (li + 1 < 9) { v |= (uint64_t)(uint32_t)a.l[li + 1] << have; have += 29; }
            if (have < 32 && li + 2 < 9) v |= (uint64_t)(uint32_t)a.l[li + 2] << have;
            w.l[j] = (uint32_t)v;
        }
        return w;
    }
    DEVFN static E r2() { E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = R2(i);
        return r; }
    DEVFN static E to_mont(const E& canon) { return mul(canon, r2()); }                    // canonical integer limbs -> Montgomery (tight)
    DEVFN static E from_mont(const E& a) { E o = zero(); o.l[0] = 1; return freeze(mul(a, o)); }   // -> canonical integer
    DEVFN static E from_u32(uint32_t v) { E o = zero(); o.l[0] = (int32_t)(v & (uint32_t)MASK); o.l[1] = (int32_t)(v >> 29); return to_mont(o); }
    DEVFN static E load(const fe* p) { return unpack(load_fe(p)); }                        // memory holds frozen Montgomery values
    DEVFN static void store(fe* p, const E& a) { store_fe(p, pack(freeze(a))); }           // a in (-8p, 24p)
    // a^e (e: 8 little-endian 32-bit words, wave-uniform); a tight
    DEVNOINL static E pow(const E& a, const uint32_t* e) {
        E acc = one(); bool started = false;
        for (int i = 255; i >= 0; i--) {
            if (started) acc = sqr(acc);
            if ((e[i >> 5] >> (i & 31)) & 1) { acc = started ? mul(acc, a) : a; started = true; }
        }
        return acc;
    }
    DEVFN static E inv(const E& a) {      // Fermat; 0 -> 0
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = Q::INVE(i);
        return pow(a, e);
    }
    // canonical value of a Montgomery element > (p-1)/2 ?
    DEVFN static bool lex_large(const E& a) {
        const E c = from_mont(a);
        for (int i = 8; i >= 0; i--) { if (c.l[i] > HALF(i)) return true; if (c.l[i] < HALF(i)) return false; }
        return false;
    }
};

using Fp29 = Field29<Fp29Q>;
using Fr29 = Field29<Fr29Q>;

// ---- Fp2 = Fp[u]/(u^2+1).  mul/sqr/inv take signed-tight components (|limb| < 2^29) and return tight ones. ----
struct fe9x2 { fe9 a0, a1; };
struct Fp2x {
    using E = fe9x2;
    DEVFN static E zero() { return E{Fp29::zero(), Fp29::zero()}; }
    DEVFN static E one() { return E{Fp29::one(), Fp29::zero()}; }
    DEVFN static E add(const E& a, const E& b) { return E{Fp29::add(a.a0, b.a0), Fp29::add(a.a1, b.a1)}; }     // loose
    DEVFN static E sub(const E& a, const E& b) { return E{Fp29::sub(a.a0, b.a0), Fp29::sub(a.a1, b.a1)}; }
    DEVFN static E neg(const E& a) { return E{Fp29::neg(a.a0), Fp29::neg(a.a1)}; }
    DEVFN static E dbl(const E& a) { return add(a, a); }
    DEVFN static E norm(const E& a) { return E{Fp29::norm(a.a0), Fp29::norm(a.a1)}; }
    DEVFN static E mul(const E& a, const E& b) {       // Karatsuba; the two sums are normalised so that no T2 x T2 product occurs
        const fe9 t0 = Fp29::mul(a.a0, b.a0), t1 = Fp29::mul(a.a1, b.a1);
        const fe9 t2 = Fp29::mul(Fp29::norm(Fp29::add(a.a0, a.a1)), Fp29::norm(Fp29::add(b.a0, b.a1)));
        return E{Fp29::norm(Fp29::sub(t0, t1)), Fp29::norm(Fp29::sub(Fp29::sub(t2, t0), t1))};
    }
    DEVFN static E sqr(const E& a) {
        const fe9 s = Fp29::norm(Fp29::add(a.a0, a.a1)), d = Fp29::norm(Fp29::sub(a.a0, a.a1));
        return E{Fp29::mul(s, d), Fp29::norm(Fp29::dbl(Fp29::mul(a.a0, a.a1)))};
    }
    DEVFN static E fmms(const E& a, const E& b, const E& c, const E& d) { return norm(sub(mul(a, b), mul(c, d))); }
    DEVFN static E freeze(const E& a) { return E{Fp29::freeze(a.a0), Fp29::freeze(a.a1)}; }
    DEVFN static bool is_zero(const E& a) { return Fp29::is_zero(a.a0) && Fp29::is_zero(a.a1); }
    DEVFN static bool eq(const E& a, const E& b) { return is_zero(sub(a, b)); }
    DEVFN static E inv(const E& a) {
        const fe9 n = Fp29::norm(Fp29::add(Fp29::sqr(a.a0), Fp29::sqr(a.a1)));
        const fe9 ni = Fp29::inv(n);
        return E{Fp29::mul(a.a0, ni), Fp29::neg(Fp29::mul(a.a1, ni))};
    }
    DEVFN static E load(const fe* p) { return E{Fp29::load(p), Fp29::load(p + 1)}; }
    DEVFN static void store(fe* p, const E& a) { Fp29::store(p, a.a0); Fp29::store(p + 1, a.a1); }
    static constexpr int WORDS = 2;     // field elements per coordinate in memory
};
// helpers so that the curve template can treat both fields alike
struct Fp29f : Fp29 {
    DEVFN static E load(const fe* p) { return Fp29::load(p); }
    static constexpr int WORDS = 1;
};

// ---- points: affine (finite) and XYZZ with an explicit infinity flag (a lazily reduced ZZ cannot be tested for zero cheaply) ----
template <class F> struct Aff9 { typename F::E x, y; };
template <class F> struct Xyzz9 { typename F::E x, y, zz, zzz; bool inf; };

// All coordinates of an Xyzz9 are kept TIGHT (normalised) between operations.
template <class F>
struct Curve9 {
    using E = typename F::E;
    using A = Aff9<F>;
    using X = Xyzz9<F>;
    DEVFN static X infinity() { return X{F::zero(), F::zero(), F::zero(), F::zero(), true}; }
    DEVFN static X from_aff(const A& a) { return X{a.x, a.y, F::one(), F::one(), false}; }
    DEVFN static A neg(const A& a) { return A{a.x, F::norm(F::neg(a.y))}; }
    // dbl-2008-s-1
    DEVFN static X dbl(const X& p) {
        if (p.inf) return p;
        const E U = F::norm(F::dbl(p.y)), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.x, V);
        const E XX = F::sqr(p.x), M = F::norm(F::add(F::dbl(XX), XX));
        X r; r.inf = false;
        r.x = F::norm(F::sub(F::sqr(M), F::dbl(S)));
        r.y = F::norm(F::sub(F::mul(M, F::norm(F::sub(S, r.x))), F::mul(W, p.y)));
        r.zz = F::mul(V, p.zz);
        r.zzz = F::mul(W, p.zzz);
        return r;
    }
    // madd-2008-s.  EXACT: detect P == +-Q (needs two canonical comparisons per call).  Without EXACT the caller must check
    // afterwards that ZZ is not 0 mod p: any degenerate step zeroes ZZ for good (ZZ3 = ZZ1 * P^2), so one test at the end of a
    // long accumulation is enough, and the rare offender is recomputed with EXACT.
    template <bool EXACT>
    DEVFN static X madd(const X& p, const A& q) {
        if (p.inf) return from_aff(q);
        const E U2 = F::mul(q.x, p.zz), S2 = F::mul(q.y, p.zzz);
        const E Pd = F::sub(U2, p.x), Rd = F::sub(S2, p.y);            // signed-tight
        if (EXACT) {
            if (F::is_zero(Pd)) {
                if (F::is_zero(Rd)) return dbl(from_aff(q));
                return infinity();
            }
        }
        const E PP = F::sqr(Pd), PPP = F::mul(Pd, PP), Q = F::mul(p.x, PP);
        X r; r.inf = false;
        r.x = F::norm(F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q)));
        r.y = F::fmms(Rd, F::sub(Q, r.x), p.y, PPP);
        r.zz = F::mul(p.zz, PP);
        r.zzz = F::mul(p.zzz, PPP);
        return r;
    }
    // add-2008-s, always exact (used by the reductions and the proof assembly, never in the hot loop)
    DEVFN static X add(const X& p, const X& q) {
        if (p.inf) return q;
        if (q.inf) return p;
        const E U1 = F::mul(p.x, q.zz), U2 = F::mul(q.x, p.zz), S1 = F::mul(p.y, q.zzz), S2 = F::mul(q.y, p.zzz);
        const E Pd = F::sub(U2, U1), Rd = F::sub(S2, S1);
        if (F::is_zero(Pd)) {
            if (F::is_zero(Rd)) return dbl(p);
            return infinity();
        }
        const E PP = F::sqr(Pd), PPP = F::mul(Pd, PP), Q = F::mul(U1, PP);
        X r; r.inf = false;
        r.x = F::norm(F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q)));
        r.y = F::norm(F::sub(F::mul(Rd, F::sub(Q, r.x)), F::mul(S1, PPP)));
        r.zz = F::mul(F::mul(p.zz, q.zz), PP);
        r.zzz = F::mul(F::mul(p.zzz, q.zzz), PPP);
        return r;
    }
    // caller guarantees !inf.  1/ZZ = ZZ^2 / ZZZ^2
    DEVFN static A to_aff(const X& p) {
        const E i = F::inv(p.zzz), i2 = F::sqr(i), izz = F::mul(F::sqr(p.zz), i2);
        return A{F::mul(p.x, izz), F::mul(p.y, i)};
    }
    // memory images: affine = 2 coordinates, XYZZ = 4 coordinates (ZZ == 0 <=> infinity), all frozen Montgomery values
    DEVFN static A load_aff(const fe* p) { return A{F::load(p), F::load(p + F::WORDS)}; }
    DEVFN static void store_aff(fe* p, const A& a) { F::store(p, a.x); F::store(p + F::WORDS, a.y); }
    DEVFN static X load_xyzz(const fe* p) {
        X r; r.x = F::load(p); r.y = F::load(p + F::WORDS); r.zz = F::load(p + 2 * F::WORDS); r.zzz = F::load(p + 3 * F::WORDS);
        r.inf = F::is_zero(r.zz);
        return r;
    }
    DEVFN static void store_xyzz(fe* p, const X& v) {
        if (v.inf) { const typename F::E z = F::zero(); F::store(p, z); F::store(p + F::WORDS, z); F::store(p + 2 * F::WORDS, z); F::store(p + 3 * F::WORDS, z); return; }
        F::store(p, v.x); F::store(p + F::WORDS, v.y); F::store(p + 2 * F::WORDS, v.zz); F::store(p + 3 * F::WORDS, v.zzz);
    }
};
using G1x = Curve9<Fp29f>;
using G2x = Curve9<Fp2x>;

}  // namespace bn254
