// libverify — C-ABI drop-in for the reference's verifier library (libraries/verifier/libverify.go:14-17 ->
// impl.Verify, libraries/verifier/impl/verify_impl.go:62-82 -> ChachaVerifier / AESVerifier, verifiers.go:50-152).
//
// The reference's verifier is a CPU-side component (three pairings per proof), and so is this one:
// it is NOT part of the GPU hot path and the prover never calls it.  Groth16 verification over BN254 with gnark's
// key / proof encodings and gnark's commitment extension for the AES circuits (App. H).
// Everything numeric is host code (host_field.hpp: 4 x 64-bit Montgomery arithmetic; here: Fp2/Fp12 towers, optimal-ate
// pairing); JSON, base64 and SHA-256 come from the same host sources as libprove.
#include "../../include/libverify.h"
#include "host_ciphers.hpp"
#include "host_field.hpp"
#include "json.hpp"
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

using namespace gsc::hostf;


struct Fp2 {
    Fp a, b;   // a + b u, u^2 = -1
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
    bool operator==(const Fp2& o) const { return a == o.a && b == o.b; }
    Fp2 operator+(const Fp2& o) const { return {a + o.a, b + o.b}; }
    Fp2 operator-(const Fp2& o) const { return {a - o.a, b - o.b}; }
    Fp2 neg() const { return {a.neg(), b.neg()}; }
    Fp2 conj() const { return {a, b.neg()}; }
    Fp2 operator*(const Fp2& o) const { Fp t0 = a * o.a, t1 = b * o.b; return {t0 - t1, (a + b) * (o.a + o.b) - t0 - t1}; }
    Fp2 scale(const Fp& k) const { return {a * k, b * k}; }
    Fp2 sq() const { return {(a + b) * (a - b), (a * b) + (a * b)}; }
    Fp2 inv() const { Fp n = (a.sq() + b.sq()).inv(); return {a * n, (b * n).neg()}; }
    Fp2 pow(const u64* e, int limbs) const { Fp2 acc = one(); for (int i = limbs * 64 - 1; i >= 0; i--) { acc = acc.sq(); if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this; } return acc; }
    bool lex_large() const { return b.is_zero() ? a.lex_large() : b.lex_large(); }
};

// exponents derived from p (computed once)
struct Consts {
    u64 sqrt_e[4], pm1_3[4], pm1_2[4], pm1_6[4]; Fp2 xi, g2, g3, twist_b; Fp three;
    Fp2 frob1[6]; Fp frob2[6];       // w^(i p) = frob1[i] w^i, w^(i p^2) = frob2[i] w^i  (w^6 = xi)
    std::vector<int8_t> hard_naf;    // (p^4 - p^2 + 1) / r in non-adjacent form, least significant digit first
    static std::vector<u64> mul_big(const std::vector<u64>& a, const std::vector<u64>& b) {
        std::vector<u64> r(a.size() + b.size(), 0);
        for (size_t i = 0; i < a.size(); i++) { u128 c = 0; for (size_t j = 0; j < b.size(); j++) { c += (u128)a[i] * b[j] + r[i + j]; r[i + j] = (u64)c; c >>= 64; } r[i + b.size()] += (u64)c; }
        return r;
    }
    static void div_small(const u64* in, int n, u64 d, u64* out) { u128 rem = 0; for (int i = n - 1; i >= 0; i--) { u128 cur = (rem << 64) | in[i]; out[i] = (u64)(cur / d); rem = cur % d; } }
    Consts() {
        Fp::init(); Fr::init();
        U256 p = Fp::MOD, t = p;
        add_into(t, U256{{1, 0, 0, 0}}); div_small(t.w, 4, 4, sqrt_e);          // (p+1)/4
        t = p; t.w[0] -= 1; div_small(t.w, 4, 3, pm1_3); div_small(t.w, 4, 2, pm1_2); div_small(t.w, 4, 6, pm1_6);
        three = Fp::from_u64(3);
        xi = {Fp::from_u64(9), Fp::one()};
        twist_b = Fp2{three, Fp::zero()} * xi.inv();
        g2 = xi.pow(pm1_3, 4); g3 = xi.pow(pm1_2, 4);
        // Frobenius on Fp12 = Fp2[w]/(w^6 - xi): w^p = xi^((p-1)/6) w
        const Fp2 gam = xi.pow(pm1_6, 4);
        frob1[0] = Fp2::one(); for (int i = 1; i < 6; i++) frob1[i] = frob1[i - 1] * gam;
        const Fp nrm = (gam * gam.conj()).a;                                   // w^(p^2) = gam^p gam w = N(gam) w, N(gam) in Fp
        frob2[0] = Fp::one(); for (int i = 1; i < 6; i++) frob2[i] = frob2[i - 1] * nrm;
        // hard part of the final exponentiation: (p^4 - p^2 + 1) / r by schoolbook big-number arithmetic on 64-bit limbs
        const std::vector<u64> pv(p.w, p.w + 4), p2 = mul_big(pv, pv), p4 = mul_big(p2, p2);
        std::vector<u64> acc = p4;
        { u128 br = 0; for (size_t i = 0; i < acc.size(); i++) { const u128 d = (u128)acc[i] - (i < p2.size() ? p2[i] : 0) - br; acc[i] = (u64)d; br = (d >> 64) & 1; } }   // - p^2
        for (size_t i = 0; i < acc.size(); i++) { if (++acc[i]) break; }                                                                                                   // + 1
        // long division by r (256-bit) : shift-subtract
        const U256 r = Fr::MOD; std::vector<u64> q(acc.size(), 0); U256 rem{{0, 0, 0, 0}}; u64 ext = 0;
        for (int bit = (int)acc.size() * 64 - 1; bit >= 0; bit--) {
            ext = rem.w[3] >> 63;
            for (int k = 3; k > 0; k--) rem.w[k] = (rem.w[k] << 1) | (rem.w[k - 1] >> 63);
            rem.w[0] = (rem.w[0] << 1) | ((acc[bit / 64] >> (bit % 64)) & 1);
            if (ext || geq(rem, r)) { sub_into(rem, r); q[bit / 64] |= 1ull << (bit % 64); }
        }
        if (rem.w[0] | rem.w[1] | rem.w[2] | rem.w[3]) throw std::runtime_error("internal: r does not divide p^4 - p^2 + 1");
        // non-adjacent form: a third of the digits are non-zero (an inverse is free in the cyclotomic subgroup)
        q.push_back(0);
        auto is_zero = [&] { for (u64 x : q) if (x) return false; return true; };
        while (!is_zero()) {
            int8_t d = 0;
            if (q[0] & 1) { d = (q[0] & 2) ? -1 : 1; if (d == 1) q[0] -= 1; else { for (size_t i = 0; i < q.size(); i++) { if (++q[i]) break; } } }
            hard_naf.push_back(d);
            for (size_t i = 0; i + 1 < q.size(); i++) q[i] = (q[i] >> 1) | (q[i + 1] << 63);
            q.back() >>= 1;
        }
    }
};
const Consts& K() { static Consts c; return c; }

bool fp_sqrt(const Fp& a, Fp& out) { Fp s = a.pow(K().sqrt_e, 4); if (!(s.sq() == a)) return false; out = s; return true; }
bool fp2_sqrt(const Fp2& a, Fp2& out) {      // norm method
    Fp2 x;
    if (a.b.is_zero()) {
        Fp s; if (fp_sqrt(a.a, s)) x = {s, Fp::zero()}; else { if (!fp_sqrt(a.a.neg(), s)) return false; x = {Fp::zero(), s}; }
    } else {
        Fp n = a.a.sq() + a.b.sq(), s, half = Fp::from_u64(2).inv(), x0;
        if (!fp_sqrt(n, s)) return false;
        if (!fp_sqrt((a.a + s) * half, x0) && !fp_sqrt((a.a - s) * half, x0)) return false;
        x = {x0, a.b * (x0 + x0).inv()};
    }
    if (!(x.sq() == a)) return false;
    out = x; return true;
}

struct G1 { Fp x, y; bool inf; };
struct G2 { Fp2 x, y; bool inf; };
G1 g1_neg(const G1& p) { return {p.x, p.y.neg(), p.inf}; }
G1 g1_add(const G1& p, const G1& q) {      // affine, one inversion (verification adds at most ~1.2k points)
    if (p.inf) return q;
    if (q.inf) return p;
    Fp lam;
    if (p.x == q.x) { if (!(p.y == q.y) || p.y.is_zero()) return {Fp::zero(), Fp::zero(), true}; lam = (p.x.sq() * Fp::from_u64(3)) * (p.y + p.y).inv(); }
    else lam = (q.y - p.y) * (q.x - p.x).inv();
    Fp x3 = lam.sq() - p.x - q.x; return {x3, lam * (p.x - x3) - p.y, false};
}
// sum of many affine points with batched inversions (pairwise tree)
G1 g1_sum(std::vector<G1> pts) {
    while (pts.size() > 1) {
        std::vector<G1> nx; size_t pairs = pts.size() / 2;
        std::vector<Fp> den(pairs), pre(pairs); std::vector<int> kind(pairs, 0);
        for (size_t i = 0; i < pairs; i++) {
            const G1 &a = pts[2 * i], &b = pts[2 * i + 1];
            if (a.inf || b.inf) { kind[i] = 1; den[i] = Fp::one(); }
            else if (a.x == b.x) { if (a.y == b.y && !a.y.is_zero()) { kind[i] = 2; den[i] = a.y + a.y; } else { kind[i] = 3; den[i] = Fp::one(); } }
            else den[i] = b.x - a.x;
        }
        Fp run = Fp::one(); for (size_t i = 0; i < pairs; i++) { pre[i] = run; run = run * den[i]; }
        Fp inv = run.inv();
        std::vector<Fp> dinv(pairs); for (size_t i = pairs; i-- > 0;) { dinv[i] = inv * pre[i]; inv = inv * den[i]; }
        for (size_t i = 0; i < pairs; i++) {
            const G1 &a = pts[2 * i], &b = pts[2 * i + 1];
            if (kind[i] == 1) nx.push_back(a.inf ? b : a);
            else if (kind[i] == 3) nx.push_back({Fp::zero(), Fp::zero(), true});
            else { Fp lam = kind[i] == 2 ? a.x.sq() * Fp::from_u64(3) * dinv[i] : (b.y - a.y) * dinv[i]; Fp x3 = lam.sq() - a.x - b.x; nx.push_back({x3, lam * (a.x - x3) - a.y, false}); }
        }
        if (pts.size() & 1) nx.push_back(pts.back());
        pts.swap(nx);
    }
    return pts.empty() ? G1{Fp::zero(), Fp::zero(), true} : pts[0];
}
G1 g1_mul(const G1& p, const U256& k) { G1 acc{Fp::zero(), Fp::zero(), true}; for (int i = 255; i >= 0; i--) { acc = g1_add(acc, acc); if ((k.w[i / 64] >> (i % 64)) & 1) acc = g1_add(acc, p); } return acc; }

// gnark-crypto encodings: flags in the two top bits of byte 0
bool g1_decode(const uint8_t* b, G1& p) {
    const uint8_t flag = b[0] & 0xC0; uint8_t xb[32]; memcpy(xb, b, 32); xb[0] &= 0x3F;
    if (flag == 0x40) {      // point at infinity: gnark-crypto rejects the encoding unless every other bit is zero
        for (int i = 0; i < 32; i++) if (xb[i]) return false;
        p = {Fp::zero(), Fp::zero(), true}; return true;
    }
    if (flag == 0) return false;
    if (!Fp::from_be(xb, p.x)) return false;
    Fp y; if (!fp_sqrt(p.x.sq() * p.x + K().three, y)) return false;
    if ((flag == 0xC0) != y.lex_large()) y = y.neg();
    p.y = y; p.inf = false; return true;
}
// The twist E'(Fp2) has order r * (2p - r): a point that satisfies the curve equation need not lie in the r-torsion subgroup G2.
// gnark-crypto's decoder rejects such points (G2Affine.SetBytes -> IsInSubGroup) and so does groth16.Verify (proof.isValid()),
// so the drop-in must too.  Plain test [r]Q == O in Jacobian coordinates (381 group operations, no inversion).
bool g2_in_subgroup(const G2& q) {
    if (q.inf) return true;
    Fp2 X = Fp2::zero(), Y = Fp2::zero(), Z = Fp2::zero(); bool inf = true;
    const U256 r = Fr::MOD;
    for (int i = 255; i >= 0; i--) {
        if (!inf) {      // dbl-2009-l (a = 0)
            const Fp2 A = X.sq(), B = Y.sq(), C = B.sq(); Fp2 D = (X + B).sq() - A - C; D = D + D;
            const Fp2 E = A + A + A, F = E.sq(), Z3 = (Y * Z) + (Y * Z); Fp2 C8 = C + C; C8 = C8 + C8; C8 = C8 + C8;
            X = F - D - D; Y = E * (D - X) - C8; Z = Z3;
            if (Z.is_zero()) inf = true;
        }
        if (!((r.w[i / 64] >> (i % 64)) & 1)) continue;
        if (inf) { X = q.x; Y = q.y; Z = Fp2::one(); inf = false; continue; }
        // madd-2007-bl
        const Fp2 Z1Z1 = Z.sq(), U2 = q.x * Z1Z1, S2 = q.y * Z * Z1Z1, H = U2 - X; Fp2 rr = S2 - Y; rr = rr + rr;
        if (H.is_zero()) {
            if (!rr.is_zero()) { inf = true; continue; }
            const Fp2 A = X.sq(), B = Y.sq(), C = B.sq(); Fp2 D = (X + B).sq() - A - C; D = D + D;      // accumulator == Q: double
            const Fp2 E = A + A + A, F = E.sq(), Z3 = (Y * Z) + (Y * Z); Fp2 C8 = C + C; C8 = C8 + C8; C8 = C8 + C8;
            X = F - D - D; Y = E * (D - X) - C8; Z = Z3; continue;
        }
        const Fp2 HH = H.sq(); Fp2 I = HH + HH; I = I + I; const Fp2 J = H * I, V = X * I;
        const Fp2 X3 = rr.sq() - J - V - V; Fp2 YJ = Y * J; YJ = YJ + YJ;
        const Fp2 Y3 = rr * (V - X3) - YJ, Z3 = (Z + H).sq() - Z1Z1 - HH;
        X = X3; Y = Y3; Z = Z3;
    }
    return inf;
}
bool g2_decode(const uint8_t* b, G2& p) {
    const uint8_t flag = b[0] & 0xC0; uint8_t xb[32]; memcpy(xb, b, 32); xb[0] &= 0x3F;
    if (flag == 0x40) {
        for (int i = 0; i < 32; i++) if (xb[i] || b[32 + i]) return false;
        p = {Fp2::zero(), Fp2::zero(), true}; return true;
    }
    if (flag == 0) return false;
    if (!Fp::from_be(xb, p.x.b) || !Fp::from_be(b + 32, p.x.a)) return false;
    Fp2 y; if (!fp2_sqrt(p.x.sq() * p.x + K().twist_b, y)) return false;
    if ((flag == 0xC0) != y.lex_large()) y = y.neg();
    p.y = y; p.inf = false;
    return g2_in_subgroup(p);
}

// Fp12 = Fp2[w]/(w^6 - xi)
struct Fp12 {
    Fp2 c[6];
    static Fp12 one() { Fp12 r; for (auto& x : r.c) x = Fp2::zero(); r.c[0] = Fp2::one(); return r; }
    bool is_one() const { if (!(c[0] == Fp2::one())) return false; for (int i = 1; i < 6; i++) if (!c[i].is_zero()) return false; return true; }
    Fp12 operator*(const Fp12& o) const {
        Fp2 t[11]; for (auto& x : t) x = Fp2::zero();
        for (int i = 0; i < 6; i++) { if (c[i].is_zero()) continue; for (int j = 0; j < 6; j++) { if (o.c[j].is_zero()) continue; t[i + j] = t[i + j] + c[i] * o.c[j]; } }
        Fp12 r; for (int k = 0; k < 6; k++) { r.c[k] = t[k]; if (k + 6 < 11) r.c[k] = r.c[k] + t[k + 6] * K().xi; }
        return r;
    }
    Fp12 sq() const {      // cross products once: 15 products + 6 squarings instead of 36 products
        Fp2 t[11]; for (auto& x : t) x = Fp2::zero();
        for (int i = 0; i < 6; i++) { t[2 * i] = t[2 * i] + c[i].sq(); for (int j = i + 1; j < 6; j++) { const Fp2 m = c[i] * c[j]; t[i + j] = t[i + j] + m + m; } }
        Fp12 r; for (int k = 0; k < 6; k++) { r.c[k] = t[k]; if (k + 6 < 11) r.c[k] = r.c[k] + t[k + 6] * K().xi; }
        return r;
    }
    Fp12 conj6() const { Fp12 r = *this; r.c[1] = r.c[1].neg(); r.c[3] = r.c[3].neg(); r.c[5] = r.c[5].neg(); return r; }      // x^(p^6): w -> -w
    Fp12 frob() const { Fp12 r; for (int i = 0; i < 6; i++) r.c[i] = c[i].conj() * K().frob1[i]; return r; }                      // x^p
    Fp12 frob2() const { Fp12 r; for (int i = 0; i < 6; i++) r.c[i] = c[i].scale(K().frob2[i]); return r; }                       // x^(p^2)
    // 1 / x through Fp6 = Fp2[u]/(u^3 - xi), u = w^2: x = A + w B, 1/x = (A - w B) / (A^2 - u B^2)
    struct F6 {
        Fp2 a, b, c;
        F6 operator*(const F6& o) const { const Fp2& xi = K().xi; return {a * o.a + (b * o.c + c * o.b) * xi, a * o.b + b * o.a + (c * o.c) * xi, a * o.c + b * o.b + c * o.a}; }
        F6 operator-(const F6& o) const { return {a - o.a, b - o.b, c - o.c}; }
        F6 mul_u() const { return {c * K().xi, a, b}; }
        F6 inv() const {
            const Fp2& xi = K().xi;
            const Fp2 t0 = a.sq() - (b * c) * xi, t1 = c.sq() * xi - a * b, t2 = b.sq() - a * c;
            const Fp2 n = (a * t0 + (c * t1 + b * t2) * xi).inv();
            return {t0 * n, t1 * n, t2 * n};
        }
    };
    Fp12 inv() const {
        const F6 A{c[0], c[2], c[4]}, B{c[1], c[3], c[5]};
        const F6 n = (A * A - (B * B).mul_u()).inv(), ra = A * n, rb = B * n;
        Fp12 r; r.c[0] = ra.a; r.c[2] = ra.b; r.c[4] = ra.c; r.c[1] = rb.a.neg(); r.c[3] = rb.b.neg(); r.c[5] = rb.c.neg();
        return r;
    }
};
// line through psi(T), psi(Q) evaluated at P (D-type twist, psi(x,y) = (x w^2, y w^3)): yP - lambda xP w + (lambda xT - yT) w^3
struct Tw { Fp2 x, y; };
Fp12 line(const Fp2& lam, const Tw& T, const G1& P) { Fp12 l; for (auto& x : l.c) x = Fp2::zero(); l.c[0] = {P.y, Fp::zero()}; l.c[1] = lam.scale(P.x).neg(); l.c[3] = lam * T.x - T.y; return l; }
// Product of pairings == 1 ?  One Miller loop for all pairs: the accumulator is squared once per step whatever the number of pairs,
// and the slopes of a step share ONE field inversion (Montgomery's trick over Fp2).  Final exponentiation (p^12 - 1) / r =
// (p^6 - 1)(p^2 + 1) * (p^4 - p^2 + 1) / r: the first two factors by conjugation, one inversion and a Frobenius map, the last by
// square-and-multiply over its non-adjacent form (761 squarings, ~250 products; inverse = conjugate in the cyclotomic subgroup).
bool pairing_product_is_one(const std::vector<std::pair<G1, G2>>& v) {
    struct Pair { G1 P; Tw T, Q; };
    std::vector<Pair> ps;
    for (auto& pq : v) if (!pq.first.inf && !pq.second.inf) ps.push_back({pq.first, {pq.second.x, pq.second.y}, {pq.second.x, pq.second.y}});
    const size_t n = ps.size();
    Fp12 f = Fp12::one();
    std::vector<Fp2> num(n), den(n), pre(n);
    auto slopes = [&]() {      // num[k] / den[k] -> num[k] (den == 0 cannot occur for points of order r; it would give slope 0 and a rejected proof)
        Fp2 run = Fp2::one(); for (size_t k = 0; k < n; k++) { pre[k] = run; run = run * den[k]; }
        Fp2 inv = run.inv();
        for (size_t k = n; k-- > 0;) { num[k] = num[k] * (inv * pre[k]); inv = inv * den[k]; }
    };
    auto apply = [&](size_t k, const Tw& other, bool dbl) {      // line through T (and `other`), then T <- T + other (or 2 T)
        Tw& T = ps[k].T; const Fp2& lam = num[k];
        f = f * line(lam, T, ps[k].P);
        const Fp2 x3 = lam.sq() - T.x - (dbl ? T.x : other.x);
        T = {x3, lam * (T.x - x3) - T.y};
    };
    auto dbl_all = [&]() { for (size_t k = 0; k < n; k++) { const Fp2 x2 = ps[k].T.x.sq(); num[k] = x2 + x2 + x2; den[k] = ps[k].T.y + ps[k].T.y; } slopes(); for (size_t k = 0; k < n; k++) apply(k, ps[k].T, true); };
    auto add_all = [&](const std::vector<Tw>& o) { for (size_t k = 0; k < n; k++) { num[k] = o[k].y - ps[k].T.y; den[k] = o[k].x - ps[k].T.x; } slopes(); for (size_t k = 0; k < n; k++) apply(k, o[k], false); };
    if (n) {
        const u64 loop[2] = {0x9d797039be763ba8ull, 1};     // 6x + 2, x = 4965661367192848881
        std::vector<Tw> q0(n), q1(n), q2(n);
        for (size_t k = 0; k < n; k++) {
            q0[k] = ps[k].Q;
            q1[k] = {q0[k].x.conj() * K().g2, q0[k].y.conj() * K().g3};
            q2[k] = {q1[k].x.conj() * K().g2, (q1[k].y.conj() * K().g3).neg()};
        }
        for (int i = 63; i >= 0; i--) { f = f.sq(); dbl_all(); if ((loop[i / 64] >> (i % 64)) & 1) add_all(q0); }
        add_all(q1); add_all(q2);
    }
    const Fp12 easy1 = f.conj6() * f.inv();
    const Fp12 g = easy1.frob2() * easy1, gi = g.conj6();
    const auto& e = K().hard_naf; Fp12 r = Fp12::one();
    for (size_t i = e.size(); i-- > 0;) { r = r.sq(); if (e[i] > 0) r = r * g; else if (e[i] < 0) r = r * gi; }
    return r.is_one();
}

// ---- verifying key ----
struct VerifyingKey { G1 alpha; G2 beta, gamma, delta; std::vector<G1> K; bool has_commitment = false; G2 ped_g, ped_gsn; };
uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
std::unique_ptr<VerifyingKey> parse_vk(const uint8_t* b, size_t n) {
    auto vk = std::make_unique<VerifyingKey>(); size_t i = 0; G1 skip1;
    auto need = [&](size_t k) { if (i + k > n) throw std::runtime_error("vk: truncated"); };
    auto g1 = [&](G1& p) { need(32); if (!g1_decode(b + i, p)) throw std::runtime_error("vk: bad G1 point"); i += 32; };
    auto g2 = [&](G2& p) { need(64); if (!g2_decode(b + i, p)) throw std::runtime_error("vk: bad G2 point"); i += 64; };
    g1(vk->alpha); g1(skip1); g2(vk->beta); g2(vk->gamma); g1(skip1); g2(vk->delta);
    need(4); uint32_t nk = be32(b + i); i += 4; vk->K.resize(nk); for (auto& p : vk->K) g1(p);
    need(4); uint32_t outer = be32(b + i); i += 4; if (outer > 1) throw std::runtime_error("vk: more than one commitment");
    for (uint32_t o = 0; o < outer; o++) { need(4); uint32_t inner = be32(b + i); i += 4; if (inner) throw std::runtime_error("vk: public committed wires are not supported"); }
    need(4); uint32_t nck = be32(b + i); i += 4; if (nck != outer) throw std::runtime_error("vk: commitment key count");
    if (nck) { vk->has_commitment = true; g2(vk->ped_g); g2(vk->ped_gsn); }
    if (i != n) throw std::runtime_error("vk: trailing bytes");
    return vk;
}

std::mutex g_mu; std::unique_ptr<VerifyingKey> g_vk[3]; bool g_dir_tried = false;
const char* kNames[3] = {"chacha20", "aes-128-ctr", "aes-256-ctr"};
const char* kFiles[3] = {"vk.chacha20", "vk.aes128", "vk.aes256"};
void load_dir_once() {
    if (g_dir_tried) return;
    g_dir_tried = true;
    const char* dir = getenv("GSC_VK_DIR"); if (!dir) return;
    for (int k = 0; k < 3; k++) if (!g_vk[k]) {
        std::ifstream f(std::string(dir) + "/" + kFiles[k], std::ios::binary); if (!f) continue;
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        try { g_vk[k] = parse_vk(buf.data(), buf.size()); } catch (const std::exception& e) { printf("%s\n", e.what()); }
    }
}

Fr hash_to_fr(const uint8_t* msg, size_t n, const char* dst) {      // gnark hash_to_field, one element: 48 xmd bytes mod r
    uint8_t x[48]; gsc::expand_message_xmd_sha256(msg, n, dst, x, 48);
    Fr acc = Fr::zero(), b256 = Fr::from_u64(256);
    for (int i = 0; i < 48; i++) acc = acc * b256 + Fr::from_u64(x[i]);
    return acc;
}

bool bytes_field(const gsc::JsonValue& v, std::vector<uint8_t>& out) {
    if (v.kind == gsc::JsonValue::String) { size_t bad; return gsc::base64_decode(v.text, out, bad); }
    if (v.kind == gsc::JsonValue::Null) { out.clear(); return true; }
    if (v.kind != gsc::JsonValue::Array) return false;
    out.clear();
    for (auto& e : v.items) { if (e.kind != gsc::JsonValue::Number || e.text.find_first_not_of("0123456789") != std::string::npos || e.text.size() > 3 || atoi(e.text.c_str()) > 255) return false; out.push_back((uint8_t)atoi(e.text.c_str())); }
    return true;
}
bool fold_eq(const std::string& a, const char* b) { if (a.size() != strlen(b)) return false; for (size_t i = 0; i < a.size(); i++) if ((a[i] | 32) != (b[i] | 32)) return false; return true; }

bool verify_impl(const char* data, size_t len) {
    gsc::JsonValue root = gsc::json_parse(data, len);
    if (root.kind != gsc::JsonValue::Object) return false;
    std::string cipher; std::vector<uint8_t> proof, sig;
    for (auto& kv : root.members) {
        if (fold_eq(kv.first, "cipher")) { if (kv.second.kind == gsc::JsonValue::String) cipher = kv.second.text; else if (kv.second.kind != gsc::JsonValue::Null) return false; }
        else if (fold_eq(kv.first, "proof")) { if (!bytes_field(kv.second, proof)) return false; }
        else if (fold_eq(kv.first, "publicSignals")) { if (!bytes_field(kv.second, sig)) return false; }
    }
    int id = -1; for (int k = 0; k < 3; k++) if (cipher == kNames[k]) id = k;
    if (id < 0) return false;                                   // verify_impl.go:78-81: unknown cipher -> false
    const VerifyingKey* vk;
    { std::lock_guard<std::mutex> l(g_mu); load_dir_once(); vk = g_vk[id].get(); }
    if (!vk) { printf("verifying key for %s is not loaded\n", kNames[id]); return false; }
    if (sig.size() != 144) { printf("public signals must be 144 bytes, not %zu\n", sig.size()); return false; }   // verifiers.go:52-55
    const uint8_t *ct = sig.data(), *nonce = ct + 64, *ctr = ct + 76, *pt = ct + 80;
    // public inputs in circuit order (verifiers.go:18-23 / :35-40), as (base index into vk.K[1..], scalar)
    std::vector<G1> terms; terms.push_back(vk->K.at(0));
    size_t npub = 0;
    if (id == 0) {        // bits: Counter[32] (LE value), Nonce[3][32] (LE words), In[16][32] (BE words), Out[16][32] (BE words)
        auto word = [](const uint8_t* p, bool be) { return be ? (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3] : (uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]; };
        std::vector<uint32_t> words; words.push_back(word(ctr, false));
        for (int i = 0; i < 3; i++) words.push_back(word(nonce + 4 * i, false));
        for (int i = 0; i < 16; i++) words.push_back(word(pt + 4 * i, true));
        for (int i = 0; i < 16; i++) words.push_back(word(ct + 4 * i, true));
        npub = 32 * words.size();
        if (vk->K.size() != 1 + npub) return false;
        for (size_t w = 0; w < words.size(); w++) for (int bit = 0; bit < 32; bit++) if ((words[w] >> bit) & 1) terms.push_back(vk->K[1 + 32 * w + bit]);
    } else {              // bytes: Nonce[12], Counter (BE u32), Plaintext[64], Ciphertext[64]
        std::vector<uint32_t> vals; for (int i = 0; i < 12; i++) vals.push_back(nonce[i]);
        vals.push_back((uint32_t)ctr[0] << 24 | (uint32_t)ctr[1] << 16 | (uint32_t)ctr[2] << 8 | ctr[3]);
        for (int i = 0; i < 64; i++) vals.push_back(pt[i]);
        for (int i = 0; i < 64; i++) vals.push_back(ct[i]);
        npub = vals.size();
        if (vk->K.size() != 1 + npub + (vk->has_commitment ? 1 : 0)) return false;
        for (size_t i = 0; i < npub; i++) if (vals[i]) terms.push_back(g1_mul(vk->K[1 + i], U256{{vals[i], 0, 0, 0}}));
    }
    // proof (App. B.3)
    const size_t nc = vk->has_commitment ? 1 : 0;
    if (proof.size() != 164 + 32 * nc || be32(proof.data() + 128) != nc) return false;
    G1 Ar, Krs, D{Fp::zero(), Fp::zero(), true}, pok; G2 Bs;
    if (!g1_decode(proof.data(), Ar) || !g2_decode(proof.data() + 32, Bs) || !g1_decode(proof.data() + 96, Krs)) return false;
    if (nc && !g1_decode(proof.data() + 132, D)) return false;
    if (!g1_decode(proof.data() + 132 + 32 * nc, pok)) return false;
    if (nc) {
        uint8_t msg[64]; D.x.to_be(msg); D.y.to_be(msg + 32); if (D.inf) { memset(msg, 0, 64); msg[0] = 0x40; }
        const Fr c = hash_to_fr(msg, 64, "bsb22-commitment");
        terms.push_back(g1_mul(vk->K[1 + npub], c.canon())); terms.push_back(D);
        if (!pairing_product_is_one({{D, vk->ped_gsn}, {pok, vk->ped_g}})) return false;       // proof of knowledge of the commitment
    }
    const G1 L = g1_sum(terms);
    return pairing_product_is_one({{Ar, Bs}, {g1_neg(vk->alpha), vk->beta}, {g1_neg(L), vk->gamma}, {g1_neg(Krs), vk->delta}});
}

}  // namespace

extern "C" {

GoUint8 Verify(GoSlice params) {
    try { K(); return verify_impl((const char*)params.data, params.len > 0 ? (size_t)params.len : 0) ? 1 : 0; }
    catch (const std::exception& e) { printf("%s\n", e.what()); return 0; }      // verify_impl.go:64-69: any panic -> false
}

GoUint8 InitVerifier(GoUint8 algorithmID, GoSlice verifyingKey) {
    if (algorithmID > 2 || !verifyingKey.data || verifyingKey.len <= 0) return 0;
    try { K(); auto vk = parse_vk((const uint8_t*)verifyingKey.data, (size_t)verifyingKey.len); std::lock_guard<std::mutex> l(g_mu); g_vk[algorithmID] = std::move(vk); return 1; }
    catch (const std::exception& e) { printf("%s\n", e.what()); return 0; }
}

}  // extern "C"
