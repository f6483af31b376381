// Groth16 Setup, host side (see setup.hpp).  Algebra: SURVEY.md App. D (QAP at tau), App. H (commitment extension),
// file layouts App. B.1 / B.2.  Reference call sites: keygen.go:345,384,423 (groth16.Setup), :341-352 (WriteTo).
#include "setup.hpp"
#include "formats.hpp"
#include "host_ciphers.hpp"
#include "host_field.hpp"
#include <sys/random.h>
#include <cerrno>
#include <cstring>
#include <stdexcept>
#include <string>

namespace gsc {
using namespace hostf;

namespace {

Fr fr_from_be_reduce(const uint8_t* b, size_t n) {      // big-endian integer of any length mod r (fr.Hash reduces its 48 xmd bytes the same way)
    Fr acc = Fr::zero(); const Fr k = Fr::from_u64(256);
    for (size_t i = 0; i < n; i++) acc = acc * k + Fr::from_u64(b[i]);
    return acc;
}
Fr fr_from_hex(const char* hex) {
    uint8_t b[32]; memset(b, 0, 32); const size_t n = strlen(hex);
    for (size_t i = 0; i < n; i++) { const char ch = hex[n - 1 - i]; const int v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; b[31 - i / 2] |= (uint8_t)(v << (4 * (i & 1))); }
    Fr r; if (!Fr::from_be(b, r)) throw std::runtime_error("setup: bad constant"); return r;
}
Fp fp_from_hex(const char* hex) {
    uint8_t b[32]; memset(b, 0, 32); const size_t n = strlen(hex);
    for (size_t i = 0; i < n; i++) { const char ch = hex[n - 1 - i]; const int v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; b[31 - i / 2] |= (uint8_t)(v << (4 * (i & 1))); }
    Fp r; if (!Fp::from_be(b, r)) throw std::runtime_error("setup: bad constant"); return r;
}
// Toxic waste and everything derived from it in scalar form is overwritten before its memory is released — also when an exception
// unwinds the stack (a HIP error, bad_alloc): secrets live in SecretVec / under a WipeOnExit guard, whose destructors do the wiping.
void wipe_bytes(void* p, size_t n) { volatile uint8_t* q = static_cast<volatile uint8_t*>(p); for (size_t i = 0; i < n; i++) q[i] = 0; }
void wipe(Fr& v) { wipe_bytes(v.v.w, sizeof v.v.w); }
template <class T> struct SecretVec : std::vector<T> {
    using std::vector<T>::vector;
    SecretVec(const SecretVec&) = delete; SecretVec& operator=(const SecretVec&) = delete;
    ~SecretVec() { wipe_bytes(this->data(), this->size() * sizeof(T)); }
};
struct WipeOnExit {      // wipes the listed scalars when the scope is left, normally or by an exception
    std::vector<Fr*> frs;
    explicit WipeOnExit(std::initializer_list<Fr*> l) : frs(l) {}
    WipeOnExit(const WipeOnExit&) = delete; WipeOnExit& operator=(const WipeOnExit&) = delete;
    ~WipeOnExit() { for (Fr* f : frs) wipe(*f); }
};
// One toxic scalar: hash_to_field(seed | label) — uniform in Fr up to 2^-128; never zero.
Fr toxic(const uint8_t seed[32], const char* label) {
    uint8_t msg[64], h[48]; memcpy(msg, seed, 32); memset(msg + 32, 0, 32); strncpy((char*)msg + 32, label, 31);
    expand_message_xmd_sha256(msg, 64, "gsc-test-setup", h, 48);
    Fr t = fr_from_be_reduce(h, 48);
    wipe_bytes(msg, sizeof msg); wipe_bytes(h, sizeof h);      // the seed and the integer that reduces to the scalar
    return t.is_zero() ? Fr::one() : t;
}
void fr_to_le(const Fr& v, uint8_t* out) { const U256 c = v.canon(); memcpy(out, c.w, 32); }
void store_mont(const Fp& v, uint8_t* out) { memcpy(out, v.v.w, 32); }

struct Out {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { for (int i = 3; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 7; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
    void fr(const Fr& v) { uint8_t t[32]; v.to_be(t); b.insert(b.end(), t, t + 32); }
    void raw(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
};
// (p-1)/2 big-endian: a compressed point carries the "larger y" flag iff y > (p-1)/2
const uint8_t kHalfP[32] = {0x18, 0x32, 0x27, 0x39, 0x70, 0x98, 0xd0, 0x14, 0xdc, 0x28, 0x22, 0xdb, 0x40, 0xc0, 0xac, 0x2e,
                            0xcb, 0xc0, 0xb5, 0x48, 0xb4, 0x38, 0xe5, 0x46, 0x9e, 0x10, 0x46, 0x0b, 0x6c, 0x3e, 0x7e, 0xa3};
void le_to_be(const uint8_t* le, uint8_t* be) { for (int i = 0; i < 32; i++) be[i] = le[31 - i]; }
bool be_zero(const uint8_t* a) { for (int i = 0; i < 32; i++) if (a[i]) return false; return true; }
// gnark-crypto compressed encodings from canonical little-endian affine coordinates
void put_g1(Out& o, const uint8_t* xy_le, bool inf) {
    uint8_t x[32], y[32];
    if (inf) { memset(x, 0, 32); x[0] = 0x40; o.raw(x, 32); return; }
    le_to_be(xy_le, x); le_to_be(xy_le + 32, y);
    x[0] |= memcmp(y, kHalfP, 32) > 0 ? 0xC0 : 0x80;
    o.raw(x, 32);
}
void put_g2(Out& o, const uint8_t* p_le, bool inf) {      // p_le: x.a0, x.a1, y.a0, y.a1
    uint8_t x0[32], x1[32], y0[32], y1[32];
    if (inf) { uint8_t z[64]; memset(z, 0, 64); z[0] = 0x40; o.raw(z, 64); return; }
    le_to_be(p_le, x0); le_to_be(p_le + 32, x1); le_to_be(p_le + 64, y0); le_to_be(p_le + 96, y1);
    const bool large = be_zero(y1) ? memcmp(y0, kHalfP, 32) > 0 : memcmp(y1, kHalfP, 32) > 0;
    x1[0] |= large ? 0xC0 : 0x80;
    o.raw(x1, 32); o.raw(x0, 32);
}

}  // namespace

SetupKeys groth16_setup(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* seed32, int device) {
    Fr::init(); Fp::init();
    const R1csFile cs = parse_r1cs(r1cs, r1cs_len);
    const size_t m = cs.n_constraints, nw = cs.n_wires(), npub = cs.n_public;
    if (cs.n_public_committed) throw std::runtime_error("setup: public committed wires are not supported");
    size_t n = 1; int lg = 0; while (n < m) { n <<= 1; lg++; }
    if (lg > 28) throw std::runtime_error("setup: constraint system too large");
    uint8_t seed[32];
    struct SeedWipe { uint8_t* s; ~SeedWipe() { wipe_bytes(s, 32); } } seed_wipe{seed};
    if (seed32) memcpy(seed, seed32, 32);
    else { size_t got = 0; while (got < 32) { const ssize_t k = getrandom(seed + got, 32 - got, 0); if (k > 0) got += (size_t)k; else if (!(k < 0 && errno == EINTR)) throw std::runtime_error(std::string("getrandom failed: ") + strerror(errno)); } }
    Fr tau = toxic(seed, "tau"), alpha = toxic(seed, "alpha"), beta = toxic(seed, "beta"), gamma = toxic(seed, "gamma"), delta = toxic(seed, "delta"),
       sigma = toxic(seed, "sigma"), ped_g = toxic(seed, "pedersen-g");
    Fr gamma_inv = Fr::zero(), delta_inv = Fr::zero(), tn = Fr::zero(), zt = Fr::zero();
    WipeOnExit toxic_guard{&tau, &alpha, &beta, &gamma, &delta, &sigma, &ped_g, &gamma_inv, &delta_inv, &tn, &zt};
    wipe_bytes(seed, 32);
    // domain (gnark-crypto fft.NewDomain): generator of the 2^lg-th roots from the 2^28-th root, coset shift 5
    Fr omega = fr_from_hex("2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0");
    for (int i = lg; i < 28; i++) omega = omega.sq();
    const Fr omega_inv = omega.inv(), n_inv = Fr::from_u64(n).inv(), g = Fr::from_u64(5), g_inv = g.inv(), one = Fr::one();
    // Lagrange basis at tau: L_j = (tau^n - 1) / n * w^j / (tau - w^j); one batch inversion
    tn = tau; for (int i = 0; i < lg; i++) tn = tn.sq();
    zt = tn - one;
    SecretVec<Fr> L(n);
    {
        SecretVec<Fr> den(n), pre(n); std::vector<Fr> wj(n);
        Fr w = one;
        for (size_t j = 0; j < n; j++) { wj[j] = w; den[j] = tau - w; w = w * omega; }
        Fr run = one, inv = Fr::zero(), scale = Fr::zero(), dj = Fr::zero();
        WipeOnExit guard{&run, &inv, &scale, &dj};
        for (size_t j = 0; j < n; j++) { pre[j] = run; run = run * den[j]; }
        if (run.is_zero()) throw std::runtime_error("setup: tau is a root of unity");
        inv = run.inv(); scale = zt * n_inv;
        for (size_t j = n; j-- > 0;) { dj = inv * pre[j]; inv = inv * den[j]; L[j] = dj * wj[j] * scale; }
    }
    // A_i(tau), B_i(tau), C_i(tau): column sums of the R1CS matrices against the Lagrange basis
    SecretVec<Fr> A(nw, Fr::zero()), B(nw, Fr::zero()), C(nw, Fr::zero());
    std::vector<Fr> coeff(cs.n_coeff());
    for (size_t i = 0; i < coeff.size(); i++) memcpy(coeff[i].v.w, cs.coeff_limbs.data() + 8 * i, 32);      // stored in Montgomery form already
    for (size_t ii = 0; ii < cs.n_instr(); ii++) {
        if (cs.bp_kind[cs.blueprint[ii]] != BP_R1C) continue;
        const uint32_t* cd = cs.calldata.data() + cs.instr_start[ii];
        const uint32_t cnt[3] = {cd[1], cd[2], cd[3]}; const uint32_t* t = cd + 4; SecretVec<Fr>* dst[3] = {&A, &B, &C};
        if (cs.constraint_off[ii] >= m) throw std::runtime_error("setup: constraint offset out of range");
        const Fr& Lj = L[cs.constraint_off[ii]];
        for (int side = 0; side < 3; side++) for (uint32_t k = 0; k < cnt[side]; k++, t += 2) {
            if (t[0] >= coeff.size()) throw std::runtime_error("setup: coefficient id out of range");
            const uint32_t wid = t[1] == WIRE_CONST ? 0 : t[1];          // a constant term multiplies the ONE wire
            if (wid >= nw) throw std::runtime_error("setup: wire id out of range");
            (*dst[side])[wid] = (*dst[side])[wid] + coeff[t[0]] * Lj;
        }
    }
    std::vector<uint8_t> committed(nw, 0);
    for (uint32_t w : cs.commit_private) committed[w] = 1;
    gamma_inv = gamma.inv(); delta_inv = delta.inv();
    // ---- scalars of every G1 point of the keys, in one list:  [alpha, beta, delta | A (nw) | B (nw) | K (nw) | Z (n-1) | sigma * basis (ncp)]
    const size_t ncp = cs.commit_private.size();
    const size_t oA = 3, oB = oA + nw, oK = oB + nw, oZ = oK + nw, oS = oZ + (n - 1), n1 = oS + ncp;
    SecretVec<uint8_t> sc1(32 * n1);
    fr_to_le(alpha, &sc1[0]); fr_to_le(beta, &sc1[32]); fr_to_le(delta, &sc1[64]);
    SecretVec<Fr> K(nw);
    for (size_t i = 0; i < nw; i++) {
        fr_to_le(A[i], &sc1[32 * (oA + i)]); fr_to_le(B[i], &sc1[32 * (oB + i)]);
        const bool to_vk = i < npub || (cs.has_commitment && i == cs.commit_wire) || committed[i];
        K[i] = (beta * A[i] + alpha * B[i] + C[i]) * (to_vk ? gamma_inv : delta_inv);
        fr_to_le(K[i], &sc1[32 * (oK + i)]);
    }
    {   // Z[k] = tau^bitrev(k) * Z(tau) / delta, k < n - 1
        SecretVec<Fr> tp(n); tp[0] = zt * delta_inv; for (size_t j = 1; j < n; j++) tp[j] = tp[j - 1] * tau;
        for (size_t k = 0; k + 1 < n; k++) { size_t br = 0; for (int b = 0; b < lg; b++) if ((k >> b) & 1) br |= (size_t)1 << (lg - 1 - b); fr_to_le(tp[br], &sc1[32 * (oZ + k)]); }
    }
    for (size_t j = 0; j < ncp; j++) fr_to_le(K[cs.commit_private[j]] * sigma, &sc1[32 * (oS + j)]);
    // ---- G2: [beta, gamma, delta | B (nw) | pedersen G, -sigma * G]
    const size_t o2B = 3, o2P = o2B + nw, n2 = o2P + 2;
    SecretVec<uint8_t> sc2(32 * n2);
    fr_to_le(beta, &sc2[0]); fr_to_le(gamma, &sc2[32]); fr_to_le(delta, &sc2[64]);
    for (size_t i = 0; i < nw; i++) memcpy(&sc2[32 * (o2B + i)], &sc1[32 * (oB + i)], 32);
    fr_to_le(ped_g, &sc2[32 * o2P]); fr_to_le((ped_g * sigma).neg(), &sc2[32 * (o2P + 1)]);
    // ---- generator multiples on the GPU
    uint8_t g1m[64], g2m[128];
    store_mont(Fp::from_u64(1), g1m); store_mont(Fp::from_u64(2), g1m + 32);
    store_mont(fp_from_hex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"), g2m);
    store_mont(fp_from_hex("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2"), g2m + 32);
    store_mont(fp_from_hex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"), g2m + 64);
    store_mont(fp_from_hex("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b"), g2m + 96);
    std::vector<uint8_t> p1(64 * n1), i1(n1), p2(128 * n2), i2(n2);
    setup_generator_muls(device, false, g1m, sc1.data(), n1, p1.data(), i1.data());
    setup_generator_muls(device, true, g2m, sc2.data(), n2, p2.data(), i2.data());
    // from here on only group elements are needed: discard the toxic waste and every scalar derived from it now (the guards above
    // would do it at the end of the function, and do it on any early exit)
    for (SecretVec<uint8_t>* v : {&sc1, &sc2}) wipe_bytes(v->data(), v->size());
    for (SecretVec<Fr>* v : {&L, &A, &B, &C, &K}) wipe_bytes(v->data(), v->size() * sizeof(Fr));
    for (Fr* f : toxic_guard.frs) wipe(*f);
    auto g1 = [&](Out& o, size_t idx) { put_g1(o, &p1[64 * idx], i1[idx] != 0); };
    auto g2 = [&](Out& o, size_t idx) { put_g2(o, &p2[128 * idx], i2[idx] != 0); };
    auto in_pkK = [&](size_t i) { return i >= npub && !committed[i] && !(cs.has_commitment && i == cs.commit_wire); };
    // ---- pk (App. B.1)
    SetupKeys keys; Out pk;
    pk.u64(n); pk.fr(n_inv); pk.fr(omega); pk.fr(omega_inv); pk.fr(g); pk.fr(g_inv); pk.b.push_back(1);
    g1(pk, 0); g1(pk, 1); g1(pk, 2);
    size_t nA = 0, nB = 0, nK = 0;
    for (size_t i = 0; i < nw; i++) { nA += !i1[oA + i]; nB += !i1[oB + i]; nK += in_pkK(i); }
    pk.u32((uint32_t)nA); for (size_t i = 0; i < nw; i++) if (!i1[oA + i]) g1(pk, oA + i);
    pk.u32((uint32_t)nB); for (size_t i = 0; i < nw; i++) if (!i1[oB + i]) g1(pk, oB + i);
    pk.u32((uint32_t)(n - 1)); for (size_t k = 0; k + 1 < n; k++) g1(pk, oZ + k);
    pk.u32((uint32_t)nK); for (size_t i = 0; i < nw; i++) if (in_pkK(i)) g1(pk, oK + i);
    g2(pk, 0); g2(pk, 2);
    pk.u32((uint32_t)nB); for (size_t i = 0; i < nw; i++) if (!i1[oB + i]) g2(pk, o2B + i);
    pk.u64(nw); pk.u64(nw - nA); pk.u64(nw - nB);
    for (size_t i = 0; i < nw; i++) pk.b.push_back(i1[oA + i] ? 1 : 0);
    for (size_t i = 0; i < nw; i++) pk.b.push_back(i1[oB + i] ? 1 : 0);
    pk.u32(cs.has_commitment ? 1u : 0u);
    if (cs.has_commitment) {      // Pedersen: Basis_j = K_j / gamma for the committed wires, BasisExpSigma = sigma * Basis
        pk.u32((uint32_t)ncp); for (size_t j = 0; j < ncp; j++) g1(pk, oK + cs.commit_private[j]);
        pk.u32((uint32_t)ncp); for (size_t j = 0; j < ncp; j++) g1(pk, oS + j);
    }
    // ---- vk (App. B.2)
    Out vk;
    g1(vk, 0); g1(vk, 1); g2(vk, 0); g2(vk, 1); g1(vk, 2); g2(vk, 2);
    vk.u32((uint32_t)(npub + (cs.has_commitment ? 1 : 0)));
    for (size_t i = 0; i < npub; i++) g1(vk, oK + i);
    if (cs.has_commitment) g1(vk, oK + cs.commit_wire);
    vk.u32(cs.has_commitment ? 1u : 0u); if (cs.has_commitment) vk.u32(0);
    vk.u32(cs.has_commitment ? 1u : 0u); if (cs.has_commitment) { g2(vk, o2P); g2(vk, o2P + 1); }
    keys.pk.swap(pk.b); keys.vk.swap(vk.b);
    return keys;
}

}  // namespace gsc
