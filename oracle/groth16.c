/* TEST ORACLE — not product code (see bn254.h).
 *
 * CPU restatement of the Groth16 prove / verify path the reference runs through
 * groth16.Prove (libraries/prover/impl/provers.go:148,216), proof.WriteTo (:152-157, :220-226),
 * ProvingKey.ReadFrom (prove_impl.go:86-87), VerifyingKey.ReadFrom (libraries/verifier/impl/verify_impl.go:36-37)
 * and groth16.Verify (libraries/verifier/impl/verifiers.go:99,145).  Those live in the un-vendored
 * module github.com/consensys/gnark v0.11.0 (go.mod:8); what is restated here is the key/proof
 * layout of SURVEY.md App. B and the prover algebra of App. D.  Pinned by the pairing check against
 * the reference-authored tests/golden/vk.chacha20 and the byte-level vectors of App. E.
 * The AES commitment transcript (App. H) is "parity unpinned": no reference AES proving key ships.
 */
#include "groth16.h"
#include "ciphers.h"
#include <stdio.h>

static uint64_t rd64be(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
static uint32_t rd32be(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

typedef struct { const uint8_t *b; size_t n, i; int err; } rdr;
static int take_g1(rdr *r, g1aff *p) { int k = g1_decode(p, r->b + r->i, r->n - r->i); if (k != 32) { r->err = 1; return -1; } r->i += 32; return 0; }
static int take_g2(rdr *r, g2aff *p) { int k = g2_decode(p, r->b + r->i, r->n - r->i); if (k != 64) { r->err = 1; return -1; } r->i += 64; return 0; }
static g1aff *take_g1_slice(rdr *r, size_t *n) {
    if (r->i + 4 > r->n) { r->err = 1; return NULL; }
    uint32_t k = rd32be(r->b + r->i); r->i += 4;
    if (r->i + 32ull * k > r->n) { r->err = 1; return NULL; }
    g1aff *a = (g1aff *)malloc(sizeof(g1aff) * (k ? k : 1)); int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : bad)
#endif
    for (long j = 0; j < (long)k; j++) if (g1_decode(&a[j], r->b + r->i + 32 * (size_t)j, 32) != 32) bad |= 1;
    r->i += 32ull * k; if (bad) r->err = 1; *n = k; return a;
}
static g2aff *take_g2_slice(rdr *r, size_t *n) {
    if (r->i + 4 > r->n) { r->err = 1; return NULL; }
    uint32_t k = rd32be(r->b + r->i); r->i += 4;
    if (r->i + 64ull * k > r->n) { r->err = 1; return NULL; }
    g2aff *a = (g2aff *)malloc(sizeof(g2aff) * (k ? k : 1)); int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : bad)
#endif
    for (long j = 0; j < (long)k; j++) if (g2_decode(&a[j], r->b + r->i + 64 * (size_t)j, 64) != 64) bad |= 1;
    r->i += 64ull * k; if (bad) r->err = 1; *n = k; return a;
}
static void take_fr(rdr *r, fe *v) { if (r->i + 32 > r->n || !fr_from_be(v, r->b + r->i)) { r->err = 1; return; } r->i += 32; }

/* App. B.1 */
int pk_parse(pk_t *pk, const uint8_t *buf, size_t len) {
    memset(pk, 0, sizeof *pk);
    rdr r = {buf, len, 0, 0};
    if (len < 169) return -1;
    pk->n = rd64be(buf); r.i = 8;
    take_fr(&r, &pk->n_inv); take_fr(&r, &pk->omega); take_fr(&r, &pk->omega_inv); take_fr(&r, &pk->g); take_fr(&r, &pk->g_inv);
    r.i += 1;  /* "with precompute" flag */
    take_g1(&r, &pk->alpha); take_g1(&r, &pk->beta); take_g1(&r, &pk->delta);
    if (r.err) return -1;
    pk->A = take_g1_slice(&r, &pk->nA); pk->B = take_g1_slice(&r, &pk->nB);
    pk->Z = take_g1_slice(&r, &pk->nZ); pk->K = take_g1_slice(&r, &pk->nK);
    take_g2(&r, &pk->beta2); take_g2(&r, &pk->delta2);
    pk->B2 = take_g2_slice(&r, &pk->nB2);
    if (r.err || r.i + 24 > len) return -2;
    pk->n_wires = rd64be(buf + r.i); pk->n_infA = rd64be(buf + r.i + 8); pk->n_infB = rd64be(buf + r.i + 16); r.i += 24;
    if (r.i + 2 * pk->n_wires + 4 > len) return -3;
    pk->infA = (uint8_t *)malloc(pk->n_wires); memcpy(pk->infA, buf + r.i, pk->n_wires); r.i += pk->n_wires;
    pk->infB = (uint8_t *)malloc(pk->n_wires); memcpy(pk->infB, buf + r.i, pk->n_wires); r.i += pk->n_wires;
    pk->n_ck = (int)rd32be(buf + r.i); r.i += 4;
    if (pk->n_ck > 1) return -4;
    if (pk->n_ck == 1) {
        pk->basis = take_g1_slice(&r, &pk->n_basis);
        size_t n2; pk->basis_sigma = take_g1_slice(&r, &n2);
        if (r.err || n2 != pk->n_basis) return -5;
    }
    if (r.i != len) return -6;
    size_t ca = 0, cb = 0;
    for (size_t i = 0; i < pk->n_wires; i++) { ca += !pk->infA[i]; cb += !pk->infB[i]; }
    if (ca != pk->nA || cb != pk->nB || cb != pk->nB2 || pk->nZ + 1 != pk->n) return -7;
    return 0;
}
void pk_free(pk_t *pk) {
    free(pk->A); free(pk->B); free(pk->Z); free(pk->K); free(pk->B2); free(pk->infA); free(pk->infB); free(pk->basis); free(pk->basis_sigma);
    memset(pk, 0, sizeof *pk);
}
/* App. B.2 */
int vk_parse(vk_t *vk, const uint8_t *buf, size_t len) {
    memset(vk, 0, sizeof *vk);
    rdr r = {buf, len, 0, 0};
    take_g1(&r, &vk->alpha); take_g1(&r, &vk->beta1); take_g2(&r, &vk->beta2); take_g2(&r, &vk->gamma2);
    take_g1(&r, &vk->delta1); take_g2(&r, &vk->delta2);
    if (r.err) return -1;
    vk->K = take_g1_slice(&r, &vk->nK);
    if (r.err || r.i + 4 > len) return -2;
    uint32_t outer = rd32be(buf + r.i); r.i += 4;
    if (outer > 1) return -3;
    vk->n_commit = (int)outer;
    for (uint32_t o = 0; o < outer; o++) {
        if (r.i + 4 > len) return -3;
        uint32_t inner = rd32be(buf + r.i); r.i += 4;
        if (inner) return -3;   /* public committed wires: none in the reference circuits */
        r.i += 8ull * inner;
    }
    if (r.i + 4 > len) return -4;
    uint32_t nck = rd32be(buf + r.i); r.i += 4;
    if (nck != outer) return -4;
    if (nck) { g2aff a, b; if (g2_decode(&a, buf + r.i, len - r.i) != 64) return -5; r.i += 64; if (g2_decode(&b, buf + r.i, len - r.i) != 64) return -5; r.i += 64; vk->ped_g = a; vk->ped_gsigma_neg = b; }
    if (r.i != len) return -6;
    return 0;
}
void vk_free(vk_t *vk) { free(vk->K); memset(vk, 0, sizeof *vk); }

/* ---- NTT (natural order in/out) ---- */
static void ntt(fe *a, size_t n, const fe *w) {
    int lg = 0; while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0; for (int b = 0; b < lg; b++) if (i >> b & 1) j |= (size_t)1 << (lg - 1 - b);
        if (j > i) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    fe *tw = (fe *)malloc(sizeof(fe) * (n / 2 ? n / 2 : 1));
    fr_set_one(&tw[0]); for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], w);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n / len >= 8)
#endif
        for (long s = 0; s < (long)n; s += (long)len)
            for (size_t k = 0; k < half; k++) {
                fe u = a[s + k], v; fr_mul(&v, &a[s + k + half], &tw[k * step]);
                fr_add(&a[s + k], &u, &v); fr_sub(&a[s + k + half], &u, &v);
            }
    }
    free(tw);
}

/* App. D: H = (A*B - C)/(X^n - 1) through the coset g*<omega>; h gets n coefficients in NATURAL order */
void compute_h(const pk_t *pk, const fe *A, const fe *B, const fe *C, size_t m, fe *h) {
    size_t n = pk->n;
    fe *a = (fe *)calloc(n, sizeof(fe)), *b = (fe *)calloc(n, sizeof(fe)), *c = (fe *)calloc(n, sizeof(fe));
    memcpy(a, A, m * sizeof(fe)); memcpy(b, B, m * sizeof(fe)); memcpy(c, C, m * sizeof(fe));
    fe *v[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        ntt(v[k], n, &pk->omega_inv);
        fe gp = pk->n_inv;                      /* coefficient j * n^-1 * g^j */
        for (size_t j = 0; j < n; j++) { fr_mul(&v[k][j], &v[k][j], &gp); fr_mul(&gp, &gp, &pk->g); }
        ntt(v[k], n, &pk->omega);
    }
    /* g^n - 1 */
    fe gn = pk->g, one, den; for (size_t t = n; t > 1; t >>= 1) fr_sqr(&gn, &gn);
    fr_set_one(&one); fr_sub(&den, &gn, &one); fr_inv(&den, &den);
    for (size_t j = 0; j < n; j++) { fe t; fr_mul(&t, &a[j], &b[j]); fr_sub(&t, &t, &c[j]); fr_mul(&h[j], &t, &den); }
    ntt(h, n, &pk->omega_inv);
    fe gp = pk->n_inv;
    for (size_t j = 0; j < n; j++) { fr_mul(&h[j], &h[j], &gp); fr_mul(&gp, &gp, &pk->g_inv); }
    free(a); free(b); free(c);
}

static size_t bitrev(size_t i, int lg) { size_t j = 0; for (int b = 0; b < lg; b++) if (i >> b & 1) j |= (size_t)1 << (lg - 1 - b); return j; }

/* gnark hash_to_field with one output element: expand_message_xmd(SHA-256, 48 bytes) mod r */
void hash_to_fr(fe *out, const uint8_t *msg, size_t len, const char *dst) {
    uint8_t x[48]; expand_message_xmd(msg, len, (const uint8_t *)dst, strlen(dst), x, 48);
    /* big-endian 384-bit integer mod r: Horner by bytes */
    fe acc, b256, t; fr_set_zero(&acc); fr_from_u64(&b256, 256);
    for (int i = 0; i < 48; i++) { fr_mul(&acc, &acc, &b256); fr_from_u64(&t, x[i]); fr_add(&acc, &acc, &t); }
    *out = acc;
}

typedef struct { const pk_t *pk; g1aff commitment; fe *vals; size_t n; const r1cs_t *cs; } commit_ctx;
static int commit_cb(void *vctx, const fe *vals, size_t n, fe *out) {
    commit_ctx *c = (commit_ctx *)vctx;
    size_t npub = c->cs->n_pub_committed;
    if (!c->pk->n_ck || n != npub + c->pk->n_basis) return -1;
    const fe *priv = vals + npub; size_t np = n - npub;
    uint64_t *sc = (uint64_t *)malloc(32 * (np ? np : 1));
    for (size_t i = 0; i < np; i++) fr_to_canon(sc + 4 * i, &priv[i]);
    g1jac D; g1_msm(&D, c->pk->basis, sc, np); free(sc);
    g1_jac_to_aff(&c->commitment, &D);
    c->vals = (fe *)malloc(sizeof(fe) * (np ? np : 1)); memcpy(c->vals, priv, sizeof(fe) * np); c->n = np;
    uint8_t *msg = (uint8_t *)malloc(64 + 32 * npub);
    g1_encode_uncompressed(msg, &c->commitment);
    for (size_t i = 0; i < npub; i++) fr_to_be(msg + 64 + 32 * i, &vals[i]);
    hash_to_fr(out, msg, 64 + 32 * npub, "bsb22-commitment");
    free(msg);
    return 0;
}

static void msm_filtered_g1(g1jac *out, const g1aff *pts, size_t npts, const fe *W, const uint8_t *skip, size_t lo, size_t hi) {
    uint64_t *sc = (uint64_t *)malloc(32 * (npts ? npts : 1)); size_t k = 0;
    for (size_t i = lo; i < hi; i++) if (!skip[i]) { if (k < npts) fr_to_canon(sc + 4 * k, &W[i]); k++; }
    if (k != npts) { g1_jac_set_inf(out); free(sc); return; }
    g1_msm(out, pts, sc, npts); free(sc);
}

int groth16_prove(const r1cs_t *cs, const pk_t *pk, const fe *witness, const fe *r, const fe *s, const fe *mask,
                  uint8_t *proof_out, size_t *proof_len, prove_dump_t *dump) {
    size_t n = pk->n, m = cs->n_constraints, nw = cs->n_wires;
    if (nw != pk->n_wires || m > n) return -1;
    fe *W = (fe *)malloc(sizeof(fe) * nw), *A = (fe *)malloc(sizeof(fe) * m), *B = (fe *)malloc(sizeof(fe) * m), *C = (fe *)malloc(sizeof(fe) * m);
    commit_ctx cc; memset(&cc, 0, sizeof cc); cc.pk = pk; cc.cs = cs;
    solve_opts_t so = {mask, cs->n_commit ? commit_cb : NULL, &cc};
    long rc = r1cs_solve(cs, witness, W, A, B, C, &so);
    if (rc) { free(W); free(A); free(B); free(C); free(cc.vals); return -2; }
    fe *h = (fe *)malloc(sizeof(fe) * n);
    compute_h(pk, A, B, C, m, h);
    if (!fr_is_zero(&h[n - 1])) { free(W); free(A); free(B); free(C); free(h); free(cc.vals); return -3; }
    if (dump) {
        if (dump->W) for (size_t i = 0; i < nw; i++) fr_to_be(dump->W + 32 * i, &W[i]);
        if (dump->A) for (size_t i = 0; i < m; i++) { fr_to_be(dump->A + 32 * i, &A[i]); fr_to_be(dump->B + 32 * i, &B[i]); fr_to_be(dump->C + 32 * i, &C[i]); }
        if (dump->h) for (size_t i = 0; i < n; i++) fr_to_be(dump->h + 32 * i, &h[i]);
    }
    /* MSMs */
    int lg = 0; while (((size_t)1 << lg) < n) lg++;
    g1jac sA, sB1, sK, sZ; g2jac sB2;
    msm_filtered_g1(&sA, pk->A, pk->nA, W, pk->infA, 0, nw);
    msm_filtered_g1(&sB1, pk->B, pk->nB, W, pk->infB, 0, nw);
    {
        uint64_t *sc = (uint64_t *)malloc(32 * (pk->nB2 ? pk->nB2 : 1)); size_t k = 0;
        for (size_t i = 0; i < nw; i++) if (!pk->infB[i]) fr_to_canon(sc + 4 * k++, &W[i]);
        g2_msm(&sB2, pk->B2, sc, pk->nB2); free(sc);
    }
    {
        uint8_t *skip = (uint8_t *)calloc(nw, 1);
        if (cs->n_commit) { for (size_t i = 0; i < cs->n_commit_priv; i++) skip[cs->commit_priv[i]] = 1; skip[cs->commit_index] = 1; }
        msm_filtered_g1(&sK, pk->K, pk->nK, W, skip, cs->n_public, nw);
        free(skip);
    }
    {
        uint64_t *sc = (uint64_t *)malloc(32 * pk->nZ);
        for (size_t k = 0; k < pk->nZ; k++) fr_to_canon(sc + 4 * k, &h[bitrev(k, lg)]);
        g1_msm(&sZ, pk->Z, sc, pk->nZ); free(sc);
    }
    uint64_t rc4[4], sc4[4], rs4[4]; fe rs; fr_mul(&rs, r, s); fr_neg(&rs, &rs);
    fr_to_canon(rc4, r); fr_to_canon(sc4, s); fr_to_canon(rs4, &rs);
    g1jac d1, t, Ar, Bs1, Krs; g2jac d2, t2, Bs2;
    g1_jac_from_aff(&d1, &pk->delta); g2_jac_from_aff(&d2, &pk->delta2);
    g1_jac_mul(&t, &d1, rc4); g1_jac_add(&Ar, &sA, &t); g1_jac_add_aff(&Ar, &Ar, &pk->alpha);
    g1_jac_mul(&t, &d1, sc4); g1_jac_add(&Bs1, &sB1, &t); g1_jac_add_aff(&Bs1, &Bs1, &pk->beta);
    g2_jac_mul(&t2, &d2, sc4); g2_jac_add(&Bs2, &sB2, &t2); g2_jac_add_aff(&Bs2, &Bs2, &pk->beta2);
    g1_jac_add(&Krs, &sK, &sZ);
    g1_jac_mul(&t, &Ar, sc4); g1_jac_add(&Krs, &Krs, &t);
    g1_jac_mul(&t, &Bs1, rc4); g1_jac_add(&Krs, &Krs, &t);
    g1_jac_mul(&t, &d1, rs4); g1_jac_add(&Krs, &Krs, &t);
    g1aff aAr, aKrs; g2aff aBs; g1_jac_to_aff(&aAr, &Ar); g1_jac_to_aff(&aKrs, &Krs); g2_jac_to_aff(&aBs, &Bs2);
    /* App. B.3 serialization */
    uint8_t *o = proof_out; size_t p = 0;
    g1_encode_compressed(o + p, &aAr); p += 32; g2_encode_compressed(o + p, &aBs); p += 64; g1_encode_compressed(o + p, &aKrs); p += 32;
    uint32_t nc = (uint32_t)cs->n_commit; o[p++] = (uint8_t)(nc >> 24); o[p++] = (uint8_t)(nc >> 16); o[p++] = (uint8_t)(nc >> 8); o[p++] = (uint8_t)nc;
    g1aff pok; memset(&pok, 0, sizeof pok); pok.inf = 1;
    if (nc) {
        g1_encode_compressed(o + p, &cc.commitment); p += 32;
        uint64_t *sc = (uint64_t *)malloc(32 * (cc.n ? cc.n : 1));
        for (size_t i = 0; i < cc.n; i++) fr_to_canon(sc + 4 * i, &cc.vals[i]);
        g1jac P; g1_msm(&P, pk->basis_sigma, sc, cc.n); free(sc); g1_jac_to_aff(&pok, &P);
    }
    g1_encode_compressed(o + p, &pok); p += 32;
    *proof_len = p;
    free(W); free(A); free(B); free(C); free(h); free(cc.vals);
    return 0;
}

/* App. B.4 / App. H.  pub: n_pub elements (without the ONE wire). */
int groth16_verify(const vk_t *vk, const uint8_t *proof, size_t proof_len, const fe *pub, size_t n_pub) {
    if (proof_len < 164) return 0;
    g1aff Ar, Krs, pok, D; g2aff Bs; memset(&D, 0, sizeof D); D.inf = 1;
    if (g1_decode(&Ar, proof, 32) != 32 || g2_decode(&Bs, proof + 32, 64) != 64 || g1_decode(&Krs, proof + 96, 32) != 32) return 0;
    uint32_t nc = rd32be(proof + 128);
    if (nc != (uint32_t)vk->n_commit || proof_len != 164 + 32ull * nc) return 0;
    size_t p = 132;
    if (nc) { if (g1_decode(&D, proof + p, 32) != 32) return 0; p += 32; }
    if (g1_decode(&pok, proof + p, 32) != 32) return 0;
    if (vk->nK != 1 + n_pub + nc) return 0;
    size_t ns = n_pub + nc;
    uint64_t *sc = (uint64_t *)malloc(32 * (ns ? ns : 1));
    for (size_t i = 0; i < n_pub; i++) fr_to_canon(sc + 4 * i, &pub[i]);
    if (nc) {
        uint8_t msg[64]; g1_encode_uncompressed(msg, &D);
        fe c; hash_to_fr(&c, msg, 64, "bsb22-commitment"); fr_to_canon(sc + 4 * n_pub, &c);
        g1aff Pp[2] = {D, pok}; g2aff Qq[2] = {vk->ped_gsigma_neg, vk->ped_g};
        if (!pairing_product_is_one(Pp, Qq, 2)) { free(sc); return 0; }
    }
    g1jac L; g1_msm(&L, vk->K + 1, sc, ns); free(sc);
    g1_jac_add_aff(&L, &L, &vk->K[0]);
    if (nc) g1_jac_add_aff(&L, &L, &D);
    g1aff La; g1_jac_to_aff(&La, &L);
    g1aff P[4]; g2aff Q[4];
    P[0] = Ar; Q[0] = Bs;
    g1_aff_neg(&P[1], &vk->alpha); Q[1] = vk->beta2;
    g1_aff_neg(&P[2], &La); Q[2] = vk->gamma2;
    g1_aff_neg(&P[3], &Krs); Q[3] = vk->delta2;
    return pairing_product_is_one(P, Q, 4);
}

/* ---- witness assignment (libraries/prover/impl/provers.go:106-142, :194-210; utils/bytes.go:11-47) ---- */
static void put_bits32(fe *dst, uint32_t v) { for (int i = 0; i < 32; i++) fr_from_u64(&dst[i], (v >> i) & 1); }
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t le32(const uint8_t *p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
/* public: Counter[32], Nonce[3][32], In[16][32], Out[16][32]; secret: Key[8][32]  -> 1408 elements */
void assign_chacha(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t pt[64], uint8_t ct[64], fe *wit) {
    chacha20_xor(key, nonce, counter, pt, ct, 64);
    fe *p = wit;
    put_bits32(p, counter); p += 32;
    for (int i = 0; i < 3; i++, p += 32) put_bits32(p, le32(nonce + 4 * i));
    for (int i = 0; i < 16; i++, p += 32) put_bits32(p, be32(pt + 4 * i));
    for (int i = 0; i < 16; i++, p += 32) put_bits32(p, be32(ct + 4 * i));
    for (int i = 0; i < 8; i++, p += 32) put_bits32(p, le32(key + 4 * i));
}
/* public: Nonce[12], Counter, Plaintext[64], Ciphertext[64]; secret: Key[16|32] -> 141 + keylen elements */
void assign_aes(const uint8_t *key, int keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t pt[64], uint8_t ct[64], fe *wit) {
    aes_ctr_xor(key, keylen, nonce, counter, pt, ct, 64);
    fe *p = wit;
    for (int i = 0; i < 12; i++) fr_from_u64(p++, nonce[i]);
    fr_from_u64(p++, counter);
    for (int i = 0; i < 64; i++) fr_from_u64(p++, pt[i]);
    for (int i = 0; i < 64; i++) fr_from_u64(p++, ct[i]);
    for (int i = 0; i < keylen; i++) fr_from_u64(p++, key[i]);
}
