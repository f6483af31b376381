/* TEST ORACLE — not product code (see bn254.h).
 *
 * Groth16 Setup in gnark's key layout, needed because the reference does not ship pk.aes128 / pk.aes256
 * (reference .MISSING_LARGE_BLOBS:1-2; they are produced by groth16.Setup at keygen.go:384,423).  Restates the
 * algebra of SURVEY.md App. D / App. H; the output files follow App. B.1 / B.2 so that the product's and the oracle's
 * key decoders read them exactly like reference-made keys.  Toxic waste is derived from a caller-supplied seed:
 * keys made here are TEST keys.  Self-consistency only: no reference AES key exists to compare with.
 */
#include "groth16.h"
#include "ciphers.h"
#include <stdio.h>

static void put32be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put64be(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); }

static void toxic(fe *out, const uint8_t seed[32], const char *label) {
    uint8_t msg[64]; memcpy(msg, seed, 32); memset(msg + 32, 0, 32); strncpy((char *)msg + 32, label, 31);
    hash_to_fr(out, msg, 64, "gsc-test-setup");
    if (fr_is_zero(out)) fr_set_one(out);
}

/* fixed-base tables for the generators: T[j][d-1] = d * 256^j * G, d = 1..255 */
typedef struct { g1aff *t1; g2aff *t2; } gen_tables;
static void build_gen_tables(gen_tables *gt, int want_g2) {
    gt->t1 = (g1aff *)malloc(sizeof(g1aff) * 32 * 255);
    gt->t2 = want_g2 ? (g2aff *)malloc(sizeof(g2aff) * 32 * 255) : NULL;
    g1jac b1; g1_jac_from_aff(&b1, &G1_GEN);
    g2jac b2; g2_jac_from_aff(&b2, &G2_GEN);
    for (int j = 0; j < 32; j++) {
        g1jac a1; g1_jac_set_inf(&a1); g2jac a2; g2_jac_set_inf(&a2);
        for (int d = 1; d <= 255; d++) {
            g1_jac_add(&a1, &a1, &b1); g1_jac_to_aff(&gt->t1[j * 255 + d - 1], &a1);
            if (want_g2) { g2_jac_add(&a2, &a2, &b2); g2_jac_to_aff(&gt->t2[j * 255 + d - 1], &a2); }
        }
        for (int k = 0; k < 8; k++) { g1_jac_dbl(&b1, &b1); if (want_g2) g2_jac_dbl(&b2, &b2); }
    }
}
static void g1_gen_mul(g1aff *out, const gen_tables *gt, const fe *s) {
    uint64_t c[4]; fr_to_canon(c, s);
    g1jac acc; g1_jac_set_inf(&acc);
    for (int j = 0; j < 32; j++) { unsigned d = (unsigned)(c[j / 8] >> (8 * (j % 8))) & 255; if (d) g1_jac_add_aff(&acc, &acc, &gt->t1[j * 255 + d - 1]); }
    g1_jac_to_aff(out, &acc);
}
static void g2_gen_mul(g2aff *out, const gen_tables *gt, const fe *s) {
    uint64_t c[4]; fr_to_canon(c, s);
    g2jac acc; g2_jac_set_inf(&acc);
    for (int j = 0; j < 32; j++) { unsigned d = (unsigned)(c[j / 8] >> (8 * (j % 8))) & 255; if (d) g2_jac_add_aff(&acc, &acc, &gt->t2[j * 255 + d - 1]); }
    g2_jac_to_aff(out, &acc);
}

typedef struct { uint8_t *b; size_t n, cap; } obuf;
static void ob_need(obuf *o, size_t k) { if (o->n + k > o->cap) { o->cap = (o->n + k) * 2 + 4096; o->b = (uint8_t *)realloc(o->b, o->cap); } }
static void ob_g1(obuf *o, const g1aff *p) { ob_need(o, 32); g1_encode_compressed(o->b + o->n, p); o->n += 32; }
static void ob_g2(obuf *o, const g2aff *p) { ob_need(o, 64); g2_encode_compressed(o->b + o->n, p); o->n += 64; }
static void ob_fr(obuf *o, const fe *v) { ob_need(o, 32); fr_to_be(o->b + o->n, v); o->n += 32; }
static void ob_u32(obuf *o, uint32_t v) { ob_need(o, 4); put32be(o->b + o->n, v); o->n += 4; }
static void ob_u64(obuf *o, uint64_t v) { ob_need(o, 8); put64be(o->b + o->n, v); o->n += 8; }

/* 2^28-th root of unity of Fr used by gnark-crypto */
static const char *ROOT_2_28_HEX = "2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0";

static void fr_from_hex(fe *r, const char *hex) {
    uint8_t b[32]; memset(b, 0, 32); size_t n = strlen(hex);
    for (size_t i = 0; i < n; i++) { char ch = hex[n - 1 - i]; int v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; b[31 - i / 2] |= (uint8_t)(v << (4 * (i & 1))); }
    fr_from_be(r, b);
}

int groth16_setup(const r1cs_t *cs, const uint8_t seed[32], uint8_t **pk_out, size_t *pk_len, uint8_t **vk_out, size_t *vk_len) {
    bn254_init();
    const size_t m = cs->n_constraints, nw = cs->n_wires, npub = cs->n_public;
    size_t n = 1; int lg = 0; while (n < m) { n <<= 1; lg++; }
    fe tau, alpha, beta, gamma, delta, sigma;
    toxic(&tau, seed, "tau"); toxic(&alpha, seed, "alpha"); toxic(&beta, seed, "beta"); toxic(&gamma, seed, "gamma"); toxic(&delta, seed, "delta"); toxic(&sigma, seed, "sigma");
    /* domain */
    fe omega, omega_inv, n_inv, g, g_inv, nfe;
    fr_from_hex(&omega, ROOT_2_28_HEX);
    for (int i = lg; i < 28; i++) fr_sqr(&omega, &omega);
    fr_inv(&omega_inv, &omega); fr_from_u64(&nfe, n); fr_inv(&n_inv, &nfe); fr_from_u64(&g, 5); fr_inv(&g_inv, &g);
    /* Lagrange basis at tau: L_j = (tau^n - 1)/n * w^j / (tau - w^j) */
    fe tn = tau; for (int i = 0; i < lg; i++) fr_sqr(&tn, &tn);
    fe one, zt; fr_set_one(&one); fr_sub(&zt, &tn, &one);           /* Z(tau) = tau^n - 1 */
    fe *den = (fe *)malloc(sizeof(fe) * n), *pre = (fe *)malloc(sizeof(fe) * n), *L = (fe *)malloc(sizeof(fe) * n), *wj = (fe *)malloc(sizeof(fe) * n);
    fe w = one;
    for (size_t j = 0; j < n; j++) { wj[j] = w; fr_sub(&den[j], &tau, &w); fr_mul(&w, &w, &omega); }
    fe run = one; for (size_t j = 0; j < n; j++) { pre[j] = run; fr_mul(&run, &run, &den[j]); }
    fe inv; fr_inv(&inv, &run);
    fe scale; fr_mul(&scale, &zt, &n_inv);
    for (size_t j = n; j-- > 0;) { fe dj; fr_mul(&dj, &inv, &pre[j]); fr_mul(&inv, &inv, &den[j]); fr_mul(&L[j], &dj, &wj[j]); fr_mul(&L[j], &L[j], &scale); }
    free(den); free(pre); free(wj);
    /* A_i(tau), B_i(tau), C_i(tau) */
    fe *A = (fe *)calloc(nw, sizeof(fe)), *B = (fe *)calloc(nw, sizeof(fe)), *C = (fe *)calloc(nw, sizeof(fe));
    for (size_t ii = 0; ii < cs->n_instr; ii++) {
        if (cs->bp_kind[cs->bp[ii]] != BP_R1C) continue;
        const uint32_t *cd = cs->calldata + cs->cstart[ii];
        uint32_t cnt[3] = {cd[1], cd[2], cd[3]}; const uint32_t *t = cd + 4; fe *dst[3] = {A, B, C};
        const fe *Lj = &L[cs->coff[ii]];
        for (int side = 0; side < 3; side++) for (uint32_t k = 0; k < cnt[side]; k++, t += 2) {
            fe v; fr_mul(&v, &cs->coeff[t[0]], Lj);
            uint32_t wid = t[1] == 0xFFFFFFFFu ? 0 : t[1];      /* a constant term multiplies the ONE wire */
            fr_add(&dst[side][wid], &dst[side][wid], &v);
        }
    }
    free(L);
    /* classification of wires */
    uint8_t *committed = (uint8_t *)calloc(nw, 1);
    for (size_t i = 0; i < cs->n_commit_priv; i++) committed[cs->commit_priv[i]] = 1;
    gen_tables gt; build_gen_tables(&gt, 1);
    fe gamma_inv, delta_inv; fr_inv(&gamma_inv, &gamma); fr_inv(&delta_inv, &delta);
    g1aff *pA = (g1aff *)malloc(sizeof(g1aff) * nw), *pB = (g1aff *)malloc(sizeof(g1aff) * nw), *pK = (g1aff *)malloc(sizeof(g1aff) * nw);
    g2aff *pB2 = (g2aff *)malloc(sizeof(g2aff) * nw);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (long i = 0; i < (long)nw; i++) {
        g1_gen_mul(&pA[i], &gt, &A[i]); g1_gen_mul(&pB[i], &gt, &B[i]); g2_gen_mul(&pB2[i], &gt, &B[i]);
        fe k, t; fr_mul(&k, &beta, &A[i]); fr_mul(&t, &alpha, &B[i]); fr_add(&k, &k, &t); fr_add(&k, &k, &C[i]);
        int to_vk = (size_t)i < npub || (cs->n_commit && (uint32_t)i == cs->commit_index) || committed[i];
        fr_mul(&k, &k, to_vk ? &gamma_inv : &delta_inv);
        g1_gen_mul(&pK[i], &gt, &k);
    }
    /* Z[k] = tau^bitrev(k) * Z(tau) / delta, k < n-1 */
    g1aff *pZ = (g1aff *)malloc(sizeof(g1aff) * n);
    fe *tp = (fe *)malloc(sizeof(fe) * n); { fe zd; fr_mul(&zd, &zt, &delta_inv); tp[0] = zd; for (size_t j = 1; j < n; j++) fr_mul(&tp[j], &tp[j - 1], &tau); }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (long k = 0; k < (long)n - 1; k++) {
        size_t br = 0; for (int b = 0; b < lg; b++) if ((size_t)k >> b & 1) br |= (size_t)1 << (lg - 1 - b);
        g1_gen_mul(&pZ[k], &gt, &tp[br]);
    }
    free(tp);
    g1aff a1, b1, d1; g2aff b2, g2, d2; g1_gen_mul(&a1, &gt, &alpha); g1_gen_mul(&b1, &gt, &beta); g1_gen_mul(&d1, &gt, &delta);
    g2_gen_mul(&b2, &gt, &beta); g2_gen_mul(&g2, &gt, &gamma); g2_gen_mul(&d2, &gt, &delta);
    /* ---- pk (App. B.1) ---- */
    obuf pk = {0, 0, 0};
    ob_u64(&pk, n); ob_fr(&pk, &n_inv); ob_fr(&pk, &omega); ob_fr(&pk, &omega_inv); ob_fr(&pk, &g); ob_fr(&pk, &g_inv);
    ob_need(&pk, 1); pk.b[pk.n++] = 1;
    ob_g1(&pk, &a1); ob_g1(&pk, &b1); ob_g1(&pk, &d1);
    size_t nA = 0, nB = 0, nK = 0;
    for (size_t i = 0; i < nw; i++) { nA += !pA[i].inf; nB += !pB[i].inf; }
    for (size_t i = npub; i < nw; i++) if (!committed[i] && !(cs->n_commit && i == cs->commit_index)) nK++;
    ob_u32(&pk, (uint32_t)nA); for (size_t i = 0; i < nw; i++) if (!pA[i].inf) ob_g1(&pk, &pA[i]);
    ob_u32(&pk, (uint32_t)nB); for (size_t i = 0; i < nw; i++) if (!pB[i].inf) ob_g1(&pk, &pB[i]);
    ob_u32(&pk, (uint32_t)(n - 1)); for (size_t k = 0; k + 1 < n; k++) ob_g1(&pk, &pZ[k]);
    ob_u32(&pk, (uint32_t)nK); for (size_t i = npub; i < nw; i++) if (!committed[i] && !(cs->n_commit && i == cs->commit_index)) ob_g1(&pk, &pK[i]);
    ob_g2(&pk, &b2); ob_g2(&pk, &d2);
    ob_u32(&pk, (uint32_t)nB); for (size_t i = 0; i < nw; i++) if (!pB[i].inf) ob_g2(&pk, &pB2[i]);
    ob_u64(&pk, nw); ob_u64(&pk, nw - nA); ob_u64(&pk, nw - nB);
    ob_need(&pk, 2 * nw); for (size_t i = 0; i < nw; i++) pk.b[pk.n++] = (uint8_t)pA[i].inf; for (size_t i = 0; i < nw; i++) pk.b[pk.n++] = (uint8_t)pB[i].inf;
    ob_u32(&pk, (uint32_t)cs->n_commit);
    g2aff pedG, pedGS; memset(&pedG, 0, sizeof pedG); memset(&pedGS, 0, sizeof pedGS);
    if (cs->n_commit) {
        /* Pedersen: Basis_j = K_j/gamma for the committed wires, BasisExpSigma = sigma*Basis; vk: G, -sigma*G */
        ob_u32(&pk, (uint32_t)cs->n_commit_priv); for (size_t j = 0; j < cs->n_commit_priv; j++) ob_g1(&pk, &pK[cs->commit_priv[j]]);
        uint64_t sc[4]; fr_to_canon(sc, &sigma);
        ob_u32(&pk, (uint32_t)cs->n_commit_priv);
        g1aff *bs = (g1aff *)malloc(sizeof(g1aff) * cs->n_commit_priv);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
        for (long j = 0; j < (long)cs->n_commit_priv; j++) { g1jac t; g1_jac_from_aff(&t, &pK[cs->commit_priv[j]]); g1_jac_mul(&t, &t, sc); g1_jac_to_aff(&bs[j], &t); }
        for (size_t j = 0; j < cs->n_commit_priv; j++) ob_g1(&pk, &bs[j]);
        free(bs);
        fe gs; toxic(&gs, seed, "pedersen-g"); g2_gen_mul(&pedG, &gt, &gs);
        fe ns; fr_mul(&ns, &gs, &sigma); fr_neg(&ns, &ns); g2_gen_mul(&pedGS, &gt, &ns);
    }
    /* ---- vk (App. B.2) ---- */
    obuf vk = {0, 0, 0};
    ob_g1(&vk, &a1); ob_g1(&vk, &b1); ob_g2(&vk, &b2); ob_g2(&vk, &g2); ob_g1(&vk, &d1); ob_g2(&vk, &d2);
    ob_u32(&vk, (uint32_t)(npub + (cs->n_commit ? 1 : 0)));
    for (size_t i = 0; i < npub; i++) ob_g1(&vk, &pK[i]);
    if (cs->n_commit) ob_g1(&vk, &pK[cs->commit_index]);
    ob_u32(&vk, (uint32_t)cs->n_commit); for (int c = 0; c < cs->n_commit; c++) ob_u32(&vk, 0);
    ob_u32(&vk, (uint32_t)cs->n_commit); if (cs->n_commit) { ob_g2(&vk, &pedG); ob_g2(&vk, &pedGS); }
    *pk_out = pk.b; *pk_len = pk.n; *vk_out = vk.b; *vk_len = vk.n;
    free(A); free(B); free(C); free(committed); free(pA); free(pB); free(pK); free(pB2); free(pZ); free(gt.t1); free(gt.t2);
    return 0;
}
