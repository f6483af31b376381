/* TEST ORACLE — not product code (see bn254.h).  Flat C entry points for tests/ (ctypes),
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg. */
#include "groth16.h"
#include "ciphers.h"
#include <stdio.h>

#define EXPORT __attribute__((visibility("default")))

EXPORT void orc_init(void) { bn254_init(); }
EXPORT void *orc_r1cs_new(const uint8_t *b, size_t n) {
    bn254_init(); r1cs_t *cs = (r1cs_t *)malloc(sizeof *cs);
    int rc = r1cs_parse(cs, b, n);
    if (rc) { fprintf(stderr, "oracle: r1cs_parse rc=%d\n", rc); free(cs); return NULL; }
    return cs;
}
EXPORT void orc_r1cs_free(void *p) { if (p) { r1cs_free((r1cs_t *)p); free(p); } }
/* what: 0 wires 1 constraints 2 public(incl ONE) 3 secret 4 instructions 5 levels 6 calldata 7 coeffs 8 commitments 9 committed 10 commit wire */
EXPORT size_t orc_r1cs_info(const void *p, int what) {
    const r1cs_t *c = (const r1cs_t *)p;
    switch (what) { case 0: return c->n_wires; case 1: return c->n_constraints; case 2: return c->n_public; case 3: return c->n_secret;
        case 4: return c->n_instr; case 5: return c->n_levels; case 6: return c->n_calldata; case 7: return c->n_coeff;
        case 8: return (size_t)c->n_commit; case 9: return c->n_commit_priv; case 10: return c->commit_index; }
    return 0;
}
EXPORT int orc_r1cs_levels_are_permutation(const void *p) {
    const r1cs_t *c = (const r1cs_t *)p; uint8_t *seen = (uint8_t *)calloc(c->n_instr, 1); int ok = 1;
    for (size_t i = 0; i < c->n_instr; i++) { uint32_t v = c->level_instr[i]; if (v >= c->n_instr || seen[v]) { ok = 0; break; } seen[v] = 1; }
    free(seen); return ok;
}
EXPORT void *orc_pk_new(const uint8_t *b, size_t n) {
    bn254_init(); pk_t *pk = (pk_t *)malloc(sizeof *pk);
    int rc = pk_parse(pk, b, n);
    if (rc) { fprintf(stderr, "oracle: pk_parse rc=%d\n", rc); free(pk); return NULL; }
    return pk;
}
EXPORT void orc_pk_free(void *p) { if (p) { pk_free((pk_t *)p); free(p); } }
/* what: 0 n 1 nA 2 nB 3 nZ 4 nK 5 nB2 6 n_wires 7 n_ck 8 n_basis */
EXPORT size_t orc_pk_info(const void *p, int what) {
    const pk_t *k = (const pk_t *)p;
    switch (what) { case 0: return k->n; case 1: return k->nA; case 2: return k->nB; case 3: return k->nZ; case 4: return k->nK;
        case 5: return k->nB2; case 6: return k->n_wires; case 7: return (size_t)k->n_ck; case 8: return k->n_basis; }
    return 0;
}
EXPORT void *orc_vk_new(const uint8_t *b, size_t n) {
    bn254_init(); vk_t *vk = (vk_t *)malloc(sizeof *vk);
    int rc = vk_parse(vk, b, n);
    if (rc) { fprintf(stderr, "oracle: vk_parse rc=%d\n", rc); free(vk); return NULL; }
    return vk;
}
EXPORT void orc_vk_free(void *p) { if (p) { vk_free((vk_t *)p); free(p); } }
EXPORT size_t orc_vk_nk(const void *p) { return ((const vk_t *)p)->nK; }

EXPORT void orc_chacha20_xor(const uint8_t *key, const uint8_t *nonce, uint32_t counter, const uint8_t *in, uint8_t *out, size_t len) { chacha20_xor(key, nonce, counter, in, out, len); }
EXPORT void orc_chacha20_block(const uint8_t *key, const uint8_t *nonce, uint32_t counter, uint8_t *out) { chacha20_block(key, nonce, counter, out); }
EXPORT void orc_aes_ctr_xor(const uint8_t *key, int keylen, const uint8_t *nonce, uint32_t counter, const uint8_t *in, uint8_t *out, size_t len) { aes_ctr_xor(key, keylen, nonce, counter, in, out, len); }
EXPORT void orc_aes_encrypt_block(const uint8_t *key, int keylen, const uint8_t *in, uint8_t *out) { aes_encrypt_block(key, keylen, in, out); }
EXPORT void orc_sha256(const uint8_t *m, size_t n, uint8_t *out) { sha256(m, n, out); }
EXPORT int orc_expand_message_xmd(const uint8_t *m, size_t n, const uint8_t *dst, size_t dn, uint8_t *out, size_t on) { return expand_message_xmd(m, n, dst, dn, out, on); }

static int fe_from_be_or_zero(fe *o, const uint8_t *b) { if (!b) { fr_set_zero(o); return 1; } return fr_from_be(o, b); }

/* cipher: 0 chacha20, 1 aes-128-ctr, 2 aes-256-ctr.  r,s,mask: 32-byte BE canonical (NULL = 0).
 * dumps may be NULL.  Returns 0 on success. */
EXPORT int orc_prove(const void *csp, const void *pkp, int cipher, const uint8_t *key, const uint8_t *nonce, uint32_t counter,
                     const uint8_t *pt, const uint8_t *r_be, const uint8_t *s_be, const uint8_t *mask_be,
                     uint8_t *proof_out, size_t *proof_len, uint8_t *ct_out,
                     uint8_t *dW, uint8_t *dA, uint8_t *dB, uint8_t *dC, uint8_t *dh) {
    const r1cs_t *cs = (const r1cs_t *)csp; const pk_t *pk = (const pk_t *)pkp;
    size_t nin = cs->n_public - 1 + cs->n_secret;
    fe *wit = (fe *)malloc(sizeof(fe) * nin);
    int keylen = cipher == 0 ? 32 : cipher == 1 ? 16 : 32;
    if (cipher == 0) { if (nin != 1408) { free(wit); return -10; } assign_chacha(key, nonce, counter, pt, ct_out, wit); }
    else { if (nin != (size_t)(141 + keylen)) { free(wit); return -10; } assign_aes(key, keylen, nonce, counter, pt, ct_out, wit); }
    fe r, s, mask;
    if (!fe_from_be_or_zero(&r, r_be) || !fe_from_be_or_zero(&s, s_be) || !fe_from_be_or_zero(&mask, mask_be)) { free(wit); return -11; }
    prove_dump_t d = {dW, dA, dB, dC, dh};
    int rc = groth16_prove(cs, pk, wit, &r, &s, &mask, proof_out, proof_len, &d);
    free(wit); return rc;
}
/* witness generation only (no proving key needed): W/A/B/C dumps, BE canonical.  commit_out_be: value forced
 * onto the commitment wire (NULL = 0) — any value satisfies the R1CS. */
typedef struct { fe v; } fixed_commit;
static int fixed_commit_cb(void *ctx, const fe *vals, size_t n, fe *out) { (void)vals; (void)n; *out = ((fixed_commit *)ctx)->v; return 0; }
EXPORT long orc_solve(const void *csp, int cipher, const uint8_t *key, const uint8_t *nonce, uint32_t counter, const uint8_t *pt,
                      const uint8_t *mask_be, const uint8_t *commit_be, uint8_t *ct_out, uint8_t *dW, uint8_t *dA, uint8_t *dB, uint8_t *dC) {
    const r1cs_t *cs = (const r1cs_t *)csp;
    size_t nin = cs->n_public - 1 + cs->n_secret;
    fe *wit = (fe *)malloc(sizeof(fe) * nin);
    int keylen = cipher == 0 ? 32 : cipher == 1 ? 16 : 32;
    if (cipher == 0) assign_chacha(key, nonce, counter, pt, ct_out, wit); else assign_aes(key, keylen, nonce, counter, pt, ct_out, wit);
    fe mask; fixed_commit fc; fe_from_be_or_zero(&mask, mask_be); fe_from_be_or_zero(&fc.v, commit_be);
    solve_opts_t so = {&mask, fixed_commit_cb, &fc};
    fe *W = (fe *)malloc(sizeof(fe) * cs->n_wires), *A = (fe *)malloc(sizeof(fe) * cs->n_constraints), *B = (fe *)malloc(sizeof(fe) * cs->n_constraints), *C = (fe *)malloc(sizeof(fe) * cs->n_constraints);
    long rc = r1cs_solve(cs, wit, W, A, B, C, &so);
    if (!rc) {
        for (size_t i = 0; i < cs->n_constraints; i++) { fe ab; fr_mul(&ab, &A[i], &B[i]); if (!fr_eq(&ab, &C[i])) { rc = -(long)i - 2; break; } }
        if (dW) for (size_t i = 0; i < cs->n_wires; i++) fr_to_be(dW + 32 * i, &W[i]);
        if (dA) for (size_t i = 0; i < cs->n_constraints; i++) { fr_to_be(dA + 32 * i, &A[i]); fr_to_be(dB + 32 * i, &B[i]); fr_to_be(dC + 32 * i, &C[i]); }
    }
    free(wit); free(W); free(A); free(B); free(C);
    return rc;
}
/* public-input reconstruction as libraries/verifier/impl/verifiers.go:50-104 (ChaCha) / :110-152 (AES):
 * publicSignals = ct(64) | nonce(12) | counter(4; LE for ChaCha, BE for AES) | pt(64). */
EXPORT int orc_verify(const void *vkp, int cipher, const uint8_t *proof, size_t proof_len, const uint8_t *signals, size_t nsig) {
    const vk_t *vk = (const vk_t *)vkp;
    if (nsig != 144) return 0;
    const uint8_t *ct = signals, *nonce = signals + 64, *ctr = signals + 76, *pt = signals + 80;
    fe pub[1152]; size_t np;
    if (cipher == 0) {
        fe wit[1408]; uint8_t dummy_ct[64], zero_key[32] = {0};
        uint32_t counter = (uint32_t)ctr[0] | ((uint32_t)ctr[1] << 8) | ((uint32_t)ctr[2] << 16) | ((uint32_t)ctr[3] << 24);
        assign_chacha(zero_key, nonce, counter, pt, dummy_ct, wit);
        memcpy(pub, wit, sizeof(fe) * 1152);
        /* Out = the caller's ciphertext, not a recomputed one */
        for (int i = 0; i < 16; i++) { uint32_t v = ((uint32_t)ct[4 * i] << 24) | ((uint32_t)ct[4 * i + 1] << 16) | ((uint32_t)ct[4 * i + 2] << 8) | ct[4 * i + 3]; for (int j = 0; j < 32; j++) fr_from_u64(&pub[32 + 96 + 512 + 32 * i + j], (v >> j) & 1); }
        np = 1152;
    } else {
        uint32_t counter = ((uint32_t)ctr[0] << 24) | ((uint32_t)ctr[1] << 16) | ((uint32_t)ctr[2] << 8) | ctr[3];
        fe *p = pub;
        for (int i = 0; i < 12; i++) fr_from_u64(p++, nonce[i]);
        fr_from_u64(p++, counter);
        for (int i = 0; i < 64; i++) fr_from_u64(p++, pt[i]);
        for (int i = 0; i < 64; i++) fr_from_u64(p++, ct[i]);
        np = 141;
    }
    return groth16_verify(vk, proof, proof_len, pub, np);
}
/* computeH alone on caller-supplied vectors (canonical big-endian, m <= domain size); h_be gets n elements, natural order.
 * The vectors need not satisfy a*b = c: the pipeline of App. D is defined for any input. */
EXPORT int orc_compute_h(const void *pkp, const uint8_t *a_be, const uint8_t *b_be, const uint8_t *c_be, size_t m, uint8_t *h_be) {
    const pk_t *pk = (const pk_t *)pkp;
    if (m > pk->n) return -1;
    fe *A = (fe *)malloc(sizeof(fe) * (m ? m : 1)), *B = (fe *)malloc(sizeof(fe) * (m ? m : 1)), *C = (fe *)malloc(sizeof(fe) * (m ? m : 1)), *h = (fe *)malloc(sizeof(fe) * pk->n);
    int ok = 1;
    for (size_t i = 0; i < m; i++) ok &= fr_from_be(&A[i], a_be + 32 * i) & fr_from_be(&B[i], b_be + 32 * i) & fr_from_be(&C[i], c_be + 32 * i);
    if (ok) { compute_h(pk, A, B, C, m, h); for (size_t i = 0; i < pk->n; i++) fr_to_be(h_be + 32 * i, &h[i]); }
    free(A); free(B); free(C); free(h);
    return ok ? 0 : -2;
}

/* Setup with a seed; returns malloc'd key files (free with orc_free) */
EXPORT int orc_setup(const void *csp, const uint8_t *seed32, uint8_t **pk, size_t *pk_len, uint8_t **vk, size_t *vk_len) {
    return groth16_setup((const r1cs_t *)csp, seed32, pk, pk_len, vk, vk_len);
}
EXPORT void orc_free(void *p) { free(p); }
/* self-tests of the pairing: bilinearity on the generators */
EXPORT int orc_pairing_selftest(void) {
    bn254_init();
    uint64_t a[4] = {0x1234567, 0, 0, 0}, b[4] = {0xabcdef01, 7, 0, 0};
    g1jac g1, ag1; g2jac g2, bg2; g1aff P[2]; g2aff Q[2];
    g1_jac_from_aff(&g1, &G1_GEN); g2_jac_from_aff(&g2, &G2_GEN);
    if (!g1_aff_on_curve(&G1_GEN, &G1_B) || !g2_aff_on_curve(&G2_GEN, &G2_B)) return -1;
    /* e(a*G1, b*G2) * e(-(ab)*G1, G2) == 1 */
    fe fa, fb, fab; fr_from_canon(&fa, a); fr_from_canon(&fb, b); fr_mul(&fab, &fa, &fb);
    uint64_t ab[4]; fr_to_canon(ab, &fab);
    g1_jac_mul(&ag1, &g1, a); g2_jac_mul(&bg2, &g2, b);
    g1_jac_to_aff(&P[0], &ag1); g2_jac_to_aff(&Q[0], &bg2);
    g1jac abg1; g1_jac_mul(&abg1, &g1, ab); g1_jac_neg(&abg1, &abg1); g1_jac_to_aff(&P[1], &abg1); Q[1] = G2_GEN;
    if (!pairing_product_is_one(P, Q, 2)) return -2;
    /* non-degeneracy: e(G1,G2) != 1 */
    if (pairing_product_is_one(&G1_GEN, &G2_GEN, 1)) return -3;
    /* r*G2 = inf (subgroup) */
    g2jac rg2; g2_jac_mul(&rg2, &g2, FR_MOD_LIMBS); if (!g2_jac_is_inf(&rg2)) return -4;
    return 0;
}
/* field constants for cross-checks against SURVEY.md App. I: which: 0 R mod p, 1 R^2 mod p, 2 R mod r, 3 R^2 mod r (32B BE) */
EXPORT void orc_field_const(int which, uint8_t out[32]) {
    bn254_init(); fe v; uint64_t one[4] = {1, 0, 0, 0};
    /* from_canon(1) = R; from_canon(R as canonical) = R^2 */
    if (which == 0 || which == 1) { fp_from_canon(&v, one); if (which == 1) { uint64_t c[4]; memcpy(c, v.l, 32); fp_from_canon(&v, c); } }
    else { fr_from_canon(&v, one); if (which == 3) { uint64_t c[4]; memcpy(c, v.l, 32); fr_from_canon(&v, c); } }
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) out[(3 - i) * 8 + k] = (uint8_t)(v.l[i] >> (56 - 8 * k));
}
