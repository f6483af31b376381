/* TEST ORACLE — not product code (see bn254.h).
 *
 * BN254 fields, groups, point codec and optimal-ate pairing, restated from the published
 * definitions.  Replaces, for checking only, what the reference calls
 * in gnark-crypto v0.14.0 ecc/bn254 (go.mod:9): fp/fr arithmetic, G1/G2 group law, MultiExp,
 * point (de)compression (used by groth16 ProvingKey.ReadFrom — libraries/prover/impl/prove_impl.go:86-87)
 * and the pairing behind groth16.Verify (libraries/verifier/impl/verifiers.go:99,145).
 */
#include "bn254.h"
#include "constants.inc"

/* ---- Fp ---- */
#define F(n) fp_##n
#define FMOD0 0x3c208c16d87cfd47ULL
#define FMOD1 0x97816a916871ca8dULL
#define FMOD2 0xb85045b68181585dULL
#define FMOD3 0x30644e72e131a029ULL
#include "field_impl.inc"
#undef F
#undef FMOD0
#undef FMOD1
#undef FMOD2
#undef FMOD3
/* ---- Fr ---- */
#define F(n) fr_##n
#define FMOD0 0x43e1f593f0000001ULL
#define FMOD1 0x2833e84879b97091ULL
#define FMOD2 0xb85045b68181585dULL
#define FMOD3 0x30644e72e131a029ULL
#include "field_impl.inc"
#undef F
#undef FMOD0
#undef FMOD1
#undef FMOD2
#undef FMOD3

const uint64_t FR_MOD_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

/* ---- Fp2 = Fp[u]/(u^2+1) ---- */
void fp2_add(fe2 *r, const fe2 *a, const fe2 *b) { fp_add(&r->a0, &a->a0, &b->a0); fp_add(&r->a1, &a->a1, &b->a1); }
void fp2_sub(fe2 *r, const fe2 *a, const fe2 *b) { fp_sub(&r->a0, &a->a0, &b->a0); fp_sub(&r->a1, &a->a1, &b->a1); }
void fp2_dbl(fe2 *r, const fe2 *a) { fp2_add(r, a, a); }
void fp2_neg(fe2 *r, const fe2 *a) { fp_neg(&r->a0, &a->a0); fp_neg(&r->a1, &a->a1); }
void fp2_conj(fe2 *r, const fe2 *a) { r->a0 = a->a0; fp_neg(&r->a1, &a->a1); }
void fp2_mul(fe2 *r, const fe2 *a, const fe2 *b) {
    fe t0, t1, t2, t3;
    fp_mul(&t0, &a->a0, &b->a0); fp_mul(&t1, &a->a1, &b->a1);
    fp_add(&t2, &a->a0, &a->a1); fp_add(&t3, &b->a0, &b->a1); fp_mul(&t2, &t2, &t3);
    fp_sub(&t2, &t2, &t0); fp_sub(&t2, &t2, &t1);
    fp_sub(&r->a0, &t0, &t1); r->a1 = t2;
}
void fp2_sqr(fe2 *r, const fe2 *a) { fp2_mul(r, a, a); }
void fp2_mul_fp(fe2 *r, const fe2 *a, const fe *b) { fp_mul(&r->a0, &a->a0, b); fp_mul(&r->a1, &a->a1, b); }
void fp2_inv(fe2 *r, const fe2 *a) {
    fe n, t; fp_sqr(&n, &a->a0); fp_sqr(&t, &a->a1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->a0, &a->a0, &n); fp_mul(&t, &a->a1, &n); fp_neg(&r->a1, &t);
}
int fp2_is_zero(const fe2 *a) { return fp_is_zero(&a->a0) && fp_is_zero(&a->a1); }
int fp2_eq(const fe2 *a, const fe2 *b) { return fp_eq(&a->a0, &b->a0) && fp_eq(&a->a1, &b->a1); }
void fp2_set_zero(fe2 *r) { fp_set_zero(&r->a0); fp_set_zero(&r->a1); }
void fp2_set_one(fe2 *r) { fp_set_one(&r->a0); fp_set_zero(&r->a1); }
void fp2_pow(fe2 *r, const fe2 *a, const uint64_t *e, int nlimbs) {
    fe2 acc, base = *a; fp2_set_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fp2_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp2_mul(&acc, &acc, &base);
    }
    *r = acc;
}
int fp_sqrt(fe *r, const fe *a) {
    fe s, c; fp_pow(&s, a, FP_SQRT_EXP, 4); fp_sqr(&c, &s);
    if (!fp_eq(&c, a)) return 0;
    *r = s; return 1;
}
/* norm method: a = (x0 + x1 u)^2  =>  x0^2 = (a0 +- sqrt(a0^2+a1^2))/2, x1 = a1/(2 x0) */
int fp2_sqrt(fe2 *r, const fe2 *a) {
    fe2 x;
    if (fp_is_zero(&a->a1)) {
        if (fp_sqrt(&x.a0, &a->a0)) { fp_set_zero(&x.a1); }
        else { fe na; fp_neg(&na, &a->a0); if (!fp_sqrt(&x.a1, &na)) return 0; fp_set_zero(&x.a0); }
    } else {
        fe n, t, s, two, half, x0;
        fp_sqr(&n, &a->a0); fp_sqr(&t, &a->a1); fp_add(&n, &n, &t);
        if (!fp_sqrt(&s, &n)) return 0;
        fp_from_u64(&two, 2); fp_inv(&half, &two);
        fp_add(&t, &a->a0, &s); fp_mul(&t, &t, &half);
        if (!fp_sqrt(&x0, &t)) {
            fp_sub(&t, &a->a0, &s); fp_mul(&t, &t, &half);
            if (!fp_sqrt(&x0, &t)) return 0;
        }
        fe d; fp_dbl(&d, &x0); fp_inv(&d, &d);
        x.a0 = x0; fp_mul(&x.a1, &a->a1, &d);
    }
    fe2 c; fp2_sqr(&c, &x);
    if (!fp2_eq(&c, a)) return 0;
    *r = x; return 1;
}
int fp2_lex_large(const fe2 *a) { return fp_is_zero(&a->a1) ? fp_lex_large(&a->a0) : fp_lex_large(&a->a1); }
static void fp2_inv_wrap(fe2 *r, const fe2 *a) { fp2_inv(r, a); }

/* ---- groups ---- */
#define KE fe
#define K(n) fp_##n
#define G(n) g1_##n
#define GAFF g1aff
#define GJAC g1jac
#include "curve_body.inc"
#undef KE
#undef K
#undef G
#undef GAFF
#undef GJAC
#define KE fe2
#define K(n) fp2_##n
#define G(n) g2_##n
#define GAFF g2aff
#define GJAC g2jac
#include "curve_body.inc"
#undef KE
#undef K
#undef G
#undef GAFF
#undef GJAC

fe G1_B; fe2 G2_B; g1aff G1_GEN; g2aff G2_GEN;
static fe2 XI, GAMMA2, GAMMA3;   /* xi = 9+u, xi^((p-1)/3), xi^((p-1)/2) */
static int inited = 0;

static void fp_from_hex(fe *r, const char *hex) {
    uint8_t b[32]; memset(b, 0, 32);
    size_t n = strlen(hex);
    for (size_t i = 0; i < n; i++) {
        char ch = hex[n - 1 - i]; int v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        b[31 - i / 2] |= (uint8_t)(v << (4 * (i & 1)));
    }
    fp_from_be(r, b);
}

void bn254_init(void) {
    if (inited) return;
    fp_init(); fr_init();
    fp_from_u64(&G1_B, 3);
    fp_from_u64(&XI.a0, 9); fp_from_u64(&XI.a1, 1);
    fe2 b3, xinv; fp_from_u64(&b3.a0, 3); fp_set_zero(&b3.a1);
    fp2_inv(&xinv, &XI); fp2_mul(&G2_B, &b3, &xinv);
    fp_from_u64(&G1_GEN.x, 1); fp_from_u64(&G1_GEN.y, 2); G1_GEN.inf = 0;
    /* standard alt_bn128 G2 generator */
    fp_from_hex(&G2_GEN.x.a0, "1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed");
    fp_from_hex(&G2_GEN.x.a1, "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2");
    fp_from_hex(&G2_GEN.y.a0, "12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa");
    fp_from_hex(&G2_GEN.y.a1, "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b");
    G2_GEN.inf = 0;
    fp2_pow(&GAMMA2, &XI, FP_P_MINUS1_DIV3, 4);
    fp2_pow(&GAMMA3, &XI, FP_P_MINUS1_DIV2, 4);
    inited = 1;
}

/* ---- point codec: big-endian, top two bits of byte 0 are flags ---- */
int g1_decode(g1aff *p, const uint8_t *b, size_t avail) {
    if (avail < 32) return -1;
    uint8_t flag = b[0] & 0xC0;
    uint8_t xb[32]; memcpy(xb, b, 32); xb[0] &= 0x3F;
    if (flag == 0x40) {   /* infinity: every other bit must be zero (gnark-crypto rejects anything else) */
        for (int i = 0; i < 32; i++) if (xb[i]) return -1;
        fp_set_zero(&p->x); fp_set_zero(&p->y); p->inf = 1; return 32;
    }
    if (!fp_from_be(&p->x, xb)) return -1;
    p->inf = 0;
    if (flag == 0x00) {
        if (avail < 64) return -1;
        if (!fp_from_be(&p->y, b + 32)) return -1;
        return g1_aff_on_curve(p, &G1_B) ? 64 : -1;
    }
    fe rhs, y; fp_sqr(&rhs, &p->x); fp_mul(&rhs, &rhs, &p->x); fp_add(&rhs, &rhs, &G1_B);
    if (!fp_sqrt(&y, &rhs)) return -1;
    int large = fp_lex_large(&y);
    if ((flag == 0xC0) != (large != 0)) fp_neg(&y, &y);
    p->y = y;
    return 32;
}
int g2_decode(g2aff *p, const uint8_t *b, size_t avail) {
    if (avail < 64) return -1;
    uint8_t flag = b[0] & 0xC0;
    uint8_t xb[32]; memcpy(xb, b, 32); xb[0] &= 0x3F;
    if (flag == 0x40) {
        for (int i = 0; i < 32; i++) if (xb[i] || b[32 + i]) return -1;
        fp2_set_zero(&p->x); fp2_set_zero(&p->y); p->inf = 1; return 64;
    }
    if (!fp_from_be(&p->x.a1, xb) || !fp_from_be(&p->x.a0, b + 32)) return -1;
    p->inf = 0;
    if (flag == 0x00) {
        if (avail < 128) return -1;
        if (!fp_from_be(&p->y.a1, b + 64) || !fp_from_be(&p->y.a0, b + 96)) return -1;
        return g2_aff_on_curve(p, &G2_B) ? 128 : -1;
    }
    fe2 rhs, y; fp2_sqr(&rhs, &p->x); fp2_mul(&rhs, &rhs, &p->x); fp2_add(&rhs, &rhs, &G2_B);
    if (!fp2_sqrt(&y, &rhs)) return -1;
    int large = fp2_lex_large(&y);
    if ((flag == 0xC0) != (large != 0)) fp2_neg(&y, &y);
    p->y = y;
    return 64;
}
void g1_encode_compressed(uint8_t out[32], const g1aff *p) {
    if (p->inf) { memset(out, 0, 32); out[0] = 0x40; return; }
    fp_to_be(out, &p->x);
    out[0] |= fp_lex_large(&p->y) ? 0xC0 : 0x80;
}
void g2_encode_compressed(uint8_t out[64], const g2aff *p) {
    if (p->inf) { memset(out, 0, 64); out[0] = 0x40; return; }
    fp_to_be(out, &p->x.a1); fp_to_be(out + 32, &p->x.a0);
    out[0] |= fp2_lex_large(&p->y) ? 0xC0 : 0x80;
}
void g1_encode_uncompressed(uint8_t out[64], const g1aff *p) {
    if (p->inf) { memset(out, 0, 64); out[0] = 0x40; return; }
    fp_to_be(out, &p->x); fp_to_be(out + 32, &p->y);
}

/* ---- Fp12 = Fp2[w]/(w^6 - xi), flat representation ---- */
void fp12_set_one(fe12 *r) { for (int i = 0; i < 6; i++) fp2_set_zero(&r->c[i]); fp2_set_one(&r->c[0]); }
int fp12_is_one(const fe12 *a) {
    fe2 one; fp2_set_one(&one);
    if (!fp2_eq(&a->c[0], &one)) return 0;
    for (int i = 1; i < 6; i++) if (!fp2_is_zero(&a->c[i])) return 0;
    return 1;
}
void fp12_mul(fe12 *r, const fe12 *a, const fe12 *b) {
    fe2 t[11]; for (int i = 0; i < 11; i++) fp2_set_zero(&t[i]);
    for (int i = 0; i < 6; i++) {
        if (fp2_is_zero(&a->c[i])) continue;
        for (int j = 0; j < 6; j++) {
            if (fp2_is_zero(&b->c[j])) continue;
            fe2 m; fp2_mul(&m, &a->c[i], &b->c[j]); fp2_add(&t[i + j], &t[i + j], &m);
        }
    }
    fe12 o;
    for (int k = 0; k < 6; k++) {
        o.c[k] = t[k];
        if (k + 6 < 11) { fe2 m; fp2_mul(&m, &t[k + 6], &XI); fp2_add(&o.c[k], &o.c[k], &m); }
    }
    *r = o;
}

/* affine arithmetic on the twist E'(Fp2): y^2 = x^3 + 3/xi, with the line through psi(T),psi(Q)
 * evaluated at P in G1: l = yP - lambda*xP*w + (lambda*xT - yT)*w^3   (untwist psi(x,y) = (x w^2, y w^3)) */
typedef struct { fe2 x, y; } twpt;
static void line_eval(fe12 *l, const fe2 *lambda, const twpt *T, const g1aff *P) {
    for (int i = 0; i < 6; i++) fp2_set_zero(&l->c[i]);
    l->c[0].a0 = P->y;
    fe2 t; fp2_mul_fp(&t, lambda, &P->x); fp2_neg(&l->c[1], &t);
    fp2_mul(&t, lambda, &T->x); fp2_sub(&l->c[3], &t, &T->y);
}
static void tw_dbl_line(fe12 *l, twpt *T, const g1aff *P) {
    fe2 lam, t, x3, y3;
    fp2_sqr(&t, &T->x); fp2_dbl(&lam, &t); fp2_add(&lam, &lam, &t);
    fp2_dbl(&t, &T->y); fp2_inv_wrap(&t, &t); fp2_mul(&lam, &lam, &t);
    line_eval(l, &lam, T, P);
    fp2_sqr(&x3, &lam); fp2_sub(&x3, &x3, &T->x); fp2_sub(&x3, &x3, &T->x);
    fp2_sub(&t, &T->x, &x3); fp2_mul(&y3, &lam, &t); fp2_sub(&y3, &y3, &T->y);
    T->x = x3; T->y = y3;
}
static void tw_add_line(fe12 *l, twpt *T, const twpt *Q, const g1aff *P) {
    fe2 lam, t, x3, y3;
    fp2_sub(&lam, &Q->y, &T->y); fp2_sub(&t, &Q->x, &T->x); fp2_inv_wrap(&t, &t); fp2_mul(&lam, &lam, &t);
    line_eval(l, &lam, T, P);
    fp2_sqr(&x3, &lam); fp2_sub(&x3, &x3, &T->x); fp2_sub(&x3, &x3, &Q->x);
    fp2_sub(&t, &T->x, &x3); fp2_mul(&y3, &lam, &t); fp2_sub(&y3, &y3, &T->y);
    T->x = x3; T->y = y3;
}
static void tw_frob(twpt *r, const twpt *q) {
    fe2 t; fp2_conj(&t, &q->x); fp2_mul(&r->x, &t, &GAMMA2);
    fp2_conj(&t, &q->y); fp2_mul(&r->y, &t, &GAMMA3);
}
void miller_loop(fe12 *f, const g1aff *P, const g2aff *Q) {
    fp12_set_one(f);
    if (P->inf || Q->inf) return;
    twpt T = {Q->x, Q->y}, Qa = {Q->x, Q->y};
    fe12 l;
    for (int i = 63; i >= 0; i--) {           /* 6x+2 has 65 bits; top bit consumed by T = Q */
        fp12_mul(f, f, f);
        tw_dbl_line(&l, &T, P); fp12_mul(f, f, &l);
        if ((ATE_LOOP[i / 64] >> (i % 64)) & 1) { tw_add_line(&l, &T, &Qa, P); fp12_mul(f, f, &l); }
    }
    twpt Q1, Q2; tw_frob(&Q1, &Qa); tw_frob(&Q2, &Q1); fp2_neg(&Q2.y, &Q2.y);
    tw_add_line(&l, &T, &Q1, P); fp12_mul(f, f, &l);
    tw_add_line(&l, &T, &Q2, P); fp12_mul(f, f, &l);
}
void final_exp(fe12 *r, const fe12 *f) {
    fe12 acc, base = *f; fp12_set_one(&acc);
    int top = 44 * 64 - 1;
    while (!((FINAL_EXP[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        fp12_mul(&acc, &acc, &acc);
        if ((FINAL_EXP[i / 64] >> (i % 64)) & 1) fp12_mul(&acc, &acc, &base);
    }
    *r = acc;
}
int pairing_product_is_one(const g1aff *P, const g2aff *Q, int n) {
    fe12 acc, f; fp12_set_one(&acc);
    for (int i = 0; i < n; i++) { miller_loop(&f, &P[i], &Q[i]); fp12_mul(&acc, &acc, &f); }
    final_exp(&acc, &acc);
    return fp12_is_one(&acc);
}
