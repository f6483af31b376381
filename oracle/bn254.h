/* TEST ORACLE — not product code.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call anything under oracle/.
 *
 * CPU restatement of the BN254 arithmetic the reference obtains from the un-vendored Go module
 * github.com/consensys/gnark-crypto v0.14.0 (reference go.mod:9).  Parity is pinned by the
 * reference-authored artefacts under tests/golden (pk.chacha20, r1cs.*, vk.*) through the
 * pairing check and the known-answer vectors of SURVEY.md App. E — see oracle/README.md.
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

typedef struct { uint64_t l[4]; } fe;      /* Montgomery form, little-endian limbs */
typedef struct { fe a0, a1; } fe2;         /* a0 + a1*u, u^2 = -1 */
typedef struct { fe2 c[6]; } fe12;         /* sum c[i] w^i, w^6 = 9 + u */

void bn254_init(void);

#define DECL_FIELD(P) \
    void P##add(fe *, const fe *, const fe *); void P##sub(fe *, const fe *, const fe *); \
    void P##dbl(fe *, const fe *); void P##neg(fe *, const fe *); \
    void P##mul(fe *, const fe *, const fe *); void P##sqr(fe *, const fe *); \
    void P##inv(fe *, const fe *); void P##pow(fe *, const fe *, const uint64_t *, int); \
    int P##is_zero(const fe *); int P##eq(const fe *, const fe *); \
    void P##set_zero(fe *); void P##set_one(fe *); \
    void P##from_canon(fe *, const uint64_t[4]); void P##to_canon(uint64_t[4], const fe *); \
    void P##from_u64(fe *, uint64_t); int P##from_be(fe *, const uint8_t *); \
    void P##to_be(uint8_t *, const fe *); int P##lex_large(const fe *);
DECL_FIELD(fp_)
DECL_FIELD(fr_)

void fp2_add(fe2 *, const fe2 *, const fe2 *); void fp2_sub(fe2 *, const fe2 *, const fe2 *);
void fp2_dbl(fe2 *, const fe2 *); void fp2_neg(fe2 *, const fe2 *);
void fp2_mul(fe2 *, const fe2 *, const fe2 *); void fp2_sqr(fe2 *, const fe2 *);
void fp2_inv(fe2 *, const fe2 *); void fp2_conj(fe2 *, const fe2 *);
int fp2_is_zero(const fe2 *); int fp2_eq(const fe2 *, const fe2 *);
void fp2_set_zero(fe2 *); void fp2_set_one(fe2 *);
void fp2_mul_fp(fe2 *, const fe2 *, const fe *);
void fp2_pow(fe2 *, const fe2 *, const uint64_t *, int);
int fp2_sqrt(fe2 *, const fe2 *);
int fp_sqrt(fe *, const fe *);
int fp2_lex_large(const fe2 *);

typedef struct { fe x, y; int inf; } g1aff;
typedef struct { fe x, y, z; } g1jac;
typedef struct { fe2 x, y; int inf; } g2aff;
typedef struct { fe2 x, y, z; } g2jac;

#define DECL_GROUP(P, AFF, JAC, KE) \
    void P##jac_set_inf(JAC *); int P##jac_is_inf(const JAC *); void P##jac_from_aff(JAC *, const AFF *); \
    void P##jac_dbl(JAC *, const JAC *); void P##jac_add(JAC *, const JAC *, const JAC *); \
    void P##jac_add_aff(JAC *, const JAC *, const AFF *); void P##jac_neg(JAC *, const JAC *); \
    void P##aff_neg(AFF *, const AFF *); void P##jac_to_aff(AFF *, const JAC *); \
    void P##jac_mul(JAC *, const JAC *, const uint64_t[4]); int P##aff_on_curve(const AFF *, const KE *); \
    void P##msm(JAC *, const AFF *, const uint64_t *, size_t);
DECL_GROUP(g1_, g1aff, g1jac, fe)
DECL_GROUP(g2_, g2aff, g2jac, fe2)

extern fe G1_B;     /* 3 */
extern fe2 G2_B;    /* 3/(9+u) */
extern g1aff G1_GEN;
extern g2aff G2_GEN;
extern const uint64_t FR_MOD_LIMBS[4];

/* gnark-crypto point encoding.  Return bytes consumed, or -1. */
int g1_decode(g1aff *p, const uint8_t *b, size_t avail);     /* 32 (compressed/inf) or 64 (uncompressed) */
int g2_decode(g2aff *p, const uint8_t *b, size_t avail);     /* 64 or 128 */
void g1_encode_compressed(uint8_t out[32], const g1aff *p);
void g2_encode_compressed(uint8_t out[64], const g2aff *p);
void g1_encode_uncompressed(uint8_t out[64], const g1aff *p);

/* pairing */
void fp12_set_one(fe12 *); void fp12_mul(fe12 *, const fe12 *, const fe12 *);
int fp12_is_one(const fe12 *);
void miller_loop(fe12 *f, const g1aff *P, const g2aff *Q);
void final_exp(fe12 *r, const fe12 *f);
/* prod e(P_i,Q_i) == 1 ? */
int pairing_product_is_one(const g1aff *P, const g2aff *Q, int n);

#endif
