/* TEST ORACLE — not product code (see bn254.h). */
#ifndef ORACLE_GROTH16_H
#define ORACLE_GROTH16_H
#include "bn254.h"
#include "r1cs.h"

typedef struct {
    uint64_t n; fe n_inv, omega, omega_inv, g, g_inv;
    g1aff alpha, beta, delta; g1aff *A, *B, *Z, *K; size_t nA, nB, nZ, nK;
    g2aff beta2, delta2; g2aff *B2; size_t nB2;
    uint64_t n_wires, n_infA, n_infB; uint8_t *infA, *infB;
    int n_ck; g1aff *basis, *basis_sigma; size_t n_basis;
} pk_t;
typedef struct {
    g1aff alpha, beta1, delta1; g2aff beta2, gamma2, delta2; g1aff *K; size_t nK;
    int n_commit; g2aff ped_g, ped_gsigma_neg;
} vk_t;
typedef struct { uint8_t *W, *A, *B, *C, *h; } prove_dump_t;   /* canonical 32-byte big-endian dumps (optional) */

int pk_parse(pk_t *pk, const uint8_t *buf, size_t len); void pk_free(pk_t *pk);
int vk_parse(vk_t *vk, const uint8_t *buf, size_t len); void vk_free(vk_t *vk);
void compute_h(const pk_t *pk, const fe *A, const fe *B, const fe *C, size_t m, fe *h);
void hash_to_fr(fe *out, const uint8_t *msg, size_t len, const char *dst);
int groth16_prove(const r1cs_t *cs, const pk_t *pk, const fe *witness, const fe *r, const fe *s, const fe *mask,
                  uint8_t *proof_out, size_t *proof_len, prove_dump_t *dump);
int groth16_verify(const vk_t *vk, const uint8_t *proof, size_t proof_len, const fe *pub, size_t n_pub);
/* test keys in gnark layout from a seed; caller frees *pk_out / *vk_out */
int groth16_setup(const r1cs_t *cs, const uint8_t seed[32], uint8_t **pk_out, size_t *pk_len, uint8_t **vk_out, size_t *vk_len);
void assign_chacha(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t pt[64], uint8_t ct[64], fe *wit);
void assign_aes(const uint8_t *key, int keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t pt[64], uint8_t ct[64], fe *wit);
#endif
