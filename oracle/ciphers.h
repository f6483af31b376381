/* TEST ORACLE — not product code (see bn254.h). */
#ifndef ORACLE_CIPHERS_H
#define ORACLE_CIPHERS_H
#include <stdint.h>
#include <stddef.h>
void chacha20_block(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, uint8_t out[64]);
void chacha20_xor(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t *in, uint8_t *out, size_t len);
void aes_encrypt_block(const uint8_t *key, int keylen, const uint8_t in[16], uint8_t out[16]);
void aes_ctr_xor(const uint8_t *key, int keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t *in, uint8_t *out, size_t len);
void sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
int expand_message_xmd(const uint8_t *msg, size_t msg_len, const uint8_t *dst, size_t dst_len, uint8_t *out, size_t out_len);
#endif
