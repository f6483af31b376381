// Native stream ciphers run on the host to obtain the ciphertext half of the public witness, exactly where
// the reference runs them: libraries/prover/impl/provers.go:93-101 (golang.org/x/crypto/chacha20,
// SetCounter + XORKeyStream) and :184-192 (crypto/aes + cipher.NewCTR with IV = nonce || BE32(counter)).
#pragma once
#include <cstdint>
#include <cstddef>
namespace gsc {
void chacha20_xor_stream(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t* in, uint8_t* out, size_t len);
void aes_ctr_xor_stream(const uint8_t* key, size_t keylen, const uint8_t nonce[12], uint32_t counter, const uint8_t* in, uint8_t* out, size_t len);
// SHA-256 and RFC 9380 expand_message_xmd(SHA-256): gnark derives the Groth16 commitment challenge with
// hash_to_field over the uncompressed commitment point (gnark backend/groth16 prove, DST "bsb22-commitment").
void sha256_digest(const uint8_t* msg, size_t len, uint8_t out[32]);
void expand_message_xmd_sha256(const uint8_t* msg, size_t msg_len, const char* dst, uint8_t* out, size_t out_len);
}  // namespace gsc
