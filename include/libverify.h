/* libverify.h — C-ABI of the verifier library, drop-in for the header cgo generates from the reference's
 * libraries/verifier/libverify.go:14-17 (`go build -buildmode=c-shared`).  CPU-side, like the reference's
 * (three pairings per proof); not part of the GPU prover's hot path.
 */
#ifndef GSC_LIBVERIFY_H
#define GSC_LIBVERIFY_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef GSC_LIBPROVE_H
typedef unsigned char GoUint8;
typedef long long GoInt;
typedef struct { void *data; GoInt len; GoInt cap; } GoSlice;   /* passed BY VALUE */
#endif

/* libverify.go:14-17 -> impl.Verify (libraries/verifier/impl/verify_impl.go:62-82).
 * params: JSON {"cipher","proof","publicSignals"} (verify_impl.go:18-22); byte fields are base64 strings or arrays of
 * 0..255.  publicSignals = ciphertext(64) | nonce(12) | counter(4: little-endian for chacha20, big-endian for AES) |
 * plaintext(64) (verifiers.go:59-62, :120-123).  Returns 1 iff the Groth16 proof verifies; any malformed input -> 0. */
extern GoUint8 Verify(GoSlice params);

/* Addition.  The reference embeds its three verifying keys at build time (verify_impl.go:24-31); this library loads them:
 * either explicitly (gnark VerifyingKey.WriteTo bytes) or, on first use, from the directory named by
 * the environment variable GSC_VK_DIR (files vk.chacha20, vk.aes128, vk.aes256 — the reference's generated/ folder).
 * algorithmID: 0 chacha20, 1 aes-128-ctr, 2 aes-256-ctr.  Returns 1 on success. */
extern GoUint8 InitVerifier(GoUint8 algorithmID, GoSlice verifyingKey);

#ifdef __cplusplus
}
#endif
#endif
