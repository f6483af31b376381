/* libprove.h — C-ABI of the MI355X-native Groth16 prover for the gnark-symmetric-crypto circuits.
 *
 * Drop-in for the header cgo generates from the reference's libraries/prover/libprove.go
 * (`go build -buildmode=c-shared`, reference README.md:83-96): same symbol names, same argument
 * layout (GoSlice by value, GoUint8 for bool, struct Prove_return), same ownership rules.
 * Additions that the reference does not have are grouped at the end and prefixed gsc_ / named ProveBatch.
 */
#ifndef GSC_LIBPROVE_H
#define GSC_LIBPROVE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned char GoUint8;
typedef long long GoInt;
typedef struct { void *data; GoInt len; GoInt cap; } GoSlice;   /* passed BY VALUE */
struct Prove_return { void *r0; /* proofRes */ GoInt r1; /* resLen */ };

/* libprove.go:17-18 — no-op that forces the dynamic loader to bind the library. */
extern void enforce_binding(void);

/* libprove.go:20-23 -> impl.InitAlgorithm (prove_impl.go:65-114).
 * algorithmID: 0 chacha20, 1 aes-128-ctr, 2 aes-256-ctr (prove_impl.go:15-25).
 * provingKey / r1cs: the gnark v0.11.0 files written by keygen.go:341-352 (read only during the call).
 * Returns 1 on success or if the algorithm is already initialised, 0 on unknown id / parse failure
 * (message on stdout, as the reference prints it). */
extern GoUint8 InitAlgorithm(GoUint8 algorithmID, GoSlice provingKey, GoSlice r1cs);

/* libprove.go:25-28 — releases a buffer returned by Prove / ProveBatch (C free()). */
extern void Free(void *pointer);

/* libprove.go:30-47 -> impl.Prove (prove_impl.go:116-143).
 * params: JSON {"cipher","key","nonce","counter","input"} (provers.go:53-59).
 * Returns a malloc'd, NOT NUL-terminated JSON buffer and its length:
 *   success: {"proof":{"proofJson":"<base64>"},"publicSignals":"<base64 ciphertext>"}
 *   failure: the JSON encoding of the Go panic value (a quoted string for message panics, an object for
 *            decode errors); never unwinds across the ABI. */
extern struct Prove_return Prove(GoSlice params);

/* ---- additions (not in the reference) ---- */

/* Proves many independent statements in one device batch.  params: JSON array of Prove inputs (ciphers may be
 * mixed).  Returns a JSON array whose i-th element is exactly what Prove would have returned for element i. */
extern struct Prove_return ProveBatch(GoSlice params);

/* Binary batch entry used by bench.py / tests (no JSON on the timed path).
 * cipher: algorithm id.  inputs: n records of 112 bytes {key[32] (AES-128: first 16 used), nonce[12],
 * counter u32 little-endian, input[64]}.  proofs: n x 196 bytes, proof_lens: n (0 on failure),
 * ciphertexts: n x 64 bytes.  Returns the number of proofs produced, or -1 if the algorithm is not initialised.
 * A call of up to 32 statements (here and in ProveBatch) shares device batches with concurrent callers, like single Prove calls do. */
extern long long gsc_prove_raw(GoUint8 cipher, const uint8_t *inputs, size_t n, uint8_t *proofs, uint32_t *proof_lens, uint8_t *ciphertexts);

/* Groth16 Setup for one of the reference's circuits (stands in for groth16.Setup, keygen.go:345,384,423; needed because the
 * reference ships no pk.aes128 / pk.aes256).  r1cs: the gnark v0.11.0 constraint system file.  On success (0) *pk / *vk are malloc'd
 * buffers in groth16.ProvingKey.WriteTo / VerifyingKey.WriteTo layout (what InitAlgorithm and the verifier read); release both
 * with Free.  The group elements are computed on the GPU.  seed32 == NULL: toxic waste from the OS CSPRNG, discarded before the
 * call returns.  A non-NULL 32-byte seed makes the keys a deterministic function of (r1cs, seed) — TEST keys — and is refused (-1)
 * unless test hooks are enabled (below). */
extern int gsc_setup(GoSlice r1cs, const uint8_t *seed32, void **pk, size_t *pk_len, void **vk, size_t *vk_len);

/* TEST HOOKS.  gsc_set_deterministic_randomness and every gsc_debug_* function below refuse to work (return -1, message on
 * stdout) unless the process was started with GSC_ENABLE_TEST_HOOKS=1 in its environment; the variable is read once, when the
 * library is loaded.  A production host never sets it: fixed prover randomness voids zero-knowledge for the whole process. */

/* TEST HOOK: fixes the prover randomness (r, s, AES commitment mask; 32-byte big-endian, < Fr modulus) for every
 * subsequent proof of this process; pass NULLs to return to the OS CSPRNG (the default).  With it fixed the proof is a
 * deterministic function of the inputs, which is what byte-level parity with gnark is defined on.
 * Returns 0, or -1 when test hooks are disabled. */
extern int gsc_set_deterministic_randomness(const uint8_t *r_be32, const uint8_t *s_be32, const uint8_t *mask_be32);

/* TEST HOOK: runs one proof and copies the intermediate vectors of the device pipeline for parity tests.
 * which: 0 W (n_wires), 1 A, 2 B, 3 C (n_constraints), 4 h (domain size; element k = h_{bitrev(k)}).
 * Elements are 32 bytes, little-endian 32-bit limbs; W/A/B/C are in Montgomery form (x*2^256 mod r), h is canonical.
 * Call gsc_debug_prove first; gsc_debug_vector returns the element count (or -1) and copies min(cap, size) bytes. */
extern long long gsc_debug_prove(GoSlice params);
extern long long gsc_debug_vector(int which, uint8_t *out, size_t cap);

/* TEST HOOK: element-wise operations of the device's radix-2^29 field arithmetic, for unit tests against big integers.
 * field: 0 = Fp, 1 = Fr.  op: 0 mul, 1 add, 2 sub, 3 sqr, 4 inverse, 5 r*b - b*a (fused), 6 neg, 7 (r-b)*(a+b); the op is applied
 * `chain` times to a running value r that starts at a.  a, b, out: n canonical 32-byte little-endian values.  0 on success. */
extern int gsc_debug_field_ops(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, int chain);

/* TEST HOOK: the quotient-polynomial kernels (computeH) alone, on caller-supplied vectors, 64 independent columns at once.
 * abc_be: a, b, c one after the other, each [m][64] canonical big-endian 32-byte values (m <= constraints of the algorithm).
 * h_out (cap bytes, at least domain*64*32): [domain][64] canonical little-endian values, row k = coefficient bitrev(k).
 * Returns the domain size (also when h_out is NULL: size query), -1 on error. */
extern long long gsc_debug_compute_h(GoUint8 algorithmID, const uint8_t *abc_be, size_t m, uint8_t *h_out, size_t cap);
/* TEST HOOK: the quotient kernels of the EVALUATION form alone (what batch calls run: four transforms instead of six; see
 * csrc/k_quot_bases.hip).  ab_be: a, then b, each [m][64] canonical big-endian values.  d_out: [domain][64] canonical little-endian
 * values, row i = A(zeta w^i) * B(zeta w^i) * 2^261 mod r in natural order (zeta: the primitive 2n-th root of unity, w = zeta^2 the
 * domain generator; A, B the interpolation polynomials of a, b).  Returns the domain size (also for d_out == NULL), -1 on error. */
/* TEST HOOK: n samples {100 MHz clock, shader clock}, interval_us apart, taken by a one-wave kernel that stays resident beside whatever the
 * device runs meanwhile (call it from a thread of its own around a Prove): the shader clock each kernel of a call is granted.  Blocks. */
extern int gsc_debug_clock_trace(uint32_t n, uint32_t interval_us, unsigned long long *out);
/* TEST HOOK: bytes of a finished call's secrets (key wires of the witness, r, s, -rs, the raw input records, commitment masks) that are still
 * non-zero in the algorithm's device buffers — the engine clears them behind the last kernel of every call, so 0; -1 on error / hooks disabled. */
extern long long gsc_debug_secret_residue(GoUint8 algorithmID);
extern long long gsc_debug_compute_d(GoUint8 algorithmID, const uint8_t *ab_be, size_t m, uint8_t *d_out, size_t cap);
/* TEST HOOK (host arithmetic only, no GPU): the GLV split the latency path feeds to its scalar multiplications
 * (csrc/glv.hpp).  k: canonical scalar < r, 32 bytes little-endian.  out: 20 bytes |k1|, 20 bytes |k2| (little-endian), 4 bytes
 * flags (bit 0: k1 < 0, bit 1: k2 < 0) with k = k1 + k2 * lambda (mod r).  Returns 0, -1 on error. */
extern int gsc_debug_glv_split(const uint8_t *k, uint8_t *out);

/* Human-readable description of an initialised algorithm (sizes, table memory, and per engine replica of GSC_DEVICES the calls /
 * statements it has served so far: "served(calls/statements)=a/b,c/d"); returns bytes written. */
extern size_t gsc_describe(GoUint8 algorithmID, char *out, size_t cap);
/* Device milliseconds of the four stages (witness, quotient, msm, assembly) of the batch of that algorithm that finished last. */
extern int gsc_last_stage_ms(GoUint8 algorithmID, float out[4]);
/* The dominant kernel of the batch of that algorithm that finished last, timed with HIP events on the kernel's own stream (bench.py's
 * roofline object).  Batch calls: "k_msm_win<Fp29f>", the Z-table gather-accumulate.  Calls of a handful of statements (the latency
 * path): the resident witness kernel "k_solver_few".  name: NUL-terminated kernel name (cap bytes); *ms: milliseconds; *statements:
 * statements the call proved; *columns: the 64-padded batch the kernels ran on; *nbases: fixed bases per proof of the Z set.
 * Any out pointer may be NULL.  Returns 0, -1 when the algorithm is not initialised. */
extern int gsc_last_dominant_kernel(GoUint8 algorithmID, char *name, size_t cap, float *ms, size_t *statements, size_t *columns, size_t *nbases);
/* For the same batch: *clock_mhz = the shader clock the Z-table kernel ran at (clock stamps of one wave in the middle of the launch; 0 when
 * the call took the latency path), *windows = the digit windows of its Z set.  bench.py prices the kernel's VALU-issue roofline with them.
 * The three gsc_last_* functions report the calling thread's own last gsc_prove_raw / ProveBatch call of more than 32 statements; on a thread
 * that made none, the chunk that finished last on any replica. */
extern int gsc_last_kernel_clock(GoUint8 algorithmID, float *clock_mhz, int *windows);

#ifdef __cplusplus
}
#endif
#endif
