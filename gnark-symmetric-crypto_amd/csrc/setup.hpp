// Groth16 Setup for the reference's circuits, in gnark's key layout.
//
// Stands in for groth16.Setup as the reference's key generator calls it (keygen.go:345,384,423) for the R1CS files it ships:
// needed in practice because pk.aes128 / pk.aes256 are absent from the reference (.MISSING_LARGE_BLOBS:1-2), so the AES-V2
// algorithms can only be deployed with keys made here.  Scalars (Lagrange basis at tau, A_i/B_i/C_i(tau), K_i, Z_k) are host
// arithmetic (host_field.hpp); the ~5*10^5 generator multiples are computed on the GPU (k_fixed_mul over window rows of G).
// Output: pk in groth16.ProvingKey.WriteTo layout (incl. InfinityA/B filtering, bit-reversed Z truncated to
// n-1, commitment keys), vk in VerifyingKey.WriteTo layout (App. B.2), commitment extension per App. H.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace gsc {

struct SetupKeys { std::vector<uint8_t> pk, vk; };
// seed32 == nullptr: toxic waste from the OS CSPRNG (the only mode a production host can reach).  With a seed the keys are a
// deterministic function of (r1cs, seed): TEST keys.  Throws std::runtime_error on malformed input / device failure.
SetupKeys groth16_setup(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* seed32, int device);

// scalars_le: n canonical 32-byte little-endian scalars; gen_mont: the generator's affine coordinates as Montgomery (R = 2^256)
// little-endian images (G1: 64 B; G2: 128 B = x.a0, x.a1, y.a0, y.a1).  out: n affine points, canonical little-endian coordinates
// (same order); inf[i] = 1 where the scalar was zero.
void setup_generator_muls(int device, bool g2, const uint8_t* gen_mont, const uint8_t* scalars_le, size_t n, uint8_t* out, uint8_t* inf);

}  // namespace gsc
