// Host-only check of kernels.hpp quot_digit_index: the order in which the evaluation-form quotient lays out its bases V (engine_tables.hip)
// must be a permutation of the domain and must match the index arithmetic of the last quotient kernel (k_ntt.hip, EVAL == 2): thread
// (g, u4) of a strided tile holds the elements u4 + kq * G/4, kq = 0..3, i.e. the natural indices ((u4 + kq * G/4) << Llo) + g, and writes
// them as the bases 4 m + kq, m = g * G/4 + u4.
#include "../../gnark-symmetric-crypto_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>

int main() {
    for (int L = gsc::NTT_MIN_LOG2; L <= gsc::NTT_MAX_LOG2; L++) {
        const uint32_t n = 1u << L; const int Lhi = (L + 1) / 2, Llo = L - Lhi; const uint32_t G = 1u << Lhi, Cn = 1u << Llo;
        std::vector<uint8_t> seen(n, 0);
        for (uint32_t t = 0; t < n; t++) {
            const uint32_t i = gsc::quot_digit_index(L, t);
            if (i >= n || seen[i]) { printf("L=%d: not a permutation at t=%u\n", L, t); return 1; }
            seen[i] = 1;
        }
        for (uint32_t g = 0; g < Cn; g++) for (uint32_t u4 = 0; u4 < G / 4; u4++) for (uint32_t kq = 0; kq < 4; kq++) {
            const uint32_t m = g * (G / 4) + u4, t = 4 * m + kq, want = ((u4 + kq * (G / 4)) << Llo) + g;
            if (gsc::quot_digit_index(L, t) != want) { printf("L=%d: position %u is index %u, the kernel holds %u\n", L, t, gsc::quot_digit_index(L, t), want); return 1; }
        }
    }
    // digits per scalar: the windows of c-bit signed digits must cover 254 bits plus the last carry
    for (int c = 4; c <= gsc::MSM_MAX_WINDOW; c++) if (gsc::msm_windows(c) * c < 254) { printf("c=%d: %d windows do not cover a scalar\n", c, gsc::msm_windows(c)); return 1; }
    if (gsc::msm_windows(17) != 15 || gsc::msm_windows(16) != 16 || gsc::msm_windows(15) != 17) { printf("unexpected window counts\n"); return 1; }
    printf("QUOT-INDEX-OK\n");
    return 0;
}
