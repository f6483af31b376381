// Launchers for the gfx950 kernels (defined in k_*.hip).  Everything takes a stream and returns
// immediately; no launcher allocates, synchronises or copies (hipGraph-capturable, guide §6 G9).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include "bn254_dev.hpp"

namespace gsc {
using bn254::fe;
using bn254::fe2;

struct G1Aff { fe x, y; };                 // 64 B, Montgomery
struct G2Aff { fe2 x, y; };                // 128 B
struct G1Xyzz { fe x, y, zz, zzz; };       // 128 B
struct G2Xyzz { fe2 x, y, zz, zzz; };      // 256 B

// ---- InitAlgorithm-time kernels (k_init.hip) ----
// in: n x 32 B big-endian compressed X (flag bits in byte 0).  status[i] != 0 => not on curve / bad encoding.
// Infinity encodings produce the generator and status 2 (caller maps the base to the zero scalar row).
void launch_decompress_g1(const uint8_t* in, G1Aff* out, uint8_t* status, size_t n, hipStream_t s);
void launch_decompress_g2(const uint8_t* in, G2Aff* out, uint8_t* status, size_t n, hipStream_t s);
// canonical big-endian Fr -> Montgomery limbs
void launch_fr_from_be(const uint8_t* in, fe* out, size_t n, hipStream_t s);
void launch_fr_inverse(const fe* in, fe* out, size_t n, hipStream_t s);
// Fixed-base digit tables: table[(k*nwin + j)*D + (d-1)] = d * 2^(c*j) * base[k],  D = 2^(c-1), affine.
// Builds table rows [row0, row0 + nrows) (row = k*nwin + j); scratch: nrows * D projective points.
void launch_build_table_g1(const G1Aff* bases, size_t row0, size_t nrows, int c, int nwin, G1Aff* table, G1Xyzz* scratch, hipStream_t s);
void launch_build_table_g2(const G2Aff* bases, size_t row0, size_t nrows, int c, int nwin, G2Aff* table, G2Xyzz* scratch, hipStream_t s);
// NTT constants for a domain of size 2^L: tw_fwd[i] = w^i, tw_inv[i] = w^-i (i < n/2, Montgomery);
// scale_mid[pos] = n^-1 * g^bitrev(pos) (Montgomery); scale_out[pos] = n^-1 * g^-bitrev(pos) (plain, so that the
// Montgomery product with it leaves the result in canonical form).
void launch_ntt_constants(const fe* omega, const fe* omega_inv, const fe* g, const fe* g_inv, const fe* n_inv, int L,
                          int32_t* tw_fwd, int32_t* tw_inv, fe* scale_mid, fe* scale_out, fe* den_inv, int32_t* qr, hipStream_t s);

// TEST HOOK: radix-2^29 field self-test.  field 0 = Fp, 1 = Fr; a, b, out: n canonical 32-byte little-endian values (device memory).
void launch_field_ops(int field, int op, const fe* a, const fe* b, fe* out, size_t n, int chain, hipStream_t s);

// ---- witness generation (k_solver.hip) ----
// inputs: batch x 176 B records {key[32], nonce[12], counter u32 LE, pt[64], ct[64]} (ChaCha) laid out per proof.
// W layout: W[wire * batch + proof].
void launch_assign_chacha(const uint8_t* inputs, fe* W, size_t batch, hipStream_t s);
// AES records: {key[32] (zero padded), nonce[12], counter u32 LE, pt[64], ct[64]}; keylen 16 or 32
void launch_assign_aes(const uint8_t* inputs, int keylen, fe* W, size_t batch, hipStream_t s);
// rs: batch x 2 x 32 B little-endian canonical (r, s).  Fills rows nw..nw+3 of W: r, s, -r*s, 0 (Montgomery).
// mask_in (optional): batch x 32 B little-endian canonical value for hints.Randomize -> mask_out[proof] (Montgomery)
void launch_prep_rs(const uint8_t* rs, fe* W, size_t n_wires, size_t batch, const uint8_t* mask_in, fe* mask_out, hipStream_t s);
struct SolverArgs {
    const uint32_t* prog; const uint32_t* sched;   // instruction words; level schedule (formats.hpp SolverProgram::sched)
    uint32_t first_level, end_level;               // the level this launch executes (end_level unused by the kernel)
    const fe* coeff; const fe* coeff_inv; const uint32_t* lookup_coeff;
    fe* W; fe* A; fe* B; fe* C; size_t batch;
    uint32_t* status;                               // per proof: 0xFFFFFFFF satisfied (caller initialises), else index of the first failing op
    const fe* mask;                                 // per proof value for hints.Randomize (Montgomery) or nullptr
    const fe* commit;                               // per proof commitment challenge (Montgomery) or nullptr
    int has_div;                                    // program contains divisions (R1C solved in L or R)
    uint32_t n_long;                                // the first n_long ops of the level get a whole workgroup each
};
// executes level a.first_level, which holds `level_width` instructions
void launch_solver_level(const SolverArgs& a, uint32_t level_width, hipStream_t s);
// same for a level made of OP_COUNT ops (LDS histogram kernel)
void launch_solver_count_level(const SolverArgs& a, uint32_t level_width, hipStream_t s);
// *flag |= 1 if some OP_COUNT table's index column is not 0,1,2,... (InitAlgorithm-time validation)
void launch_check_count_tables(const uint32_t* prog, const fe* coeff, const uint32_t* count_ops, uint32_t nops, uint32_t* flag, hipStream_t s);

constexpr uint32_t MSM_GROUP_ENTRIES = 3280;      // (3^8 - 1) / 2
void launch_build_subset_g1(const G1Aff* bases, size_t ngroups, G1Aff* table, G1Xyzz* scratch, uint8_t* ok, hipStream_t s);
void launch_build_subset_g2(const G2Aff* bases, size_t ngroups, G2Aff* table, G2Xyzz* scratch, uint8_t* ok, hipStream_t s);

// Calibration: cls[w] = 0 if wire w is 0 or 1 in every accepted proof of the batch (status == 0xFFFFFFFF), otherwise the largest bit
// length of its sign-normalised value (1 = the wire also takes -1, .. 254).  W: [n_wires][batch] Montgomery.
void launch_classify_wires(const fe* W, size_t n_wires, size_t batch, const uint32_t* status, uint8_t* cls, hipStream_t s);

// ---- quotient polynomial (k_ntt.hip) ----
struct NttPlan { int L; const int32_t* tw_fwd; const int32_t* tw_inv; const fe* scale_mid; const fe* scale_out; const fe* den_inv; const int32_t* qr; };   // tw_*, qr: 12 int32 per entry (limbs)
constexpr int NTT_QMAX = 512;      // qr[q + NTT_QMAX] = q*r as limbs, q = -NTT_QMAX .. NTT_QMAX (range reduction by the top limb)
// a,b,c: [n][batch] Montgomery, first m rows valid (rows >= m are treated as zero and need not be initialised).
// On return `a` holds h in canonical form: a[pos] = h_{bitrev(pos)} — the order pk.G1.Z is stored in.
// Domains the four kernels are written (and tested) for: workgroups of 2^ceil(L/2) and 2^floor(L/2) threads within their launch bounds.
constexpr int NTT_MIN_LOG2 = 15, NTT_MAX_LOG2 = 17;
// Returns the first launch-configuration error (nothing is launched when the domain is unsupported).
hipError_t launch_compute_h(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s);

// ---- multi-scalar multiplication (k_msm.hip) ----
// partial[slice * batch + proof] = sum over bases k in slice of scalar[rows[k]][proof] * base_k
struct MsmArgs {
    const void* table; int c; int nwin; size_t nbases;
    const uint32_t* rows;          // scalar row per base (nullptr: row k)
    const fe* scalars;             // [row][batch]
    int scalars_mont;              // 1: Montgomery form, 0: canonical
    size_t batch; size_t nslices;  // slices of ceil(nbases/nslices) consecutive bases, rounded up to a multiple of 8
    void* partial;                 // G1Xyzz / G2Xyzz [nslices][batch]
    // Layout of a wire set, predicted at InitAlgorithm (never trusted for correctness):
    //   bases [0, nwide)            full-width scalars: a second digit table with wider digits (table2, c2, nwin2; base index k)
    //   bases [nwide, nwide + nbit) scalars that are -1, 0 or 1 in nearly every proof (nwide, nbit multiples of 8): group g = bases
    //                               nwide+8g .. +7 has a table of signed subset sums sub[g][v-1] = sum_i t_i * base_i for the balanced-ternary
    //                               value v = sum_i t_i 3^i > 0 (MSM_GROUP_ENTRIES affine entries; v < 0 is the negated entry), group_ok[g] != 0
    //   the rest                    everything else, through `table`
    size_t nbit; const void* sub; const uint8_t* group_ok;
    size_t nwide; const void* table2; int c2, nwin2;
};
void launch_msm_g1(const MsmArgs& a, hipStream_t s);
void launch_msm_g2(const MsmArgs& a, hipStream_t s);
// One reduction level: out[g][proof] = sum of the partials of group g of slices; returns the number of groups (1 = out[proof] is the
// final sum).  Groups hold 64 slices (butterfly over lanes, small batches) or MSM_REDUCE_FANIN (lanes = proofs, large batches), so
// `out` must have room for ceil(nslices / MSM_REDUCE_FANIN) * batch points; it must not alias `partial`.
constexpr size_t MSM_REDUCE_FANIN = 32;
inline bool msm_reduce_by_proof(size_t nslices, size_t batch) { return (batch / 64) * ((nslices + MSM_REDUCE_FANIN - 1) / MSM_REDUCE_FANIN) >= 128; }   // by-proof does 8x less work; the butterfly only wins when there are too few (proof group, chunk) waves
inline size_t msm_reduce_groups(size_t nslices, size_t batch) { const size_t f = msm_reduce_by_proof(nslices, batch) ? MSM_REDUCE_FANIN : 64; return (nslices + f - 1) / f; }
size_t launch_msm_reduce_g1(const G1Xyzz* partial, size_t nslices, size_t batch, G1Xyzz* out, hipStream_t s);
size_t launch_msm_reduce_g2(const G2Xyzz* partial, size_t nslices, size_t batch, G2Xyzz* out, hipStream_t s);
// ---- windowed MSM over one table per base (k_msm_win.hip) ----
// table[k * D + (d - 1)] = d * base[k], d = 1 .. D = 2^(c-1), affine.  Built by threads (base, segment of `seg` entries; seg divides D);
// one launch covers threads [t0, t0 + nthreads) of that grid and needs nthreads * seg projective scratch points.
void launch_build_base_table_g1(const G1Aff* bases, size_t t0, size_t nthreads, int c, uint32_t seg, G1Aff* table, G1Xyzz* scratch, hipStream_t s);
void launch_build_base_table_g2(const G2Aff* bases, size_t t0, size_t nthreads, int c, uint32_t seg, G2Aff* table, G2Xyzz* scratch, hipStream_t s);
// Signed-digit recoding of the scalars of one MSM, once per batch: digits[(j * noct + o) * batch + p] holds the eight int16 digits
// e_{8o..8o+7, j} in [-D, D-1] of proof p (noct = ceil(nbases / 8); bases beyond nbases get zero digits).  nwin is chosen by the
// host so that the top window never overflows (msm_windows below).
struct MsmRecodeArgs {
    const fe* scalars; const uint32_t* rows;   // [row][batch]; scalar row per base (nullptr: row k)
    int mont;                                  // 1: Montgomery residues of wire values (sign-normalised before recoding), 0: canonical integers < r
    size_t nbases, batch; int c, nwin;
    uint4* digits;
};
void launch_msm_recode(const MsmRecodeArgs& a, hipStream_t s);
// Number of windows for c-bit signed digits of scalars below r (the top window must absorb the last carry without overflow:
// floor((r-1) / 2^(c (nwin-1))) + 1 <= 2^(c-1) - 1).  r < 2^254, top bits 0x30644e72e131a029...
inline int msm_windows(int c) {
    int nwin = (254 + c - 1) / c;
    const int top_bits = 254 - c * (nwin - 1);                 // bits of r in the top window (1 .. c)
    // value of r's top window: r >> (254 - top_bits), with r = 0x30644e72e131a029b85045b6... * 2^(254-64) (leading 64 bits are enough for c <= 32)
    const unsigned long long lead = 0xc19139cb84c680a6ull;      // floor(r / 2^190): the 64 leading bits of r counted from bit 253
    const unsigned long long top = lead >> (64 - top_bits);
    if (top + 1 > (1ull << (c - 1)) - 1) nwin++;
    return nwin;
}
// partial[(slice * nwin + j) * batch + p] = sum over the bases k of the slice of sign(e_kj) * table[k][|e_kj| - 1]
struct MsmWinArgs {
    const void* table; int c, nwin; size_t nbases;
    const uint4* digits; size_t batch;
    size_t nslices, per;           // slices of `per` consecutive bases (per a multiple of 8)
    void* partial;                 // G1Xyzz / G2Xyzz [nslices][nwin][batch]
    int placement;                 // workgroup -> (slice, window, proofs) map: 0 one XCD per slice, 1 four XCDs per slice (speed only)
    int exp_same_entry;            // MEASUREMENT ONLY (GSC_MSM_EXP=1 with test hooks on): every gather reads entry 0 — wrong sums, pure VALU time
};
void launch_msm_win_g1(const MsmWinArgs& a, hipStream_t s);
void launch_msm_win_g2(const MsmWinArgs& a, hipStream_t s);
// out[p] = sum_j 2^(c j) S[j * batch + p]
void launch_msm_horner_g1(const G1Xyzz* S, int nwin, int c, size_t batch, G1Xyzz* out, hipStream_t s);
void launch_msm_horner_g2(const G2Xyzz* S, int nwin, int c, size_t batch, G2Xyzz* out, hipStream_t s);

// Commitment helpers (AES-V2).  points: batch XYZZ sums -> out: batch x 64 B big-endian canonical X|Y
// (gnark's uncompressed G1 encoding, the prefix of the commitment hash); flags[proof] |= bit if the point is infinity.
void launch_points_to_affine_be(const G1Xyzz* points, size_t batch, uint8_t* out, uint8_t* flags, uint32_t bit, hipStream_t s);
// h48: batch x 48 bytes (big-endian integers from expand_message_xmd) -> commit[proof] = value mod r, Montgomery form
void launch_challenge_from_hash(const uint8_t* h48, fe* commit, size_t batch, hipStream_t s);

// Proof assembly: inputs are the completed sums
//   sumA = alpha + sum A + r*delta, sumB1 = beta + sum B + s*delta, sumB2 (G2), sumK = sum K - rs*delta, sumZ.
// rs: batch x 64 B (r, s little-endian canonical).  out: batch x 256 B = Ar.x Ar.y | Bs.x.a0 Bs.x.a1 Bs.y.a0 Bs.y.a1 | Krs.x Krs.y,
// canonical little-endian limbs; flags[proof] bit0 Ar inf, bit1 Bs inf, bit2 Krs inf (buffer zeroed by the caller,
// 4-byte aligned, length rounded up to 4).  tmp: scratch of 2 * batch points.
void launch_finalize(const G1Xyzz* sumA, const G1Xyzz* sumB1, const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ,
                     const uint8_t* rs, size_t batch, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s);

}  // namespace gsc
