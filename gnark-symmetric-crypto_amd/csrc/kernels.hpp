// Launchers for the gfx950 kernels (defined in k_*.hip).  Everything takes a stream and returns
// immediately; no launcher allocates, synchronises or copies (hipGraph-capturable, guide §6 G9).
#pragma once
#include "glv.hpp"
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
// NTT constants for a domain of size 2^L with generator omega (Montgomery): tw_fwd[i] = w^i, tw_inv[i] = w^-i (i < n/2);
// with zeta the primitive 2n-th root of unity whose square is omega (zeta^n = -1; derived from gnark-crypto's 2^28-th root):
// scale_mid[pos] = n^-1 * zeta^bitrev(pos) (times the factor that moves the solver's 2^256-domain values into the NTT kernels' 2^261
// domain); scale_out[pos] = (2n)^-1 * zeta^-bitrev(pos) and half_c = 16 / n as plain integers, so that the Montgomery products with
// them leave the result in canonical form.  *flag |= 1 if omega is not that root's 2^(28-L)-th power (not a gnark domain).
// tw_inv_plain[i] = w^-i as a canonical integer in limbs (same 12-word entries); scale_mid_plain[pos] = scale_mid[pos] * 2^256: for inputs that are plain
// small integers instead of 2^256-domain images (NttNarrow::plain).
void launch_ntt_constants(const fe* omega, const fe* omega_inv, const fe* n_inv, int L,
                          int32_t* tw_fwd, int32_t* tw_inv, fe* scale_mid, fe* scale_out, fe* half_c, int32_t* qr, uint32_t* flag, int32_t* tw_inv_plain, fe* scale_mid_plain, hipStream_t s);

// TEST HOOK: radix-2^29 field self-test.  field 0 = Fp, 1 = Fr; a, b, out: n canonical 32-byte little-endian values (device memory).
void launch_field_ops(int field, int op, const fe* a, const fe* b, fe* out, size_t n, int chain, hipStream_t s);
void launch_wave_inverse(const fe* a, fe* out, size_t n, hipStream_t s);      // TEST HOOK (field 1, op 8 of launch_field_ops): k_solver's division, 64 values per inversion

// TEST HOOK / diagnostics: one resident wave records n samples {100 MHz clock, shader clock} `interval` 100 MHz ticks apart into out[2 n]
void launch_clock_trace(unsigned long long* out, uint32_t n, uint32_t interval_100mhz_ticks, hipStream_t s);

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
    unsigned long long* trace;                      // diagnostics (GSC_SOLVER_TRACE): per level 8 x 100 MHz clock stamps (min entry, max of 5 stages), or nullptr
};
// executes level a.first_level, which holds `level_width` instructions
void launch_solver_level(const SolverArgs& a, uint32_t level_width, hipStream_t s);
// Calls with at most MSM_FEW_PROOFS statements: levels [from, to) (none of them an OP_COUNT level) in one launch of a resident grid,
// one wave per (statement, op), lanes = terms; program layout: formats.hpp FewProgram.  sync: two words; word 0 (arrival counter) zeroed before every
// launch, word 1 (a launch gave up at a barrier) zeroed once per call: a launch that finds it set returns at once.
struct SolverFewArgs {
    const uint32_t* ops; const uint32_t* terms; const uint32_t* level_start;
    uint32_t from, to;
    const fe* coeff; const fe* coeff_inv; const uint32_t* lookup_coeff;
    fe* W; fe* A; fe* B; fe* C; size_t batch; uint32_t n_real;
    uint32_t* status; const fe* mask; const fe* commit;
    uint32_t* sync;
    uint32_t poll_limit, test_missing;              // barrier polls before giving up; test hook: arrivals that never come (exercises the give-up path)
    uint32_t nlev_trace;                            // diagnostics: slot of the whole-launch stamps in trace (the program's level count)
    unsigned long long* trace;                      // diagnostics (GSC_SOLVER_TRACE): per level stamps of workgroup 0 (100 MHz clock: level in, first item done, workgroup done, released, all arrived, acquired), or nullptr
};
void launch_solver_few(const SolverFewArgs& a, int has_div, uint32_t workgroups, hipStream_t s);
// same for a level made of OP_COUNT ops (LDS histogram kernel)
void launch_solver_count_level(const SolverArgs& a, uint32_t level_width, hipStream_t s);
// the same for the first nproofs columns of a latency-path call: lanes = queries (FewProgram::count_ops / count_qoff; first_op = count_first[level])
void launch_solver_count_few(const SolverArgs& a, const uint32_t* count_ops, const uint32_t* count_qoff, uint32_t first_op, uint32_t level_width, size_t nproofs, hipStream_t s);
// *flag |= 1 if some OP_COUNT table's index column is not 0,1,2,... (InitAlgorithm-time validation)
void launch_check_count_tables(const uint32_t* prog, const fe* coeff, const uint32_t* count_ops, uint32_t nops, uint32_t* flag, hipStream_t s);

// ---- small-integer witness path (k_wit_small.hip; program layout: wit_small.hpp) ----
// Byte planes: W8[(group * rows_per_group + wire) * 64 + lane], A8 / B8 / C8[(group * crows + constraint) * 64 + lane], proof = 64 * group + lane;
// values 0, 1, 0xFF (-1).  *flag |= 1 when a value predicted to fit a byte plane does not (the engine then solves the chunk again generically).
struct WitChainArgs {
    const uint32_t* tiny; const uint32_t* parts; const uint32_t* bits; const uint32_t* twire; const long long* tcoef; const uint32_t* levels; uint32_t nlevels;
    int8_t* W8; size_t rows_per_group; uint32_t* flag;
};
// one workgroup of 16 waves per proof group; lds_bytes = 512 * SmallProgram::max_slots
void launch_wit_chain(const WitChainArgs& a, size_t groups, size_t lds_bytes, hipStream_t s);
struct WitRowsArgs {
    const uint32_t* rtiny; uint32_t n_rtiny, tiny_per_chunk, n_tiny_chunks; const uint32_t* rgen; uint32_t n_rgen; const uint32_t* rtwire; const long long* rtcoef;
    const int8_t* W8; size_t rows_per_group;
    int8_t* A8; int8_t* B8; int8_t* C8; size_t crows;      // byte planes of the rows predicted narrow
    fe* A; fe* B; fe* C; size_t batch;                     // [constraint][batch] Montgomery elements: the rows predicted wide
    uint32_t* status; uint32_t* flag;
};
void launch_wit_rows(const WitRowsArgs& a, size_t groups, hipStream_t s);
// plane[(group * rows_per_group + row) * 64 + lane] = WS_PLANE_WIDE for every row with cls[row] != 0 (once per lane buffer: nothing else writes those entries)
void launch_wit_mark_wide(int8_t* plane, size_t rows_per_group, size_t nrows, const uint8_t* cls, size_t batch, hipStream_t s);
// mat[row * batch + p] = the plane's value as a Montgomery element, for rows 0 .. nrows-1 (cls != nullptr: only rows with cls[row] == 0)
void launch_wit_expand(const int8_t* plane, size_t rows_per_group, size_t nrows, const uint8_t* cls, fe* mat, size_t batch, hipStream_t s);
// rows 0 .. nrows-1 of W (32-byte elements: the input wires as k_assign_* wrote them) into the byte plane; a value outside {-1, 0, 1} raises *flag
void launch_wit_narrow(const fe* W, size_t batch, size_t nrows, int8_t* W8, size_t rows_per_group, uint32_t* flag, hipStream_t s);
// out[i] = coefficient i as an integer when |c| < 2^62 (ok[i] = 1), from its Montgomery image
void launch_coeff_small(const fe* coeff, size_t n, long long* out, uint8_t* ok, hipStream_t s);

// Calibration: cls[w] = 0 if wire w is 0 or 1 in every accepted proof of the batch (status == 0xFFFFFFFF), otherwise the largest bit
// length of its sign-normalised value (1 = the wire also takes -1, .. 254).  W: [n_wires][batch] Montgomery.
void launch_classify_wires(const fe* W, size_t n_wires, size_t batch, const uint32_t* status, uint8_t* cls, hipStream_t s);

// ---- quotient polynomial (k_ntt.hip) ----
// Byte planes of the small-integer witness path as inputs of the first transform kernel: plane[k] (or nullptr) for vector k = a, b, c;
// an entry 0, 1, -1 stands for that value (its 2^256 Montgomery image), WS_PLANE_WIDE says the row's 32-byte element is in the vector itself.
constexpr int8_t WS_PLANE_WIDE = -128;
// plain: the first transform works on the byte planes' small integers THEMSELVES (not on their 2^256 Montgomery images): the first two butterfly stages of
// ternary inputs are then scalings of twiddles by integers in [-4, 4] instead of field products; the second kernel's scale table absorbs the change of
// domain (NttPlan::scale_mid_plain).  Evaluation-form launches only (launch_compute_d*).
struct NttNarrow { const int8_t* plane[3]; size_t crows; int plain = 0; };      // plane[(group * crows + row) * 64 + lane]
struct NttPlan { int L; const int32_t* tw_fwd; const int32_t* tw_inv; const fe* scale_mid; const fe* scale_out; const fe* half_c; const int32_t* qr;
                 unsigned long long* clk = nullptr;
                 const int32_t* tw_inv_plain = nullptr; const fe* scale_mid_plain = nullptr; };      // (NttNarrow::plain) tw_inv as canonical integers; scale_mid x 2^256      // clk (diagnostics): kernel k's middle workgroup stamps {100 MHz clock, shader clock} at its start and end into clk[4 k ..]   // tw_*, qr: 12 int32 per entry (limbs)
constexpr int NTT_QMAX = 512;      // qr[q + NTT_QMAX] = q*r as limbs, q = -NTT_QMAX .. NTT_QMAX (range reduction by the top limb)
// a,b,c: [n][batch] Montgomery, first m rows valid (rows >= m are treated as zero and need not be initialised), c = a*b row by row
// (a satisfied constraint system; otherwise the result is not gnark's).  On return `a` holds h in canonical form:
// a[pos] = h_{bitrev(pos)} — the order pk.G1.Z is stored in; b and c are overwritten.
// Domains the four kernels are written (and tested) for: workgroups of 2^ceil(L/2) and 2^floor(L/2) threads within their launch bounds.
constexpr int NTT_MIN_LOG2 = 15, NTT_MAX_LOG2 = 17;
// Returns the first launch-configuration error (nothing is launched when the domain is unsupported).
// ncols > 0: only the first ncols columns (rounded up to the tile's 4) are transformed — a call with a handful of statements in a 64-column batch
hipError_t launch_compute_h(const NttPlan& p, fe* a, fe* b, fe* c, size_t m, size_t batch, hipStream_t s, size_t ncols = 0, const NttNarrow* narrow = nullptr);
// The quotient in EVALUATION form: only the four transforms of a and b; on return a[i] = A(zeta w^i) B(zeta w^i) * 2^261 mod r as a
// canonical integer, natural order (b overwritten, c not touched): the scalars of the bases V_i of launch_quot_bases.
hipError_t launch_compute_d(const NttPlan& p, fe* a, fe* b, size_t m, size_t batch, hipStream_t s, size_t ncols = 0, const NttNarrow* narrow = nullptr);
// The same, but the last kernel recodes d itself: it writes the signed c-bit digits of the windowed MSM (launch_msm_win_g1's format, see
// launch_msm_recode) instead of d — no scalar vector in memory, no recoding pass.  A thread of that kernel holds d at four indices, which
// become four consecutive bases: table position t of the MSM set belongs to the index quot_digit_index(L, t) (the engine lays the bases
// V out in that order).  Whole batches only (ncols = 0).
struct QuotDigits { uint4* digits; int c, nwin; };
hipError_t launch_compute_d_digits(const NttPlan& p, fe* a, fe* b, size_t m, size_t batch, const QuotDigits& qd, hipStream_t s, const NttNarrow* narrow = nullptr);
inline uint32_t quot_digit_index(int L, uint32_t t) {
    const int Lhi = (L + 1) / 2, Llo = L - Lhi; const uint32_t quarter = (1u << Lhi) / 4;
    const uint32_t kq = t & 3, m = t >> 2, u4 = m % quarter, g = m / quarter;
    return ((u4 + kq * quarter) << Llo) + g;
}
// InitAlgorithm: the key's quotient bases in evaluation form (k_quot_bases.hip).  zfile: the n - 1 points of pk.G1.Z as stored
// (bit-reversed order), zstatus[i] = 2 for a point at infinity; n = 2^L.  mode 0: out[i] = U_i = (1/2n) sum_k w^(-ik) Z_k (scalars: the
// solver's c_i); mode 1: out[i] = V_i = -(1 / (2n 2^261)) sum_k zeta^(-k) w^(-ik) Z_k (scalars: launch_compute_d's output).  Natural order,
// n points, affine (8 x 32-bit Montgomery images); status[i] = 2 for the point at infinity.  tw: n/2 elements, scratch: n points.
// perm (device, n entries, or nullptr): out[i] = the point of index perm[i] instead of i.
void launch_quot_bases(const G1Aff* zfile, const uint8_t* zstatus, int L, int mode, const fe* omega_inv, const fe* n_inv,
                       fe* tw, G1Xyzz* scratch, const uint32_t* perm, G1Aff* out, uint8_t* status, hipStream_t s);

// ---- multi-scalar multiplication (k_msm_win.hip, k_msm.hip) ----
// Every MSM of the prover is a fixed-base sum over a set of the proving key (A, B1, B2, K, Z, commitment bases) for a batch of
// independent proofs.  Each base has ONE table row of multiples, T_k[d - 1] = d * P_k (affine).  Two kernels (k_msm_win.hip):
//   windowed: full-width scalars.  sum_k s_k P_k = sum_j 2^(c j) S_j,  S_j = sum_k sign(e_kj) T_k[|e_kj| - 1], signed c-bit digits
//             e_kj in [-D, D-1], D = 2^(c-1); rows are uniform: row k = entries [k * D, (k + 1) * D).
//   flat:     small scalars (bits, ternary values, bytes — nearly every wire).  One signed 16-bit value per (base, proof), one
//             accumulator; rows have their own length rowlen[k] at entry rowoff[k]; bit groups of eight bases share a table of signed
//             subset sums.
struct MsmRowSeg { uint32_t base, first, count, pad; uint64_t entry; };     // build work item: `count` multiples of base, from `first`, written at table entry `entry`
// Builds the rows described by segs (each at most `cap` entries); scratch: nsegs * cap projective points.
void launch_build_rows_g1(const G1Aff* bases, const MsmRowSeg* segs, size_t nsegs, uint32_t cap, G1Aff* table, G1Xyzz* scratch, hipStream_t s);
void launch_build_rows_g2(const G2Aff* bases, const MsmRowSeg* segs, size_t nsegs, uint32_t cap, G2Aff* table, G2Xyzz* scratch, hipStream_t s);
// out[i] = 2^shift[i] * in[src[i]]
void launch_shift_bases_g1(const G1Aff* in, const uint32_t* src, const uint32_t* shift, size_t n, G1Aff* out, hipStream_t s);
void launch_shift_bases_g2(const G2Aff* in, const uint32_t* src, const uint32_t* shift, size_t n, G2Aff* out, hipStream_t s);

// Windowed sets.  Signed-digit recoding once per batch: digits[(j * noct + o) * batch + p] holds the eight int16 digits
// e_{8o..8o+7, j} of proof p (noct = ceil(nbases / 8); bases beyond nbases get zero digits).  nwin = msm_windows(c).  c = 17 (MSM_MAX_WINDOW):
// digits need 18 bits, so an octet takes TWO 16-byte words of eight int32 — digits[2 * index], digits[2 * index + 1]: buffers twice the size.
constexpr int MSM_MAX_WINDOW = 17;
inline size_t msm_digit_words(int c) { return c > 16 ? 2 : 1; }      // 16-byte words per (window, octet, proof)
struct MsmRecodeArgs {
    const fe* scalars; const uint32_t* rows;   // [row][batch]; scalar row per base (nullptr: row k)
    int mont;                                  // 1: Montgomery residues of wire values (sign-normalised before recoding), 0: canonical integers < r
    size_t nbases, batch; int c, nwin;
    uint4* digits;
};
void launch_msm_recode(const MsmRecodeArgs& a, hipStream_t s);
// Number of windows for c-bit signed digits of scalars below r (the top window must absorb the last carry without overflow:
// floor((r-1) / 2^(c (nwin-1))) + 1 <= 2^(c-1) - 1).
inline int msm_windows(int c) {
    int nwin = (254 + c - 1) / c;
    const int top_bits = 254 - c * (nwin - 1);                 // bits of r in the top window (1 .. c)
    const unsigned long long lead = 0xc19139cb84c680a6ull;      // floor(r / 2^190): the 64 leading bits of r counted from bit 253
    const unsigned long long top = lead >> (64 - top_bits);
    if (top + 1 > (1ull << (c - 1)) - 1) nwin++;
    return nwin;
}
// partial[(slice * nwin + j) * batch + p] = sum over the bases k of the slice of sign(e_kj) * T_k[|e_kj| - 1]
// One wave per (slice of `per` consecutive bases, window j, 64 proofs).
struct MsmWinArgs {
    const void* table; int c, nwin; size_t nbases;
    const uint4* digits; size_t batch;
    size_t nslices, per;           // per: a multiple of 8
    void* partial;                 // G1Xyzz / G2Xyzz [nslices][nwin][batch]
    // diagnostics (bench.py's VALU roofline): eight waves spread over the grid stamp {100 MHz clock, shader clock} when they start and when they
    // end -> clk[4 k .. 4 k + 3], k < 8; the ratios are the shader clock the launch really ran at (the chip is power-limited).  nullptr: no stamps.
    unsigned long long* clk = nullptr;
    // TEST HOOK (timing experiment, WRONG sums): every gather's entry index is masked to this many bits, i.e. the kernel does the same arithmetic over rows of
    // 2^bits entries — the L2 hit rate it would have with that many lanes per entry in flight.  0 = off.
    uint32_t exp_entry_mask = 0;
};
void launch_msm_win_g1(const MsmWinArgs& a, hipStream_t s);
void launch_msm_win_g2(const MsmWinArgs& a, hipStream_t s);
// The same partial sums for the first `nproofs` proofs only (columns nproofs .. batch-1 of `partial` are not written): lanes are bases
// instead of proofs — the latency path of a single Prove call.  Worth it below MSM_FEW_PROOFS proofs.
constexpr size_t MSM_FEW_PROOFS = 32;     // layout bound of the latency kernels; the engine's threshold is EngineConfig::few_max
void launch_msm_win_few_g1(const MsmWinArgs& a, size_t nproofs, hipStream_t s);
void launch_msm_win_few_g2(const MsmWinArgs& a, size_t nproofs, hipStream_t s);
// out[p] = sum_j 2^(c j) S[j * batch + p] (+ addend[p] when addend != nullptr), for up to MSM_HORNER_JOBS independent sets in one launch
constexpr int MSM_HORNER_JOBS = 6;
struct MsmHornerJob { const void* S; const void* addend; void* out; int nwin, c; };
struct MsmHornerJobs { MsmHornerJob job[MSM_HORNER_JOBS]; int n; };
void launch_msm_horner_g1(const MsmHornerJobs& jobs, size_t batch, hipStream_t s);
void launch_msm_horner_g2(const MsmHornerJobs& jobs, size_t batch, hipStream_t s);

// Flat sets.  digits[o * batch + p] = eight int16 values of octet o for proof p (k_recode_flat):
//   bit groups  the first nbit bases (a multiple of 8) are predicted to carry scalars in {-1, 0, 1} and are taken eight at a time:
//               group o has a table of signed subset sums sub[o][v - 1] = sum_i t_i * base_{8o+i} for the balanced-ternary value
//               v = sum_i t_i 3^i > 0 (MSM_GROUP_ENTRIES affine entries; v < 0 is the negated entry).  The prediction is checked for
//               every wave of 64 proofs: gok[o * (batch/64) + wave] = 1 and slot 0 carries v, or 0 and the octet holds ordinary values.
//               group_ok[o] = 0 disables a group for good (its subset sums hit the point at infinity);
//   ordinary    slot i = the sign-normalised scalar of base 8o + i if it fits 15 bits, else MSM_FLAT_ESCAPE;
//   window      octwin[o] >= 0: slots = digits octwin[o] .. +7 (c bits, c <= 15) of the single scalar rows[8 o]; the bases of such an
//               octet are the points 2^(c j) P of one wide wire.
// A value beyond its row (a wrong prediction) or an escape is multiplied out from the row's first entry: results never depend on
// rowlen or on the grouping.
constexpr uint32_t MSM_GROUP_ENTRIES = 3280;      // (3^8 - 1) / 2
constexpr int32_t MSM_FLAT_ESCAPE = -32768;
void launch_build_subset_g1(const G1Aff* bases, size_t ngroups, G1Aff* table, G1Xyzz* scratch, uint8_t* ok, hipStream_t s);
void launch_build_subset_g2(const G2Aff* bases, size_t ngroups, G2Aff* table, G2Xyzz* scratch, uint8_t* ok, hipStream_t s);
struct MsmFlatRecodeArgs {
    const fe* scalars; const uint32_t* rows; const int32_t* octwin;   // Montgomery wire values [row][batch]; row per base; per octet first window or -1 (nullptr: none)
    size_t nbases, batch; int c;
    uint4* digits;
    size_t nbit; const uint8_t* group_ok; uint8_t* gok;
    int mont;                                       // scalars are Montgomery values (wires) or canonical integers (h)
    // byte plane of the small-integer witness path (or nullptr): a scalar row r < plane_rows is read from plane[(group * plane_stride + r) * 64 + lane]
    // (0, 1, -1; WS_PLANE_WIDE = the row's 32-byte element is in `scalars` after all) instead of scalars[r * batch + p]
    const int8_t* plane = nullptr; size_t plane_rows = 0, plane_stride = 0;
};
void launch_msm_recode_flat(const MsmFlatRecodeArgs& a, hipStream_t s);
// lanes = octets of one proof; gok[octet * MSM_FEW_PROOFS + proof]; pairs with launch_msm_flat_few_* (nslices = ceil(octets / 64))
void launch_msm_recode_flat_few(const MsmFlatRecodeArgs& a, size_t nproofs, hipStream_t s);
// partial[slice * batch + p]; one wave per (slice of `per` consecutive bases, 64 proofs); per a multiple of 8, at most 512
struct MsmFlatArgs {
    const void* table; const uint64_t* rowoff; const uint32_t* rowlen; size_t nbases;
    const uint4* digits; size_t batch;
    size_t nslices, per;
    void* partial;
    size_t nbit; const void* sub; const uint8_t* gok;
    const fe* scalars; const uint32_t* rows;          // for escapes
};
void launch_msm_flat_g1(const MsmFlatArgs& a, hipStream_t s);
void launch_msm_flat_g2(const MsmFlatArgs& a, hipStream_t s);
void launch_msm_flat_few_g1(const MsmFlatArgs& a, size_t nproofs, hipStream_t s);
void launch_msm_flat_few_g2(const MsmFlatArgs& a, size_t nproofs, hipStream_t s);

// One reduction level over slices: out[g][column] = sum of the partials of group g of slices; returns the number of groups (1 = out
// is the final sum).  `batch` counts independent columns (proofs, or windows x proofs).  Groups hold 64 slices (butterfly over
// lanes, small batches) or MSM_REDUCE_FANIN (lanes = columns, large batches), so `out` must have room for
// ceil(nslices / MSM_REDUCE_FANIN) * batch points; it must not alias `partial`.
constexpr size_t MSM_REDUCE_FANIN = 32;
inline bool msm_reduce_by_proof(size_t nslices, size_t batch) { return (batch / 64) * ((nslices + MSM_REDUCE_FANIN - 1) / MSM_REDUCE_FANIN) >= 128; }   // by-proof does 8x less work; the butterfly only wins when there are too few (proof group, chunk) waves
inline size_t msm_reduce_groups(size_t nslices, size_t batch) { const size_t f = msm_reduce_by_proof(nslices, batch) ? MSM_REDUCE_FANIN : 64; return (nslices + f - 1) / f; }
size_t launch_msm_reduce_g1(const G1Xyzz* partial, size_t nslices, size_t batch, G1Xyzz* out, hipStream_t s);
size_t launch_msm_reduce_g2(const G2Xyzz* partial, size_t nslices, size_t batch, G2Xyzz* out, hipStream_t s);
// calls with a handful of statements: `cols` columns in rows of `stride` (= the batch), only the first npr columns of every row are
// summed (lanes = slices, 64 per wave); returns ceil(nslices / 64); `out` as above
size_t launch_msm_reduce_few_g1(const G1Xyzz* partial, size_t nslices, size_t cols, size_t stride, size_t npr, G1Xyzz* out, hipStream_t s);
size_t launch_msm_reduce_few_g2(const G2Xyzz* partial, size_t nslices, size_t cols, size_t stride, size_t npr, G2Xyzz* out, hipStream_t s);

// Groth16 Setup: out[i] = scalars[i] * G for n independent canonical scalars (8 little-endian words each), from the window rows
// table[j * D + d - 1] = d * 2^(c j) * G, D = 2^(c-1), nwin = msm_windows(c).  out: affine coordinates as canonical integers
// (G1: x, y = 2 x 32 B; G2: x.a0, x.a1, y.a0, y.a1 = 4 x 32 B); inf[i] = 1 for a zero scalar.
void launch_fixed_mul_g1(const G1Aff* table, int c, int nwin, const fe* scalars, size_t n, fe* out, uint8_t* inf, hipStream_t s);
void launch_fixed_mul_g2(const G2Aff* table, int c, int nwin, const fe* scalars, size_t n, fe* out, uint8_t* inf, hipStream_t s);

// Commitment helpers (AES-V2).  points: batch XYZZ sums -> out: batch x 64 B big-endian canonical X|Y
// (gnark's uncompressed G1 encoding, the prefix of the commitment hash); flags[proof] |= bit if the point is infinity.
void launch_points_to_affine_be(const G1Xyzz* points, size_t batch, uint8_t* out, uint8_t* flags, uint32_t bit, hipStream_t s);
// cpts: batch x 64 B (big-endian X | Y) -> commit[proof] = hash_to_field(cpts[proof]) with expand_message_xmd(SHA-256), DST "bsb22-commitment",
// 48 bytes reduced mod r, Montgomery form: the commitment challenge, computed on the device (no host round trip inside a proof)
void launch_challenge_from_point(const uint8_t* cpts, fe* commit, size_t batch, hipStream_t s);

// Proof assembly: inputs are the completed sums
//   sumA = alpha + sum A + r*delta, sumB1 = beta + sum B + s*delta, sumB2 (G2), sumK = sum K - rs*delta, sumZ.
// rs: batch x 64 B (r, s little-endian canonical).  out: batch x 256 B = Ar.x Ar.y | Bs.x.a0 Bs.x.a1 Bs.y.a0 Bs.y.a1 | Krs.x Krs.y,
// canonical little-endian limbs; flags[proof] bit0 Ar inf, bit1 Bs inf, bit2 Krs inf (buffer zeroed by the caller,
// 4-byte aligned, length rounded up to 4).  tmp: scratch of 2 * batch points.
// Two kernels: the scalar multiplications s * Ar and r * Bs1 (254 doublings each: the longest serial chain of the assembly) only need
// sumA and sumB1, so they can run beside the remaining MSMs; the combine step needs everything.
void launch_fin_scalarmul(const G1Xyzz* sumA, const G1Xyzz* sumB1, const uint8_t* rs, size_t batch, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s);
// the same for the first nproofs columns only, one wave per (statement, role): lanes share the doubling chain and split the two GLV
// halves of the scalar (glv[2 * proof + role], role 0 = s, role 1 = r: glv.hpp glv_split on the host)
void launch_fin_scalarmul_few(const G1Xyzz* sumA, const G1Xyzz* sumB1, const GlvSplit* glv, size_t batch, size_t nproofs, uint8_t* out, uint8_t* flags, G1Xyzz* tmp, hipStream_t s);
// sumC: a further addend of Krs (the c-part of the evaluation-form quotient) or nullptr
void launch_fin_combine(const G2Xyzz* sumB2, const G1Xyzz* sumK, const G1Xyzz* sumZ, const G1Xyzz* sumC, const G1Xyzz* tmp, size_t batch, uint8_t* out, uint8_t* flags, hipStream_t s);

}  // namespace gsc
