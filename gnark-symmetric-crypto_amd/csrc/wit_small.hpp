// The "small-integer" witness path: a second, much cheaper solver for constraint systems whose whole witness is small integers.
//
// Replaces, for such circuits, the same reference steps as k_solver.hip (cs.Solve inside groth16.Prove, reference
// libraries/prover/impl/provers.go:148) — the generic lanes-are-proofs interpreter there computes every term in
// 256-bit field arithmetic and stores every value as a 32-byte element, one kernel launch per level.  ChaCha20-V3 never needs
// that: all 40 coefficients of its R1CS are integers below 2^34, every wire is a bit or a value in {-1, 0, 1}, every linear
// expression is a sum below 2^36.  The program below is the same instruction list re-laid for that case:
//   * wires live in a byte plane W8[group of 64 proofs][wire][64] (0, 1, 0xFF = -1) — a proof group's whole witness is 1.5 MB;
//   * the CHAIN (k_wit_chain): one workgroup per proof group walks the levels that PRODUCE wires (XOR-style products, nBits
//     hints) with a workgroup barrier between levels — no kernel boundary, no device-wide barrier, no field arithmetic;
//   * the ROWS (k_wit_rows): a = L(w), b = R(w), c = O(w) of every constraint, fully parallel, checked (a b == c) and written as
//     byte planes A8 / B8 / C8 — or as 32-byte Montgomery elements for the few rows that are predicted wide (ChaCha20's 336
//     add32 sums).
// Which wires / rows are narrow is the calibration witness's PREDICTION (engine_tables.hip calibrate()); both kernels check every
// value they store and raise `flag` when one does not fit, and the engine then proves the chunk again with the generic solver:
// a wrong prediction costs time, never correctness.  A circuit that does not qualify (AES-V2: lookups, inverses, a commitment)
// simply has no small program.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "formats.hpp"

namespace gsc {

constexpr uint32_t WS_IB = 8;                 // tiny items per bundle: item lists are padded to a multiple of this
constexpr uint32_t WS_TINY_WORDS = 16;        // [flags, out | constraint, wire x 6 (L0 L1 R0 R1 O0 O1), coefficient x 6 (int32), 0, 0]
constexpr uint32_t WS_CHUNK = 8;              // terms per load round of the general sums; term lists are padded to a multiple of this
constexpr uint32_t WS_PART_CHUNKS = 4;        // a part of an nBits sum: at most 32 terms, one load round
constexpr uint32_t WS_BITS_PER_ITEM = 9;      // output bits written by one wave of the second phase
constexpr uint32_t WS_F_ITEM = 1u;            // flags: a real item (0 = padding)
constexpr uint32_t WS_F_NEG = 1u << 8;        // chain: the solved wire's coefficient is -1
constexpr int WS_CLS_SHIFT_A = 16, WS_CLS_SHIFT_B = 18, WS_CLS_SHIFT_C = 20;      // rows: 0 = byte plane, 1 = 32-byte element
constexpr int64_t WS_COEF_TINY = (int64_t)1 << 28;      // |coefficient| of a tiny item: sums of two stay in 32 bits

struct SmallProgram {
    bool ok = false; std::string why;         // why: the first reason the circuit does not qualify
    uint32_t n_wires = 0, n_constraints = 0;
    uint32_t rows_per_group = 0;              // rows of W8 per proof group: the wires, then `scratch_row` (padding items write there)
    uint32_t scratch_row = 0;
    // chain: per level [tiny0, tiny1, part0, part1, bits0, bits1] (item indices)
    std::vector<uint32_t> levels; uint32_t n_levels = 0, max_slots = 0;
    std::vector<uint32_t> tiny;               // WS_TINY_WORDS per item: out = +-(L R - O)
    std::vector<uint32_t> parts;              // 4 per part: [LDS slot, first term, chunks, 0]
    std::vector<uint32_t> bits;               // 4 per item: [first output wire, first slot | parts << 16, first bit | bits << 8, 0]
    std::vector<uint32_t> twire; std::vector<int64_t> tcoef;      // terms of the parts (padded with wire 0 x 0)
    // rows
    std::vector<uint32_t> rtiny;              // WS_TINY_WORDS per constraint, padded to a multiple of WS_IB
    std::vector<uint32_t> rgen;               // 8 per constraint: [flags, constraint, first term, chunks L, chunks R, chunks O, 0, 0]
    std::vector<uint32_t> rtwire; std::vector<int64_t> rtcoef;
    std::vector<uint8_t> cls_a, cls_b, cls_c; // per constraint row: 0 = byte plane, 1 = 32-byte element (written to A / B / C)
    size_t n_chain_items = 0, n_nbits = 0;
    uint32_t n_rtiny = 0, n_rgen = 0;         // items of the rows kernel (n_rtiny: a multiple of WS_IB)
};

// coef[cid] / coef_ok[cid]: the R1CS coefficients as integers (|c| < 2^62) where they are that small (computed on the device at
// InitAlgorithm: launch_coeff_small); class_w / class_a / class_b / class_c: the calibration classes (0 = bit, 1 = also -1, more = wide)
SmallProgram build_small_program(const SolverProgram& sp, size_t n_wires, size_t n_constraints, const std::vector<int64_t>& coef, const std::vector<uint8_t>& coef_ok,
                                 const std::vector<uint8_t>& class_w, const std::vector<uint8_t>& class_a, const std::vector<uint8_t>& class_b, const std::vector<uint8_t>& class_c);

}  // namespace gsc
