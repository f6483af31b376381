// Host-side decoders for the reference's binary artefacts (no field arithmetic on the host:
// everything numeric is shipped to the GPU as raw limbs / compressed coordinates).
//
//  * R1CS  : gnark v0.11.0 constraint.ConstraintSystem.WriteTo layout, read by the reference at
//            libraries/prover/impl/prove_impl.go:102-103 (format: SURVEY.md App. A).
//  * pk    : groth16.ProvingKey.WriteTo layout, read at prove_impl.go:86-87.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

namespace gsc {

constexpr uint32_t HINT_NBITS = 4115454955u;      // std/math/bits.nBits
constexpr uint32_t HINT_COUNT = 2138922168u;      // std/internal/logderivarg.countHint
constexpr uint32_t HINT_RANDOMIZE = 1774611027u;  // internal/hints.Randomize
constexpr uint32_t HINT_BSB22 = 4156202267u;      // frontend/cs.Bsb22CommitmentComputePlaceholder
constexpr uint32_t WIRE_CONST = 0xFFFFFFFFu;

enum BlueprintKind { BP_HINT = 0, BP_R1C = 1, BP_LOOKUP = 2 };

struct R1csFile {
    size_t n_public = 0, n_secret = 0, n_internal = 0, n_constraints = 0;
    size_t n_wires() const { return n_public + n_secret + n_internal; }
    std::vector<uint32_t> calldata;
    std::vector<size_t> instr_start;                 // n_instr + 1
    std::vector<uint32_t> blueprint, constraint_off, wire_off;
    std::vector<std::vector<uint32_t>> levels;
    std::vector<BlueprintKind> bp_kind;
    std::vector<std::vector<uint32_t>> bp_entries;   // lookup blueprints: EntriesCalldata
    std::vector<uint32_t> coeff_limbs;               // n_coeff * 8 u32 limbs, little-endian, Montgomery (as stored)
    size_t n_coeff() const { return coeff_limbs.size() / 8; }
    bool has_commitment = false;
    uint32_t commit_wire = 0;
    std::vector<uint32_t> commit_private;
    size_t n_public_committed = 0;
    size_t n_instr() const { return blueprint.size(); }
};
// throws std::runtime_error on malformed input
R1csFile parse_r1cs(const uint8_t* buf, size_t len);

// One compressed point as stored in the key file, split into (x bytes big-endian with the flag bits
// cleared, flag).  flag: 0x80 smaller y, 0xC0 larger y, 0x40 infinity.
struct PkFile {
    uint64_t domain_n = 0;
    uint8_t n_inv[32], omega[32], omega_inv[32], coset_g[32], coset_g_inv[32];   // Fr, big-endian canonical
    // G1 points: 32 bytes each (flag still in the top bits of byte 0)
    std::vector<uint8_t> g1_alpha, g1_beta, g1_delta, g1_A, g1_B, g1_Z, g1_K;
    // G2 points: 64 bytes each (X.A1 | X.A0)
    std::vector<uint8_t> g2_beta, g2_delta, g2_B;
    uint64_t n_wires = 0;
    std::vector<uint8_t> inf_A, inf_B;
    bool has_commitment_key = false;
    std::vector<uint8_t> ped_basis, ped_basis_sigma;   // G1, 32 bytes each
};
PkFile parse_pk(const uint8_t* buf, size_t len);

// ---- device solver program (built once per algorithm from the R1CS instruction list) ----
// Word stream; every op starts with a header word: opcode | (total_words << 8).  The stream is padded with 64 zero
// words so that a wave may always fetch a full 64-word window starting at any op.
enum SolverOp : uint32_t {
    OP_END = 0,
    OP_R1C = 1,       // [hdr, loc, constraint, unk_wire, unk_coeff, L, R, O]   loc 0 none,1 L,2 R,3 O; L/R/O = linear expressions
                      //   (n, then n x (cid, wid)) without the term of the wire being solved
    OP_NBITS = 2,     // [hdr, out0, nOut, linear expression]
    OP_COUNT = 3,     // [hdr, out0, nTable, nVars(=2), nQueries, nTable x {[1,cid_index,CONST],[1,cid_value,CONST]}, nQueries x {expr, expr}]
    OP_LOOKUP = 4,    // [hdr, out0, nIn, table_id, then nIn linear expressions]
    OP_RANDOMIZE = 5, // [hdr, out0, nOut]
    OP_COMMIT = 6,    // [hdr, out0, nOut]   value supplied by the host (commitment challenge) per proof
};
struct SolverProgram {
    std::vector<uint32_t> words;
    // lookup tables flattened: table t entry e -> coefficient id (entries are constant expressions)
    std::vector<uint32_t> lookup_coeff;      // n_tables * 256
    size_t n_tables = 0;
    // Level schedule (ops of one level are mutually independent; a level only reads wires produced by earlier levels):
    // sched = [n_levels, level_start[0..n_levels], op_word_offset[0..n_ops)] with ops listed level by level.
    // When the circuit has a commitment, its OP_COMMIT sits alone in level `commit_level`; the host runs levels
    // [0, commit_level), computes the commitment, then runs [commit_level, n_levels).  Otherwise commit_level == n_levels.
    std::vector<uint32_t> sched;
    std::vector<uint8_t> level_kind;         // per level: 0 = generic ops (k_solver), 1 = OP_COUNT histogram ops (k_solver_count)
    std::vector<uint32_t> level_long;        // per level: its first level_long[l] ops are "long" (more than LONG_OP_WORDS words)
    static constexpr uint32_t LONG_OP_WORDS = 48;
    std::vector<uint32_t> count_ops;         // word offsets of the OP_COUNT ops (checked on the device at init)
    size_t n_levels = 0, commit_level = 0, max_level_width = 0;
    size_t n_ops = 0, n_inversions = 0;
};
SolverProgram build_solver_program(const R1csFile& cs);

// The same program re-laid for calls with a handful of statements (k_solver_few: one wave per (statement, op), lanes = terms):
// fixed 8-word descriptors so that an op costs one scalar load instead of a walk through its words, the terms of its linear
// expressions contiguous (L | R | O), a lookup op split into one descriptor per expression.
//   R1C:        [OP_R1C | loc << 8, constraint, unknown wire, unknown coeff, term offset, nL, nR, nO]
//   NBITS:      [OP_NBITS, out0, nOut, 0, term offset, n, 0, 0]
//   LOOKUP:     [OP_LOOKUP, out wire, table, 0, term offset, n, 0, 0]
//   RANDOMIZE / COMMIT: [op, out0, nOut, 0 ...]
// level_start has n_levels + 1 entries; OP_COUNT levels are empty here (they keep their histogram kernel).
struct FewProgram {
    std::vector<uint32_t> ops;           // 8 words per descriptor
    std::vector<uint32_t> terms;         // (coefficient id, wire id) pairs
    std::vector<uint32_t> level_start;
    size_t max_level_width = 0;
    // OP_COUNT levels for k_solver_count_few (lanes = queries): per op [word offset of the op, first entry in count_qoff, queries, 0];
    // count_qoff: word offset of every query's first expression; count_first[l]: first op of level l in count_ops (OP_COUNT levels only)
    std::vector<uint32_t> count_ops, count_qoff, count_first;
};
FewProgram build_few_program(const SolverProgram& sp);

}  // namespace gsc
