/* TEST ORACLE — not product code (see bn254.h). */
#ifndef ORACLE_R1CS_H
#define ORACLE_R1CS_H
#include "bn254.h"

#define HINT_NBITS      4115454955u  /* github.com/consensys/gnark/std/math/bits.nBits */
#define HINT_COUNT      2138922168u  /* std/internal/logderivarg.countHint */
#define HINT_RANDOMIZE  1774611027u  /* internal/hints.Randomize */
#define HINT_BSB22      4156202267u  /* frontend/cs.Bsb22CommitmentComputePlaceholder */

enum { BP_HINT = 0, BP_R1C = 1, BP_LOOKUP = 2 };

typedef struct {
    size_t n_instr;
    uint32_t *bp, *coff, *woff;      /* per instruction: blueprint id, constraint offset, wire offset */
    size_t *cstart;                  /* per instruction: first calldata word */
    uint32_t *calldata; size_t n_calldata;
    fe *coeff; size_t n_coeff;       /* Montgomery form, as stored in the file */
    size_t n_public, n_secret, n_internal, n_wires, n_constraints;
    int n_bp; int bp_kind[32]; uint32_t *bp_entries[32]; size_t bp_nentries[32];
    size_t n_levels; size_t *level_off; uint32_t *level_instr;
    int n_commit; uint32_t commit_index; uint32_t *commit_priv; size_t n_commit_priv; size_t n_pub_committed;
} r1cs_t;

int r1cs_parse(r1cs_t *cs, const uint8_t *buf, size_t len);   /* 0 on success */
void r1cs_free(r1cs_t *cs);

typedef int (*commit_cb_t)(void *ctx, const fe *vals, size_t n, fe *out);
typedef struct {
    const fe *randomize;          /* value returned by hints.Randomize (NULL -> 0) */
    commit_cb_t commit_cb; void *commit_ctx;   /* Bsb22 hint override (NULL -> output 0) */
} solve_opts_t;

/* witness: n_public-1 + n_secret elements (public first).  W: n_wires.  A,B,C: n_constraints.
 * returns 0 on success, otherwise 1 + index of the failing instruction. */
long r1cs_solve(const r1cs_t *cs, const fe *witness, fe *W, fe *A, fe *B, fe *C, const solve_opts_t *opts);

#endif
