// R1CS witness generation on the GPU: one wavefront lane per proof, one uniform instruction stream.
//
// Replaces the cs.Solve step inside groth16.Prove (reference libraries/prover/impl/provers.go:148,216;
// gnark constraint/bn254 solver — SURVEY.md §8(a) a5, semantics App. C) and the witness assignment of
// provers.go:106-142 / :194-210 (bit / byte layout: utils/bytes.go:11-47).
//
// Layout: W[wire][proof], A/B/C[constraint][proof], 32-byte elements: the 64 lanes of a wave read and write
// 2 KiB contiguous per access (coalesced), and the instruction words are wave-uniform (scalar registers).
// Parallelism inside a proof comes from the circuit's level structure (ChaCha20-V3: 163 levels of ~147 mutually
// independent instructions): one launch per level, one wave per (64 proofs, instruction), so a 1024-proof batch
// puts ~2 400 waves on the chip per level; the kernel boundary is the barrier between levels.
#include "kernels.hpp"
#include "formats.hpp"
#include <atomic>

namespace gsc {
using namespace bn254;

namespace {

// WPB = waves per workgroup (template parameter of k_solver): 8 for small batches (the level's long operations are on the critical
// path and split over more waves), 4 for large ones (workgroups of short operations retire sooner)

// A wave-wide 64-word window onto the instruction stream: one coalesced load, fields are read with v_readlane
// (the op and all of its fields are wave-uniform, so they live in scalar registers).
struct Window {
    const uint32_t* prog; uint32_t base; uint32_t w; uint32_t lane;
    __device__ __forceinline__ void load(uint32_t b) { base = b; w = prog[b + lane]; }
    __device__ __forceinline__ uint32_t get(uint32_t k) {
        if (k - base >= 64u) load(k);
        return (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(k - base));
    }
    // k must be inside the window.  Unlike get() this has no reload path, so the compiler can prove the window
    // register is not pending and does NOT put an s_waitcnt vmcnt(0) in front of it — which matters because that
    // wait would also drain the wire loads issued in between (measured: it serialised every term, ~1.2 us each).
    __device__ __forceinline__ uint32_t peek(uint32_t k) const { return (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(k - base)); }
    __device__ __forceinline__ void ensure(uint32_t first, uint32_t count) { if (first + count - base > 64u) load(first); }
};

// element j of v, j wave-uniform: a select chain keeps v in registers (dynamic indexing would go to scratch)
__device__ __forceinline__ fe pick(const fe (&v)[8], uint32_t j) {
    fe r = v[0];
#pragma unroll
    for (int k = 1; k < 8; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = j == (uint32_t)k ? v[k].l[i] : r.l[i];
    }
    return r;
}
__device__ __forceinline__ uint32_t limb_at(const fe& v, uint32_t i) {
    uint32_t r = v.l[0];
#pragma unroll
    for (int k = 1; k < 8; k++) r = i == (uint32_t)k ? v.l[k] : r;
    return r;
}

// Linear expression [n, (cid, wid) x n] at word `at`: loads of up to 8 wires are issued back to back (one memory
// round), then folded by a rolled loop so that the code — in particular the Montgomery product — exists once.
// Code size matters here: every wave runs this straight-line path once, so instruction-cache misses dominate
// if the body is unrolled (measured: 330 KB of code made a level take 170 us instead of ~20).
// part / nparts: the chunks of the expression are dealt round-robin to `nparts` cooperating waves; the caller adds the partial sums.
__device__ __forceinline__ fe eval_expr(Window& win, uint32_t at, const fe* coeff, const fe* W, size_t batch, size_t p, uint32_t& next,
                                        uint32_t part = 0, uint32_t nparts = 1) {
    const uint32_t n = win.get(at);
    next = at + 1 + 2 * n;
    fe acc = Fr::zero();
    for (uint32_t k0 = 8 * part; k0 < n; k0 += 8 * nparts) {
        const uint32_t t0 = at + 1 + 2 * k0;
        win.ensure(t0, 16);                                   // the whole chunk inside the window
        uint32_t cidv[8], widv[8];                            // wave-uniform: scalar registers
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool live = k0 + j < n;
            cidv[j] = live ? win.peek(t0 + 2 * j) : 0u;
            widv[j] = live ? win.peek(t0 + 2 * j + 1) : WIRE_CONST;
        }
        // Wire values and coefficients: 32 loads requested back to back, branch-free (dead slots read coefficient 0, which is
        // cached), so that no control-flow join forces an early s_waitcnt and the whole chunk costs one memory round.
        fe v[8], cfs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const fe* src = widv[j] == WIRE_CONST ? coeff + cidv[j] : W + (size_t)widv[j] * batch + p;
            v[j] = load_fe(src);
            cfs[j] = load_fe(coeff + cidv[j]);
        }
        // Fold.  A lone wave issues one VALU instruction every ~9 cycles, so the per-term instruction count is what a long
        // expression (ChaCha's 130-term add32 rows) costs: the common cases are unrolled with static register indices;
        // only terms that really need a Montgomery product (general coefficient times a non-bit value) take the rolled path.
        uint32_t slow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (k0 + j < n) {
                const uint32_t c = widv[j] == WIRE_CONST ? 1u : cidv[j];
                // gnark reserves coefficient ids 0..4 for 0, 1, 2, -1, -2 (checked on the host at InitAlgorithm)
                if (c == 1) acc = Fr::add(acc, v[j]);
                else if (c == 3) acc = Fr::sub(acc, v[j]);
                else if (c == 2) acc = Fr::add(acc, Fr::dbl(v[j]));
                else if (c == 4) acc = Fr::sub(acc, Fr::dbl(v[j]));
                else if (c > 4) {
                    // Most wires of these circuits are bits: when every lane's value is 0 or 1 the product is a select.
                    const bool is0 = Fr::is_zero(v[j]), is1 = Fr::eq(v[j], Fr::one());
                    if (__builtin_amdgcn_ballot_w64(!(is0 || is1)) != 0) slow |= 1u << j;
                    else acc = Fr::add(acc, is1 ? cfs[j] : Fr::zero());
                }
            }
        }
#pragma unroll 1
        while (slow) {
            const uint32_t j = __builtin_ctz(slow); slow &= slow - 1;
            acc = Fr::add(acc, Fr::mul(pick(cfs, j), pick(v, j)));
        }
    }
    return acc;
}

// One launch per level (the kernel boundary is the inter-level barrier and makes the previous level's stores visible
// chip-wide); grid = (proof groups of 64, ops in the level); one op per workgroup.  The WPB waves of the workgroup split
// the chunks of the op's linear expressions (ChaCha's add32 rows have 130 terms and would otherwise be one wave's serial
// work — a lone wave issues an instruction only every ~9 cycles) and combine the partial sums through LDS.
__device__ fe few_inverse(const fe& a_mont);

// 1 / x for the 64 proofs of a full wave at once (every lane active, x != 0 in every lane; Montgomery in and out): Montgomery's trick
// ACROSS the lanes — inclusive prefix and suffix products by doubling strides, ONE inversion of the product of all 64 (the same value in
// every lane, so Kaliski's data-dependent branches do not diverge), then 1 / x_i = (x_0 .. x_{i-1}) (x_{i+1} .. x_63) / (x_0 .. x_63).
// 14 products + ~25 k instructions for the shared inversion instead of the 134 k of a lane-wise power x^(r-2): AES-V2's
// log-derivative argument divides 2 080 / 2 384 times per proof, all of them in six levels (1.05 ms each at 512 statements).
__device__ __noinline__ fe wave_batch_inverse(const fe& x) {
    const uint32_t lane = threadIdx.x & 63;
    auto from = [](const fe& v, uint32_t src) { fe r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.l[k] = (uint32_t)__shfl((int)v.l[k], (int)(src & 63u));
        return r; };
    auto pick2 = [](bool c, const fe& a, const fe& b) { fe r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.l[k] = c ? a.l[k] : b.l[k];
        return r; };
    fe P = x, S = x;
#pragma unroll 1
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const fe tp = Fr::mul(P, from(P, lane - d)), ts = Fr::mul(S, from(S, lane + d));
        P = pick2(lane >= d, tp, P); S = pick2(lane + d < 64, ts, S);
    }
    fe total;
#pragma unroll
    for (int k = 0; k < 8; k++) total.l[k] = (uint32_t)__builtin_amdgcn_readlane((int)P.l[k], 63);
    const fe ti = few_inverse(total);
    const fe one = Fr::one();
    const fe before = pick2(lane > 0, from(P, lane - 1), one), after = pick2(lane < 63, from(S, lane + 1), one);
    return Fr::mul(Fr::mul(before, after), ti);
}

template <bool HAS_DIV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_solver(SolverArgs a) {
    __shared__ uint32_t s_part[3][WPB][8][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const size_t batch = a.batch;
    const uint32_t nlev = a.sched[0];
    const uint32_t* lstart = a.sched + 1;
    const uint32_t* ops = a.sched + 2 + nlev;
    const uint32_t lev = a.first_level;
    auto stamp = [&](int k) {
        if (a.trace && lane == 0) { const unsigned long long t = wall_clock64(); if (k == 0) atomicMin(a.trace + 16 * lev, t); else atomicMax(a.trace + 16 * lev + k, t); }
    };
    stamp(0);
    // workgroups [0, n_long): one long op each, its terms split over the WPB waves;
    // workgroups [n_long, ...): WPB short ops each, one per wave (the LDS exchange then has a single contributor)
    const bool coop = blockIdx.y < a.n_long;
    const uint32_t i = lstart[lev] + (coop ? blockIdx.y : a.n_long + (blockIdx.y - a.n_long) * WPB + wave);
    const uint32_t part = coop ? wave : 0u, nparts = coop ? (uint32_t)WPB : 1u, slot = coop ? 0u : wave;
    if (i >= lstart[lev + 1]) return;          // only in wave-per-op workgroups, which never reach a barrier
    bool bad = false;
    Window win{a.prog, 0, 0, lane};
    const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)ops[i]);
    win.load(at);
    const uint32_t op = win.get(at) & 0xFF;
    stamp(1);
    if (op == OP_R1C || op == OP_NBITS) {
        const uint32_t f1 = win.get(at + 1), f2 = win.get(at + 2), f3 = win.get(at + 3), f4 = win.get(at + 4);
        const uint32_t nexpr = op == OP_R1C ? 3u : 1u;
        uint32_t q = op == OP_R1C ? at + 5 : at + 3;
#pragma unroll 1
        for (uint32_t e = 0; e < nexpr; e++) {
            uint32_t next;
            const fe r = eval_expr(win, q, a.coeff, a.W, batch, p, next, part, nparts);
            q = next;
#pragma unroll
            for (int k = 0; k < 8; k++) s_part[e][wave][k][lane] = r.l[k];       // own slot: no cross-wave hazard in wave-per-op mode
        }
        fe v[3];
        stamp(2);
        if (coop) { __syncthreads(); if (wave != 0) return; }
        stamp(3);
#pragma unroll 1
        for (uint32_t e = 0; e < nexpr; e++) {
            fe acc;
#pragma unroll
            for (int k = 0; k < 8; k++) acc.l[k] = s_part[e][slot][k][lane];
#pragma unroll 1
            for (uint32_t w = 1; w < nparts; w++) {
                fe t;
#pragma unroll
                for (int k = 0; k < 8; k++) t.l[k] = s_part[e][w][k][lane];
                acc = Fr::add(acc, t);
            }
            if (e == 0) v[0] = acc; else if (e == 1) v[1] = acc; else v[2] = acc;
        }
        stamp(4);
        if (op == OP_NBITS) {                               // [hdr, out0, nOut, expr]
            const fe r = Fr::from_mont(v[0]);
            const fe one = Fr::one(), zero = Fr::zero();
            for (uint32_t k = 0; k < f2; k++) {
                const uint32_t bit = k < 256 ? (limb_at(r, k >> 5) >> (k & 31)) & 1u : 0u;
                store_fe(a.W + (size_t)(f1 + k) * batch + p, bit ? one : zero);
            }
        } else {                                            // [hdr, loc, constraint, unk_wire, unk_coeff, L, R, O]
            fe va = v[0], vb = v[1], vc = v[2];
            const uint32_t loc = f1, cidx = f2, uw = f3, uc = f4;
            // a*b: in these circuits the factors are mostly bits, or one of them is the constant 1 — a wave whose 64 proofs all
            // fall in such a case replaces the 353-instruction product by selects (the solver is half VALU-bound at large batches)
            fe ab;
            {
                const fe one = Fr::one();
                const bool a0 = Fr::is_zero(va), a1 = Fr::eq(va, one), b0 = Fr::is_zero(vb), b1 = Fr::eq(vb, one);
                if (__builtin_amdgcn_ballot_w64(!(b0 || b1)) == 0) {            // b is a bit everywhere: a*b = b ? a : 0
#pragma unroll
                    for (int k = 0; k < 8; k++) ab.l[k] = b1 ? va.l[k] : 0u;
                } else if (__builtin_amdgcn_ballot_w64(!(a0 || a1)) == 0) {     // a is a bit everywhere
#pragma unroll
                    for (int k = 0; k < 8; k++) ab.l[k] = a1 ? vb.l[k] : 0u;
                } else ab = Fr::mul(va, vb);
            }
            if (loc == 0) bad = !Fr::eq(ab, vc);
            else {
                fe wire;
                if (loc == 3) { wire = Fr::sub(ab, vc); vc = ab; }
                else {
                    const fe known = loc == 1 ? vb : va;
                    fe part = loc == 1 ? va : vb;
                    const bool kz = Fr::is_zero(known);
                    fe kinv = Fr::zero();
                    if (HAS_DIV) {                          // loc is wave-uniform: all 64 lanes are here (a lane that cannot divide lends a 1)
                        fe safe;
#pragma unroll
                        for (int k = 0; k < 8; k++) safe.l[k] = kz ? (k == 0 ? 1u : 0u) : known.l[k];      // any non-zero value will do; this one is 1/R
                        kinv = wave_batch_inverse(safe);
                    }
                    if (kz) { wire = Fr::zero(); bad = !Fr::eq(ab, vc); }   // gnark: cannot divide; the constraint must already hold
                    else if (HAS_DIV) { wire = Fr::sub(Fr::mul(vc, kinv), part); part = Fr::add(part, wire); }
                    else { wire = Fr::zero(); bad = true; }     // the host selects HAS_DIV whenever the program divides
                    if (loc == 1) va = part; else vb = part;
                }
                // divide by the unknown wire's coefficient; ids 1 and 3 are the constants 1 and -1 (checked at InitAlgorithm)
                if (uc == 3) wire = Fr::neg(wire);
                else if (uc != 1) wire = Fr::mul(wire, load_fe(a.coeff_inv + uc));
                store_fe(a.W + (size_t)uw * batch + p, wire);
            }
            store_fe(a.A + (size_t)cidx * batch + p, va);
            store_fe(a.B + (size_t)cidx * batch + p, vb);
            store_fe(a.C + (size_t)cidx * batch + p, vc);
        }
    } else if (!coop || wave == 0) {
        if (op == OP_LOOKUP) {                              // [hdr, out0, nIn, table, exprs]: out[e] = table[value of expr e]
            const uint32_t o0 = win.get(at + 1), nin = win.get(at + 2), table = win.get(at + 3);
            uint32_t q = at + 4;
#pragma unroll 1
            for (uint32_t e = 0; e < nin; e++) {
                uint32_t next;
                const fe r = Fr::from_mont(eval_expr(win, q, a.coeff, a.W, batch, p, next));
                q = next;
                const uint32_t hi = r.l[1] | r.l[2] | r.l[3] | r.l[4] | r.l[5] | r.l[6] | r.l[7];
                if (hi != 0 || r.l[0] >= 256) bad = true;
                const uint32_t cid = a.lookup_coeff[table * 256 + (r.l[0] & 255)];
                store_fe(a.W + (size_t)(o0 + e) * batch + p, load_fe(a.coeff + cid));
            }
        } else if (op == OP_RANDOMIZE || op == OP_COMMIT) { // [hdr, out0, nOut]
            const uint32_t o0 = win.get(at + 1), nout = win.get(at + 2);
            const fe* src = op == OP_RANDOMIZE ? a.mask : a.commit;
            fe v = src ? load_fe(src + p) : Fr::zero();
            for (uint32_t k = 0; k < nout; k++) store_fe(a.W + (size_t)(o0 + k) * batch + p, v);
        }
    }
    if (bad) atomicMin(a.status + p, i);     // status: 0xFFFFFFFF = satisfied, else first failing op
    stamp(5);
}

// ---- calls with a handful of statements (single Prove): k_solver_few -----------------------------------------------------------------
// With a handful of statements (EngineConfig::few_max: up to 32) in a 64-column batch, the lanes-are-proofs kernel above leaves the chip idle and pays, per level, a kernel
// launch, a cold walk through the instruction words and a lone wave's serial fold of every term (measured: 26 us per level, 163
// levels for ChaCha20, 445 for AES).  Here the whole level range is ONE launch of a fixed grid that stays resident: a wave takes
// one (statement, op) at a time with lanes = terms of its linear expressions (one load round, one Montgomery product, a short
// reduction), and levels are separated by a device-wide barrier on an arrival counter instead of a kernel boundary.  Only the
// statements' own columns are written (one lane's store per value): the other columns of the 64-wide batch keep what they held —
// zero from the allocation or an earlier call's witness, field elements either way, which is all the transforms and MSMs ask.
// The grid is far smaller than the chip (<= 1 workgroup per CU), so every workgroup is resident; a barrier that is not reached
// within ~2 s of polling (another process's resident kernel holding CUs; the engine chains its own launches) raises the abort
// bit, every workgroup leaves — no wave waits forever — and the host solves the call again with one launch per level.
constexpr uint32_t FEW_WAVES = 8;
constexpr uint32_t FEW_ABORT = 0x80000000u;

__device__ __forceinline__ fe readlane_fe(const fe& v, uint32_t src) {
    fe r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = (uint32_t)__builtin_amdgcn_readlane((int)v.l[k], (int)src);
    return r;
}
// Wire values cross workgroups (and XCDs, each with an L2 of its own) inside one launch: they are read and written with device-scope
// accesses (sc1: loads are served at the memory side, stores write through), so no barrier has to write an L2 back or invalidate one
// (the plain-access variant — the cooperative-groups recipe — was measured 7 % slower and started every level with a cold L2).
__device__ __forceinline__ fe load_wire(const fe* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    fe r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long v = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.l[2 * k] = (uint32_t)v; r.l[2 * k + 1] = (uint32_t)(v >> 32);
    }
    return r;
}
__device__ __forceinline__ void store_wire(fe* p, const fe& v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
#pragma unroll
    for (int k = 0; k < 4; k++) __hip_atomic_store(q + k, (unsigned long long)v.l[2 * k] | ((unsigned long long)v.l[2 * k + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// false: the barrier was abandoned (abort bit), the caller returns
// Ordering (HSA / LLVM AMDGPU memory model, agent scope).  Producer side: every wave first waits until its own stores have been
// acknowledged (s_waitcnt vmcnt(0): wire values are sc1 write-through stores, so "acknowledged" means they have reached the memory
// side all XCDs share — a workgroup barrier alone does not wait for stores in flight), then the workgroup barrier collects the waves,
// then ONE thread arrives with a RELEASE read-modify-write.  Consumer side: the poll is relaxed (no cache maintenance per poll) and
// is followed by an ACQUIRE fence before the workgroup is let go, so no wire load of the next level can be satisfied by a line that
// was cached before the arrivals were seen.
__device__ __forceinline__ bool few_grid_barrier(uint32_t* sync, uint32_t target, unsigned long long* stamps, uint32_t poll_limit) {
    __shared__ uint32_t s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's wire stores are complete at the memory side
    __syncthreads();
    if (threadIdx.x == 0) {
        if (stamps) stamps[2] = stamps[3] = wall_clock64();
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t polls = 0, seen;
        while ((seen = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if (++polls > poll_limit) { seen = __hip_atomic_fetch_or(sync, FEW_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | FEW_ABORT; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stamps) stamps[4] = wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (stamps) stamps[5] = wall_clock64();
        s_ok = (seen & FEW_ABORT) ? 0u : 1u;
    }
    __syncthreads();
    return s_ok != 0;
}

// One op's descriptor (wave-uniform: scalar registers) and the first FEW_SLOTS x 64 of its terms, lane t holding terms t, t + 64,
// ..: static data, fetched for a wave's next item before the barrier so that only the wire values remain to be loaded after it.
constexpr int FEW_SLOTS = 3;             // ChaCha20's add32 rows have 96 + 1 + 33 terms
struct FewFetch { uint32_t d[8]; uint2 cw[FEW_SLOTS]; };
__device__ __forceinline__ FewFetch few_fetch(const SolverFewArgs& a, uint32_t i, uint32_t lane) {
    FewFetch f;
    const uint32_t* d = a.ops + 8 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 8; k++) f.d[k] = d[k];
    const uint2* terms = reinterpret_cast<const uint2*>(a.terms) + f.d[4];          // the term list is padded by FEW_SLOTS x 64 pairs
#pragma unroll
    for (int j = 0; j < FEW_SLOTS; j++) f.cw[j] = terms[64 * j + lane];
    return f;
}
__device__ __forceinline__ fe few_wave_sum(fe sum) {
#pragma unroll 1
    for (int m = 32; m >= 1; m >>= 1) {
        fe o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.l[k] = (uint32_t)__shfl_xor((int)sum.l[k], m);
        sum = Fr::add(sum, o);
    }
    return sum;
}
__device__ __forceinline__ fe pick_slot(const fe (&v)[FEW_SLOTS], uint32_t j) {
    fe r = v[0];
#pragma unroll
    for (int k = 1; k < FEW_SLOTS; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = j == (uint32_t)k ? v[k].l[i] : r.l[i];
    }
    return r;
}

// 1 / a for the latency kernel (a != 0, Montgomery in and out): Kaliski's almost-inverse — a binary Euclid whose cofactors are only
// shifted and added, never reduced, giving (a R)^-1 2^k with 254 <= k <= 508 — followed by 512 - k modular doublings:
// (a R)^-1 2^k 2^(512-k) = a^-1 R.  ~25 k instructions instead of the 128 k of Fermat's a^(r-2) (254 squarings on a lone wave:
// 0.43 ms, and AES-V2's log-derivative argument puts 400 divisions in a level).  Variable time, like gnark-crypto's Inverse; the
// batch kernel keeps the constant-shape power (its lanes hold different values).
__device__ __noinline__ fe few_inverse(const fe& a_mont) {
    fe u, v = a_mont, r = Fr::zero(), s = Fr::zero();
#pragma unroll
    for (int i = 0; i < 8; i++) u.l[i] = FrParams::mod(i);
    s.l[0] = 1;
    auto shr1 = [](fe& t) { for (int i = 0; i < 7; i++) t.l[i] = (t.l[i] >> 1) | (t.l[i + 1] << 31); t.l[7] >>= 1; };
    auto shl1 = [](fe& t) { for (int i = 7; i > 0; i--) t.l[i] = (t.l[i] << 1) | (t.l[i - 1] >> 31); t.l[0] <<= 1; };
    auto gt = [](const fe& a, const fe& b) { for (int i = 7; i >= 0; i--) { if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; } return false; };
    auto sub_plain = [](fe& a, const fe& b) { uint64_t br = 0; for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)a.l[i] - b.l[i] - br; a.l[i] = (uint32_t)d; br = (d >> 32) & 1; } };
    auto add_plain = [](fe& a, const fe& b) { uint64_t c = 0; for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; a.l[i] = (uint32_t)c; c >>= 32; } };      // cofactors stay below 2 r < 2^255
    uint32_t k = 0;
#pragma unroll 1
    while (!Fr::is_zero(v) && k < 512) {
        if (!(u.l[0] & 1u)) { shr1(u); shl1(s); }
        else if (!(v.l[0] & 1u)) { shr1(v); shl1(r); }
        else if (gt(u, v)) { sub_plain(u, v); shr1(u); add_plain(r, s); shl1(s); }
        else { sub_plain(v, u); shr1(v); add_plain(s, r); shl1(r); }
        k++;
    }
    fe m;
#pragma unroll
    for (int i = 0; i < 8; i++) m.l[i] = FrParams::mod(i);
    if (!gt(m, r)) sub_plain(r, m);                             // r < 2 r_mod
    fe x = Fr::neg(r);                                          // r_mod - r = (a R)^-1 2^k
#pragma unroll 1
    for (uint32_t i = k; i < 512; i++) x = Fr::dbl(x);
    return x;
}

template <bool HAS_DIV>
__device__ __forceinline__ void few_item(const SolverFewArgs& a, const FewFetch& f, uint32_t i, uint32_t p, uint32_t lane, unsigned long long* stamps = nullptr) {
    const uint32_t w0 = f.d[0], w1 = f.d[1], w2 = f.d[2], w3 = f.d[3], toff = f.d[4], n0 = f.d[5], n1 = f.d[6], n2 = f.d[7];
    const uint32_t op = w0 & 0xFF, T = n0 + n1 + n2;
    const size_t batch = a.batch;
    constexpr uint32_t CHUNK = 64 * FEW_SLOTS;
    fe va = Fr::zero(), vb = va, vc = va;
    const uint2* terms = reinterpret_cast<const uint2*>(a.terms) + toff;
#pragma unroll 1
    for (uint32_t t0 = 0; t0 < T; t0 += CHUNK) {
        const uint32_t nslot = T - t0 > CHUNK ? (uint32_t)FEW_SLOTS : (T - t0 + 63) / 64;       // wave-uniform: most ops have a handful of terms
        // coefficient x wire value, one term per (lane, slot).  Nearly every wire of these circuits is a bit and most coefficients
        // are +-1: such a product is a select; the terms that need the 353-instruction Montgomery multiplication are queued per lane
        fe cf[FEW_SLOTS], x[FEW_SLOTS], prod[FEW_SLOTS];
        uint32_t cid[FEW_SLOTS];
#pragma unroll
        for (int j = 0; j < FEW_SLOTS; j++) {
            if ((uint32_t)j < nslot) {
                const uint2 cw = t0 ? terms[t0 + 64 * j + lane] : f.cw[j];
                const bool live = t0 + 64 * j + lane < T;
                // a dead lane multiplies coefficient 0 (the constant 0) by coefficient 1 (the constant 1): ids checked at InitAlgorithm
                cid[j] = live ? cw.x : 0u;
                const bool cst = !live || cw.y == WIRE_CONST;
                cf[j] = load_fe(a.coeff + cid[j]);
                x[j] = load_wire(cst ? a.coeff + 1 : a.W + (size_t)cw.y * batch + p);       // one branch-free load: no early wait
            }
        }
        uint32_t hard = 0;
#pragma unroll
        for (int j = 0; j < FEW_SLOTS; j++) {
            prod[j] = Fr::zero();
            if ((uint32_t)j < nslot) {
                const bool z = Fr::is_zero(x[j]), o = Fr::eq(x[j], Fr::one());
                const fe nx = Fr::neg(x[j]);
#pragma unroll
                for (int k = 0; k < 8; k++) prod[j].l[k] = z ? 0u : o ? cf[j].l[k] : cid[j] == 1 ? x[j].l[k] : nx.l[k];
                if (!(z || o || cid[j] == 1 || cid[j] == 3)) hard |= 1u << j;
            }
        }
#pragma unroll 1
        while (__builtin_amdgcn_ballot_w64(hard != 0) != 0) {          // usually not at all, else once
            const uint32_t j = hard ? (uint32_t)__builtin_ctz(hard) : 0u;
            const fe m = Fr::mul(pick_slot(cf, j), pick_slot(x, j));
#pragma unroll
            for (int q = 0; q < FEW_SLOTS; q++) {
                const bool take = hard != 0 && j == (uint32_t)q;
#pragma unroll
                for (int k = 0; k < 8; k++) prod[q].l[k] = take ? m.l[k] : prod[q].l[k];
            }
            hard &= hard - 1;
        }
        if (stamps && lane == 0) stamps[6] = wall_clock64() + (prod[0].l[0] & prod[FEW_SLOTS - 1].l[0] & 0u);
        // per expression: a handful of terms are read lane by lane; more are masked into a per-lane value and folded by a butterfly —
        // the butterflies of L and O (ChaCha20's add32 rows: 96 and 33 terms) run as one loop, two independent chains interleaved
        fe ls[3]; uint32_t longm = 0;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t s = e == 0 ? 0u : e == 1 ? n0 : n0 + n1, n = e == 0 ? n0 : e == 1 ? n1 : n2;
            // the expression's terms inside this chunk: [lo, hi) relative to t0
            const uint32_t lo = s > t0 ? s - t0 : 0u, hi = s + n > t0 ? (s + n - t0 < CHUNK ? s + n - t0 : CHUNK) : 0u;
            ls[e] = Fr::zero();
            if (hi <= lo) continue;
            if (hi - lo <= 6) {
                fe sum = readlane_fe(pick_slot(prod, lo >> 6), lo & 63);
#pragma unroll 1
                for (uint32_t k = lo + 1; k < hi; k++) sum = Fr::add(sum, readlane_fe(pick_slot(prod, k >> 6), k & 63));
                if (e == 0) va = Fr::add(va, sum); else if (e == 1) vb = Fr::add(vb, sum); else vc = Fr::add(vc, sum);
            } else {
                longm |= 1u << e;
#pragma unroll
                for (int j = 0; j < FEW_SLOTS; j++) {
                    if (hi > 64u * j && lo < 64u * (j + 1)) {            // wave-uniform
                        const bool in = lane + 64u * j >= lo && lane + 64u * j < hi;
                        fe t;
#pragma unroll
                        for (int k = 0; k < 8; k++) t.l[k] = in ? prod[j].l[k] : 0u;
                        ls[e] = Fr::add(ls[e], t);
                    }
                }
            }
        }
        if ((longm & 5u) == 5u) {
#pragma unroll 1
            for (int m = 32; m >= 1; m >>= 1) {
                fe o0, o2;
#pragma unroll
                for (int k = 0; k < 8; k++) { o0.l[k] = (uint32_t)__shfl_xor((int)ls[0].l[k], m); o2.l[k] = (uint32_t)__shfl_xor((int)ls[2].l[k], m); }
                ls[0] = Fr::add(ls[0], o0); ls[2] = Fr::add(ls[2], o2);
            }
            va = Fr::add(va, ls[0]); vc = Fr::add(vc, ls[2]);
            longm &= ~5u;
        }
        if (longm & 1u) va = Fr::add(va, few_wave_sum(ls[0]));
        if (longm & 2u) vb = Fr::add(vb, few_wave_sum(ls[1]));
        if (longm & 4u) vc = Fr::add(vc, few_wave_sum(ls[2]));
    }
    bool bad = false;
    if (stamps && lane == 0) stamps[7] = wall_clock64() + (va.l[0] & vb.l[0] & vc.l[0] & 0u);
    if (op == OP_R1C) {                                 // same rules as k_solver's OP_R1C
        const uint32_t loc = w0 >> 8, cidx = w1, uw = w2, uc = w3;
        fe ab;
        {
            const fe one = Fr::one();
            const bool a0 = Fr::is_zero(va), a1 = Fr::eq(va, one), b0 = Fr::is_zero(vb), b1 = Fr::eq(vb, one);      // the values are wave-uniform here
            if (b0 || b1) {
#pragma unroll
                for (int k = 0; k < 8; k++) ab.l[k] = b1 ? va.l[k] : 0u;
            } else if (a0 || a1) {
#pragma unroll
                for (int k = 0; k < 8; k++) ab.l[k] = a1 ? vb.l[k] : 0u;
            } else ab = Fr::mul(va, vb);
        }
        if (stamps && lane == 0) stamps[8] = wall_clock64() + (ab.l[0] & 0u);
        if (loc == 0) bad = !Fr::eq(ab, vc);
        else {
            fe wire;
            if (loc == 3) { wire = Fr::sub(ab, vc); vc = ab; }
            else {
                const fe known = loc == 1 ? vb : va;
                fe part = loc == 1 ? va : vb;
                if (Fr::is_zero(known)) { wire = Fr::zero(); bad = !Fr::eq(ab, vc); }
                else if (HAS_DIV) { wire = Fr::sub(Fr::mul(vc, few_inverse(known)), part); part = Fr::add(part, wire); }
                else { wire = Fr::zero(); bad = true; }
                if (loc == 1) va = part; else vb = part;
            }
            if (uc == 3) wire = Fr::neg(wire);
            else if (uc != 1) wire = Fr::mul(wire, load_fe(a.coeff_inv + uc));
            if (stamps && lane == 0) stamps[9] = wall_clock64() + (wire.l[0] & 0u);
            if (lane == 0) store_wire(a.W + (size_t)uw * batch + p, wire);
        }
        {       // one store instruction for the three rows: lanes 0..2
            fe out;
#pragma unroll
            for (int k = 0; k < 8; k++) out.l[k] = lane == 0 ? va.l[k] : lane == 1 ? vb.l[k] : vc.l[k];
            fe* row = lane == 0 ? a.A : lane == 1 ? a.B : a.C;
            if (lane < 3) store_fe(row + (size_t)cidx * batch + p, out);
        }
        if (stamps && lane == 0) stamps[10] = wall_clock64();
    } else if (op == OP_NBITS) {
        if (stamps && lane == 0) stamps[11] = wall_clock64();
        const fe r = Fr::from_mont(va);
        const fe one = Fr::one();
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < w2; k0 += 64) {                   // lane k: bit k0 + k
            const uint32_t k = k0 + lane;
            const bool bit = k < 256 && ((limb_at(r, (k >> 5) & 7) >> (k & 31)) & 1u);
            fe v;
#pragma unroll
            for (int j = 0; j < 8; j++) v.l[j] = bit ? one.l[j] : 0u;
            if (k < w2) store_wire(a.W + (size_t)(w1 + k) * batch + p, v);
        }
    } else if (op == OP_LOOKUP) {
        const fe r = Fr::from_mont(va);
        const uint32_t hi = r.l[1] | r.l[2] | r.l[3] | r.l[4] | r.l[5] | r.l[6] | r.l[7];
        if (hi != 0 || r.l[0] >= 256) bad = true;
        const uint32_t cid = a.lookup_coeff[w2 * 256 + (r.l[0] & 255)];
        const fe v = load_fe(a.coeff + cid);
        if (lane == 0) store_wire(a.W + (size_t)w1 * batch + p, v);
    } else if (op == OP_RANDOMIZE || op == OP_COMMIT) {
        const fe* src = op == OP_RANDOMIZE ? a.mask : a.commit;
        const fe v = src ? load_fe(src + p) : Fr::zero();
#pragma unroll 1
        for (uint32_t k = lane; k < w2; k += 64) store_wire(a.W + (size_t)(w1 + k) * batch + p, v);
    }
    if (bad && lane == 0) atomicMin(a.status + p, i);
}

template <bool HAS_DIV>
__global__ __launch_bounds__(64 * FEW_WAVES) void k_solver_few(SolverFewArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = wave * gridDim.x + blockIdx.x, nw = gridDim.x * FEW_WAVES;       // neighbouring items (the long ops come first) go to different CUs
    // an earlier resident launch of this call gave up (sync[1], written before this launch started): the host will solve the call
    // again level by level, so this launch must not spin through the same timeouts
    if (__hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    uint32_t epoch = 0;
    auto op_of = [&](uint32_t l0, uint32_t it) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + it / a.n_real)); };
    uint32_t l0 = a.level_start[a.from], l1 = a.level_start[a.from + 1];
    FewFetch f = few_fetch(a, op_of(l0, gw < (l1 - l0) * a.n_real ? gw : 0u), lane);       // (a wave without an item fetches op 0: harmless)
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) { a.trace[16 * a.nlev_trace + 0] = wall_clock64(); a.trace[16 * a.nlev_trace + 1] = clock64(); }
    for (uint32_t lev = a.from; lev < a.to; lev++) {
        const uint32_t items = (l1 - l0) * a.n_real;
        unsigned long long* stamps = a.trace && blockIdx.x == 0 ? a.trace + 16 * lev : nullptr;       // workgroup 0 only: it holds the level's first (longest) op
        if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();
        for (uint32_t it = gw; it < items; it += nw) {
            const uint32_t i = op_of(l0, it);
            const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)(it % a.n_real));
            if (it != gw) f = few_fetch(a, i, lane);
            few_item<HAS_DIV>(a, f, i, p, lane, it == 0 ? stamps : nullptr);
        }
        if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
        if (a.trace && lane == 0 && gw < items) atomicMax(a.trace + 16 * lev + 12, wall_clock64());
        if (lev + 1 == a.to) { if (stamps && threadIdx.x == 0) { a.trace[16 * a.nlev_trace + 2] = wall_clock64(); a.trace[16 * a.nlev_trace + 3] = clock64(); } break; }
        l0 = l1; l1 = a.level_start[lev + 2];
        f = few_fetch(a, op_of(l0, gw < (l1 - l0) * a.n_real ? gw : 0u), lane);            // static data of the next level's item: in flight across the barrier
        if (!few_grid_barrier(a.sync, ++epoch * (gridDim.x + a.test_missing), stamps, a.poll_limit)) {
            if (threadIdx.x < a.n_real && blockIdx.x == 0) atomicMin(a.status + threadIdx.x, 0u);      // unsatisfied, unless the host solves the call again:
            if (threadIdx.x == 0) atomicOr(a.sync + 1, 1u);                                              // sync[1] tells it that the launch gave up
            return;
        }
    }
}

// OP_COUNT (logderivarg.countHint): out[i] = number of query rows equal to table row i.  One wave per (64 proofs, op); the
// 256 x 64-lane histogram lives in LDS (32 KiB).  Table rows are constants (index i, value T[i]) — shape checked on the host,
// index column checked by k_check_count_tables at InitAlgorithm — so a query is matched by reading row `index` directly.
constexpr uint32_t COUNT_WAVES = 8;      // waves sharing one histogram: the queries (two dependent wire loads each) are dealt round-robin
__global__ __launch_bounds__(64 * COUNT_WAVES) void k_solver_count(SolverArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];     // [row][lane]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const size_t batch = a.batch;
    const uint32_t nlev = a.sched[0];
    const uint32_t* lstart = a.sched + 1;
    const uint32_t* ops = a.sched + 2 + nlev;
    const uint32_t i = lstart[a.first_level] + blockIdx.y;
    if (i >= lstart[a.first_level + 1]) return;                         // (uniform over the workgroup)
    Window win{a.prog, 0, 0, lane};
    const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)ops[i]);
    win.load(at);
    const uint32_t o0 = win.get(at + 1), ntab = win.get(at + 2), nq = win.get(at + 4);
    for (uint32_t r = wave; r < ntab; r += COUNT_WAVES) s_cnt[r * 64 + lane] = 0;
    __syncthreads();
    const uint32_t rows_base = at + 5;
    uint32_t q = rows_base + 6 * ntab;
    bool bad = false;
#pragma unroll 1
    for (uint32_t k = 0; k < nq; k++) {
        if (k % COUNT_WAVES != wave) {                                   // somebody else's query: step over its two expressions
            q += 1 + 2 * win.get(q);
            q += 1 + 2 * win.get(q);
            continue;
        }
        uint32_t next;
        fe x0 = eval_expr(win, q, a.coeff, a.W, batch, p, next); q = next;
        fe x1 = eval_expr(win, q, a.coeff, a.W, batch, p, next); q = next;
        const fe c0 = Fr::from_mont(x0);
        const uint32_t hi = c0.l[1] | c0.l[2] | c0.l[3] | c0.l[4] | c0.l[5] | c0.l[6] | c0.l[7];
        const bool in_range = hi == 0 && c0.l[0] < ntab;
        const uint32_t idx = in_range ? c0.l[0] : 0u;
        const uint32_t cidv = a.prog[rows_base + 6 * idx + 4];          // value column of table row idx
        const fe tv = load_fe(a.coeff + cidv);
        if (in_range && Fr::eq(x1, tv)) atomicAdd(&s_cnt[idx * 64 + lane], 1u);
        else bad = true;                                                  // gnark: "query not in table"
    }
    __syncthreads();
#pragma unroll 1
    for (uint32_t r = wave; r < ntab; r += COUNT_WAVES) {
        const uint32_t c = s_cnt[r * 64 + lane];
        fe v = Fr::zero();
        if (__builtin_amdgcn_ballot_w64(c != 0) != 0) v = Fr::from_u32(c);
        store_fe(a.W + (size_t)(o0 + r) * batch + p, v);
    }
    if (bad) atomicMin(a.status + p, i);
}
// The same for calls with a handful of statements: one workgroup per (op, statement), lanes = QUERIES (their positions in the instruction
// words come from the host: FewProgram::count_qoff), one 32-bit LDS counter per table row.  0.4 ms -> 0.05 ms per level for one statement.
constexpr uint32_t COUNT_FEW_THREADS = 1024;
__device__ __forceinline__ fe count_few_expr(const uint32_t* prog, uint32_t q, const fe* coeff, const fe* W, size_t batch, size_t p, uint32_t& next) {
    const uint32_t n = prog[q];
    fe acc = Fr::zero();
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t cid = prog[q + 1 + 2 * k], wid = prog[q + 2 + 2 * k];
        const fe cf = load_fe(coeff + cid);
        if (wid == WIRE_CONST) acc = Fr::add(acc, cf);
        else {
            const fe v = load_fe(W + (size_t)wid * batch + p);
            acc = Fr::add(acc, cid == 1 ? v : Fr::mul(cf, v));
        }
    }
    next = q + 1 + 2 * n;
    return acc;
}
__global__ __launch_bounds__(COUNT_FEW_THREADS) void k_solver_count_few(SolverArgs a, const uint32_t* count_ops, const uint32_t* count_qoff, uint32_t first_op) {
    __shared__ uint32_t s_cnt[256];
    const uint32_t* d = count_ops + 4 * (size_t)(first_op + blockIdx.x);
    const uint32_t at = d[0], qbase = d[1], nq = d[2];
    const size_t p = blockIdx.y, batch = a.batch;
    const uint32_t o0 = a.prog[at + 1], ntab = a.prog[at + 2], rows_base = at + 5;
    for (uint32_t r = threadIdx.x; r < ntab; r += COUNT_FEW_THREADS) s_cnt[r] = 0;
    __syncthreads();
    bool bad = false;
    for (uint32_t k = threadIdx.x; k < nq; k += COUNT_FEW_THREADS) {
        uint32_t q = count_qoff[qbase + k], next;
        const fe x0 = count_few_expr(a.prog, q, a.coeff, a.W, batch, p, next);
        const fe x1 = count_few_expr(a.prog, next, a.coeff, a.W, batch, p, next);
        const fe c0 = Fr::from_mont(x0);
        const uint32_t hi = c0.l[1] | c0.l[2] | c0.l[3] | c0.l[4] | c0.l[5] | c0.l[6] | c0.l[7];
        const bool in_range = hi == 0 && c0.l[0] < ntab;
        const uint32_t idx = in_range ? c0.l[0] : 0u;
        const fe tv = load_fe(a.coeff + a.prog[rows_base + 6 * idx + 4]);      // value column of table row idx
        if (in_range && Fr::eq(x1, tv)) atomicAdd(&s_cnt[idx], 1u);
        else bad = true;                                                      // gnark: "query not in table"
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < ntab; r += COUNT_FEW_THREADS) store_fe(a.W + (size_t)(o0 + r) * batch + p, Fr::from_u32(s_cnt[r]));
    if (bad) atomicMin(a.status + p, 0u);
}
// InitAlgorithm-time check for k_solver_count: the index constant of table row r must be r.
__global__ void k_check_count_tables(const uint32_t* prog, const fe* coeff, const uint32_t* count_ops, uint32_t nops, uint32_t* flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t op = t >> 8, row = t & 255;
    if (op >= nops) return;
    const uint32_t at = count_ops[op];
    if (row >= prog[at + 2]) return;
    const uint32_t cid = prog[at + 5 + 6 * row + 1];
    if (!Fr::eq(load_fe(coeff + cid), Fr::from_u32(row))) atomicOr(flag, 1u);
}

__global__ void k_assign_chacha(const uint8_t* inputs, fe* W, size_t batch) {
    // 44 words per proof: Counter, Nonce[3] (LE), In[16] (BE), Out[16] (BE) public; Key[8] (LE) secret
    size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= 45 * batch) return;
    size_t p = idx % batch; uint32_t w = (uint32_t)(idx / batch);
    const fe one = Fr::one(), zero = Fr::zero();
    if (w == 44) { store_fe(W + p, one); return; }     // wire 0 = ONE
    const uint8_t* rec = inputs + 176 * p;
    const uint8_t *key = rec, *nonce = rec + 32, *ctr = rec + 44, *pt = rec + 48, *ct = rec + 112;
    const uint8_t* src; bool be;
    if (w == 0) { src = ctr; be = false; }
    else if (w < 4) { src = nonce + 4 * (w - 1); be = false; }
    else if (w < 20) { src = pt + 4 * (w - 4); be = true; }
    else if (w < 36) { src = ct + 4 * (w - 20); be = true; }
    else { src = key + 4 * (w - 36); be = false; }
    uint32_t v = be ? ((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3]
                    : ((uint32_t)src[3] << 24) | ((uint32_t)src[2] << 16) | ((uint32_t)src[1] << 8) | src[0];
    for (uint32_t b = 0; b < 32; b++) store_fe(W + (size_t)(1 + 32 * w + b) * batch + p, ((v >> b) & 1u) ? one : zero);
}

__global__ void k_assign_aes(const uint8_t* inputs, int keylen, fe* W, size_t batch) {
    // Nonce[12], Counter, Plaintext[64], Ciphertext[64] public; Key[keylen] secret
    const uint32_t nvals = 141 + keylen;
    size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= (size_t)(nvals + 1) * batch) return;
    size_t p = idx % batch; uint32_t i = (uint32_t)(idx / batch);
    if (i == nvals) { store_fe(W + p, Fr::one()); return; }
    const uint8_t* rec = inputs + 176 * p;
    const uint8_t *key = rec, *nonce = rec + 32, *ctr = rec + 44, *pt = rec + 48, *ct = rec + 112;
    uint32_t v;
    if (i < 12) v = nonce[i];
    else if (i == 12) v = (uint32_t)ctr[0] | ((uint32_t)ctr[1] << 8) | ((uint32_t)ctr[2] << 16) | ((uint32_t)ctr[3] << 24);
    else if (i < 77) v = pt[i - 13];
    else if (i < 141) v = ct[i - 77];
    else v = key[i - 141];
    store_fe(W + (size_t)(1 + i) * batch + p, Fr::from_u32(v));
}

__global__ void k_prep_rs(const uint8_t* rs, fe* W, size_t n_wires, size_t batch, const uint8_t* mask_in, fe* mask_out) {
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= batch) return;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(rs + 64 * p);
    fe r, s;
    for (int i = 0; i < 8; i++) { r.l[i] = q[i]; s.l[i] = q[8 + i]; }
    r = Fr::to_mont(r); s = Fr::to_mont(s);
    store_fe(W + (n_wires + 0) * batch + p, r);
    store_fe(W + (n_wires + 1) * batch + p, s);
    store_fe(W + (n_wires + 2) * batch + p, Fr::neg(Fr::mul(r, s)));
    store_fe(W + (n_wires + 3) * batch + p, Fr::zero());
    if (mask_in) {
        const uint32_t* mq = reinterpret_cast<const uint32_t*>(mask_in + 32 * p);
        fe m; for (int i = 0; i < 8; i++) m.l[i] = mq[i];
        store_fe(mask_out + p, Fr::to_mont(m));
    }
}

}  // namespace

void launch_assign_chacha(const uint8_t* inputs, fe* W, size_t batch, hipStream_t s) {
    size_t n = 45 * batch;
    hipLaunchKernelGGL(k_assign_chacha, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, inputs, W, batch);
}
void launch_assign_aes(const uint8_t* inputs, int keylen, fe* W, size_t batch, hipStream_t s) {
    size_t n = (size_t)(142 + keylen) * batch;
    hipLaunchKernelGGL(k_assign_aes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, inputs, keylen, W, batch);
}
void launch_prep_rs(const uint8_t* rs, fe* W, size_t n_wires, size_t batch, const uint8_t* mask_in, fe* mask_out, hipStream_t s) {
    hipLaunchKernelGGL(k_prep_rs, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, rs, W, n_wires, batch, mask_in, mask_out);
}
// TEST HOOK kernel: the batch solver's wave-wide inversion on canonical inputs, canonical outputs; 0 stays 0 (the solver lends such a lane a 1)
namespace {
__global__ __launch_bounds__(64) void k_wave_inverse(const fe* a, fe* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    fe x = i < n ? Fr::to_mont(load_fe(a + i)) : Fr::one();
    const bool z = Fr::is_zero(x);
    if (z) x = Fr::one();
    const fe r = wave_batch_inverse(x);
    if (i < n) store_fe(out + i, z ? Fr::zero() : Fr::from_mont(r));
}
}  // namespace
void launch_wave_inverse(const fe* a, fe* out, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_wave_inverse, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a, out, n);
}

void launch_solver_count_level(const SolverArgs& a, uint32_t level_width, hipStream_t s) {
    if (!level_width) return;
    hipLaunchKernelGGL(k_solver_count, dim3((unsigned)(a.batch / 64), level_width), dim3(64 * COUNT_WAVES), 256 * 64 * sizeof(uint32_t), s, a);
}
void launch_solver_count_few(const SolverArgs& a, const uint32_t* count_ops, const uint32_t* count_qoff, uint32_t first_op, uint32_t level_width, size_t nproofs, hipStream_t s) {
    if (!level_width) return;
    hipLaunchKernelGGL(k_solver_count_few, dim3(level_width, (unsigned)nproofs), dim3(COUNT_FEW_THREADS), 0, s, a, count_ops, count_qoff, first_op);
}
void launch_check_count_tables(const uint32_t* prog, const fe* coeff, const uint32_t* count_ops, uint32_t nops, uint32_t* flag, hipStream_t s) {
    if (!nops) return;
    hipLaunchKernelGGL(k_check_count_tables, dim3(nops), dim3(256), 0, s, prog, coeff, count_ops, nops, flag);
}
void launch_solver_few(const SolverFewArgs& a, int has_div, uint32_t workgroups, hipStream_t s) {
    if (a.from >= a.to) return;
    // 96 KiB of (unused) dynamic LDS: more than half of a CU's 160 KiB, so that no two workgroups share a CU and every wave has a
    // SIMD nearly to itself — the level time is the longest op's serial instruction stream
    constexpr size_t lds = 96 * 1024;
    static std::atomic<int> state[64];                       // per device (function attributes are per device): 0 not asked yet, 1 granted, 2 refused
    int dev = 0; (void)hipGetDevice(&dev);
    std::atomic<int>& st = state[dev & 63];
    if (st.load() == 0) {
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_solver_few<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(k_solver_few<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        st.store(ok ? 1 : 2);
    }
    const bool attr = st.load() == 1;
    const size_t dyn = attr ? lds : 0;
    if (has_div) hipLaunchKernelGGL(k_solver_few<true>, dim3(workgroups), dim3(64 * FEW_WAVES), dyn, s, a);
    else hipLaunchKernelGGL(k_solver_few<false>, dim3(workgroups), dim3(64 * FEW_WAVES), dyn, s, a);
}
void launch_solver_level(const SolverArgs& a, uint32_t level_width, hipStream_t s) {
    if (!level_width) return;
    const uint32_t n_short = level_width - a.n_long;
    const uint32_t wpb = a.batch <= 2048 ? 8 : 4;      // measured: 8 waves cut the witness time by 12 % at batch <= 1024 and cost 15 % at 8192
    const dim3 grid((unsigned)(a.batch / 64), a.n_long + (n_short + wpb - 1) / wpb), block(64 * wpb);
    // the division-free variant (ChaCha20-V3 never divides) carries no call to the inversion routine and so needs no
    // scratch memory: a kernel with scratch pays a per-dispatch setup that dominated the 163 short level launches
    if (a.has_div) { if (wpb == 8) hipLaunchKernelGGL((k_solver<true, 8>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_solver<true, 4>), grid, block, 0, s, a); }
    else { if (wpb == 8) hipLaunchKernelGGL((k_solver<false, 8>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_solver<false, 4>), grid, block, 0, s, a); }
}

}  // namespace gsc
