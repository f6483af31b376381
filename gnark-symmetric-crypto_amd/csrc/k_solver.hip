// R1CS witness generation on the GPU: one wavefront lane per proof, one uniform instruction stream.
//
// Replaces the cs.Solve step inside groth16.Prove (reference libraries/prover/impl/provers.go:148,216;
// gnark constraint/bn254 solver — SURVEY.md §8(a) a5, semantics App. C) and the witness assignment of
// provers.go:106-142 / :194-210 (bit / byte layout: utils/bytes.go:11-47).
//
// Layout: W[wire][proof], A/B/C[constraint][proof], 32-byte elements: the 64 lanes of a wave read and write
// 2 KiB contiguous per access (coalesced), and the instruction words are wave-uniform (scalar registers).
#include "kernels.hpp"
#include "formats.hpp"

namespace gsc {
using namespace bn254;

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct SolverCtx {
    const fe* coeff; const fe* W; size_t batch; size_t p;
};

// sum of n terms (coeff id, wire id); wave-uniform control flow
__device__ __forceinline__ fe eval_terms(const uint32_t* t, uint32_t n, const fe* coeff, const fe* W, size_t batch, size_t p) {
    fe acc = Fr::zero();
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t cid = uni(t[2 * k]), wid = uni(t[2 * k + 1]);
        if (wid == WIRE_CONST) { acc = Fr::add(acc, load_fe(coeff + cid)); continue; }
        fe w = load_fe(W + (size_t)wid * batch + p);
        // gnark reserves coefficient ids 0..4 for 0, 1, 2, -1, -2 (checked on the host at InitAlgorithm)
        if (cid == 1) acc = Fr::add(acc, w);
        else if (cid == 3) acc = Fr::sub(acc, w);
        else if (cid == 2) acc = Fr::add(acc, Fr::dbl(w));
        else if (cid == 4) acc = Fr::sub(acc, Fr::dbl(w));
        else if (cid != 0) acc = Fr::add(acc, Fr::mul(load_fe(coeff + cid), w));
    }
    return acc;
}

__global__ __launch_bounds__(64) void k_solver(SolverArgs a) {
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t batch = a.batch;
    const uint32_t* pc = a.prog + a.first_word;
    uint32_t fail = a.resume ? a.status[p] : 0u;
    uint32_t op_index = 0;
    bool first = true;
    for (;; op_index++) {
        const uint32_t hdr = uni(pc[0]);
        const uint32_t op = hdr & 0xFF, len = hdr >> 8;
        if (op == OP_END) break;
        if (op == OP_COMMIT && !(first && a.resume)) break;      // the host finishes the commitment, then resumes here
        first = false;
        if (op == OP_R1C) {
            const uint32_t loc = uni(pc[1]), nL = uni(pc[2]), nR = uni(pc[3]), nO = uni(pc[4]);
            const uint32_t cidx = uni(pc[5]), uw = uni(pc[6]), uc = uni(pc[7]);
            const uint32_t* t = pc + 8;
            fe va = eval_terms(t, nL, a.coeff, a.W, batch, p);
            fe vb = eval_terms(t + 2 * nL, nR, a.coeff, a.W, batch, p);
            fe vc = eval_terms(t + 2 * (nL + nR), nO, a.coeff, a.W, batch, p);
            if (loc == 0) {
                if (!Fr::eq(Fr::mul(va, vb), vc) && !fail) fail = 1 + op_index;
            } else {
                fe wire;
                if (loc == 3) { fe ab = Fr::mul(va, vb); wire = Fr::sub(ab, vc); vc = ab; }
                else {
                    fe known = loc == 1 ? vb : va;
                    fe part = loc == 1 ? va : vb;
                    if (Fr::is_zero(known)) {
                        wire = Fr::zero();
                        if (!Fr::eq(Fr::mul(va, vb), vc) && !fail) fail = 1 + op_index;
                    } else {
                        wire = Fr::sub(Fr::mul(vc, Fr::inv(known)), part);
                        part = Fr::add(part, wire);
                    }
                    if (loc == 1) va = part; else vb = part;
                }
                wire = Fr::mul(wire, load_fe(a.coeff_inv + uc));
                store_fe(a.W + (size_t)uw * batch + p, wire);
            }
            store_fe(a.A + (size_t)cidx * batch + p, va);
            store_fe(a.B + (size_t)cidx * batch + p, vb);
            store_fe(a.C + (size_t)cidx * batch + p, vc);
        } else if (op == OP_NBITS) {
            const uint32_t o0 = uni(pc[1]), nout = uni(pc[2]), nt = uni(pc[3]);
            fe v = Fr::from_mont(eval_terms(pc + 4, nt, a.coeff, a.W, batch, p));
            const fe one = Fr::one(), zero = Fr::zero();
            for (uint32_t k = 0; k < nout; k++) {
                const uint32_t bit = k < 256 ? (v.l[k >> 5] >> (k & 31)) & 1u : 0u;
                store_fe(a.W + (size_t)(o0 + k) * batch + p, bit ? one : zero);
            }
        } else if (op == OP_LOOKUP) {
            const uint32_t o0 = uni(pc[1]), nin = uni(pc[2]), table = uni(pc[3]);
            const uint32_t* q = pc + 4;
            for (uint32_t k = 0; k < nin; k++) {
                const uint32_t nt = uni(q[0]);
                fe v = Fr::from_mont(eval_terms(q + 1, nt, a.coeff, a.W, batch, p));
                q += 1 + 2 * nt;
                uint32_t hi = v.l[1] | v.l[2] | v.l[3] | v.l[4] | v.l[5] | v.l[6] | v.l[7];
                uint32_t idx = v.l[0];
                if ((hi != 0 || idx >= 256) && !fail) { fail = 1 + op_index; }
                idx &= 255;
                const uint32_t cid = a.lookup_coeff[table * 256 + idx];
                store_fe(a.W + (size_t)(o0 + k) * batch + p, load_fe(a.coeff + cid));
            }
        } else if (op == OP_COUNT) {
            // out[i] = number of query rows equal to table row i.  Rows are nvars linear expressions each.
            const uint32_t o0 = uni(pc[1]), ntab = uni(pc[2]), nvars = uni(pc[3]), nq = uni(pc[4]);
            const uint32_t* rows = pc + 5;
            // walk to the first query row
            const uint32_t* qrows = rows;
            for (uint32_t k = 0; k < ntab * nvars; k++) qrows += 1 + 2 * uni(qrows[0]);
            const uint32_t* tr = rows;
            for (uint32_t i = 0; i < ntab; i++) {
                fe trow[2];
                for (uint32_t v = 0; v < nvars && v < 2; v++) { const uint32_t nt = uni(tr[0]); trow[v] = eval_terms(tr + 1, nt, a.coeff, a.W, batch, p); tr += 1 + 2 * nt; }
                uint32_t cnt = 0;
                const uint32_t* qr = qrows;
                for (uint32_t q = 0; q < nq; q++) {
                    bool same = true;
                    for (uint32_t v = 0; v < nvars && v < 2; v++) { const uint32_t nt = uni(qr[0]); fe qv = eval_terms(qr + 1, nt, a.coeff, a.W, batch, p); qr += 1 + 2 * nt; same = same && Fr::eq(qv, trow[v]); }
                    cnt += same ? 1u : 0u;
                }
                store_fe(a.W + (size_t)(o0 + i) * batch + p, Fr::from_u32(cnt));
            }
        } else if (op == OP_RANDOMIZE) {
            const uint32_t o0 = uni(pc[1]), nout = uni(pc[2]);
            fe v = a.mask ? load_fe(a.mask + p) : Fr::zero();
            for (uint32_t k = 0; k < nout; k++) store_fe(a.W + (size_t)(o0 + k) * batch + p, v);
        } else if (op == OP_COMMIT) {
            const uint32_t o0 = uni(pc[1]), nout = uni(pc[2]);
            fe v = a.commit ? load_fe(a.commit + p) : Fr::zero();
            for (uint32_t k = 0; k < nout; k++) store_fe(a.W + (size_t)(o0 + k) * batch + p, v);
        }
        pc += len;
    }
    a.status[p] = fail;
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

__global__ void k_prep_rs(const uint8_t* rs, fe* W, size_t n_wires, size_t batch) {
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
void launch_prep_rs(const uint8_t* rs, fe* W, size_t n_wires, size_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_prep_rs, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, rs, W, n_wires, batch);
}
void launch_solver(const SolverArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_solver, dim3((unsigned)(a.batch / 64)), dim3(64), 0, s, a);
}

}  // namespace gsc
