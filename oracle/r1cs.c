/* TEST ORACLE — not product code (see bn254.h).
 *
 * gnark v0.11.0 R1CS file decoder and sequential solver, restated from SURVEY.md App. A / App. C.
 * Stands in for constraint.ConstraintSystem.ReadFrom (libraries/prover/impl/prove_impl.go:102-103)
 * and for the cs.Solve step inside groth16.Prove (libraries/prover/impl/provers.go:148,216); the
 * Go modules that implement them (gnark v0.11.0, ronanh/intcomp v1.1.0, fxamacker/cbor v2.7.0 —
 * go.mod:8,19,27) are not vendored, so the on-disk format is the authority.  Pinned by: every
 * constraint a*b=c holding on the shipped r1cs.* files, LEVELS being a permutation of the
 * instruction indices, and sha256(W), sha256(a|b|c) of SURVEY.md App. E.
 */
#include "r1cs.h"
#include <stdio.h>

static uint64_t rd64le(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
static uint32_t rd32le(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* intcomp-u32 stream (App. A): [BINPACK]? [VARBYTE]? trailer.  Appends decoded values to out.
 * Returns number of values, or -1. */
static long intcomp_u32(const uint32_t *w, size_t nw, uint32_t *out, size_t cap) {
    size_t p = 0, n_out = 0;
    if (nw == 0) return 0;
    if (w[0] >= 128) {
        if (nw < 3) return -1;
        uint32_t n = w[0], len = w[1], prev = w[2];
        if (n % 128 || len > nw || len < 3) return -1;
        size_t q = 3, end = len;
        for (uint32_t g = 0; g < n / 128; g++) {
            if (q >= end) return -1;
            uint32_t hdr = w[q++];
            for (int s = 0; s < 4; s++) {
                uint32_t hb = (hdr >> (24 - 8 * s)) & 0xFF;
                uint32_t zz = hb >> 7, bl = hb & 0x7F;
                if (bl > 32 || q + bl > end) return -1;
                for (int j = 0; j < 32; j++) {
                    uint32_t v = 0;
                    if (bl) {
                        size_t bit = (size_t)j * bl; size_t wi = bit / 32, sh = bit % 32;
                        uint64_t two = w[q + wi];
                        if (sh + bl > 32) two |= (uint64_t)w[q + wi + 1] << 32;
                        v = (uint32_t)((two >> sh) & (bl == 32 ? 0xFFFFFFFFull : ((1ull << bl) - 1)));
                    }
                    uint32_t delta = zz ? ((v >> 1) ^ (uint32_t)(0 - (v & 1))) : v;
                    prev += delta;
                    if (n_out >= cap) return -1;
                    out[n_out++] = prev;
                }
                q += bl;
            }
        }
        p = len;
    }
    if (p + 1 < nw) {           /* VARBYTE block (something besides the trailer remains) */
        uint32_t n = w[p], len = w[p + 1];
        if (n >= 128 || p + len > nw || len < 2) return -1;
        size_t nbytes = (size_t)(len - 2) * 4, bi = 0;
        uint32_t acc = 0;
        for (uint32_t k = 0; k < n; k++) {
            uint32_t v = 0; int shift = 0;
            for (;;) {
                if (bi >= nbytes) return -1;
                uint32_t word = w[p + 2 + bi / 4];
                uint8_t b = (uint8_t)(word >> (24 - 8 * (bi % 4))); bi++;
                v |= (uint32_t)(b & 0x7F) << shift; shift += 7;
                if (!(b & 0x80)) break;
                if (shift > 35) return -1;
            }
            acc += v;
            if (n_out >= cap) return -1;
            out[n_out++] = acc;
        }
    }
    return (long)n_out;
}

/* ---- minimal CBOR reader ---- */
typedef struct { const uint8_t *b; size_t n, i; int err; } cbor_t;
static uint64_t cb_head(cbor_t *c, int *major) {
    if (c->i >= c->n) { c->err = 1; *major = -1; return 0; }
    uint8_t ib = c->b[c->i++]; *major = ib >> 5; int ai = ib & 31; uint64_t v = 0;
    if (ai < 24) return ai;
    int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : -1;
    if (nb < 0 || c->i + nb > c->n) { c->err = 1; return 0; }
    for (int k = 0; k < nb; k++) v = (v << 8) | c->b[c->i++];
    return v;
}
static void cb_skip(cbor_t *c) {
    int m; uint64_t v = cb_head(c, &m);
    if (c->err) return;
    switch (m) {
    case 0: case 1: case 7: return;
    case 2: case 3: if (c->i + v > c->n) { c->err = 1; return; } c->i += v; return;
    case 4: for (uint64_t k = 0; k < v && !c->err; k++) cb_skip(c); return;
    case 5: for (uint64_t k = 0; k < 2 * v && !c->err; k++) cb_skip(c); return;
    case 6: cb_skip(c); return;
    }
}
/* reads a text-string key into buf; returns 0 if the item is not a text string (item consumed) */
static int cb_text(cbor_t *c, char *buf, size_t cap) {
    int m; size_t save = c->i; uint64_t v = cb_head(c, &m);
    if (m != 3) { c->i = save; cb_skip(c); buf[0] = 0; return 0; }
    if (c->i + v > c->n) { c->err = 1; return 0; }
    size_t k = v < cap - 1 ? v : cap - 1; memcpy(buf, c->b + c->i, k); buf[k] = 0; c->i += v; return 1;
}
static uint64_t cb_uint(cbor_t *c) { int m; uint64_t v = cb_head(c, &m); if (m != 0) c->err = 1; return v; }
static uint64_t cb_array(cbor_t *c) { int m; uint64_t v = cb_head(c, &m); if (m == 7 && v == 22) return 0; if (m != 4) c->err = 1; return v; }
static uint64_t cb_map(cbor_t *c) { int m; uint64_t v = cb_head(c, &m); if (m == 7 && v == 22) return 0; if (m != 5) c->err = 1; return v; }
static uint32_t *cb_u32_array(cbor_t *c, size_t *n) {
    uint64_t k = cb_array(c); if (c->err) return NULL;
    uint32_t *a = (uint32_t *)malloc(sizeof(uint32_t) * (k ? k : 1));
    for (uint64_t i = 0; i < k; i++) a[i] = (uint32_t)cb_uint(c);
    *n = k; return a;
}

static int parse_body(r1cs_t *cs, const uint8_t *b, size_t n) {
    cbor_t c = {b, n, 0, 0};
    uint64_t nk = cb_map(&c); char key[64];
    for (uint64_t k = 0; k < nk && !c.err; k++) {
        cb_text(&c, key, sizeof key);
        if (!strcmp(key, "Public")) { cs->n_public = cb_array(&c); for (size_t i = 0; i < cs->n_public; i++) cb_skip(&c); }
        else if (!strcmp(key, "Secret")) { cs->n_secret = cb_array(&c); for (size_t i = 0; i < cs->n_secret; i++) cb_skip(&c); }
        else if (!strcmp(key, "NbConstraints")) cs->n_constraints = cb_uint(&c);
        else if (!strcmp(key, "NbInternalVariables")) cs->n_internal = cb_uint(&c);
        else if (!strcmp(key, "Blueprints")) {
            uint64_t nb = cb_array(&c); if (nb > 32) return -1; cs->n_bp = (int)nb;
            for (uint64_t i = 0; i < nb && !c.err; i++) {
                int m; uint64_t tag = cb_head(&c, &m); if (m != 6) return -1;
                cs->bp_entries[i] = NULL; cs->bp_nentries[i] = 0;
                if (tag == 5309735) { cs->bp_kind[i] = BP_HINT; cb_skip(&c); }
                else if (tag == 5309736) { cs->bp_kind[i] = BP_R1C; cb_skip(&c); }
                else if (tag == 5309741) {
                    cs->bp_kind[i] = BP_LOOKUP;
                    uint64_t mk = cb_map(&c); char k2[64];
                    for (uint64_t j = 0; j < mk && !c.err; j++) {
                        cb_text(&c, k2, sizeof k2);
                        if (!strcmp(k2, "EntriesCalldata")) cs->bp_entries[i] = cb_u32_array(&c, &cs->bp_nentries[i]);
                        else cb_skip(&c);
                    }
                } else return -1;   /* unknown blueprint: refuse rather than mis-solve */
            }
        } else if (!strcmp(key, "CommitmentInfo")) {
            int m; uint64_t tag = cb_head(&c, &m);
            if (m != 6 || tag != 5309742) return -1;
            uint64_t nc = cb_array(&c); if (nc > 1) return -1;   /* the reference circuits have at most one */
            cs->n_commit = (int)nc;
            for (uint64_t i = 0; i < nc && !c.err; i++) {
                uint64_t mk = cb_map(&c); char k2[64];
                for (uint64_t j = 0; j < mk && !c.err; j++) {
                    cb_text(&c, k2, sizeof k2);
                    if (!strcmp(k2, "CommitmentIndex")) cs->commit_index = (uint32_t)cb_uint(&c);
                    else if (!strcmp(k2, "PrivateCommitted")) cs->commit_priv = cb_u32_array(&c, &cs->n_commit_priv);
                    else if (!strcmp(k2, "NbPublicCommitted")) cs->n_pub_committed = cb_uint(&c);
                    else cb_skip(&c);
                }
            }
        } else cb_skip(&c);
    }
    return c.err ? -1 : 0;
}

static uint32_t *read_stream(const uint8_t *sec, size_t seclen, size_t *p, size_t expect_cap, long *nvals) {
    if (*p + 8 > seclen) return NULL;
    uint64_t nw = rd64le(sec + *p); *p += 8;
    if (*p + 4 * nw > seclen) return NULL;
    uint32_t *w = (uint32_t *)malloc(4 * (nw ? nw : 1));
    for (uint64_t i = 0; i < nw; i++) w[i] = rd32le(sec + *p + 4 * i);
    *p += 4 * nw;
    uint32_t *out = (uint32_t *)malloc(4 * (expect_cap ? expect_cap : 1));
    *nvals = intcomp_u32(w, nw, out, expect_cap);
    free(w);
    if (*nvals < 0) { free(out); return NULL; }
    return out;
}

int r1cs_parse(r1cs_t *cs, const uint8_t *buf, size_t len) {
    memset(cs, 0, sizeof *cs);
    if (len < 64 || rd64le(buf) != len - 32) return -1;
    uint64_t lv = rd64le(buf + 32), ins = rd64le(buf + 40), cd = rd64le(buf + 48), body = rd64le(buf + 56);
    if (64 + lv + ins + cd + body + 8 > len) return -1;
    const uint8_t *L = buf + 64, *I = L + lv, *C = I + ins, *B = C + cd, *K = B + body;
    /* calldata: count, then LEB128 varints */
    uint64_t ncd = rd64le(C); size_t p = 8;
    cs->calldata = (uint32_t *)malloc(4 * (ncd ? ncd : 1)); cs->n_calldata = ncd;
    for (uint64_t i = 0; i < ncd; i++) {
        uint64_t v = 0; int sh = 0;
        for (;;) { if (p >= cd) return -1; uint8_t b = C[p++]; v |= (uint64_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; if (sh > 35) return -1; }
        cs->calldata[i] = (uint32_t)v;
    }
    /* instruction boundaries: every instruction's first calldata word is its own length */
    size_t ni = 0;
    for (size_t q = 0; q < ncd;) { uint32_t l = cs->calldata[q]; if (l == 0 || q + l > ncd) return -1; q += l; ni++; }
    cs->n_instr = ni;
    cs->cstart = (size_t *)malloc(sizeof(size_t) * (ni + 1));
    { size_t q = 0; for (size_t i = 0; i < ni; i++) { cs->cstart[i] = q; q += cs->calldata[q]; } cs->cstart[ni] = q; }
    /* instruction streams */
    p = 0; long nv;
    cs->bp = read_stream(I, ins, &p, ni, &nv); if (!cs->bp || (size_t)nv != ni) return -2;
    cs->coff = read_stream(I, ins, &p, ni, &nv); if (!cs->coff || (size_t)nv != ni) return -2;
    cs->woff = read_stream(I, ins, &p, ni, &nv); if (!cs->woff || (size_t)nv != ni) return -2;
    /* levels */
    p = 0; uint64_t nl = rd64le(L); p = 8;
    cs->n_levels = nl; cs->level_off = (size_t *)malloc(sizeof(size_t) * (nl + 1));
    cs->level_instr = (uint32_t *)malloc(4 * (ni ? ni : 1));
    size_t filled = 0;
    for (uint64_t l = 0; l < nl; l++) {
        cs->level_off[l] = filled;
        uint32_t *vals = read_stream(L, lv, &p, ni, &nv); if (!vals) return -3;
        if (filled + (size_t)nv > ni) { free(vals); return -3; }
        memcpy(cs->level_instr + filled, vals, 4 * (size_t)nv); filled += (size_t)nv; free(vals);
    }
    cs->level_off[nl] = filled;
    if (filled != ni) return -3;
    if (parse_body(cs, B, body)) return -4;
    cs->n_wires = cs->n_public + cs->n_secret + cs->n_internal;
    /* coefficients: u64 count, 4 LE limbs each, Montgomery form */
    uint64_t nc = rd64le(K);
    if ((size_t)(K - buf) + 8 + 32 * nc != len) return -5;
    cs->coeff = (fe *)malloc(sizeof(fe) * (nc ? nc : 1)); cs->n_coeff = nc;
    for (uint64_t i = 0; i < nc; i++) for (int j = 0; j < 4; j++) cs->coeff[i].l[j] = rd64le(K + 8 + 32 * i + 8 * j);
    return 0;
}
void r1cs_free(r1cs_t *cs) {
    free(cs->bp); free(cs->coff); free(cs->woff); free(cs->cstart); free(cs->calldata); free(cs->coeff);
    free(cs->level_off); free(cs->level_instr); free(cs->commit_priv);
    for (int i = 0; i < cs->n_bp; i++) free(cs->bp_entries[i]);
    memset(cs, 0, sizeof *cs);
}

/* ---- solver (App. C) ---- */
typedef struct { const r1cs_t *cs; fe *W; uint8_t *solved; } sctx;
/* evaluates a linear expression [n, (cid, wid) x n] starting at cd; advances *adv.  All wires must be solved. */
static int eval_linexp(sctx *s, const uint32_t *cd, fe *out, size_t *adv) {
    uint32_t n = cd[0]; fe acc; fr_set_zero(&acc);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t cid = cd[1 + 2 * k], wid = cd[2 + 2 * k];
        if (wid == 0xFFFFFFFFu) { fr_add(&acc, &acc, &s->cs->coeff[cid]); continue; }
        if (wid >= s->cs->n_wires || !s->solved[wid]) return -1;
        fe t; fr_mul(&t, &s->cs->coeff[cid], &s->W[wid]); fr_add(&acc, &acc, &t);
    }
    *out = acc; *adv = 1 + 2 * (size_t)n; return 0;
}
static int small_uint(const fe *v, uint64_t *out) {
    uint64_t c[4]; fr_to_canon(c, v);
    if (c[1] | c[2] | c[3]) return 0;
    *out = c[0]; return 1;
}

long r1cs_solve(const r1cs_t *cs, const fe *witness, fe *W, fe *A, fe *B, fe *C, const solve_opts_t *opts) {
    sctx s = {cs, W, (uint8_t *)calloc(cs->n_wires, 1)};
    long rc = 0;
    fr_set_one(&W[0]); s.solved[0] = 1;
    size_t nin = cs->n_public - 1 + cs->n_secret;
    for (size_t i = 0; i < nin; i++) { W[1 + i] = witness[i]; s.solved[1 + i] = 1; }
    for (size_t ii = 0; ii < cs->n_instr && !rc; ii++) {
        const uint32_t *cd = cs->calldata + cs->cstart[ii];
        int kind = cs->bp[ii] < (uint32_t)cs->n_bp ? cs->bp_kind[cs->bp[ii]] : -1;
        if (kind == BP_R1C) {
            uint32_t nL = cd[1], nR = cd[2], nO = cd[3];
            const uint32_t *t = cd + 4;
            fe acc[3]; int loc = 0; uint32_t uw = 0, uc = 0;
            uint32_t cnt[3] = {nL, nR, nO};
            for (int side = 0; side < 3 && !rc; side++) {
                fr_set_zero(&acc[side]);
                for (uint32_t k = 0; k < cnt[side]; k++, t += 2) {
                    uint32_t cid = t[0], wid = t[1];
                    if (wid == 0xFFFFFFFFu) { fr_add(&acc[side], &acc[side], &cs->coeff[cid]); continue; }
                    if (!s.solved[wid]) {
                        if (loc) { rc = (long)ii + 1; break; }   /* more than one wire to instantiate */
                        loc = side + 1; uw = wid; uc = cid; continue;
                    }
                    fe m; fr_mul(&m, &cs->coeff[cid], &W[wid]); fr_add(&acc[side], &acc[side], &m);
                }
            }
            if (rc) break;
            fe *a = &acc[0], *b = &acc[1], *c = &acc[2], wire, ab;
            if (loc == 0) {
                fr_mul(&ab, a, b);
                if (!fr_eq(&ab, c)) { rc = (long)ii + 1; break; }
            } else {
                fr_set_zero(&wire);
                if (loc == 3) { fr_mul(&ab, a, b); fr_sub(&wire, &ab, c); *c = ab; }
                else {
                    fe *known = loc == 1 ? b : a, *part = loc == 1 ? a : b;
                    if (!fr_is_zero(known)) {
                        fe ki; fr_inv(&ki, known); fr_mul(&wire, c, &ki); fr_sub(&wire, &wire, part);
                        fr_add(part, part, &wire);
                    } else {
                        fr_mul(&ab, a, b);
                        if (!fr_eq(&ab, c)) { rc = (long)ii + 1; break; }
                    }
                }
                fe ci; fr_inv(&ci, &cs->coeff[uc]); fr_mul(&wire, &wire, &ci);
                W[uw] = wire; s.solved[uw] = 1;
            }
            uint32_t co = cs->coff[ii];
            if (co >= cs->n_constraints) { rc = (long)ii + 1; break; }
            A[co] = *a; B[co] = *b; C[co] = *c;
        } else if (kind == BP_HINT) {
            uint32_t hid = cd[1], nIn = cd[2]; size_t q = 3;
            fe *in = (fe *)malloc(sizeof(fe) * (nIn ? nIn : 1));
            for (uint32_t k = 0; k < nIn && !rc; k++) { size_t adv; if (eval_linexp(&s, cd + q, &in[k], &adv)) rc = (long)ii + 1; else q += adv; }
            if (rc) { free(in); break; }
            uint32_t o0 = cd[q], o1 = cd[q + 1], nOut = o1 - o0;
            if (o1 > cs->n_wires || o1 < o0) { free(in); rc = (long)ii + 1; break; }
            if (hid == HINT_NBITS) {
                uint64_t cv[4]; fr_to_canon(cv, &in[0]);
                for (uint32_t k = 0; k < nOut; k++) fr_from_u64(&W[o0 + k], k < 256 ? (cv[k / 64] >> (k % 64)) & 1 : 0);
            } else if (hid == HINT_COUNT) {
                uint64_t nT = 0, nV = 0;
                if (nIn < 2 || !small_uint(&in[0], &nT) || !small_uint(&in[1], &nV) || nV == 0 || nT != nOut ||
                    nIn < 2 + nT * nV || (nIn - 2 - nT * nV) % nV) { free(in); rc = (long)ii + 1; break; }
                size_t nQ = (nIn - 2 - nT * nV) / nV;
                uint64_t *cnt = (uint64_t *)calloc(nT ? nT : 1, 8);
                for (size_t qy = 0; qy < nQ && !rc; qy++) {
                    const fe *qr = in + 2 + nT * nV + qy * nV; int found = 0;
                    for (size_t tr = 0; tr < nT && !found; tr++) {
                        const fe *row = in + 2 + tr * nV; int same = 1;
                        for (size_t v = 0; v < nV; v++) if (!fr_eq(&row[v], &qr[v])) { same = 0; break; }
                        if (same) { cnt[tr]++; found = 1; }
                    }
                    if (!found) rc = (long)ii + 1;     /* query not in table */
                }
                for (uint32_t k = 0; k < nOut; k++) fr_from_u64(&W[o0 + k], cnt[k]);
                free(cnt);
                if (rc) { free(in); break; }
            } else if (hid == HINT_RANDOMIZE) {
                for (uint32_t k = 0; k < nOut; k++) { if (opts && opts->randomize) W[o0 + k] = *opts->randomize; else fr_set_zero(&W[o0 + k]); }
            } else if (hid == HINT_BSB22) {
                fe out; fr_set_zero(&out);
                if (opts && opts->commit_cb && opts->commit_cb(opts->commit_ctx, in + 1, nIn - 1, &out)) { free(in); rc = (long)ii + 1; break; }
                for (uint32_t k = 0; k < nOut; k++) W[o0 + k] = out;
            } else { free(in); rc = (long)ii + 1; break; }
            for (uint32_t k = 0; k < nOut; k++) s.solved[o0 + k] = 1;
            free(in);
        } else if (kind == BP_LOOKUP) {
            uint32_t nE = cd[1], nIn = cd[2]; size_t q = 3;
            const uint32_t *ent = cs->bp_entries[cs->bp[ii]]; size_t nent = cs->bp_nentries[cs->bp[ii]];
            for (uint32_t k = 0; k < nIn && !rc; k++) {
                fe v; size_t adv; uint64_t idx;
                if (eval_linexp(&s, cd + q, &v, &adv) || !small_uint(&v, &idx) || idx >= nE) { rc = (long)ii + 1; break; }
                q += adv;
                /* walk the static entries to the idx-th linear expression */
                size_t e = 0; for (uint64_t z = 0; z < idx; z++) { if (e >= nent) break; e += 1 + 2 * (size_t)ent[e]; }
                if (e >= nent) { rc = (long)ii + 1; break; }
                fe val; size_t adv2;
                if (eval_linexp(&s, ent + e, &val, &adv2)) { rc = (long)ii + 1; break; }
                uint32_t w = cs->woff[ii] + k;
                if (w >= cs->n_wires) { rc = (long)ii + 1; break; }
                W[w] = val; s.solved[w] = 1;
            }
        } else rc = (long)ii + 1;
    }
    if (!rc) for (size_t i = 0; i < cs->n_wires; i++) if (!s.solved[i]) { rc = -1; break; }
    free(s.solved);
    return rc;
}
