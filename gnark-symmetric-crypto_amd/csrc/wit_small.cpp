// Host side of the small-integer witness path: turns the generic solver program into the chain / rows layout of wit_small.hpp,
// or says why the circuit does not qualify.  No field arithmetic here: coefficient values arrive as integers from the device.
#include "wit_small.hpp"
#include <algorithm>

namespace gsc {

namespace {
struct Term { uint32_t wire; int64_t c; };
using Expr = std::vector<Term>;

double bound_of(const Expr& e) { double b = 0; for (const Term& t : e) b += (double)(t.c < 0 ? -t.c : t.c); return b; }
bool tiny_shape(const Expr& l, const Expr& r, const Expr& o) {
    auto ok = [](const Expr& e) { if (e.size() > 2) return false; for (const Term& t : e) if (t.c >= WS_COEF_TINY || t.c <= -WS_COEF_TINY) return false; return true; };
    return ok(l) && ok(r) && ok(o);
}
void put_tiny(std::vector<uint32_t>& out, uint32_t flags, uint32_t where, const Expr& l, const Expr& r, const Expr& o) {
    uint32_t w[WS_TINY_WORDS] = {flags, where, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const Expr* e[3] = {&l, &r, &o};
    for (int s = 0; s < 3; s++) for (size_t k = 0; k < e[s]->size(); k++) { w[2 + 2 * s + k] = (*e[s])[k].wire; w[8 + 2 * s + k] = (uint32_t)(int32_t)(*e[s])[k].c; }
    out.insert(out.end(), w, w + WS_TINY_WORDS);
}
void pad_tiny(std::vector<uint32_t>& out, size_t first_item, uint32_t where) {
    while ((out.size() / WS_TINY_WORDS - first_item) % WS_IB) put_tiny(out, 0u, where, Expr{}, Expr{}, Expr{});
}
// appends the terms of e padded to whole chunks; returns the number of chunks
uint32_t put_terms(std::vector<uint32_t>& tw, std::vector<int64_t>& tc, const Term* t, size_t n) {
    for (size_t k = 0; k < n; k++) { tw.push_back(t[k].wire); tc.push_back(t[k].c); }
    const uint32_t chunks = (uint32_t)((n + WS_CHUNK - 1) / WS_CHUNK);
    for (size_t k = n; k < (size_t)chunks * WS_CHUNK; k++) { tw.push_back(0u); tc.push_back(0); }
    return chunks;
}
}  // namespace

SmallProgram build_small_program(const SolverProgram& sp, size_t n_wires, size_t n_constraints, const std::vector<int64_t>& coef, const std::vector<uint8_t>& coef_ok,
                                 const std::vector<uint8_t>& class_w, const std::vector<uint8_t>& class_a, const std::vector<uint8_t>& class_b, const std::vector<uint8_t>& class_c) {
    SmallProgram P;
    auto fail = [&](const std::string& why) { P.ok = false; P.why = why; return P; };
    if (sp.commit_level != sp.n_levels) return fail("the circuit has a commitment");
    if (class_w.size() < n_wires || class_a.size() < n_constraints || class_b.size() < n_constraints || class_c.size() < n_constraints) return fail("no calibration classes");
    if (n_wires >= (1u << 24) || n_constraints >= (1u << 24)) return fail("too large");
    for (uint8_t k : sp.level_kind) if (k) return fail("the circuit has lookup-count levels");
    P.n_wires = (uint32_t)n_wires; P.n_constraints = (uint32_t)n_constraints;
    P.scratch_row = (uint32_t)n_wires; P.rows_per_group = (uint32_t)n_wires + 1;
    P.cls_a.assign(n_constraints, 0); P.cls_b.assign(n_constraints, 0); P.cls_c.assign(n_constraints, 0);
    for (size_t i = 0; i < n_constraints; i++) { P.cls_a[i] = class_a[i] > 1; P.cls_b[i] = class_b[i] > 1; P.cls_c[i] = class_c[i] > 1; }
    const std::vector<uint32_t>& W = sp.words;
    const uint32_t nlev = sp.sched[0]; const uint32_t* lstart = sp.sched.data() + 1; const uint32_t* ops = sp.sched.data() + 2 + nlev;
    std::string err;
    auto coef_of = [&](uint32_t cid, int64_t& c) { if (cid >= coef.size() || !coef_ok[cid]) { err = "a coefficient is not a small integer"; return false; } c = coef[cid]; return true; };
    auto read_expr = [&](uint32_t& q, Expr& e) {
        const uint32_t n = W[q];
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t cid = W[q + 1 + 2 * k]; uint32_t wid = W[q + 2 + 2 * k]; int64_t c;
            if (wid == WIRE_CONST) wid = 0;      // a constant: the coefficient times wire 0, gnark's ONE wire (k_assign_* sets it to 1)
            if (wid >= n_wires || class_w[wid] > 1) { err = "a wire that is not in {-1, 0, 1}"; return false; }
            if (!coef_of(cid, c)) return false;
            if (c) e.push_back(Term{wid, c});
        }
        q += 1 + 2 * n;
        return true;
    };
    const double LIM = 4.0e18;      // < 2^62
    std::vector<uint8_t> seen(n_constraints, 0);
    for (uint32_t l = 0; l < nlev; l++) {
        const size_t tiny0 = P.tiny.size() / WS_TINY_WORDS, part0 = P.parts.size() / 4, bits0 = P.bits.size() / 4;
        uint32_t slots = 0;
        for (uint32_t k = lstart[l]; k < lstart[l + 1]; k++) {
            const uint32_t at = ops[k], op = W[at] & 0xFF;
            if (op == OP_R1C) {
                const uint32_t loc = W[at + 1], cidx = W[at + 2], uw = W[at + 3], uc = W[at + 4];
                uint32_t q = at + 5; Expr e[3];
                for (int s = 0; s < 3; s++) if (!read_expr(q, e[s])) return fail(err);
                if (cidx >= n_constraints || seen[cidx]) return fail("constraint rows are not a permutation");
                seen[cidx] = 1;
                if (loc == 1 || loc == 2) return fail("the circuit divides");
                if (loc == 3) {
                    int64_t c;
                    if (!coef_of(uc, c)) return fail(err);
                    if (c != 1 && c != -1) return fail("a solved wire with a coefficient other than +-1");
                    if (uw >= n_wires || class_w[uw] > 1) return fail("a solved wire that is not in {-1, 0, 1}");
                    if (!tiny_shape(e[0], e[1], e[2])) return fail("a producing constraint with more than two terms per side");
                    put_tiny(P.tiny, WS_F_ITEM | (c < 0 ? WS_F_NEG : 0u), uw, e[0], e[1], e[2]);
                    P.n_chain_items++;
                    e[2].push_back(Term{uw, c});      // the full constraint, for the rows
                } else if (loc != 0) return fail("unknown constraint shape");
                if (bound_of(e[0]) * bound_of(e[1]) >= LIM || bound_of(e[2]) >= LIM) return fail("a constraint whose values may exceed 62 bits");
                const uint32_t flags = WS_F_ITEM | ((uint32_t)P.cls_a[cidx] << WS_CLS_SHIFT_A) | ((uint32_t)P.cls_b[cidx] << WS_CLS_SHIFT_B) | ((uint32_t)P.cls_c[cidx] << WS_CLS_SHIFT_C);
                if (tiny_shape(e[0], e[1], e[2])) put_tiny(P.rtiny, flags, cidx, e[0], e[1], e[2]);
                else {
                    const uint32_t t0 = (uint32_t)P.rtwire.size();
                    uint32_t ch[3];
                    for (int s = 0; s < 3; s++) ch[s] = put_terms(P.rtwire, P.rtcoef, e[s].data(), e[s].size());
                    P.rgen.insert(P.rgen.end(), {flags, cidx, t0, ch[0], ch[1], ch[2], 0u, 0u});
                }
            } else if (op == OP_NBITS) {
                const uint32_t o0 = W[at + 1], nout = W[at + 2];
                uint32_t q = at + 3; Expr e;
                if (!read_expr(q, e)) return fail(err);
                if (nout == 0 || nout > 62) return fail("an nBits hint with more than 62 outputs");
                if ((size_t)o0 + nout > n_wires) return fail("nBits outputs out of range");
                for (uint32_t b = 0; b < nout; b++) if (class_w[o0 + b] > 1) return fail("an nBits output that is not a bit");
                if (bound_of(e) >= LIM) return fail("an nBits input that may exceed 62 bits");
                const size_t per_max = (size_t)WS_PART_CHUNKS * WS_CHUNK;
                const size_t nparts = e.empty() ? 1 : (e.size() + per_max - 1) / per_max;
                const size_t per = ((e.size() + nparts - 1) / nparts + WS_CHUNK - 1) / WS_CHUNK * WS_CHUNK;
                const uint32_t slot0 = slots;
                for (size_t pi = 0; pi < nparts; pi++) {
                    const size_t a = std::min(pi * per, e.size()), b = std::min(a + per, e.size());
                    const uint32_t t0 = (uint32_t)P.twire.size();
                    const uint32_t chunks = put_terms(P.twire, P.tcoef, e.data() + a, b - a);
                    P.parts.insert(P.parts.end(), {slots++, t0, chunks, 0u});
                }
                if (nparts >= (1u << 16) || slots >= (1u << 16)) return fail("an nBits input that is too long");
                for (uint32_t b0 = 0; b0 < nout; b0 += WS_BITS_PER_ITEM) {
                    const uint32_t nb = std::min(WS_BITS_PER_ITEM, nout - b0);
                    P.bits.insert(P.bits.end(), {o0 + b0, slot0 | ((uint32_t)nparts << 16), b0 | (nb << 8), 0u});
                }
                P.n_nbits++;
            } else return fail("an instruction other than a constraint or an nBits hint");
        }
        pad_tiny(P.tiny, tiny0, P.scratch_row);
        const size_t tiny1 = P.tiny.size() / WS_TINY_WORDS, part1 = P.parts.size() / 4, bits1 = P.bits.size() / 4;
        if (tiny1 == tiny0 && part1 == part0) continue;      // a level of checks only: nothing for the chain
        P.levels.insert(P.levels.end(), {(uint32_t)tiny0, (uint32_t)tiny1, (uint32_t)part0, (uint32_t)part1, (uint32_t)bits0, (uint32_t)bits1});
        P.n_levels++;
        if (slots > P.max_slots) P.max_slots = slots;
    }
    for (size_t i = 0; i < n_constraints; i++) if (!seen[i]) return fail("a constraint row that no instruction writes");
    pad_tiny(P.rtiny, 0, 0u);
    if (P.max_slots * 512u > 60000u) return fail("a level with too many nBits sums");
    // the padding reads of the sums and of the padding items touch wire 0 with coefficient 0: any value will do
    if (P.twire.empty()) { P.twire.assign(WS_CHUNK, 0u); P.tcoef.assign(WS_CHUNK, 0); }
    if (P.rtwire.empty()) { P.rtwire.assign(WS_CHUNK, 0u); P.rtcoef.assign(WS_CHUNK, 0); }
    P.n_rtiny = (uint32_t)(P.rtiny.size() / WS_TINY_WORDS); P.n_rgen = (uint32_t)(P.rgen.size() / 8);
    P.ok = true;
    return P;
}

}  // namespace gsc
