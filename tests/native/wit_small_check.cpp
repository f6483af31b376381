// CPU check of the small-integer witness program (csrc/wit_small.cpp, the host half of k_wit_small.hip): builds the program for a
// constraint system from witnesses the ORACLE solved (classes = what those witnesses show, as the engine's calibration does on the
// device), then replays chain and rows one proof at a time with plain integers — an independent restatement of what the kernels do
// with 64 proofs per wave — and compares every wire and every a / b / c value with the oracle's.
//   wit_small_check <r1cs file> <vectors file>
// vectors file: u32 count, then per statement W[n_wires], A, B, C[n_constraints] as 32-byte big-endian canonical values.
#include "formats.hpp"
#include "host_field.hpp"
#include "wit_small.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
using namespace gsc;
using Fr = hostf::Fe<1>;

static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }

// canonical big-endian value -> small signed integer (|v| < 2^62) if it is one
static bool small_of_be(const uint8_t* be, int64_t& out) {
    Fr x; if (!Fr::from_be(be, x)) return false;
    hostf::U256 c = x.canon(), m = x.neg().canon();
    auto fits = [](const hostf::U256& u) { return !u.w[1] && !u.w[2] && !u.w[3] && (u.w[0] >> 62) == 0; };
    if (fits(c)) { out = (int64_t)c.w[0]; return true; }
    if (fits(m)) { out = -(int64_t)m.w[0]; return true; }
    return false;
}
static uint8_t class_of(const std::vector<int64_t>& vals, const std::vector<uint8_t>& ok) {      // k_classify_wires: 0 bit, 1 also -1, else bit length
    uint8_t cls = 0;
    for (size_t i = 0; i < vals.size(); i++) {
        if (!ok[i]) return 254;
        const int64_t v = vals[i];
        if (v == 0 || v == 1) continue;
        uint64_t a = v < 0 ? (uint64_t)-v : (uint64_t)v; int bits = 0; while (a) { bits++; a >>= 1; }
        if ((uint8_t)bits > cls) cls = (uint8_t)bits;
    }
    return cls;
}

struct Emu {
    const SmallProgram& P; std::vector<int8_t> w8; bool flag = false;
    std::vector<int64_t> a, b, c; std::vector<uint8_t> abc_plane_ok; uint32_t failed = 0xFFFFFFFFu;
    explicit Emu(const SmallProgram& p) : P(p), w8(p.rows_per_group, 0), a(p.n_constraints), b(p.n_constraints), c(p.n_constraints) {}
    void tiny_sums(const uint32_t* d, int64_t& L, int64_t& R, int64_t& O) const {
        auto t = [&](int k) { return (int64_t)(int32_t)d[8 + k] * w8[d[2 + k]]; };
        L = t(0) + t(1); R = t(2) + t(3); O = t(4) + t(5);
    }
    void chain() {
        std::vector<int64_t> slots(P.max_slots ? P.max_slots : 1);
        for (uint32_t l = 0; l < P.n_levels; l++) {
            const uint32_t* lv = P.levels.data() + 6 * l;
            // a level's items only read wires of earlier levels: results are collected first, written after (what the barrier guarantees on the device)
            std::vector<std::pair<uint32_t, int8_t>> out;
            for (uint32_t i = lv[0]; i < lv[1]; i++) {
                const uint32_t* d = P.tiny.data() + (size_t)WS_TINY_WORDS * i;
                int64_t L, R, O; tiny_sums(d, L, R, O);
                int64_t w = L * R - O; if (d[0] & WS_F_NEG) w = -w;
                if (w < -1 || w > 1) flag = true;
                out.push_back({d[1], (int8_t)w});
            }
            for (uint32_t i = lv[2]; i < lv[3]; i++) {
                const uint32_t* d = P.parts.data() + 4 * i; int64_t acc = 0;
                for (uint32_t k = 0; k < d[2] * WS_CHUNK; k++) acc += P.tcoef[d[1] + k] * w8[P.twire[d[1] + k]];
                slots[d[0]] = acc;
            }
            for (uint32_t i = lv[4]; i < lv[5]; i++) {
                const uint32_t* d = P.bits.data() + 4 * i; const uint32_t slot0 = d[1] & 0xFFFF, np = d[1] >> 16, sh = d[2] & 0xFF, nb = d[2] >> 8;
                int64_t s = 0; for (uint32_t k = 0; k < np; k++) s += slots[slot0 + k];
                if (s < 0) flag = true;
                for (uint32_t q = 0; q < nb; q++) out.push_back({d[0] + q, (int8_t)((s >> (sh + q)) & 1)});
            }
            for (auto& o : out) w8[o.first] = o.second;
        }
    }
    void put(uint32_t flags, uint32_t cidx, int64_t L, int64_t R, int64_t O) {
        if (L * R != O && cidx < failed) failed = cidx;
        a[cidx] = L; b[cidx] = R; c[cidx] = O;
        auto chk = [&](int shift, int64_t v) { if (!((flags >> shift) & 1u) && (v < -1 || v > 1)) flag = true; };
        chk(WS_CLS_SHIFT_A, L); chk(WS_CLS_SHIFT_B, R); chk(WS_CLS_SHIFT_C, O);
    }
    void rows() {
        for (uint32_t i = 0; i < P.n_rtiny; i++) {
            const uint32_t* d = P.rtiny.data() + (size_t)WS_TINY_WORDS * i;
            if (!(d[0] & WS_F_ITEM)) continue;
            int64_t L, R, O; tiny_sums(d, L, R, O); put(d[0], d[1], L, R, O);
        }
        for (uint32_t i = 0; i < P.n_rgen; i++) {
            const uint32_t* d = P.rgen.data() + 8 * (size_t)i; uint32_t tt = d[2]; int64_t v[3];
            for (int s = 0; s < 3; s++) { int64_t acc = 0; for (uint32_t k = 0; k < d[3 + s] * WS_CHUNK; k++, tt++) acc += P.rtcoef[tt] * w8[P.rtwire[tt]]; v[s] = acc; }
            put(d[0], d[1], v[0], v[1], v[2]);
        }
    }
};

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const auto rb = slurp(argv[1]); const auto vb = slurp(argv[2]);
    const R1csFile cs = parse_r1cs(rb.data(), rb.size());
    const SolverProgram sp = build_solver_program(cs);
    const size_t nw = cs.n_wires(), nc = cs.n_constraints, nin = cs.n_public + cs.n_secret;
    uint32_t count = 0; memcpy(&count, vb.data(), 4);
    const size_t per = 32 * (nw + 3 * nc);
    if (vb.size() != 4 + per * count || !count) { fprintf(stderr, "vectors file: bad size\n"); return 2; }
    // coefficients as integers (the engine: launch_coeff_small on the device)
    std::vector<int64_t> coef(cs.n_coeff(), 0); std::vector<uint8_t> coef_ok(cs.n_coeff(), 0);
    for (size_t i = 0; i < cs.n_coeff(); i++) {
        Fr x; for (int k = 0; k < 4; k++) x.v.w[k] = (uint64_t)cs.coeff_limbs[8 * i + 2 * k] | ((uint64_t)cs.coeff_limbs[8 * i + 2 * k + 1] << 32);
        uint8_t be[32]; x.to_be(be); coef_ok[i] = small_of_be(be, coef[i]);
    }
    // the oracle's vectors as integers; classes from them
    std::vector<std::vector<int64_t>> val[4]; std::vector<std::vector<uint8_t>> okv[4];      // [matrix][row][statement]
    const size_t rows_of[4] = {nw, nc, nc, nc};
    for (int m = 0; m < 4; m++) { val[m].assign(rows_of[m], std::vector<int64_t>(count)); okv[m].assign(rows_of[m], std::vector<uint8_t>(count)); }
    for (uint32_t s = 0; s < count; s++) {
        const uint8_t* p = vb.data() + 4 + per * s;
        for (int m = 0; m < 4; m++) for (size_t r = 0; r < rows_of[m]; r++, p += 32) okv[m][r][s] = small_of_be(p, val[m][r][s]);
    }
    std::vector<uint8_t> cls[4];
    for (int m = 0; m < 4; m++) { cls[m].resize(rows_of[m]); for (size_t r = 0; r < rows_of[m]; r++) cls[m][r] = class_of(val[m][r], okv[m][r]); }
    const SmallProgram P = build_small_program(sp, nw, nc, coef, coef_ok, cls[0], cls[1], cls[2], cls[3]);
    if (!P.ok) { printf("WIT-SMALL-NO %s\n", P.why.c_str()); return 0; }
    size_t wide = 0; for (size_t i = 0; i < nc; i++) wide += P.cls_a[i] + P.cls_b[i] + P.cls_c[i];
    printf("levels=%u chain_items=%zu nbits=%zu rows_tiny=%u rows_general=%u wide_rows=%zu max_slots=%u\n", P.n_levels, P.n_chain_items, P.n_nbits, P.n_rtiny, P.n_rgen, wide, P.max_slots);
    size_t bad = 0;
    for (uint32_t s = 0; s < count; s++) {
        Emu e(P);
        for (size_t w = 0; w < nin; w++) e.w8[w] = (int8_t)val[0][w][s];
        e.chain(); e.rows();
        if (e.flag || e.failed != 0xFFFFFFFFu) { bad++; fprintf(stderr, "statement %u: flag=%d failed=%u\n", s, (int)e.flag, e.failed); continue; }
        for (size_t w = 0; w < nw; w++) if (!okv[0][w][s] || e.w8[w] != val[0][w][s]) { if (bad < 5) fprintf(stderr, "statement %u: wire %zu\n", s, w); bad++; break; }
        for (size_t r = 0; r < nc; r++) if (e.a[r] != val[1][r][s] || e.b[r] != val[2][r][s] || e.c[r] != val[3][r][s] || !okv[1][r][s] || !okv[2][r][s] || !okv[3][r][s]) { if (bad < 5) fprintf(stderr, "statement %u: row %zu\n", s, r); bad++; break; }
    }
    // a wrong prediction must be noticed: every row predicted narrow
    {
        std::vector<uint8_t> z(nc, 0);
        const SmallProgram Q = build_small_program(sp, nw, nc, coef, coef_ok, cls[0], z, z, z);
        Emu e(Q);
        for (size_t w = 0; w < nin; w++) e.w8[w] = (int8_t)val[0][w][0];
        if (Q.ok) { e.chain(); e.rows(); }
        if (wide && (!Q.ok || !e.flag)) { fprintf(stderr, "a wide row stored in a byte plane went unnoticed\n"); bad++; }
    }
    // an unsatisfied statement must be reported: flip an input bit
    {
        Emu e(P);
        for (size_t w = 0; w < nin; w++) e.w8[w] = (int8_t)val[0][w][0];
        e.w8[nin - 1] ^= 1;
        e.chain(); e.rows();
        if (e.failed == 0xFFFFFFFFu && !e.flag) { fprintf(stderr, "a flipped input bit went unnoticed\n"); bad++; }
    }
    if (bad) { printf("WIT-SMALL-FAIL %zu\n", bad); return 1; }
    printf("WIT-SMALL-OK %u statements\n", count);
    return 0;
}
