// Host-side decoders (see formats.hpp).  Format authority: SURVEY.md App. A / App. B, verified against
// the reference's shipped circuits/generated/{r1cs,pk}.* files.
#include "formats.hpp"
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace gsc {
namespace {

struct Cursor {
    const uint8_t* p; size_t n; size_t i = 0;
    Cursor(const uint8_t* b, size_t len) : p(b), n(len) {}
    void need(size_t k) const { if (i + k > n) throw std::runtime_error("unexpected end of data"); }
    uint8_t u8() { need(1); return p[i++]; }
    uint32_t u32le() { need(4); uint32_t v = 0; for (int k = 3; k >= 0; k--) v = (v << 8) | p[i + k]; i += 4; return v; }
    uint64_t u64le() { need(8); uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | p[i + k]; i += 8; return v; }
    uint32_t u32be() { need(4); uint32_t v = 0; for (int k = 0; k < 4; k++) v = (v << 8) | p[i + k]; i += 4; return v; }
    uint64_t u64be() { need(8); uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | p[i + k]; i += 8; return v; }
    const uint8_t* take(size_t k) { need(k); const uint8_t* q = p + i; i += k; return q; }
    uint64_t uvarint() {
        uint64_t v = 0; int sh = 0;
        for (;;) { uint8_t b = u8(); v |= uint64_t(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; if (sh > 63) throw std::runtime_error("varint overflow"); }
    }
};

// ---- github.com/ronanh/intcomp v1.1.0 u32 stream: [bin-packed deltas]? [var-byte deltas]? trailer ----
std::vector<uint32_t> decode_intcomp_u32(const std::vector<uint32_t>& w) {
    std::vector<uint32_t> out;
    size_t pos = 0;
    if (w.empty()) return out;
    if (w[0] >= 128) {
        if (w.size() < 3) throw std::runtime_error("intcomp: short bin-pack header");
        const uint32_t count = w[0], words = w[1];
        uint32_t prev = w[2];
        if (count % 128 || words < 3 || words > w.size()) throw std::runtime_error("intcomp: bad bin-pack header");
        size_t q = 3;
        out.reserve(count + 128);
        for (uint32_t done = 0; done < count; done += 128) {
            if (q >= words) throw std::runtime_error("intcomp: truncated group");
            const uint32_t hdr = w[q++];
            for (int sub = 0; sub < 4; sub++) {
                const uint32_t desc = (hdr >> (24 - 8 * sub)) & 0xFF;
                const bool zigzag = desc & 0x80;
                const uint32_t bits = desc & 0x7F;
                if (bits > 32 || q + bits > words) throw std::runtime_error("intcomp: bad bit width");
                for (uint32_t j = 0; j < 32; j++) {
                    uint32_t v = 0;
                    if (bits) {
                        const uint64_t bitpos = uint64_t(j) * bits;
                        const size_t wi = q + bitpos / 32; const unsigned sh = bitpos % 32;
                        uint64_t window = w[wi];
                        if (sh + bits > 32) window |= uint64_t(w[wi + 1]) << 32;
                        v = uint32_t((window >> sh) & (bits == 32 ? 0xFFFFFFFFull : ((1ull << bits) - 1)));
                    }
                    const uint32_t delta = zigzag ? ((v >> 1) ^ (0u - (v & 1u))) : v;
                    prev += delta;
                    out.push_back(prev);
                }
                q += bits;
            }
        }
        pos = words;
    }
    if (pos + 1 < w.size()) {
        const uint32_t count = w[pos], words = w[pos + 1];
        if (count >= 128 || words < 2 || pos + words > w.size()) throw std::runtime_error("intcomp: bad var-byte header");
        const size_t nbytes = size_t(words - 2) * 4;
        size_t b = 0; uint32_t acc = 0;
        for (uint32_t k = 0; k < count; k++) {
            uint32_t v = 0; int sh = 0;
            for (;;) {
                if (b >= nbytes) throw std::runtime_error("intcomp: truncated var-byte");
                const uint8_t byte = uint8_t(w[pos + 2 + b / 4] >> (24 - 8 * (b % 4))); b++;
                v |= uint32_t(byte & 0x7F) << sh; sh += 7;
                if (!(byte & 0x80)) break;
                if (sh > 35) throw std::runtime_error("intcomp: var-byte overflow");
            }
            acc += v; out.push_back(acc);
        }
    }
    return out;
}

std::vector<uint32_t> read_u32_stream(Cursor& c) {
    const uint64_t nw = c.u64le();
    c.need(4 * nw);
    std::vector<uint32_t> w(nw);
    for (uint64_t k = 0; k < nw; k++) w[k] = c.u32le();
    return decode_intcomp_u32(w);
}

// ---- CBOR (RFC 8949) — just enough for gnark's R1CS body ----
struct Cbor {
    Cursor c;
    Cbor(const uint8_t* b, size_t n) : c(b, n) {}
    struct Head { int major; uint64_t val; };
    Head head() {
        const uint8_t ib = c.u8(); Head h{ib >> 5, 0}; const int ai = ib & 31;
        if (ai < 24) { h.val = ai; return h; }
        int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : -1;
        if (nb < 0) throw std::runtime_error("cbor: indefinite/reserved length unsupported");
        for (int k = 0; k < nb; k++) h.val = (h.val << 8) | c.u8();
        return h;
    }
    void skip() {
        Head h = head();
        switch (h.major) {
            case 0: case 1: case 7: return;
            case 2: case 3: c.take(h.val); return;
            case 4: for (uint64_t k = 0; k < h.val; k++) skip(); return;
            case 5: for (uint64_t k = 0; k < 2 * h.val; k++) skip(); return;
            case 6: skip(); return;
        }
    }
    std::string text() { Head h = head(); if (h.major != 3) throw std::runtime_error("cbor: expected text"); const uint8_t* q = c.take(h.val); return std::string((const char*)q, h.val); }
    uint64_t uint() { Head h = head(); if (h.major != 0) throw std::runtime_error("cbor: expected uint"); return h.val; }
    // array / map length; CBOR null (0xf6) counts as empty
    uint64_t container(int major) { Head h = head(); if (h.major == 7 && h.val == 22) return 0; if (h.major != major) throw std::runtime_error("cbor: unexpected container"); return h.val; }
    std::vector<uint32_t> u32_array() { uint64_t n = container(4); std::vector<uint32_t> v(n); for (auto& x : v) x = uint32_t(uint()); return v; }
};

void parse_body(R1csFile& cs, const uint8_t* b, size_t n) {
    Cbor cb(b, n);
    const uint64_t nkeys = cb.container(5);
    for (uint64_t k = 0; k < nkeys; k++) {
        const std::string key = cb.text();
        if (key == "Public") { cs.n_public = cb.container(4); for (size_t i = 0; i < cs.n_public; i++) cb.skip(); }
        else if (key == "Secret") { cs.n_secret = cb.container(4); for (size_t i = 0; i < cs.n_secret; i++) cb.skip(); }
        else if (key == "NbConstraints") cs.n_constraints = cb.uint();
        else if (key == "NbInternalVariables") cs.n_internal = cb.uint();
        else if (key == "ScalarField") { if (cb.text() != "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001") throw std::runtime_error("r1cs: not a BN254 constraint system"); }
        else if (key == "Blueprints") {
            const uint64_t nb = cb.container(4);
            for (uint64_t i = 0; i < nb; i++) {
                Cbor::Head t = cb.head();
                if (t.major != 6) throw std::runtime_error("r1cs: untagged blueprint");
                cs.bp_entries.emplace_back();
                if (t.val == 5309735) { cs.bp_kind.push_back(BP_HINT); cb.skip(); }
                else if (t.val == 5309736) { cs.bp_kind.push_back(BP_R1C); cb.skip(); }
                else if (t.val == 5309741) {
                    cs.bp_kind.push_back(BP_LOOKUP);
                    const uint64_t mk = cb.container(5);
                    for (uint64_t j = 0; j < mk; j++) { if (cb.text() == "EntriesCalldata") cs.bp_entries.back() = cb.u32_array(); else cb.skip(); }
                } else throw std::runtime_error("r1cs: unsupported blueprint tag " + std::to_string(t.val));
            }
        } else if (key == "CommitmentInfo") {
            Cbor::Head t = cb.head();
            if (t.major != 6 || t.val != 5309742) throw std::runtime_error("r1cs: unsupported commitment info");
            const uint64_t nc = cb.container(4);
            if (nc > 1) throw std::runtime_error("r1cs: more than one commitment is not supported");
            for (uint64_t i = 0; i < nc; i++) {
                cs.has_commitment = true;
                const uint64_t mk = cb.container(5);
                for (uint64_t j = 0; j < mk; j++) {
                    const std::string k2 = cb.text();
                    if (k2 == "CommitmentIndex") cs.commit_wire = uint32_t(cb.uint());
                    else if (k2 == "PrivateCommitted") cs.commit_private = cb.u32_array();
                    else if (k2 == "NbPublicCommitted") cs.n_public_committed = cb.uint();
                    else cb.skip();
                }
            }
        } else cb.skip();
    }
}

}  // namespace

R1csFile parse_r1cs(const uint8_t* buf, size_t len) {
    R1csFile cs;
    Cursor top(buf, len);
    if (len < 64 || top.u64le() != len - 32) throw std::runtime_error("r1cs: bad length prefix");
    top.i = 32;
    const uint64_t lv = top.u64le(), ins = top.u64le(), cd = top.u64le(), body = top.u64le();
    if (64 + lv + ins + cd + body + 8 > len) throw std::runtime_error("r1cs: section sizes exceed file");
    const uint8_t* L = buf + 64; const uint8_t* I = L + lv; const uint8_t* C = I + ins; const uint8_t* B = C + cd; const uint8_t* K = B + body;

    {   // calldata: count + LEB128 words; instruction i's first word is its own length
        Cursor c(C, cd);
        const uint64_t n = c.u64le();
        cs.calldata.resize(n);
        for (auto& w : cs.calldata) { uint64_t v = c.uvarint(); if (v > 0xFFFFFFFFull) throw std::runtime_error("r1cs: calldata word overflow"); w = uint32_t(v); }
        for (size_t q = 0; q < n;) {
            const uint32_t l = cs.calldata[q];
            if (l == 0 || q + l > n) throw std::runtime_error("r1cs: corrupt instruction length");
            cs.instr_start.push_back(q); q += l;
        }
        cs.instr_start.push_back(n);
    }
    const size_t n_instr = cs.instr_start.size() - 1;
    {
        Cursor c(I, ins);
        cs.blueprint = read_u32_stream(c); cs.constraint_off = read_u32_stream(c); cs.wire_off = read_u32_stream(c);
        if (cs.blueprint.size() != n_instr || cs.constraint_off.size() != n_instr || cs.wire_off.size() != n_instr)
            throw std::runtime_error("r1cs: instruction table size mismatch");
    }
    {
        Cursor c(L, lv);
        const uint64_t nl = c.u64le();
        std::vector<uint8_t> seen(n_instr, 0); size_t total = 0;
        for (uint64_t l = 0; l < nl; l++) {
            cs.levels.push_back(read_u32_stream(c));
            for (uint32_t v : cs.levels.back()) { if (v >= n_instr || seen[v]) throw std::runtime_error("r1cs: levels are not a permutation"); seen[v] = 1; total++; }
        }
        if (total != n_instr) throw std::runtime_error("r1cs: levels do not cover all instructions");
    }
    parse_body(cs, B, body);
    {
        Cursor c(K, len - (K - buf));
        const uint64_t nc = c.u64le();
        if (c.i + 32 * nc != c.n) throw std::runtime_error("r1cs: coefficient table size mismatch");
        cs.coeff_limbs.resize(nc * 8);
        for (auto& w : cs.coeff_limbs) w = c.u32le();    // 4 x u64le == 8 x u32le
    }
    for (uint32_t b : cs.blueprint) if (b >= cs.bp_kind.size()) throw std::runtime_error("r1cs: blueprint id out of range");
    return cs;
}

PkFile parse_pk(const uint8_t* buf, size_t len) {
    PkFile pk; Cursor c(buf, len);
    pk.domain_n = c.u64be();
    if (pk.domain_n == 0 || (pk.domain_n & (pk.domain_n - 1)) || pk.domain_n > (1ull << 28)) throw std::runtime_error("pk: bad domain size");
    memcpy(pk.n_inv, c.take(32), 32); memcpy(pk.omega, c.take(32), 32); memcpy(pk.omega_inv, c.take(32), 32);
    memcpy(pk.coset_g, c.take(32), 32); memcpy(pk.coset_g_inv, c.take(32), 32);
    c.u8();   // "with precompute" flag
    auto g1 = [&](std::vector<uint8_t>& v) { const uint8_t* q = c.take(32); v.assign(q, q + 32); };
    auto g2 = [&](std::vector<uint8_t>& v) { const uint8_t* q = c.take(64); v.assign(q, q + 64); };
    auto slice = [&](std::vector<uint8_t>& v, size_t sz) { const uint32_t n = c.u32be(); const uint8_t* q = c.take(size_t(n) * sz); v.assign(q, q + size_t(n) * sz); };
    g1(pk.g1_alpha); g1(pk.g1_beta); g1(pk.g1_delta);
    slice(pk.g1_A, 32); slice(pk.g1_B, 32); slice(pk.g1_Z, 32); slice(pk.g1_K, 32);
    g2(pk.g2_beta); g2(pk.g2_delta); slice(pk.g2_B, 64);
    pk.n_wires = c.u64be(); c.u64be(); c.u64be();   // NbInfinityA / NbInfinityB are implied by the flag arrays
    { const uint8_t* q = c.take(pk.n_wires); pk.inf_A.assign(q, q + pk.n_wires); }
    { const uint8_t* q = c.take(pk.n_wires); pk.inf_B.assign(q, q + pk.n_wires); }
    const uint32_t nck = c.u32be();
    if (nck > 1) throw std::runtime_error("pk: more than one commitment key is not supported");
    if (nck == 1) { pk.has_commitment_key = true; slice(pk.ped_basis, 32); slice(pk.ped_basis_sigma, 32); if (pk.ped_basis.size() != pk.ped_basis_sigma.size()) throw std::runtime_error("pk: commitment key size mismatch"); }
    if (c.i != len) throw std::runtime_error("pk: trailing bytes");
    size_t na = 0, nb = 0;
    for (size_t i = 0; i < pk.n_wires; i++) { na += !pk.inf_A[i]; nb += !pk.inf_B[i]; }
    if (na * 32 != pk.g1_A.size() || nb * 32 != pk.g1_B.size() || nb * 64 != pk.g2_B.size() || pk.g1_Z.size() != (pk.domain_n - 1) * 32)
        throw std::runtime_error("pk: inconsistent slice sizes");
    // only the compressed encoding written by ProvingKey.WriteTo (keygen.go:350-352) is accepted
    auto check = [](const std::vector<uint8_t>& v, size_t sz) { for (size_t o = 0; o < v.size(); o += sz) if ((v[o] & 0xC0) == 0) throw std::runtime_error("pk: uncompressed point encoding is not supported"); };
    check(pk.g1_alpha, 32); check(pk.g1_beta, 32); check(pk.g1_delta, 32); check(pk.g1_A, 32); check(pk.g1_B, 32); check(pk.g1_Z, 32); check(pk.g1_K, 32);
    check(pk.g2_beta, 64); check(pk.g2_delta, 64); check(pk.g2_B, 64); check(pk.ped_basis, 32); check(pk.ped_basis_sigma, 32);
    return pk;
}

// ---- solver program ----
namespace {
struct LinExp { std::vector<uint32_t> words; };   // [n, (cid,wid)*n]
size_t linexp_words(const uint32_t* p) { return 1 + 2 * size_t(p[0]); }
}  // namespace

SolverProgram build_solver_program(const R1csFile& cs) {
    SolverProgram sp;
    const size_t nw = cs.n_wires();
    std::vector<uint8_t> solved(nw, 0);
    for (size_t i = 0; i < cs.n_public + cs.n_secret; i++) solved[i] = 1;
    // lookup tables: entries must be constant expressions [1, cid, CONST]
    std::vector<int> table_of_bp(cs.bp_kind.size(), -1);
    for (size_t b = 0; b < cs.bp_kind.size(); b++) if (cs.bp_kind[b] == BP_LOOKUP) {
        const auto& e = cs.bp_entries[b];
        if (e.size() != 256 * 3) throw std::runtime_error("solver: lookup table is not 256 constant entries");
        table_of_bp[b] = int(sp.n_tables++);
        for (size_t k = 0; k < 256; k++) { if (e[3 * k] != 1 || e[3 * k + 2] != WIRE_CONST) throw std::runtime_error("solver: non-constant lookup entry"); sp.lookup_coeff.push_back(e[3 * k + 1]); }
    }
    auto& W = sp.words;
    std::vector<uint32_t> wire_level(nw, 0);          // level at which a wire's value becomes available (inputs: 0)
    std::vector<uint32_t> op_level, op_offset, count_ops; size_t commit_op = (size_t)-1;
    uint32_t cur_level = 0;                            // max wire_level over the wires the current op reads
    auto check_wire = [&](uint32_t wid) {
        if (wid == WIRE_CONST) return;
        if (wid >= nw || !solved[wid]) throw std::runtime_error("solver: instruction reads an unsolved wire");
        if (wire_level[wid] > cur_level) cur_level = wire_level[wid];
    };
    auto check_coeff = [&](uint32_t cid) { if (cid >= cs.n_coeff()) throw std::runtime_error("solver: coefficient id out of range"); };
    auto copy_linexp = [&](const uint32_t* p, const uint32_t* end) -> size_t {
        if (p >= end || p + linexp_words(p) > end) throw std::runtime_error("solver: truncated linear expression");
        for (uint32_t k = 0; k < p[0]; k++) { check_coeff(p[1 + 2 * k]); check_wire(p[2 + 2 * k]); }
        W.insert(W.end(), p, p + linexp_words(p));
        return linexp_words(p);
    };
    for (size_t ii = 0; ii < cs.n_instr(); ii++) {
        const uint32_t* cd = cs.calldata.data() + cs.instr_start[ii];
        const uint32_t* end = cs.calldata.data() + cs.instr_start[ii + 1];
        const size_t ncd = end - cd;
        const BlueprintKind kind = cs.bp_kind[cs.blueprint[ii]];
        const size_t hdr_at = W.size();
        cur_level = 0;
        std::vector<uint32_t> produced;
        if (kind == BP_R1C) {
            if (ncd < 4) throw std::runtime_error("solver: short R1C");
            const uint32_t cnt[3] = {cd[1], cd[2], cd[3]};
            if (4 + 2 * (size_t(cnt[0]) + cnt[1] + cnt[2]) != ncd) throw std::runtime_error("solver: R1C length mismatch");
            if (cs.constraint_off[ii] >= cs.n_constraints) throw std::runtime_error("solver: constraint offset out of range");
            uint32_t loc = 0, uw = 0, uc = 0;
            W.insert(W.end(), {0u, 0u, cs.constraint_off[ii], 0u, 0u});
            const uint32_t* t = cd + 4;
            for (int side = 0; side < 3; side++) {
                const size_t cnt_at = W.size(); uint32_t kept = 0;
                W.push_back(0u);
                for (uint32_t k = 0; k < cnt[side]; k++, t += 2) {
                    check_coeff(t[0]);
                    if (t[1] != WIRE_CONST && t[1] < nw && !solved[t[1]]) {
                        if (loc) throw std::runtime_error("solver: more than one wire to instantiate");
                        loc = side + 1; uw = t[1]; uc = t[0]; continue;
                    }
                    check_wire(t[1]);
                    W.push_back(t[0]); W.push_back(t[1]); kept++;
                }
                W[cnt_at] = kept;
            }
            W[hdr_at + 1] = loc; W[hdr_at + 3] = uw; W[hdr_at + 4] = uc;
            if (loc) { solved[uw] = 1; produced.push_back(uw); }
            if (loc == 1 || loc == 2) sp.n_inversions++;
            W[hdr_at] = OP_R1C | uint32_t((W.size() - hdr_at) << 8);
        } else if (kind == BP_HINT) {
            if (ncd < 3) throw std::runtime_error("solver: short hint");
            const uint32_t hid = cd[1], nin = cd[2];
            // locate outputs first (they follow the inputs)
            const uint32_t* p = cd + 3;
            for (uint32_t k = 0; k < nin; k++) { if (p >= end) throw std::runtime_error("solver: truncated hint"); p += linexp_words(p); }
            if (p + 2 != end) throw std::runtime_error("solver: hint length mismatch");
            const uint32_t o0 = p[0], o1 = p[1];
            if (o1 < o0 || o1 > nw) throw std::runtime_error("solver: hint output range");
            const uint32_t nout = o1 - o0;
            const uint32_t* in = cd + 3;
            if (hid == HINT_NBITS) {
                if (nin != 1) throw std::runtime_error("solver: nBits expects one input");
                W.insert(W.end(), {0u, o0, nout}); copy_linexp(in, p);
                W[hdr_at] = OP_NBITS;
            } else if (hid == HINT_COUNT) {
                // inputs: [nTable, nVars, table rows..., query rows...] (std/internal/logderivarg.countHint).  Supported shape —
                // the one gnark's logderivlookup emits: rows are (index, value) pairs, every table row is a pair of constant
                // single-term expressions, and (checked on the device at InitAlgorithm) the index of table row i is i.
                if (nin < 2) throw std::runtime_error("solver: countHint inputs");
                const uint32_t nvars = 2;
                if ((nin - 2) % nvars || (nin - 2) / nvars < nout) throw std::runtime_error("solver: countHint shape");
                if (nout > 256) throw std::runtime_error("solver: countHint table larger than 256 rows");
                const uint32_t nq = (nin - 2) / nvars - nout;
                W.insert(W.end(), {0u, o0, nout, nvars, nq});
                const uint32_t* q = in;
                for (int k = 0; k < 2; k++) { if (q[0] != 1 || q[2] != WIRE_CONST) throw std::runtime_error("solver: countHint sizes must be constants"); q += linexp_words(q); }
                for (uint32_t k = 0; k < nout * nvars; k++) {
                    if (q[0] != 1 || q[2] != WIRE_CONST) throw std::runtime_error("solver: countHint table rows must be constants");
                    q += copy_linexp(q, p);
                }
                for (uint32_t k = 0; k < nq * nvars; k++) q += copy_linexp(q, p);
                W[hdr_at] = OP_COUNT;
                count_ops.push_back((uint32_t)hdr_at);
            } else if (hid == HINT_RANDOMIZE) {
                W.insert(W.end(), {0u, o0, nout}); W[hdr_at] = OP_RANDOMIZE;
            } else if (hid == HINT_BSB22) {
                if (!cs.has_commitment || nout != 1 || o0 != cs.commit_wire) throw std::runtime_error("solver: unexpected commitment hint");
                W.insert(W.end(), {0u, o0, nout}); W[hdr_at] = OP_COMMIT;
                for (uint32_t w : cs.commit_private) check_wire(w);          // the commitment reads every committed wire
                commit_op = op_offset.size();
            } else throw std::runtime_error("solver: unknown hint id " + std::to_string(hid));
            for (uint32_t k = 0; k < nout; k++) { solved[o0 + k] = 1; produced.push_back(o0 + k); }
            W[hdr_at] |= uint32_t((W.size() - hdr_at) << 8);
        } else {   // BP_LOOKUP
            if (ncd < 3) throw std::runtime_error("solver: short lookup");
            const uint32_t nent = cd[1], nin = cd[2];
            if (nent != 256) throw std::runtime_error("solver: lookup table size");
            const uint32_t o0 = cs.wire_off[ii];
            if (size_t(o0) + nin > nw) throw std::runtime_error("solver: lookup output range");
            W.insert(W.end(), {0u, o0, nin, uint32_t(table_of_bp[cs.blueprint[ii]])});
            const uint32_t* q = cd + 3;
            for (uint32_t k = 0; k < nin; k++) q += copy_linexp(q, end);
            if (q != end) throw std::runtime_error("solver: lookup length mismatch");
            for (uint32_t k = 0; k < nin; k++) { solved[o0 + k] = 1; produced.push_back(o0 + k); }
            W[hdr_at] = OP_LOOKUP | uint32_t((W.size() - hdr_at) << 8);
        }
        if ((W.size() - hdr_at) >> 24) throw std::runtime_error("solver: instruction too long");
        for (uint32_t w : produced) wire_level[w] = cur_level + 1;
        op_level.push_back(cur_level); op_offset.push_back((uint32_t)hdr_at);
        sp.n_ops++;
    }
    for (size_t i = 0; i < nw; i++) if (!solved[i]) throw std::runtime_error("solver: wire " + std::to_string(i) + " is never assigned");
    W.push_back(OP_END | (1u << 8));
    W.insert(W.end(), 64, 0u);
    // level schedule (counting sort by level; the commitment op gets a level of its own)
    uint32_t nlev = 0; for (uint32_t l : op_level) if (l + 1 > nlev) nlev = l + 1;
    std::vector<std::vector<uint32_t>> by_level(nlev), count_by_level(nlev);
    {
        std::vector<uint8_t> is_count(W.size(), 0);
        for (uint32_t o : count_ops) is_count[o] = 1;
        for (size_t i = 0; i < op_level.size(); i++) if (i != commit_op) (is_count[op_offset[i]] ? count_by_level : by_level)[op_level[i]].push_back(op_offset[i]);
    }
    std::vector<std::vector<uint32_t>> levels;
    sp.commit_level = (size_t)-1;
    for (uint32_t l = 0; l < nlev; l++) {
        if (!by_level[l].empty()) { levels.push_back(by_level[l]); sp.level_kind.push_back(0); }
        if (!count_by_level[l].empty()) { levels.push_back(count_by_level[l]); sp.level_kind.push_back(1); }     // histogram ops: their own kernel
        if (commit_op != (size_t)-1 && op_level[commit_op] == l) { sp.commit_level = levels.size(); levels.push_back({op_offset[commit_op]}); sp.level_kind.push_back(0); }
    }
    sp.count_ops = count_ops;
    sp.n_levels = levels.size();
    if (sp.commit_level == (size_t)-1) sp.commit_level = sp.n_levels;
    // inside a level, long ops first: the solver gives each of them a whole workgroup (its waves split the terms)
    auto op_words = [&](uint32_t off) { return W[off] >> 8; };
    for (size_t l = 0; l < levels.size(); l++) {
        auto& v = levels[l];
        std::stable_sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return op_words(x) > op_words(y); });
        uint32_t nlong = 0;
        if (!sp.level_kind[l]) while (nlong < v.size() && op_words(v[nlong]) > SolverProgram::LONG_OP_WORDS) nlong++;
        sp.level_long.push_back(nlong);
    }
    sp.sched.push_back((uint32_t)sp.n_levels);
    uint32_t run = 0;
    for (auto& l : levels) { sp.sched.push_back(run); run += (uint32_t)l.size(); if (l.size() > sp.max_level_width) sp.max_level_width = l.size(); }
    sp.sched.push_back(run);
    for (auto& l : levels) sp.sched.insert(sp.sched.end(), l.begin(), l.end());
    return sp;
}

FewProgram build_few_program(const SolverProgram& sp) {
    FewProgram fp;
    const std::vector<uint32_t>& W = sp.words;
    const uint32_t nlev = sp.sched[0];
    const uint32_t* lstart = sp.sched.data() + 1;
    const uint32_t* ops = sp.sched.data() + 2 + nlev;
    auto expr = [&](uint32_t& q) {                    // appends the terms of the expression at word q, returns their number
        const uint32_t n = W.at(q);
        for (uint32_t k = 0; k < n; k++) { fp.terms.push_back(W.at(q + 1 + 2 * k)); fp.terms.push_back(W.at(q + 2 + 2 * k)); }
        q += 1 + 2 * n;
        return n;
    };
    auto desc = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7) {
        const uint32_t d[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
        fp.ops.insert(fp.ops.end(), d, d + 8);
    };
    auto r1c = [&](uint32_t at) {
        const uint32_t toff = (uint32_t)(fp.terms.size() / 2);
        uint32_t q = at + 5;
        const uint32_t nl = expr(q), nr = expr(q), no = expr(q);
        desc(OP_R1C | (W[at + 1] << 8), W[at + 2], W[at + 3], W[at + 4], toff, nl, nr, no);
    };
    // Constraints that solve nothing (loc = 0: more than half of the ops, ChaCha20's 130-term add32 checks among them) only read wires
    // and write their A / B / C rows: no later op waits for them, so they leave the dependent chain of levels and run at the very
    // end, where thousands of them fill the grid (a failed check still marks the statement unsatisfied).
    uint32_t last_generic = nlev;
    for (uint32_t l = 0; l < nlev; l++) if (!sp.level_kind[l]) last_generic = l;
    std::vector<uint32_t> deferred;
    for (uint32_t l = 0; l < nlev; l++) {
        fp.level_start.push_back((uint32_t)(fp.ops.size() / 8));
        fp.count_first.push_back((uint32_t)(fp.count_ops.size() / 4));
        if (sp.level_kind[l]) {                        // histogram ops: where every query starts
            for (uint32_t i = lstart[l]; i < lstart[l + 1]; i++) {
                const uint32_t at = ops[i], ntab = W.at(at + 2), nq = W.at(at + 4);
                const uint32_t d[4] = {at, (uint32_t)fp.count_qoff.size(), nq, 0};
                fp.count_ops.insert(fp.count_ops.end(), d, d + 4);
                uint32_t q = at + 5 + 6 * ntab;
                for (uint32_t k = 0; k < nq; k++) { fp.count_qoff.push_back(q); q += 1 + 2 * W.at(q); q += 1 + 2 * W.at(q); }
            }
            continue;
        }
        for (uint32_t i = lstart[l]; i < lstart[l + 1]; i++) {
            const uint32_t at = ops[i], op = W.at(at) & 0xFF;
            const uint32_t toff = (uint32_t)(fp.terms.size() / 2);
            if (op == OP_R1C) {
                if (W[at + 1] == 0 && l != last_generic) deferred.push_back(at); else r1c(at);
            } else if (op == OP_NBITS) {
                uint32_t q = at + 3;
                const uint32_t n = expr(q);
                desc(OP_NBITS, W[at + 1], W[at + 2], 0, toff, n, 0, 0);
            } else if (op == OP_LOOKUP) {
                uint32_t q = at + 4;
                for (uint32_t e = 0; e < W[at + 2]; e++) {
                    const uint32_t t = (uint32_t)(fp.terms.size() / 2), n = expr(q);
                    desc(OP_LOOKUP, W[at + 1] + e, W[at + 3], 0, t, n, 0, 0);
                }
            } else if (op == OP_RANDOMIZE || op == OP_COMMIT) desc(op, W[at + 1], W[at + 2], 0, 0, 0, 0, 0);
            else throw std::runtime_error("solver: op " + std::to_string(op) + " in a generic level");
        }
        if (l == last_generic) {                      // longest first: they set the tail of the last round
            std::stable_sort(deferred.begin(), deferred.end(), [&](uint32_t x, uint32_t y) { return (W[x] >> 8) > (W[y] >> 8); });
            for (uint32_t at : deferred) r1c(at);
            deferred.clear();
        }
        const size_t w = fp.ops.size() / 8 - fp.level_start.back();
        if (w > fp.max_level_width) fp.max_level_width = w;
    }
    fp.level_start.push_back((uint32_t)(fp.ops.size() / 8));
    fp.count_first.push_back((uint32_t)(fp.count_ops.size() / 4));
    fp.ops.insert(fp.ops.end(), 8, 0u);                 // one all-zero descriptor past the end (a wave without work may fetch it)
    fp.terms.insert(fp.terms.end(), 2 * 256, 0u);       // a lane may fetch (not use) up to 3 x 64 pairs past an op's last term
    return fp;
}

}  // namespace gsc
