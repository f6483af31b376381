// InitAlgorithm-time half of the engine: the solver program, the calibration witness, key decoding and the fixed-base tables, and the
// batch buffers of a lane.  See engine_impl.hpp; reference: prove_impl.go:86-110 (pk.ReadFrom / r1cs.ReadFrom / SetParams).
#include "engine_impl.hpp"
#include "host_ciphers.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>

namespace gsc {

namespace {
// Montgomery images of 0, 1, 2, -1, -2 in Fr: gnark puts these at coefficient ids 0..4 of every R1CS
// (SURVEY.md App. A); the solver kernel short-cuts them to additions.
const uint32_t kSmallCoeffs[5][8] = {
    {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u},
    {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u},
    {0x9ffffff6u, 0x592c6838u, 0x3ec19a53u, 0x6df8ed2bu, 0xf0f28c5cu, 0xccdd46deu, 0x340fbe5eu, 0x1c14ef83u},
    {0xa0000006u, 0x974bc177u, 0xda58a367u, 0xf13771b2u, 0x0908122eu, 0x51e1a247u, 0x4729c0fau, 0x2259d6b1u},
    {0x5000000bu, 0xeab58d5bu, 0x3af7d63du, 0xba3afb1du, 0x908ecc00u, 0xeb72fed7u, 0xad21e1cau, 0x144f5eefu},
};
}  // namespace

AlgorithmImpl::AlgorithmImpl(Cipher c, const uint8_t* pk, size_t pk_len, const uint8_t* r1cs, size_t r1cs_len, const EngineConfig& cf) : cipher(c), cfg(cf) {
    // measured crossover with the batch kernels (one 64-column batch: 12.9 ms ChaCha20, 43.7 ms AES): 32 statements for ChaCha20 (10.2 ms), ~23 for AES (8.2 ms + 1.6 ms each: 38.4 ms for 20)
    if (!cfg.few_max) cfg.few_max = cipher == CHACHA20 ? 32 : 20;
    WIN_SLICE = (size_t)cfg.win_slice;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available: the GPU prover has no CPU fallback");
    HIP_CHECK(hipSetDevice(cfg.device));
    HIP_CHECK(hipStreamCreate(&stream));
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg.device) == hipSuccess && cus > 0) cu_count = cus; }
    const bool trace = cfg.trace_host;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    R1csFile cs = parse_r1cs(r1cs, r1cs_len);
    PkFile key = parse_pk(pk, pk_len);
    const auto t1 = now();
    {
        const std::unique_ptr<SolverProgram> sp = init_program(cs);
        calibrate();
        init_small(*sp);
    }
    const auto t2 = now();
    init_key(cs, key);
    const auto t3 = now();
    if (trace) fprintf(stderr, "InitAlgorithm(%d): parse %.0f ms, solver program + calibration %.0f ms, key tables %.0f ms (%.1f GiB)\n", (int)c, ms(t0, t1), ms(t1, t2), ms(t2, t3), table_bytes / 1073741824.0);
    // Lanes: every lane can hold a full batch (GSC_MAX_BATCH), so concurrent calls each get a lane of their own and the chunks of a
    // big call spread over the free ones.  Default: one lane for ChaCha20-V3 (its MSMs fill the chip: a second lane gains
    // nothing), two for AES-V2, whose witness stage (445+ level launches of ~56 us and the commitment round trip) is latency-bound and
    // hides under the other lane's NTT / MSM kernels.
    if (cfg.lanes <= 0) cfg.lanes = has_commitment ? 2 : 1;
    const size_t nl = (size_t)cfg.lanes, lane_cap = (cfg.max_batch + 63) / 64 * 64;
    for (size_t i = 0; i < nl; i++) { lanes.emplace_back(new Lane); alloc_lane(*lanes.back(), lane_cap); }
    full_lanes = nl;
    // Small lanes: calls of a few dozen to a few hundred statements leave the chip under-filled in every stage (163 dependent solver
    // levels of ~26 us, Horner and scalar-multiplication chains that do not shrink with the batch), so several of them must be in
    // flight at once — without paying a full lane's memory for each (46 GB at 8192 proofs): extra lanes of SMALL_LANE_CAP proofs
    // (~3 GB each for ChaCha20-V3), taken by calls that fit them.  GSC_SMALL_LANES: default 2 for ChaCha20-V3; AES-V2 has two full lanes already.
    if (cfg.small_lanes < 0) cfg.small_lanes = has_commitment ? 0 : 2;
    // Capacity: 1024 where the full lane is larger than that (two 1024-statement calls side by side prove 6 % more than one after the other on the
    // full lane, profiles/r04m_lanes.txt), 512 otherwise — with GSC_MAX_BATCH <= 1024 the all-resident configuration (three algorithms, 268 GiB)
    // has 4.7 GiB to spare, and 2 x 2.9 GB more would cost the third algorithm one bit of its Z digits (tools/r04_mem_probe.py).
    const size_t small_want = cfg.small_lane_cap ? (size_t)cfg.small_lane_cap : lane_cap > SMALL_LANE_CAP ? SMALL_LANE_CAP : SMALL_LANE_CAP / 2, small_cap = lane_cap > small_want ? small_want : lane_cap;
    for (int i = 0; i < cfg.small_lanes; i++) { lanes.emplace_back(new Lane); alloc_lane(*lanes.back(), small_cap); }
    lane_busy.assign(lanes.size(), 0);
    cap = lane_cap;
    if (trace) fprintf(stderr, "InitAlgorithm(%d): %zu lane(s) of %zu proofs + %d of %zu, %.0f ms\n", (int)c, nl, lane_cap, cfg.small_lanes, small_cap, ms(t3, now()));
    HIP_CHECK(hipStreamSynchronize(stream));
}

std::unique_ptr<SolverProgram> AlgorithmImpl::init_program(const R1csFile& cs) {
    n_wires = cs.n_wires(); n_public = cs.n_public; n_inputs = cs.n_public + cs.n_secret; n_constraints = cs.n_constraints; has_commitment = cs.has_commitment;
    const size_t expect_in = cipher == CHACHA20 ? 1408 : cipher == AES_128 ? 157 : 173;
    if (cs.n_public - 1 + cs.n_secret != expect_in) throw std::runtime_error("r1cs: witness size does not match the cipher's circuit");
    if (cs.n_coeff() < 5 || memcmp(cs.coeff_limbs.data(), kSmallCoeffs, sizeof kSmallCoeffs)) throw std::runtime_error("r1cs: coefficient ids 0..4 are not 0,1,2,-1,-2");
    std::unique_ptr<SolverProgram> keep(new SolverProgram(build_solver_program(cs)));
    const SolverProgram& sp = *keep;
    n_levels = (uint32_t)sp.n_levels; commit_level = (uint32_t)sp.commit_level; has_div = sp.n_inversions ? 1 : 0;
    level_width.resize(n_levels); for (uint32_t l = 0; l < n_levels; l++) level_width[l] = sp.sched[2 + l] - sp.sched[1 + l];
    level_kind = sp.level_kind; level_long = sp.level_long;
    prog.alloc(sp.words.size()); prog.upload(sp.words.data(), sp.words.size(), stream);
    sched.alloc(sp.sched.size()); sched.upload(sp.sched.data(), sp.sched.size(), stream);
    {
        const FewProgram fp = build_few_program(sp);
        few_ops.alloc(fp.ops.size() ? fp.ops.size() : 8); few_terms.alloc(fp.terms.size()); few_lstart.alloc(fp.level_start.size());
        if (!fp.ops.empty()) few_ops.upload(fp.ops.data(), fp.ops.size(), stream);
        few_terms.upload(fp.terms.data(), fp.terms.size(), stream); few_lstart.upload(fp.level_start.data(), fp.level_start.size(), stream);
        few_count_first = fp.count_first;
        few_count_ops.alloc(fp.count_ops.size() + 4); few_count_qoff.alloc(fp.count_qoff.size() + 1);
        if (!fp.count_ops.empty()) { few_count_ops.upload(fp.count_ops.data(), fp.count_ops.size(), stream); few_count_qoff.upload(fp.count_qoff.data(), fp.count_qoff.size(), stream); }
    }
    lookup_coeff.alloc(sp.lookup_coeff.size() ? sp.lookup_coeff.size() : 1);
    if (!sp.lookup_coeff.empty()) lookup_coeff.upload(sp.lookup_coeff.data(), sp.lookup_coeff.size(), stream);
    coeff.alloc(cs.n_coeff()); coeff_inv.alloc(cs.n_coeff());
    HIP_CHECK(hipMemcpyAsync(coeff.p, cs.coeff_limbs.data(), cs.coeff_limbs.size() * 4, hipMemcpyHostToDevice, stream));
    launch_fr_inverse(coeff.p, coeff_inv.p, cs.n_coeff(), stream);
    if (!sp.count_ops.empty()) {      // lookup histograms rely on table row i carrying index i: verify once, on the device
        DevBuf<uint32_t> d_ops(sp.count_ops.size()), d_flag(1); uint32_t flag = 0;
        d_ops.upload(sp.count_ops.data(), sp.count_ops.size(), stream);
        HIP_CHECK(hipMemsetAsync(d_flag.p, 0, 4, stream));
        launch_check_count_tables(prog.p, coeff.p, d_ops.p, (uint32_t)sp.count_ops.size(), d_flag.p, stream);
        HIP_CHECK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (flag) throw std::runtime_error("r1cs: unsupported lookup table (index column is not 0..n-1)");
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    return keep;
}

// The small-integer witness program (wit_small.hpp), when the circuit qualifies: coefficients as integers (from the device: no field
// arithmetic on the host), the calibration classes of wires and constraint rows, the item lists uploaded once.
void AlgorithmImpl::init_small(const SolverProgram& sp) {
    if (!cfg.small_witness) return;
    const size_t nc = coeff.n;
    DevBuf<long long> d_c(nc ? nc : 1); DevBuf<uint8_t> d_ok(nc ? nc : 1);
    std::vector<long long> hc(nc); std::vector<uint8_t> ok(nc);
    launch_coeff_small(coeff.p, nc, d_c.p, d_ok.p, stream);
    HIP_CHECK(hipGetLastError());
    if (nc) { HIP_CHECK(hipMemcpyAsync(hc.data(), d_c.p, nc * 8, hipMemcpyDeviceToHost, stream)); HIP_CHECK(hipMemcpyAsync(ok.data(), d_ok.p, nc, hipMemcpyDeviceToHost, stream)); }
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int64_t> ci(hc.begin(), hc.end());
    std::vector<uint8_t> ca = row_class_a, cb = row_class_b, cc = row_class_c;
    if (cfg.small_witness == 2) { std::fill(ca.begin(), ca.end(), 0); std::fill(cb.begin(), cb.end(), 0); std::fill(cc.begin(), cc.end(), 0); }      // test: wrong on purpose
    small = build_small_program(sp, n_wires, n_constraints, ci, ok, row_class, ca, cb, cc);
    if (cfg.trace_host) fprintf(stderr, "InitAlgorithm(%d): small-integer witness path: %s%s\n", (int)cipher, small.ok ? "yes" : "no — ", small.ok ? "" : small.why.c_str());
    if (!small.ok) return;
    auto up32 = [&](DevBuf<uint32_t>& d, std::vector<uint32_t>& v, size_t pad) { v.resize(v.size() + pad, 0u); d.alloc(v.size()); d.upload(v.data(), v.size(), stream); };
    auto up64 = [&](DevBuf<long long>& d, std::vector<int64_t>& v, size_t pad) { v.resize(v.size() + pad, 0); d.alloc(v.size()); d.upload(reinterpret_cast<const long long*>(v.data()), v.size(), stream); };
    // (the kernels fetch descriptors and terms a whole wave at a time: 128 / 64 words beyond the last item may be read)
    up32(ws_tiny, small.tiny, 128); up32(ws_parts, small.parts, 4); up32(ws_bits, small.bits, 4); up32(ws_twire, small.twire, 64); up64(ws_tcoef, small.tcoef, 64); up32(ws_levels, small.levels, 6);
    up32(ws_rtiny, small.rtiny, 128); up32(ws_rgen, small.rgen, 8); up32(ws_rtwire, small.rtwire, 64); up64(ws_rtcoef, small.rtcoef, 64);
    ws_cls_a.alloc(n_constraints); ws_cls_b.alloc(n_constraints); ws_cls_c.alloc(n_constraints);
    ws_cls_a.upload(small.cls_a.data(), n_constraints, stream); ws_cls_b.upload(small.cls_b.data(), n_constraints, stream); ws_cls_c.upload(small.cls_c.data(), n_constraints, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

void AlgorithmImpl::pack_inputs(const ProofRequest* reqs, size_t n, size_t B, uint8_t* h_in, uint8_t* h_rs) {
    memset(h_in, 0, 176 * B); memset(h_rs, 0, 64 * B);
    for (size_t i = 0; i < B; i++) {
        const ProofRequest& q = reqs[i < n ? i : n - 1];
        uint8_t* rec = h_in + 176 * i;
        memcpy(rec, q.key, q.keylen);
        memcpy(rec + 32, q.nonce, 12);
        rec[44] = (uint8_t)q.counter; rec[45] = (uint8_t)(q.counter >> 8); rec[46] = (uint8_t)(q.counter >> 16); rec[47] = (uint8_t)(q.counter >> 24);
        memcpy(rec + 48, q.plaintext, 64); memcpy(rec + 112, q.ciphertext, 64);
        memcpy(h_rs + 64 * i, q.r, 32); memcpy(h_rs + 64 * i + 32, q.s, 32);
    }
}

void AlgorithmImpl::calibrate() {
    row_class.assign(n_wires + 4, 255);
    row_class[n_wires] = row_class[n_wires + 1] = row_class[n_wires + 2] = 254;     // r, s, -rs: uniform scalars
    row_class_c.assign(n_constraints, 255);                                         // the rows of c (evaluation-form quotient): same prediction, same fallbacks
    row_class_a.assign(n_constraints, 255); row_class_b.assign(n_constraints, 255);  // the rows of a and b (byte planes of the small-integer witness path)
    if (cfg.bit_groups <= 0) return;
    if (cfg.bit_groups >= 2) { std::fill(row_class.begin(), row_class.begin() + n_wires, 0); std::fill(row_class_c.begin(), row_class_c.end(), 0); std::fill(row_class_a.begin(), row_class_a.end(), 0); std::fill(row_class_b.begin(), row_class_b.end(), 0); return; }
    const size_t B = 64;
    std::vector<ProofRequest> reqs(B);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto& q : reqs) {
        q = ProofRequest{};
        q.keylen = cipher == AES_128 ? 16 : 32;
        for (uint32_t i = 0; i < q.keylen; i++) q.key[i] = (uint8_t)next();
        for (auto& b : q.nonce) b = (uint8_t)next();
        for (auto& b : q.plaintext) b = (uint8_t)next();
        q.counter = (uint32_t)(next() & 0xFFFF);
        if (cipher == CHACHA20) chacha20_xor_stream(q.key, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
        else aes_ctr_xor_stream(q.key, q.keylen, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
        q.r[0] = 3; q.s[0] = 5; q.mask[0] = 7;
    }
    std::vector<uint8_t> h_in(176 * B), h_rs(64 * B); pack_inputs(reqs.data(), B, B, h_in.data(), h_rs.data());
    DevBuf<uint8_t> d_inputs(h_in.size()), d_rs(h_rs.size()), d_mask_in(32 * B); DevBuf<uint32_t> d_status(B);
    DevBuf<fe> d_W((n_wires + 4) * B), d_A(n_constraints * B), d_B(n_constraints * B), d_C(n_constraints * B), d_mask(B), d_commit(B);
    d_inputs.upload(h_in.data(), h_in.size(), stream); d_rs.upload(h_rs.data(), h_rs.size(), stream);
    if (cipher == CHACHA20) launch_assign_chacha(d_inputs.p, d_W.p, B, stream);
    else launch_assign_aes(d_inputs.p, cipher == AES_128 ? 16 : 32, d_W.p, B, stream);
    HIP_CHECK(hipMemsetAsync(d_mask_in.p, 1, d_mask_in.bytes(), stream));
    HIP_CHECK(hipMemsetAsync(d_commit.p, 1, d_commit.bytes(), stream));      // stands in for the commitment challenge: any residue will do
    launch_prep_rs(d_rs.p, d_W.p, n_wires, B, has_commitment ? d_mask_in.p : nullptr, d_mask.p, stream);
    HIP_CHECK(hipMemsetAsync(d_status.p, 0xFF, B * 4, stream));
    SolverArgs sa{prog.p, sched.p, 0, n_levels, coeff.p, coeff_inv.p, lookup_coeff.p, d_W.p, d_A.p, d_B.p, d_C.p, B, d_status.p,
                  has_commitment ? d_mask.p : nullptr, has_commitment ? d_commit.p : nullptr, has_div, 0u, nullptr};
    for (uint32_t l = 0; l < n_levels; l++) {
        sa.first_level = l; sa.n_long = level_long[l];
        if (level_kind[l]) launch_solver_count_level(sa, level_width[l], stream); else launch_solver_level(sa, level_width[l], stream);
    }
    DevBuf<uint8_t> d_cls(n_wires);
    launch_classify_wires(d_W.p, n_wires, B, d_status.p, d_cls.p, stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(row_class.data(), d_cls.p, n_wires, hipMemcpyDeviceToHost, stream));
    DevBuf<uint8_t> d_cls_c(n_constraints);
    launch_classify_wires(d_C.p, n_constraints, B, d_status.p, d_cls_c.p, stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(row_class_c.data(), d_cls_c.p, n_constraints, hipMemcpyDeviceToHost, stream));
    DevBuf<uint8_t> d_cls_a(n_constraints), d_cls_b(n_constraints);
    launch_classify_wires(d_A.p, n_constraints, B, d_status.p, d_cls_a.p, stream);
    launch_classify_wires(d_B.p, n_constraints, B, d_status.p, d_cls_b.p, stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(row_class_a.data(), d_cls_a.p, n_constraints, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(row_class_b.data(), d_cls_b.p, n_constraints, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

std::vector<uint8_t> AlgorithmImpl::decompress_g1(const std::vector<uint8_t>& raw, G1Aff* out) {
    const size_t n = raw.size() / 32; std::vector<uint8_t> st(n);
    if (!n) return st;
    DevBuf<uint8_t> d_raw(raw.size()), d_st(n);
    d_raw.upload(raw.data(), raw.size(), stream);
    launch_decompress_g1(d_raw.p, out, d_st.p, n, stream);
    HIP_CHECK(hipMemcpyAsync(st.data(), d_st.p, n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return st;
}

std::vector<uint8_t> AlgorithmImpl::decompress_g2(const std::vector<uint8_t>& raw, G2Aff* out) {
    const size_t n = raw.size() / 64; std::vector<uint8_t> st(n);
    if (!n) return st;
    DevBuf<uint8_t> d_raw(raw.size()), d_st(n);
    d_raw.upload(raw.data(), raw.size(), stream);
    launch_decompress_g2(d_raw.p, out, d_st.p, n, stream);
    HIP_CHECK(hipMemcpyAsync(st.data(), d_st.p, n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return st;
}

template <class AffT, class XyzzT>
void AlgorithmImpl::build_rows(const AffT* bases, size_t n, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, AffT* table) {
    const uint32_t cap = 256;
    std::vector<MsmRowSeg> segs;
    for (size_t i = 0; i < n; i++) for (uint32_t f = 0; f < len[i]; f += cap) segs.push_back(MsmRowSeg{(uint32_t)i, f + 1, len[i] - f < cap ? len[i] - f : cap, 0u, off[i] + f});
    if (segs.empty()) return;
    size_t chunk = ((size_t)4 << 30) / (cap * sizeof(XyzzT)); if (chunk > segs.size()) chunk = segs.size();
    DevBuf<XyzzT> scratch(chunk * cap); DevBuf<MsmRowSeg> d_segs(segs.size());
    d_segs.upload(segs.data(), segs.size(), stream);
    for (size_t t0 = 0; t0 < segs.size(); t0 += chunk) launch_build_rows(bases, d_segs.p + t0, segs.size() - t0 < chunk ? segs.size() - t0 : chunk, cap, table, scratch.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
}

template <class AffT, class XyzzT, class Decomp>
void AlgorithmImpl::build_set(MsmSet<AffT>& set, const std::vector<uint8_t>& raw, size_t point_bytes, const std::vector<uint32_t>& rows, int c, const char* what, Decomp decomp, bool uniform, int expand_cv,
                              const std::vector<uint8_t>* classes, uint32_t zero_row, bool latency_layout) {
    const std::vector<uint8_t>& cls = classes ? *classes : row_class;
    const size_t n = raw.size() / point_bytes;
    if (rows.size() != n) throw std::runtime_error(std::string("pk: row map size mismatch for ") + what);
    set.nbases = n; set.c = c; set.nwin = msm_windows(c);
    DevBuf<AffT> bases(n ? n : 1);
    const std::vector<uint8_t> st = decomp(raw, bases.p);
    for (size_t i = 0; i < n; i++) if (st[i] == 1) throw std::runtime_error(std::string("pk: invalid point in ") + what);
    const size_t D = (size_t)1 << (c - 1);
    const uint32_t ROW_ZERO = zero_row != 0xFFFFFFFFu ? zero_row : (uint32_t)(n_wires + 3);
    std::vector<uint32_t> bits, narrow, wide;            // indices into the key's order; the point at infinity contributes nothing: dropped
    std::vector<uint32_t> narrow_len;
    for (size_t i = 0; i < n; i++) {
        if (st[i] == 2) continue;
        const int k = uniform || rows[i] >= cls.size() ? 255 : cls[rows[i]];
        if (uniform || cfg.bit_groups <= 0 || k == 255 || k + cfg.row_margin_bits > NARROW_MAX_BITS) wide.push_back((uint32_t)i);
        else if (k <= 1) bits.push_back((uint32_t)i);
        else { narrow.push_back((uint32_t)i); const int lb = k + cfg.row_margin_bits; narrow_len.push_back(1u << (lb < 0 ? 0 : lb)); }
    }
    // expand_cv > 0: EVERY base becomes window octets of that digit width (the latency-path layout of the quotient bases: no Horner pass)
    if (!uniform || expand_cv > 0) {
        while (bits.size() % 8) { narrow.insert(narrow.begin(), bits.back()); narrow_len.insert(narrow_len.begin(), 2u); bits.pop_back(); }
        // flat part: [bits][narrow][padding to an octet][window octets of the expanded wide wires]
        std::vector<uint32_t> src(bits), shift, frows, len; std::vector<int32_t> octwin;
        src.insert(src.end(), narrow.begin(), narrow.end());
        for (uint32_t i : src) frows.push_back(rows[i]);
        len.assign(bits.size(), 1u); len.insert(len.end(), narrow_len.begin(), narrow_len.end());
        while (src.size() % 8) { src.push_back(src.empty() ? 0u : src[0]); frows.push_back(ROW_ZERO); len.push_back(1u); }
        shift.assign(src.size(), 0u); octwin.assign(src.size() / 8, -1);
        const bool expand = !wide.empty() && (expand_cv > 0 || wide.size() <= EXPAND_MAX) && n > 0;
        if (expand) {
            set.cv = expand_cv > 0 ? expand_cv : EXPAND_C; const int nwv = msm_windows(set.cv), octs = (nwv + 7) / 8;
            for (uint32_t w : wide) for (int q = 0; q < 8 * octs; q++) {
                src.push_back(w); frows.push_back(rows[w]); shift.push_back(q < nwv ? (uint32_t)(set.cv * q) : 0u); len.push_back(q < nwv ? 1u << (set.cv - 1) : 1u);
                if (q % 8 == 0) octwin.push_back(q);
            }
            set.nexpanded = wide.size(); wide.clear();
        }
        set.nflat = src.size(); set.nbit = bits.size();
        if (set.nflat) {
            DevBuf<AffT> fb(set.nflat); DevBuf<uint32_t> d_src(set.nflat), d_shift(set.nflat);
            d_src.upload(src.data(), src.size(), stream); d_shift.upload(shift.data(), shift.size(), stream);
            launch_shift(bases.p, d_src.p, d_shift.p, set.nflat, fb.p);
            std::vector<uint64_t> off(set.nflat); size_t entries = 0;
            for (size_t i = 0; i < set.nflat; i++) { off[i] = entries; entries += len[i]; }
            set.ftable.alloc(entries); table_bytes += set.ftable.bytes();
            set.rowoff.alloc(set.nflat); set.rowlen.alloc(set.nflat); set.frows.alloc(set.nflat); set.octwin.alloc(octwin.size());
            set.rowoff.upload(off.data(), set.nflat, stream); set.rowlen.upload(len.data(), set.nflat, stream);
            set.frows.upload(frows.data(), set.nflat, stream); set.octwin.upload(octwin.data(), octwin.size(), stream);
            build_rows<AffT, XyzzT>(fb.p, set.nflat, off, len, set.ftable.p);
            if (set.nbit) {
                const size_t ng = set.nbit / 8;
                set.sub.alloc(ng * MSM_GROUP_ENTRIES); set.group_ok.alloc(ng);
                table_bytes += set.sub.bytes();
                build_subset(fb.p, ng, set.sub.p, set.group_ok.p);
            }
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(stream));      // fb, d_src, d_shift go out of scope
        }
    }
    set.nwide = wide.size();
    if (set.nwide) {      // windowed part: uniform rows
        DevBuf<AffT> wb(set.nwide); DevBuf<uint32_t> d_src(set.nwide), d_shift(set.nwide);
        std::vector<uint32_t> zero(set.nwide, 0u), wrows(set.nwide);
        for (size_t i = 0; i < set.nwide; i++) wrows[i] = rows[wide[i]];
        d_src.upload(wide.data(), set.nwide, stream); d_shift.upload(zero.data(), set.nwide, stream);
        launch_shift(bases.p, d_src.p, d_shift.p, set.nwide, wb.p);
        set.wrows.alloc(set.nwide); set.wrows.upload(wrows.data(), set.nwide, stream);
        set.wtable.alloc(set.nwide * D); table_bytes += set.wtable.bytes();
        std::vector<uint64_t> off(set.nwide); std::vector<uint32_t> len(set.nwide, (uint32_t)D);
        for (size_t i = 0; i < set.nwide; i++) off[i] = i * D;
        build_rows<AffT, XyzzT>(wb.p, set.nwide, off, len, set.wtable.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
        if (!uniform && cfg.few_path && cfg.few_wide && latency_layout) {      // (the quotient bases have their own budgeted layout: init_key)
            std::vector<uint8_t> raw_w(set.nwide * point_bytes); std::vector<uint32_t> rows_w(set.nwide);
            for (size_t i = 0; i < set.nwide; i++) { memcpy(raw_w.data() + i * point_bytes, raw.data() + (size_t)wide[i] * point_bytes, point_bytes); rows_w[i] = rows[wide[i]]; }
            set.few_wide.reset(new MsmSet<AffT>());
            build_set<AffT, XyzzT>(*set.few_wide, raw_w, point_bytes, rows_w, c, what, decomp, true, 8, classes, zero_row);
        }
    }
}

void AlgorithmImpl::build_subset(const G1Aff* b, size_t ng, G1Aff* t, uint8_t* ok) {
    size_t chunk = ((size_t)2 << 30) / (MSM_GROUP_ENTRIES * sizeof(G1Xyzz)); if (chunk > ng) chunk = ng;
    DevBuf<G1Xyzz> sc(chunk * MSM_GROUP_ENTRIES);
    for (size_t g0 = 0; g0 < ng; g0 += chunk) launch_build_subset_g1(b + 8 * g0, ng - g0 < chunk ? ng - g0 : chunk, t + g0 * MSM_GROUP_ENTRIES, sc.p, ok + g0, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

void AlgorithmImpl::build_subset(const G2Aff* b, size_t ng, G2Aff* t, uint8_t* ok) {
    size_t chunk = ((size_t)2 << 30) / (MSM_GROUP_ENTRIES * sizeof(G2Xyzz)); if (chunk > ng) chunk = ng;
    DevBuf<G2Xyzz> sc(chunk * MSM_GROUP_ENTRIES);
    for (size_t g0 = 0; g0 < ng; g0 += chunk) launch_build_subset_g2(b + 8 * g0, ng - g0 < chunk ? ng - g0 : chunk, t + g0 * MSM_GROUP_ENTRIES, sc.p, ok + g0, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

void AlgorithmImpl::init_key(const R1csFile& cs, const PkFile& key) {
    if (key.n_wires != n_wires) throw std::runtime_error("pk: wire count does not match the r1cs");
    domain_n = key.domain_n; L = 0; while (((size_t)1 << L) < domain_n) L++;
    if (domain_n < n_constraints || domain_n != (size_t)1 << L) throw std::runtime_error("pk: domain too small for the constraint system");
    if (L < NTT_MIN_LOG2 || L > NTT_MAX_LOG2) throw std::runtime_error("pk: unsupported domain size 2^" + std::to_string(L) + " (the quotient kernels cover 2^15 .. 2^17: ChaCha20-V3 and AES-V2)");
    if (cs.has_commitment != key.has_commitment_key) throw std::runtime_error("pk: commitment keys do not match the r1cs");
    // NTT constants
    {
        uint8_t be[5 * 32];
        memcpy(be, key.omega, 32); memcpy(be + 32, key.omega_inv, 32); memcpy(be + 64, key.coset_g, 32); memcpy(be + 96, key.coset_g_inv, 32); memcpy(be + 128, key.n_inv, 32);
        DevBuf<uint8_t> d_be(sizeof be); d_be.upload(be, sizeof be, stream);
        dom.alloc(6);
        launch_fr_from_be(d_be.p, dom.p, 5, stream);
        tw_fwd.alloc(domain_n / 2 * 12); tw_inv.alloc(domain_n / 2 * 12); scale_mid.alloc(domain_n); scale_mid_plain.alloc(domain_n); tw_inv_plain.alloc(domain_n / 2 * 12); scale_out.alloc(domain_n); qr.alloc((2 * NTT_QMAX + 1) * 12);
        DevBuf<uint32_t> d_flag(1); uint32_t flag = 0;
        HIP_CHECK(hipMemsetAsync(d_flag.p, 0, 4, stream));
        launch_ntt_constants(dom.p, dom.p + 1, dom.p + 4, L, tw_fwd.p, tw_inv.p, scale_mid.p, scale_out.p, dom.p + 5, qr.p, d_flag.p, tw_inv_plain.p, scale_mid_plain.p, stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (flag) throw std::runtime_error("pk: the domain generator is not gnark-crypto's root of unity for this size");
    }
    auto cat = [](std::vector<uint8_t> a, std::initializer_list<const std::vector<uint8_t>*> more) { for (auto* m : more) a.insert(a.end(), m->begin(), m->end()); return a; };
    const uint32_t ROW_ONE = 0, ROW_R = (uint32_t)n_wires, ROW_S = ROW_R + 1, ROW_NRS = ROW_R + 2;
    std::vector<uint32_t> rowsA, rowsB, rowsK;
    for (size_t i = 0; i < n_wires; i++) { if (!key.inf_A[i]) rowsA.push_back((uint32_t)i); if (!key.inf_B[i]) rowsB.push_back((uint32_t)i); }
    {
        std::vector<uint8_t> skip(n_wires, 0);
        if (cs.has_commitment) { for (uint32_t w : cs.commit_private) skip[w] = 1; skip[cs.commit_wire] = 1; }
        for (size_t i = cs.n_public; i < n_wires; i++) if (!skip[i]) rowsK.push_back((uint32_t)i);
        if (rowsK.size() * 32 != key.g1_K.size()) throw std::runtime_error("pk: G1.K size does not match the private wires");
    }
    rowsA.push_back(ROW_ONE); rowsA.push_back(ROW_R);
    std::vector<uint32_t> rowsB2 = rowsB;
    rowsB.push_back(ROW_ONE); rowsB.push_back(ROW_S); rowsB2.push_back(ROW_ONE); rowsB2.push_back(ROW_S);
    rowsK.push_back(ROW_NRS);
    quotient_eval = cfg.quotient_eval != 0;
    // coefficient form: the n - 1 bases of the key, scalars h; evaluation form: the n bases V_i, scalars d (k_quot_bases.hip)
    // (evaluation form: table position t holds V of index quot_digit_index(L, t), the order in which the last quotient kernel's threads hold d)
    std::vector<uint32_t> rowsZ(quotient_eval ? domain_n : domain_n - 1); for (size_t i = 0; i < rowsZ.size(); i++) rowsZ[i] = quotient_eval ? quot_digit_index(L, (uint32_t)i) : (uint32_t)i;
    std::vector<uint32_t> rowsZkey(domain_n - 1); for (size_t i = 0; i < rowsZkey.size(); i++) rowsZkey[i] = (uint32_t)i;
    std::vector<uint32_t> rowsC; if (quotient_eval) { rowsC.resize(n_constraints); for (size_t i = 0; i < n_constraints; i++) rowsC[i] = (uint32_t)i; }
    // Digit widths: explicit (GSC_WINDOW_Z / GSC_WINDOW_W) or the largest that keeps the tables inside the per-algorithm HBM
    // budget (the defaults leave room for all three algorithms of the reference on one 288 GB device: 3 x (48 + 16) GB).
    // Z: uniform rows of 2^(c-1) entries of 64 B: c = 16 is 69 GB for ChaCha20-V3 (2^15 - 1 bases), c = 14 is 69 GB for AES-V2 (2^17 - 1)
    if (!cfg.window_z) {
        // ... and inside what the device has free right now: the budget is an upper bound, not a promise (another tenant of the device, a host that
        // loads several algorithms).  Left out of the Z rows: the batch buffers of the lanes (~ 5.7 KiB per wire-or-domain row and proof, see
        // alloc_lane), the other sets' tables and the latency layouts (their own budgets), and 8 GiB of slack.
        double budget = cfg.z_table_gb * 1e9;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double nlanes = cfg.lanes > 0 ? cfg.lanes : (cs.has_commitment ? 2 : 1);
            const double lane = nlanes * (double)cfg.max_batch * ((double)n_wires * 32 + 3.0 * (double)domain_n * 32 + (double)domain_n * 2 * 20 + 1e5) * 1.15;
            const double others = (cfg.w_table_gb + (cfg.few_path ? cfg.few_z_gb + (cs.has_commitment ? 8 : 0) : 0)) * 1e9 + 8.0 * 1073741824.0;
            const double room = (double)free_b - lane - others;
            if (room < budget) budget = room > 0 ? room : 0;
        }
        cfg.window_z = 4;
        for (int c = MSM_MAX_WINDOW; c >= 4; c--) if ((double)rowsZ.size() * (double)((size_t)1 << (c - 1)) * 64.0 <= budget) { cfg.window_z = c; break; }
    }
    if (!cfg.window_w) {      // wire sets: only the wide wires that get the windowed kernel (more than EXPAND_MAX per set) pay for c
        auto wide_in = [&](const std::vector<uint32_t>& rows, const std::vector<uint8_t>& cls) {
            size_t k = 0;
            for (uint32_t r : rows) { const int cl = r < cls.size() ? cls[r] : 255; if (cfg.bit_groups <= 0 || cl == 255 || cl + cfg.row_margin_bits > NARROW_MAX_BITS) k++; }
            return k > EXPAND_MAX ? (double)k : 0.0;
        };
        auto wide_of = [&](const std::vector<uint32_t>& rows) { return wide_in(rows, row_class); };
        const double g1 = wide_of(rowsA) + wide_of(rowsB) + wide_of(rowsK) + 2 * wide_of(cs.commit_private) + wide_in(rowsC, row_class_c), g2 = wide_of(rowsB2);
        cfg.window_w = 4;
        for (int c = 16; c >= 4; c--) if ((g1 * 64.0 + g2 * 128.0) * (double)((size_t)1 << (c - 1)) <= cfg.w_table_gb * 1e9) { cfg.window_w = c; break; }
    }
    auto dec1 = [this](const std::vector<uint8_t>& raw, G1Aff* out) { return decompress_g1(raw, out); };
    auto dec2 = [this](const std::vector<uint8_t>& raw, G2Aff* out) { return decompress_g2(raw, out); };
    const bool trace = cfg.trace_host;
    auto timed = [&](const char* what, auto&& fn) {
        const auto a0 = std::chrono::steady_clock::now(); const size_t b0 = table_bytes; fn();
        if (trace) fprintf(stderr, "  tables %-8s %7.0f ms %8.2f GiB\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count(), (table_bytes - b0) / 1073741824.0);
    };
    timed("G1.A", [&] { build_set<G1Aff, G1Xyzz>(mA, cat(key.g1_A, {&key.g1_alpha, &key.g1_delta}), 32, rowsA, cfg.window_w, "G1.A", dec1, false); });
    timed("G1.B", [&] { build_set<G1Aff, G1Xyzz>(mB1, cat(key.g1_B, {&key.g1_beta, &key.g1_delta}), 32, rowsB, cfg.window_w, "G1.B", dec1, false); });
    timed("G1.K", [&] { build_set<G1Aff, G1Xyzz>(mK, cat(key.g1_K, {&key.g1_delta}), 32, rowsK, cfg.window_w, "G1.K", dec1, false); });
    if (!quotient_eval) timed("G1.Z", [&] { build_set<G1Aff, G1Xyzz>(mZ, key.g1_Z, 32, rowsZ, cfg.window_z, "G1.Z", dec1, true); });      // uniform full-width scalars
    else {
        // The key's Z in evaluation form: U (scalars: the solver's c rows) and V (scalars: d on the zeta-coset), computed on the device from the
        // decompressed points; build_set then lays them out like any other set — "decompression" is a device copy of the finished bases.
        const size_t nz = key.g1_Z.size() / 32;
        if (nz + 1 != domain_n) throw std::runtime_error("pk: G1.Z does not hold n - 1 points");
        DevBuf<G1Aff> d_U(domain_n), d_V(domain_n); std::vector<uint8_t> stU(domain_n), stV(domain_n);
        timed("G1.Z -> U, V", [&] {
            DevBuf<G1Aff> zb(nz); const std::vector<uint8_t> st = decompress_g1(key.g1_Z, zb.p);
            for (size_t i = 0; i < nz; i++) if (st[i] == 1) throw std::runtime_error("pk: invalid point in G1.Z");
            DevBuf<uint8_t> d_st(nz), d_stU(domain_n), d_stV(domain_n); d_st.upload(st.data(), nz, stream);
            DevBuf<fe> tw(domain_n / 2); DevBuf<G1Xyzz> scratch(domain_n); DevBuf<uint32_t> d_perm(domain_n);
            d_perm.upload(rowsZ.data(), domain_n, stream);
            launch_quot_bases(zb.p, d_st.p, L, 0, dom.p + 1, dom.p + 4, tw.p, scratch.p, nullptr, d_U.p, d_stU.p, stream);
            launch_quot_bases(zb.p, d_st.p, L, 1, dom.p + 1, dom.p + 4, tw.p, scratch.p, d_perm.p, d_V.p, d_stV.p, stream);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(stU.data(), d_stU.p, domain_n, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(stV.data(), d_stV.p, domain_n, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        });
        auto from_dev = [this](const DevBuf<G1Aff>& src, const std::vector<uint8_t>& st) {
            return [this, &src, &st](const std::vector<uint8_t>& raw, G1Aff* out) {
                const size_t n = raw.size() / 32;
                HIP_CHECK(hipMemcpyAsync(out, src.p, n * sizeof(G1Aff), hipMemcpyDeviceToDevice, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                return std::vector<uint8_t>(st.begin(), st.begin() + n);
            };
        };
        const std::vector<uint8_t> rawV(domain_n * 32, 0), rawU(n_constraints * 32, 0);      // build_set only takes the point count from these
        timed("G1.Z (V)", [&] { build_set<G1Aff, G1Xyzz>(mZ, rawV, 32, rowsZ, cfg.window_z, "G1.Z (evaluation form)", from_dev(d_V, stV), true); });
        fuse_z_digits = mZ.nwide == domain_n && cfg.fuse_z_digits != 0;      // (a V_i at infinity would be dropped from the table: positions would shift)
        timed("G1.Z (U)", [&] { build_set<G1Aff, G1Xyzz>(mC, rawU, 32, rowsC, cfg.window_w, "G1.Z (evaluation form, c)", from_dev(d_U, stU), false, 0, &row_class_c, (uint32_t)domain_n, false); });
    }
    if (cfg.few_path && cfg.few_z_gb > 0) {
        // calls with a handful of statements: the quotient bases once more as (base, window) pairs with their own rows 2^(cv j) d P — more
        // additions per proof than the wide rows above, but no 254-doubling Horner chain behind them (1.4 ms of a 6 ms Prove)
        const size_t nz = key.g1_Z.size() / 32;
        int cv = 0;
        for (int t : {8, 6, 4}) if ((double)nz * msm_windows(t) * (double)((size_t)1 << (t - 1)) * sizeof(G1Aff) <= (double)cfg.few_z_gb * 1e9) { cv = t; break; }
        if (cv) timed("G1.Z (latency layout)", [&] { build_set<G1Aff, G1Xyzz>(mZfew, key.g1_Z, 32, rowsZkey, cfg.window_z, "G1.Z", dec1, true, cv); });
    }
    timed("G2.B", [&] { build_set<G2Aff, G2Xyzz>(mB2, cat(key.g2_B, {&key.g2_beta, &key.g2_delta}), 64, rowsB2, cfg.window_w, "G2.B", dec2, false); });
    if (cs.has_commitment) {
        if (cs.n_public_committed) throw std::runtime_error("r1cs: public committed wires are not supported");
        if (key.ped_basis.size() != cs.commit_private.size() * 32) throw std::runtime_error("pk: commitment basis size does not match the r1cs");
        build_set<G1Aff, G1Xyzz>(mPed, key.ped_basis, 32, cs.commit_private, cfg.window_w, "commitment basis", dec1, false);
        build_set<G1Aff, G1Xyzz>(mPedSigma, key.ped_basis_sigma, 32, cs.commit_private, cfg.window_w, "commitment basis (sigma)", dec1, false);
    }
}

void AlgorithmImpl::alloc_lane(Lane& ln, size_t B) {
    ln.cap = B;
    // HIP spreads a process's streams over GPU_MAX_HW_QUEUES (4) hardware queues PER PRIORITY LEVEL, round-robin: with ten streams of one
    // priority (a full lane and two small ones, three streams each) a lane's side stream lands in the queue of another lane's main stream and
    // that lane's kernels wait behind a 3 ms chain of sixteen waves.  The three roles therefore take the three priority levels — three pools
    // of queues, no sharing with the default lane counts — and the levels suit them: the side stream carries thin chains on a call's
    // critical path (scalar multiplications, G2 Horner), the third stream bulk work that only has to finish before the Z sum (the quotient
    // of a big call) or beside an otherwise idle chip (a single Prove's B2 sum).  Measured against GPU_MAX_HW_QUEUES=8 in
    // profiles/r04l_hw_queues.txt.  GSC_STREAM_PRIORITIES=0: plain streams.
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (cfg.stream_priorities && least != greatest) {
        const int code = cfg.stream_priorities == 1 ? 213 : cfg.stream_priorities;      // digits: main, side, third; 1 high, 2 normal, 3 low
        auto level = [&](int d) { return d == 1 ? greatest : d == 3 ? least : (least + greatest) / 2; };
        HIP_CHECK(hipStreamCreateWithPriority(&ln.stream, hipStreamDefault, level(code / 100)));
        HIP_CHECK(hipStreamCreateWithPriority(&ln.side, hipStreamDefault, level(code / 10 % 10)));
        HIP_CHECK(hipStreamCreateWithPriority(&ln.side2, hipStreamDefault, level(code % 10)));
    } else { HIP_CHECK(hipStreamCreate(&ln.stream)); HIP_CHECK(hipStreamCreate(&ln.side)); HIP_CHECK(hipStreamCreate(&ln.side2)); }
    for (auto& e : ln.ev) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipEventCreate(&ln.ev_ws));
    HIP_CHECK(hipEventCreateWithFlags(&ln.ev_few, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ln.ev_ab, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_fs, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_b2, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_s2, hipEventDisableTiming));
    ln.h_in.alloc(176 * B); ln.h_rs.alloc(64 * B); ln.h_glv.alloc(2 * MSM_FEW_PROOFS); ln.h_out.alloc(256 * B); ln.h_flags.alloc((B + 3) / 4 * 4); ln.h_status.alloc(B); ln.h_words.alloc(56);
    ln.d_clk.alloc(48); HIP_CHECK(hipMemsetAsync(ln.d_clk.p, 0, 48 * 8, ln.stream));      // [0, 32): the Z kernel's eight samples; [32, 44): the three transform kernels
    ln.d_inputs.alloc(176 * B); ln.d_rs.alloc(64 * B); ln.d_out.alloc(256 * B); ln.d_flags.alloc((B + 3) / 4 * 4); ln.d_status.alloc(B); ln.d_fsync.alloc(2); ln.d_glv.alloc(2 * MSM_FEW_PROOFS);
    // (the buffers that will hold secrets start out clean, so that "nothing of a call is left" can be checked from the first call on)
    HIP_CHECK(hipMemsetAsync(ln.d_inputs.p, 0, ln.d_inputs.bytes(), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_rs.p, 0, ln.d_rs.bytes(), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_glv.p, 0, ln.d_glv.bytes(), ln.stream));
    ln.d_W.alloc((n_wires + 4) * B); ln.d_A.alloc(domain_n * B); ln.d_B.alloc(domain_n * B); ln.d_C.alloc((domain_n + 1) * B);      // (+ one row that stays zero: the padding slots of mC)
    // calls with a handful of statements (k_solver_few) write their own columns only: the others must always hold field elements
    // (zero, later whatever an earlier call left there) because the transforms and MSMs run over whole 64-column batches
    HIP_CHECK(hipMemsetAsync(ln.d_W.p, 0, ln.d_W.n * sizeof(fe), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_A.p, 0, ln.d_A.n * sizeof(fe), ln.stream));
    HIP_CHECK(hipMemsetAsync(ln.d_B.p, 0, ln.d_B.n * sizeof(fe), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_C.p, 0, ln.d_C.n * sizeof(fe), ln.stream));
    if (small.ok) {      // byte planes of the small-integer witness path (values 0, 1, 0xFF)
        ln.d_W8.alloc((size_t)small.rows_per_group * B); ln.d_A8.alloc(n_constraints * B); ln.d_B8.alloc(n_constraints * B); ln.d_C8.alloc(n_constraints * B); ln.d_wsflag.alloc(1);
        HIP_CHECK(hipMemsetAsync(ln.d_W8.p, 0, ln.d_W8.n, ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_A8.p, 0, ln.d_A8.n, ln.stream));
        HIP_CHECK(hipMemsetAsync(ln.d_B8.p, 0, ln.d_B8.n, ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_C8.p, 0, ln.d_C8.n, ln.stream));
        // rows predicted wide: their plane entries say "see the 32-byte element" for good (the rows kernel never writes them)
        launch_wit_mark_wide(ln.d_A8.p, n_constraints, n_constraints, ws_cls_a.p, B, ln.stream); launch_wit_mark_wide(ln.d_B8.p, n_constraints, n_constraints, ws_cls_b.p, B, ln.stream);
        launch_wit_mark_wide(ln.d_C8.p, n_constraints, n_constraints, ws_cls_c.p, B, ln.stream);
        HIP_CHECK(hipGetLastError());
    }
    // partial-sum / digit buffers: the largest need over every batch size this context can be asked for
    size_t p1 = 0, p1b = 0, p2 = 0, p2b = 0, dg = 0, dgz = 0, sj2 = 0, gk = 0; size_t sj1[Lane::NSETS] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto need = [&](auto& m, size_t b, size_t& pa, size_t& pb, size_t& sj) {
        auto part = [&](size_t nb, size_t ns, size_t cols) {
            if (ns * cols > pa) pa = ns * cols;
            if ((ns + MSM_REDUCE_FANIN - 1) / MSM_REDUCE_FANIN * cols > pb) pb = (ns + MSM_REDUCE_FANIN - 1) / MSM_REDUCE_FANIN * cols;
        };
        size_t per = 0;
        if (m.nflat) { part(m.nflat, msm_slices(m.nflat, 1, 256, b, per), b); if (m.nflat / 8 * b > dg) dg = m.nflat / 8 * b; if (m.nbit / 8 * (b / 64) > gk) gk = m.nbit / 8 * (b / 64); if (m.nbit / 8 * MSM_FEW_PROOFS > gk) gk = m.nbit / 8 * MSM_FEW_PROOFS; }
        if (b == 64 && m.few_wide) {      // latency layout of the wide wires: digits of its octets, partial sums of both parts side by side
            const size_t o2 = (m.few_wide->nflat + 7) / 8;
            if (o2 * 64 > dg) dg = o2 * 64;
            part(0, ((m.nflat + 7) / 8 + 63) / 64 + (o2 + 63) / 64, 64);
        }
        if (m.nwide) { const size_t bw = b * (size_t)m.nwin; part(m.nwide, msm_slices(m.nwide, (size_t)m.nwin, WIN_SLICE, b, per), bw); if (bw > sj) sj = bw; const size_t d = (size_t)m.nwin * ((m.nwide + 7) / 8) * b * msm_digit_words(m.c); if (d > dg) dg = d; }
    };
    MsmSet<G1Aff>* g1sets[Lane::NSETS] = {&mA, &mB1, &mK, &mZ, &mPed, &mPedSigma, &mZfew, &mC};
    for (size_t b = 64; b <= B; b += 64) {
        for (int k = 0; k < Lane::NSETS; k++) if (g1sets[k] != &mZfew || b == 64) {      // the latency layout only serves 64-column batches
            if (fuse_z_digits && g1sets[k] == &mZ) { const size_t keep = dg; dg = 0; need(mZ, b, p1, p1b, sj1[k]); if (dg > dgz) dgz = dg; dg = keep; }      // Z's digits: a buffer of their own
            else need(*g1sets[k], b, p1, p1b, sj1[k]);
        }
        need(mB2, b, p2, p2b, sj2);
    }
    if (fuse_z_digits) { ln.d_digits_w.alloc(dg ? dg : 1); dg = dgz; }
    ln.d_part1a.alloc(p1); ln.d_part1b.alloc(p1b); ln.d_part2a.alloc(p2); ln.d_part2b.alloc(p2b);
    ln.d_digits.alloc(dg); ln.d_gok.alloc(gk ? gk : 1);
    {
        size_t dgs = 1, gks = 1, ps = 1;
        for (const MsmSet<G1Aff>* m : {&mA, &mB1}) {
            const size_t noct = (m->nflat + 7) / 8, nsl = (noct + 63) / 64, noctw = m->few_wide ? (m->few_wide->nflat + 7) / 8 : 0, nslw = (noctw + 63) / 64;
            if (noct * 64 > dgs) dgs = noct * 64;
            if (noctw * 64 > dgs) dgs = noctw * 64;
            if (m->nbit / 8 * MSM_FEW_PROOFS > gks) gks = m->nbit / 8 * MSM_FEW_PROOFS;
            if ((nsl + nslw) * 64 > ps) ps = (nsl + nslw) * 64;
        }
        { const size_t o1 = (mB2.nflat + 7) / 8, o2 = mB2.few_wide ? (mB2.few_wide->nflat + 7) / 8 : 0; ln.d_digits_s2.alloc((o1 > o2 ? o1 : o2) * 64 + 1); }
        ln.d_gok_s2.alloc(mB2.nbit / 8 * MSM_FEW_PROOFS + 1);
        ln.d_digits_s.alloc(dgs); ln.d_gok_s.alloc(gks); ln.d_part1c.alloc(ps); ln.d_part1d.alloc((ps / 64 + MSM_REDUCE_FANIN - 1) / MSM_REDUCE_FANIN * 64 + 64);
    }
    for (int k = 0; k < Lane::NSETS; k++) { ln.d_sj1[k].alloc(sj1[k] ? sj1[k] : 1); ln.d_flat1[k].alloc(g1sets[k]->nflat && g1sets[k]->nwide ? B : 1); }
    ln.d_sj2.alloc(sj2 ? sj2 : 1); ln.d_flat2.alloc(B);
    ln.d_sumA.alloc(B); ln.d_sumB1.alloc(B); ln.d_sumK.alloc(B); ln.d_sumZ.alloc(B); ln.d_sumC.alloc(B); ln.d_sumB2.alloc(B); ln.d_tmp.alloc(2 * B);
    if (has_commitment) {
        ln.d_mask_in.alloc(32 * B); ln.d_mask.alloc(B); ln.d_commit.alloc(B); ln.d_cpts.alloc(128 * B); ln.d_sumD.alloc(B); ln.d_sumPok.alloc(B);
        ln.h_mask.alloc(32 * B); ln.h_cpts.alloc(128 * B);
        HIP_CHECK(hipMemsetAsync(ln.d_mask_in.p, 0, ln.d_mask_in.bytes(), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_mask.p, 0, ln.d_mask.bytes(), ln.stream));
    }
}
}  // namespace gsc
