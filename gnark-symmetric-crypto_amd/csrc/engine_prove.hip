// Per-batch half of the engine: witness -> quotient -> MSMs -> assembly for one chunk of statements on one lane (prove_chunk), the
// MSM orchestration (run_msm) and proof serialisation.  See engine_impl.hpp; reference: groth16.Prove + proof.WriteTo behind
// libraries/prover/impl/provers.go:148-157, :216-226.
#include "engine_impl.hpp"
#include "glv.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>

namespace gsc {

FewSolverChain& few_solver_chain(int device) { static FewSolverChain* chains = new FewSolverChain[64]; return chains[device & 63]; }

namespace {
// (p-1)/2, big-endian: a compressed point carries the "larger y" flag iff y > (p-1)/2
const uint8_t kHalfP[32] = {0x18, 0x32, 0x27, 0x39, 0x70, 0x98, 0xd0, 0x14, 0xdc, 0x28, 0x22, 0xdb, 0x40, 0xc0, 0xac, 0x2e,
                            0xcb, 0xc0, 0xb5, 0x48, 0xb4, 0x38, 0xe5, 0x46, 0x9e, 0x10, 0x46, 0x0b, 0x6c, 0x3e, 0x7e, 0xa3};

void le_limbs_to_be(const uint8_t* le, uint8_t* be) { for (int i = 0; i < 32; i++) be[i] = le[31 - i]; }
bool be_greater(const uint8_t* a, const uint8_t* b) { int c = memcmp(a, b, 32); return c > 0; }
bool be_is_zero(const uint8_t* a) { for (int i = 0; i < 32; i++) if (a[i]) return false; return true; }
}  // namespace

template <class XyzzT, class LR>
void AlgorithmImpl::reduce_slices(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, XyzzT* out, LR launch_reduce) {
    XyzzT* src = pa; XyzzT* alt = pb; size_t ns = nslices;
    for (;;) {
        const size_t groups = msm_reduce_groups(ns, cols);
        XyzzT* dst = groups == 1 ? out : alt;
        launch_reduce(src, ns, cols, dst, st);
        if (groups == 1) break;
        XyzzT* t = src; src = dst; alt = t; ns = groups;
    }
}

template <class XyzzT, class LRF>
void AlgorithmImpl::reduce_slices_few(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, size_t stride, size_t npr, XyzzT* out, LRF launch_reduce_few) {
    XyzzT* src = pa; XyzzT* alt = pb; size_t ns = nslices;
    for (;;) {
        const size_t groups = (ns + 63) / 64;
        XyzzT* dst = groups == 1 ? out : alt;
        launch_reduce_few(src, ns, cols, stride, npr, dst, st);
        if (groups == 1) break;
        XyzzT* t = src; src = dst; alt = t; ns = groups;
    }
}

template <class AffT, class XyzzT, class LF, class LFF, class LW, class LWF, class LR, class LRF>
void AlgorithmImpl::run_msm(Lane& ln, const MsmCtx& ctx, const MsmSet<AffT>& set, const fe* scalars, bool wires, size_t B, size_t n_real, XyzzT* pa, XyzzT* pb, XyzzT* sj, XyzzT* flat, XyzzT* sum, bool timed, bool digits_ready,
             MsmHornerJobs& pending, LF launch_flat, LFF launch_flat_few, LW launch_win, LWF launch_win_few, LR launch_reduce, LRF launch_reduce_few) {
    size_t per = 0;
    const bool fewm = n_real <= (size_t)cfg.few_max && cfg.few_path;
    if (fewm && set.latency_flat()) {
        // a call with a handful of statements, every part of the set as flat rows: lanes = octets of bases, the partial sums of both
        // parts side by side, one reduction, no Horner pass
        size_t ns = 0;
        auto part = [&](const MsmSet<AffT>& m, bool stamp) {
            if (!m.nflat) return;
            const size_t nslices = ((m.nflat + 7) / 8 + 63) / 64;
            MsmFlatRecodeArgs ra{scalars, m.frows.p, m.octwin.p, m.nflat, B, m.cv, ctx.digits, m.nbit, m.group_ok.p, ctx.gok, wires ? 1 : 0, ctx.plane, ctx.plane_rows, ctx.plane_stride};
            launch_msm_recode_flat_few(ra, n_real, ctx.stream);
            MsmFlatArgs a{m.ftable.p, m.rowoff.p, m.rowlen.p, m.nflat, ctx.digits, B, nslices, 512, pa + ns * B, m.nbit, m.sub.p, ctx.gok, scalars, m.frows.p};
            if (stamp) HIP_CHECK(hipEventRecord(ln.ev[5], ctx.stream));
            launch_flat_few(a, n_real, ctx.stream);
            if (stamp) HIP_CHECK(hipEventRecord(ln.ev[6], ctx.stream));
            ns += nslices;
        };
        part(set, timed);
        if (set.few_wide) part(*set.few_wide, false);
        if (ns) reduce_slices_few(ctx.stream, pa, pb, ns, B, B, n_real, sum, launch_reduce_few);
        else HIP_CHECK(hipMemsetAsync(sum, 0, B * sizeof(XyzzT), ctx.stream));
        return;
    }
    if (set.nflat) {
        MsmFlatRecodeArgs ra{scalars, set.frows.p, set.octwin.p, set.nflat, B, set.cv, ctx.digits, set.nbit, set.group_ok.p, ctx.gok, wires ? 1 : 0, ctx.plane, ctx.plane_rows, ctx.plane_stride};
        if (fewm) {       // (a set whose windowed part has no latency layout: GSC_FEW_WIDE=0)
            const size_t nslices = ((set.nflat + 7) / 8 + 63) / 64;
            launch_msm_recode_flat_few(ra, n_real, ctx.stream);
            MsmFlatArgs a{set.ftable.p, set.rowoff.p, set.rowlen.p, set.nflat, ctx.digits, B, nslices, 512, pa, set.nbit, set.sub.p, ctx.gok, scalars, set.frows.p};
            if (timed) HIP_CHECK(hipEventRecord(ln.ev[5], ctx.stream));
            launch_flat_few(a, n_real, ctx.stream);
            if (timed) HIP_CHECK(hipEventRecord(ln.ev[6], ctx.stream));
            reduce_slices_few(ctx.stream, pa, pb, nslices, B, B, n_real, set.nwide ? flat : sum, launch_reduce_few);
        } else {
            const size_t nslices = msm_slices(set.nflat, 1, 256, B, per);
            launch_msm_recode_flat(ra, ctx.stream);
            MsmFlatArgs a{set.ftable.p, set.rowoff.p, set.rowlen.p, set.nflat, ctx.digits, B, nslices, per, pa, set.nbit, set.sub.p, ctx.gok, scalars, set.frows.p};
            launch_flat(a, ctx.stream);
            reduce_slices(ctx.stream, pa, pb, nslices, B, set.nwide ? flat : sum, launch_reduce);
        }
    }
    if (set.nwide) {
        // a single Prove call (lanes = bases): slices of 512 bases — 8 gathers + 6 butterfly additions per wave, and at most 64 partial
        // sums per column, which one reduction launch folds
        const bool few = fewm;
        size_t nslices = msm_slices(set.nwide, (size_t)set.nwin, WIN_SLICE, B, per);
        if (few && nslices > (set.nwide + 511) / 512) { per = 512; nslices = (set.nwide + 511) / 512; }
        const size_t Bw = B * (size_t)set.nwin;
        MsmRecodeArgs ra{scalars, set.wrows.p, wires ? 1 : 0, set.nwide, B, set.c, set.nwin, ctx.digits};
        if (!digits_ready) launch_msm_recode(ra, ctx.stream);
        MsmWinArgs a{set.wtable.p, set.c, set.nwin, set.nwide, ctx.digits, B, nslices, per, pa, timed && !few ? ln.d_clk.p : nullptr,
                     timed && cfg.z_exp_entry_bits > 0 && cfg.z_exp_entry_bits < 31 ? (1u << cfg.z_exp_entry_bits) - 1 : 0u};
        if (timed) HIP_CHECK(hipEventRecord(ln.ev[5], ctx.stream));
        if (few) launch_win_few(a, n_real, ctx.stream);
        else launch_win(a, ctx.stream);
        if (timed) HIP_CHECK(hipEventRecord(ln.ev[6], ctx.stream));
        if (few) reduce_slices_few(ctx.stream, pa, pb, nslices, Bw, B, n_real, sj, launch_reduce_few);
        else reduce_slices(ctx.stream, pa, pb, nslices, Bw, sj, launch_reduce);      // slices -> one sum per (window, proof)
        if (pending.n >= MSM_HORNER_JOBS) throw std::runtime_error("internal: too many pending Horner passes");
        pending.job[pending.n++] = MsmHornerJob{sj, set.nflat ? flat : (XyzzT*)nullptr, sum, set.nwin, set.c};
    }
    if (!set.nflat && !set.nwide) HIP_CHECK(hipMemsetAsync(sum, 0, B * sizeof(XyzzT), ctx.stream));      // empty set: the point at infinity
}

void AlgorithmImpl::run_msm_g1(Lane& ln, const MsmSet<G1Aff>& set, const fe* scalars, int mont, size_t B, G1Xyzz* sum, bool timed, bool side, bool digits_ready) {
    const int k = set_index(set);
    if (side) {
        if (!set.latency_flat() || B != 64) throw std::runtime_error("internal: side-stream MSM on a set with a windowed part");
        run_msm(ln, with_plane(MsmCtx{ln.side, ln.d_digits_s.p, ln.d_gok_s.p}, ln, scalars), set, scalars, mont != 0, B, ln.n_real, ln.d_part1c.p, ln.d_part1d.p, ln.d_sj1[k].p, ln.d_flat1[k].p, sum, false, false, ln.pending1,
                launch_msm_flat_g1, launch_msm_flat_few_g1, launch_msm_win_g1, launch_msm_win_few_g1, launch_msm_reduce_g1, launch_msm_reduce_few_g1);
        return;
    }
    run_msm(ln, with_plane(MsmCtx{ln.stream, ln.d_digits_w.p && &set != &mZ ? ln.d_digits_w.p : ln.d_digits.p, ln.d_gok.p}, ln, scalars), set, scalars, mont != 0, B, ln.n_real, ln.d_part1a.p, ln.d_part1b.p, ln.d_sj1[k].p, ln.d_flat1[k].p, sum, timed, digits_ready, ln.pending1, launch_msm_flat_g1, launch_msm_flat_few_g1, launch_msm_win_g1, launch_msm_win_few_g1, launch_msm_reduce_g1, launch_msm_reduce_few_g1);
}

void AlgorithmImpl::run_msm_g2(Lane& ln, const MsmSet<G2Aff>& set, const fe* scalars, int mont, size_t B, G2Xyzz* sum, bool side) {
    if (side && (!set.latency_flat() || B != 64)) throw std::runtime_error("internal: side-stream MSM on a set with a windowed part");
    run_msm(ln, with_plane(side ? MsmCtx{ln.side2, ln.d_digits_s2.p, ln.d_gok_s2.p} : MsmCtx{ln.stream, ln.d_digits_w.p ? ln.d_digits_w.p : ln.d_digits.p, ln.d_gok.p}, ln, scalars), set, scalars, mont != 0, B, ln.n_real, ln.d_part2a.p, ln.d_part2b.p, ln.d_sj2.p, ln.d_flat2.p, sum, false, false, ln.pending2, launch_msm_flat_g2, launch_msm_flat_few_g2, launch_msm_win_g2, launch_msm_win_few_g2, launch_msm_reduce_g2, launch_msm_reduce_few_g2);
}

void AlgorithmImpl::fetch_column(Lane& ln, const fe* mat, size_t rows, size_t B, size_t col, std::vector<uint8_t>& out) {
    out.resize(rows * 32);
    HIP_CHECK(hipMemcpy2DAsync(out.data(), 32, reinterpret_cast<const uint8_t*>(mat) + 32 * col, B * 32, 32, rows, hipMemcpyDeviceToHost, ln.stream));
    HIP_CHECK(hipStreamSynchronize(ln.stream));
}

void AlgorithmImpl::wipe_secrets(Lane& ln, size_t B, bool small_call) {
    if (cfg.keep_secrets) return;      // (test hook: shows that secret_residue() sees what the wipe removes)
    const size_t n_secret = n_inputs - n_public;
    HIP_CHECK(hipMemsetAsync(ln.d_inputs.p, 0, 176 * B, ln.stream));
    HIP_CHECK(hipMemsetAsync(ln.d_rs.p, 0, 64 * B, ln.stream));
    HIP_CHECK(hipMemsetAsync(ln.d_glv.p, 0, ln.d_glv.bytes(), ln.stream));
    if (has_commitment) { HIP_CHECK(hipMemsetAsync(ln.d_mask_in.p, 0, 32 * B, ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_mask.p, 0, B * sizeof(fe), ln.stream)); }
    HIP_CHECK(hipMemsetAsync(ln.d_W.p + n_public * B, 0, n_secret * B * sizeof(fe), ln.stream));      // the key wires
    HIP_CHECK(hipMemsetAsync(ln.d_W.p + n_wires * B, 0, 3 * B * sizeof(fe), ln.stream));               // r, s, -rs
    if (small_call) HIP_CHECK(hipMemset2DAsync(ln.d_W8.p + n_public * 64, (size_t)small.rows_per_group * 64, 0, n_secret * 64, B / 64, ln.stream));
}

size_t AlgorithmImpl::secret_residue() {
    HIP_CHECK(hipSetDevice(cfg.device));
    size_t left = 0;
    auto count = [&](const void* p, size_t bytes) {
        std::vector<uint8_t> h(bytes);
        HIP_CHECK(hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost));
        for (uint8_t b : h) left += b != 0;
    };
    const size_t n_secret = n_inputs - n_public;
    for (auto& lp : lanes) {
        Lane& ln = *lp; const size_t B = ln.cap;
        HIP_CHECK(hipStreamSynchronize(ln.stream));
        count(ln.d_inputs.p, 176 * B); count(ln.d_rs.p, 64 * B); count(ln.d_glv.p, ln.d_glv.bytes());
        if (has_commitment) { count(ln.d_mask_in.p, 32 * B); count(ln.d_mask.p, B * sizeof(fe)); }
        // a chunk lays its rows out with ITS batch as the row stride (and wiped them in that layout): look where the lane's last chunk had them
        const size_t Bl = ln.last_batch;
        if (Bl) { count(ln.d_W.p + n_public * Bl, n_secret * Bl * sizeof(fe)); count(ln.d_W.p + n_wires * Bl, 3 * Bl * sizeof(fe)); }
        if (small.ok) for (size_t g = 0; g < B / 64; g++) count(ln.d_W8.p + (g * small.rows_per_group + n_public) * 64, n_secret * 64);
    }
    return left;
}

void AlgorithmImpl::prove_chunk(Lane& ln, const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* dbg, bool allow_few_solver, bool allow_small) {
    const size_t B = (n + 63) / 64 * 64;
    ln.n_real = n;
    const bool trace = cfg.trace_host;
    const auto tc0 = std::chrono::steady_clock::now();
    // staging is the lane's pinned memory; what it holds of the statements' secrets (keys, randomness, masks) is cleared when the call leaves, however it leaves
    struct StagingWiper { Lane& l; size_t B; ~StagingWiper() { explicit_bzero(l.h_in.p, 176 * B); explicit_bzero(l.h_rs.p, 64 * B); explicit_bzero(l.h_glv.p, l.h_glv.bytes()); if (l.h_mask.p) explicit_bzero(l.h_mask.p, 32 * B); } } wipe_staging{ln, B};
    // a call that leaves by an exception must not leave work behind on the lane's other streams (the next call would share its buffers with it)
    struct Drain { Lane& l; int live = std::uncaught_exceptions(); ~Drain() { if (std::uncaught_exceptions() > live) { (void)hipStreamSynchronize(l.side); (void)hipStreamSynchronize(l.side2); (void)hipStreamSynchronize(l.stream); } } } drain{ln};
    uint8_t* const h_in = ln.h_in.p; uint8_t* const h_rs = ln.h_rs.p; GlvSplit* const h_glv = ln.h_glv.p;
    pack_inputs(reqs, n, B, h_in, h_rs);
    ln.d_inputs.upload(h_in, 176 * B, ln.stream);
    ln.d_rs.upload(h_rs, 64 * B, ln.stream);
    if (n <= (size_t)cfg.few_max && cfg.few_path && B == 64) {      // latency path: the two halves of s and r for k_fin_scalarmul_few
        for (size_t i = 0; i < n; i++) for (int role = 0; role < 2; role++) {
            uint32_t w[8]; memcpy(w, h_rs + 64 * i + (role == 0 ? 32 : 0), 32);
            if (!glv_split(w, h_glv[2 * i + role])) throw std::runtime_error("internal: scalar split out of range");
        }
        ln.d_glv.upload(h_glv, 2 * n, ln.stream);
    }
    HIP_CHECK(hipMemsetAsync(ln.d_flags.p, 0, ln.d_flags.bytes(), ln.stream));
    HIP_CHECK(hipEventRecord(ln.ev[0], ln.stream));
    // 1. witness
    if (cipher == CHACHA20) launch_assign_chacha(ln.d_inputs.p, ln.d_W.p, B, ln.stream);
    else launch_assign_aes(ln.d_inputs.p, cipher == AES_128 ? 16 : 32, ln.d_W.p, B, ln.stream);
    if (has_commitment) {
        for (size_t i = 0; i < B; i++) memcpy(ln.h_mask.p + 32 * i, reqs[i < n ? i : n - 1].mask, 32);
        ln.d_mask_in.upload(ln.h_mask.p, 32 * B, ln.stream);
    }
    launch_prep_rs(ln.d_rs.p, ln.d_W.p, n_wires, B, has_commitment ? ln.d_mask_in.p : nullptr, ln.d_mask.p, ln.stream);
    HIP_CHECK(hipMemsetAsync(ln.d_status.p, 0xFF, B * 4, ln.stream));
    SolverArgs sa{prog.p, sched.p, 0, n_levels, coeff.p, coeff_inv.p, lookup_coeff.p, ln.d_W.p, ln.d_A.p, ln.d_B.p, ln.d_C.p, B, ln.d_status.p,
                  has_commitment ? ln.d_mask.p : nullptr, has_commitment ? ln.d_commit.p : nullptr, has_div, 0u, nullptr};
    DevBuf<unsigned long long> d_trace;
    const bool strace = cfg.solver_trace;
    if (strace) {
        std::vector<unsigned long long> init(16 * ((size_t)n_levels + 1), 0ull);
        if (!few_solver_wanted(n, B)) for (uint32_t l = 0; l < n_levels; l++) init[16 * l] = ~0ull;
        d_trace.alloc(init.size()); HIP_CHECK(hipMemcpy(d_trace.p, init.data(), init.size() * 8, hipMemcpyHostToDevice)); sa.trace = d_trace.p;
    }
    const bool latency_call0 = n <= (size_t)cfg.few_max && cfg.few_path && B == 64;
    const bool small_call0 = small.ok && allow_small && !strace && (!latency_call0 || cfg.small_witness_few);
    bool few_solver = few_solver_wanted(n, B) && allow_few_solver && !small_call0;      // (the small-integer witness path needs no resident grid)
    if (few_solver) {      // a recent give-up on this replica: skip the resident kernel for a while (see few_skip)
        uint32_t k = few_skip.load();
        while (k && !few_skip.compare_exchange_weak(k, k - 1)) {}
        if (k) few_solver = false;
    }
    const bool latency_call = n <= (size_t)cfg.few_max && cfg.few_path && B == 64;      // the call takes the latency kernels
    // Circuits whose witness is small integers (ChaCha20-V3): the integer kernels on byte planes (wit_small.hpp) instead of the level launches
    const bool small_call = small_call0;
    ln.small_active = small_call;
    if (few_solver) HIP_CHECK(hipMemsetAsync(ln.d_fsync.p + 1, 0, 4, ln.stream));      // set by a resident launch that gave up at a barrier
    SolverFewArgs fa{few_ops.p, few_terms.p, few_lstart.p, 0, 0, coeff.p, coeff_inv.p, lookup_coeff.p, ln.d_W.p, ln.d_A.p, ln.d_B.p, ln.d_C.p, B, (uint32_t)n,
                     ln.d_status.p, sa.mask, sa.commit, ln.d_fsync.p, 1u << 21, 0u, n_levels, nullptr};
    if (cfg.few_test_abort) { fa.poll_limit = 256; fa.test_missing = 1; }      // test: the barrier never fills
    auto run_levels = [&](uint32_t from, uint32_t to) {
        for (uint32_t l = from; l < to; l++) {
            sa.first_level = l; sa.n_long = level_long[l];
            if (level_kind[l]) {
                if (few_solver) launch_solver_count_few(sa, few_count_ops.p, few_count_qoff.p, few_count_first[l], level_width[l], n, ln.stream);
                else launch_solver_count_level(sa, level_width[l], ln.stream);
            } else if (few_solver) {                   // a run of generic levels: one launch, device-wide barriers in between
                uint32_t e = l + 1; while (e < to && !level_kind[e]) e++;
                fa.from = l; fa.to = e; fa.trace = sa.trace;
                HIP_CHECK(hipMemsetAsync(ln.d_fsync.p, 0, 4, ln.stream));
                {
                    FewSolverChain& chain = few_solver_chain(cfg.device);
                    std::lock_guard<std::mutex> lk(chain.m);
                    if (chain.last && chain.last != ln.ev_few) HIP_CHECK(hipStreamWaitEvent(ln.stream, chain.last, 0));
                    // measured: 128 workgroups best for 1-2 statements, 256 (one per CU) beyond; never more than the device has CUs
                    // (every workgroup must be resident: one per CU by construction) — the kernel works with any grid
                    uint32_t wgs = cfg.few_workgroups ? (uint32_t)cfg.few_workgroups : (n <= 2 ? 128u : 256u);
                    if (wgs > (uint32_t)cu_count) wgs = (uint32_t)cu_count;
                    launch_solver_few(fa, has_div, wgs, ln.stream);
                    HIP_CHECK(hipEventRecord(ln.ev_few, ln.stream));
                    chain.last = ln.ev_few;
                }
                l = e - 1;
            } else launch_solver_level(sa, level_width[l], ln.stream);
        }
    };
    uint8_t* const h_cpts = ln.h_cpts.p;
    if (latency_call) HIP_CHECK(hipEventRecord(ln.ev[5], ln.stream));      // dominant kernel of a latency-path call: the witness solver
    if (has_commitment) {
        // Groth16 commitment (gnark "BSB22"): solve up to the commitment hint, D = sum w_j * Basis_j over the
        // committed wires (same MSM kernels as everything else), challenge = hash_to_field(D uncompressed) on the device, resume:
        // nothing leaves the stream.
        run_levels(0, commit_level);
        run_msm_g1(ln, mPed, ln.d_W.p, 1, B, ln.d_sumD.p);
        flush_horner_g1(ln, B, ln.stream);
        launch_points_to_affine_be(ln.d_sumD.p, B, ln.d_cpts.p, ln.d_flags.p, 8, ln.stream);
        launch_challenge_from_point(ln.d_cpts.p, ln.d_commit.p, B, ln.stream);
        run_levels(commit_level, n_levels);
    } else if (small_call) {
        HIP_CHECK(hipMemsetAsync(ln.d_wsflag.p, 0, 4, ln.stream));
        launch_wit_narrow(ln.d_W.p, B, n_inputs, ln.d_W8.p, small.rows_per_group, ln.d_wsflag.p, ln.stream);
        launch_wit_chain(WitChainArgs{ws_tiny.p, ws_parts.p, ws_bits.p, ws_twire.p, ws_tcoef.p, ws_levels.p, small.n_levels, ln.d_W8.p, small.rows_per_group, ln.d_wsflag.p}, B / 64, 512 * (size_t)small.max_slots, ln.stream);
        const uint32_t per_chunk = 4 * WS_IB;
        launch_wit_rows(WitRowsArgs{ws_rtiny.p, small.n_rtiny, per_chunk, (small.n_rtiny + per_chunk - 1) / per_chunk, ws_rgen.p, small.n_rgen, ws_rtwire.p, ws_rtcoef.p, ln.d_W8.p, small.rows_per_group,
                                    ln.d_A8.p, ln.d_B8.p, ln.d_C8.p, n_constraints, ln.d_A.p, ln.d_B.p, ln.d_C.p, B, ln.d_status.p, ln.d_wsflag.p}, B / 64, ln.stream);
        // The consumers (first transform kernel, flat recoders) read the byte planes; the 32-byte matrices only hold the rows predicted wide.
        if (dbg) {      // debug dumps want every row as a 32-byte element
            launch_wit_expand(ln.d_W8.p, small.rows_per_group, n_wires, nullptr, ln.d_W.p, B, ln.stream);
            launch_wit_expand(ln.d_A8.p, n_constraints, n_constraints, ws_cls_a.p, ln.d_A.p, B, ln.stream);
            launch_wit_expand(ln.d_B8.p, n_constraints, n_constraints, ws_cls_b.p, ln.d_B.p, B, ln.stream);
            launch_wit_expand(ln.d_C8.p, n_constraints, n_constraints, ws_cls_c.p, ln.d_C.p, B, ln.stream);
        }
    } else run_levels(0, n_levels);
    if (latency_call) HIP_CHECK(hipEventRecord(ln.ev[6], ln.stream));
    if (strace) {
        HIP_CHECK(hipStreamSynchronize(ln.stream));
        std::vector<unsigned long long> t(16 * ((size_t)n_levels + 1));
        HIP_CHECK(hipMemcpy(t.data(), d_trace.p, t.size() * 8, hipMemcpyDeviceToHost));
        { const unsigned long long* w = t.data() + 16 * (size_t)n_levels; if (w[2] > w[0]) fprintf(stderr, "last launch: %.1f us, shader clock %.0f MHz\n", (double)(w[2] - w[0]) / 100.0, (double)(w[3] - w[1]) / ((double)(w[2] - w[0]) / 100.0)); }
        fprintf(stderr, "solver trace: us after the level's first stamp (0 = not taken) | next level starts\n");
        for (uint32_t l = 0; l < n_levels; l++) {
            if (level_kind[l]) continue;
            fprintf(stderr, "level %3u w %4u long %3u |", l, level_width[l], level_long[l]);
            for (int k = 1; k < 13; k++) fprintf(stderr, " %6.2f", t[16 * l + k] ? (double)(t[16 * l + k] - t[16 * l]) / 100.0 : 0.0);
            if (l + 1 < n_levels && !level_kind[l + 1]) fprintf(stderr, " | %6.2f", (double)(t[16 * l + 16] - t[16 * l]) / 100.0);
            fprintf(stderr, "\n");
        }
    }
    HIP_CHECK(hipEventRecord(ln.ev[1], ln.stream));
    if (dbg) {
        dbg->n_wires = n_wires; dbg->n_constraints = n_constraints; dbg->n = domain_n;
        fetch_column(ln, ln.d_W.p, n_wires, B, 0, dbg->W); fetch_column(ln, ln.d_A.p, n_constraints, B, 0, dbg->A);
        fetch_column(ln, ln.d_B.p, n_constraints, B, 0, dbg->B); fetch_column(ln, ln.d_C.p, n_constraints, B, 0, dbg->C);
    }
    // A latency-path call leaves the chip mostly idle, so its A and B1 sums and the two scalar multiplications that need them (s * Ar,
    // r * Bs1: 254 serial doublings, 2 ms) start on the side stream right after the witness, beside the quotient and the other MSMs.
    const bool early_ab = ln.n_real <= (size_t)cfg.few_max && cfg.few_path && B == 64 && mA.latency_flat() && mB1.latency_flat();
    if (early_ab) {
        HIP_CHECK(hipEventRecord(ln.ev_ab, ln.stream));
        HIP_CHECK(hipStreamWaitEvent(ln.side, ln.ev_ab, 0));
        run_msm_g1(ln, mA, ln.d_W.p, 1, B, ln.d_sumA.p, false, true);
        run_msm_g1(ln, mB1, ln.d_W.p, 1, B, ln.d_sumB1.p, false, true);
        launch_fin_scalarmul_few(ln.d_sumA.p, ln.d_sumB1.p, ln.d_glv.p, B, ln.n_real, ln.d_out.p, ln.d_flags.p, ln.d_tmp.p, ln.side);
    }
    const bool early_b2 = early_ab && mB2.latency_flat();         // the G2 sum too (it only reads the witness): a third stream
    if (early_b2) {
        HIP_CHECK(hipStreamWaitEvent(ln.side2, ln.ev_ab, 0));
        run_msm_g2(ln, mB2, ln.d_W.p, 1, B, ln.d_sumB2.p, true);
        HIP_CHECK(hipEventRecord(ln.ev_s2, ln.side2));
    }
    // 2. quotient polynomial.  Coefficient form: h overwrites A, canonical, bit-reversed order (six transforms).  Evaluation form (batch calls,
    // k_quot_bases.hip): d = A B on the zeta-coset overwrites A, natural order (four transforms); c stays where the solver wrote it.
    NttPlan plan{L, tw_fwd.p, tw_inv.p, scale_mid.p, scale_out.p, dom.p + 5, qr.p, trace ? ln.d_clk.p + 32 : nullptr, tw_inv_plain.p, scale_mid_plain.p};
    const NttNarrow planes{{ln.d_A8.p, ln.d_B8.p, ln.d_C8.p}, n_constraints, cfg.ntt_plain};
    const NttNarrow* narrow = small_call ? &planes : nullptr;      // a, b (and c) of this chunk are byte planes
    HIP_CHECK(hipGetLastError());      // witness launches (launch-configuration errors are not sticky: check each group)
    const bool few_call = ln.n_real <= (size_t)cfg.few_max && cfg.few_path;
    const bool use_zfew = few_call && mZfew.nflat;                 // the latency layout holds the key's own Z: coefficient form
    const bool eval = quotient_eval && !use_zfew;
    const bool z_digits_ready = eval && fuse_z_digits && !few_call;
    // The evaluation-form quotient of a batch call reads a and b (rows or byte planes) and writes d over a and its digits into the Z set's own
    // digit buffer; the wire-set sums (A, B1, B2, K, and c over mC) read W and c and recode into d_digits_w.  Nothing is shared, so the
    // three quotient kernels go to the lane's third stream and the wire sets' thin tails (slice reductions, Horner chains, recoders: 5 ms of
    // a 1024-statement call's 54, none of it chip-filling) run under them; the Z sum waits for both.  Measured (profiles/r04k_overlap_quotient.txt):
    // 64 / 256 / 512 / 1024 statements per call +2 / +8 / +6 / +3.5 %, 8192 +0.1 %, AES-128 1024 / 256 per call +2.8 / +7 % — once the lanes'
    // streams stopped sharing hardware queues (alloc_lane); before that, calls on the small lanes lost 2 - 5 % to it.  Calls of 4096 statements
    // and more keep the one-stream order: nothing to gain (their tails are 1 % of the call), and the stage times stay those of the kernels.
    const bool overlap_q = z_digits_ready && !dbg && (cfg.overlap_quotient == 2 || (cfg.overlap_quotient && B < OVERLAP_QUOTIENT_BELOW));
    hipStream_t qs = overlap_q ? ln.side2 : ln.stream;
    if (eval) {
        if (dbg) {      // the debug vector is h itself: the coefficient-form kernels on copies (they overwrite their inputs)
            DevBuf<fe> ta(domain_n * B), tb(domain_n * B), tc(domain_n * B);
            HIP_CHECK(hipMemcpyAsync(ta.p, ln.d_A.p, n_constraints * B * sizeof(fe), hipMemcpyDeviceToDevice, ln.stream));
            HIP_CHECK(hipMemcpyAsync(tb.p, ln.d_B.p, n_constraints * B * sizeof(fe), hipMemcpyDeviceToDevice, ln.stream));
            HIP_CHECK(hipMemcpyAsync(tc.p, ln.d_C.p, n_constraints * B * sizeof(fe), hipMemcpyDeviceToDevice, ln.stream));
            HIP_CHECK(launch_compute_h(plan, ta.p, tb.p, tc.p, n_constraints, B, ln.stream, 0));
            fetch_column(ln, ta.p, domain_n, B, 0, dbg->H);
        }
        if (overlap_q) HIP_CHECK(hipStreamWaitEvent(qs, ln.ev[1], 0));      // the end of the witness stage
        if (fuse_z_digits && !few_call) HIP_CHECK(launch_compute_d_digits(plan, ln.d_A.p, ln.d_B.p, n_constraints, B, QuotDigits{ln.d_digits.p, mZ.c, mZ.nwin}, qs, narrow));
        else HIP_CHECK(launch_compute_d(plan, ln.d_A.p, ln.d_B.p, n_constraints, B, ln.stream, few_call ? ln.n_real : 0, narrow));
    } else HIP_CHECK(launch_compute_h(plan, ln.d_A.p, ln.d_B.p, ln.d_C.p, n_constraints, B, ln.stream, few_call ? ln.n_real : 0, narrow));      // latency path: the statements' columns only
    HIP_CHECK(hipEventRecord(ln.ev[2], qs));
    if (dbg && !eval) fetch_column(ln, ln.d_A.p, domain_n, B, 0, dbg->H);
    // 3. MSMs.  (With fuse_z_digits the digits of d are already in the lane's Z digit buffer; every other set recodes into d_digits_w.)
    // A and B1 first: the two scalar multiplications of the assembly only need those two sums and run on a side stream
    // beside the remaining MSMs.
    if (!early_ab) {
        run_msm_g1(ln, mA, ln.d_W.p, 1, B, ln.d_sumA.p);
        run_msm_g1(ln, mB1, ln.d_W.p, 1, B, ln.d_sumB1.p);
        flush_horner_g1(ln, B, ln.stream);                                       // (AES-V2: the wide wires of A and B1; nothing for ChaCha20-V3)
        HIP_CHECK(hipEventRecord(ln.ev_ab, ln.stream));
        HIP_CHECK(hipStreamWaitEvent(ln.side, ln.ev_ab, 0));
        launch_fin_scalarmul(ln.d_sumA.p, ln.d_sumB1.p, ln.d_rs.p, B, ln.d_out.p, ln.d_flags.p, ln.d_tmp.p, ln.side);
    }
    if (!early_b2) run_msm_g2(ln, mB2, ln.d_W.p, 1, B, ln.d_sumB2.p);
    if (ln.pending2.n) {                                                         // the G2 Horner chain (3x a G1 one) also goes beside the MSMs
        HIP_CHECK(hipEventRecord(ln.ev_b2, ln.stream));
        HIP_CHECK(hipStreamWaitEvent(ln.side, ln.ev_b2, 0));
        flush_horner_g2(ln, B, ln.side);
    }
    HIP_CHECK(hipEventRecord(ln.ev_fs, ln.side));
    run_msm_g1(ln, mK, ln.d_W.p, 1, B, ln.d_sumK.p);
    if (eval) {                                                                  // sum c_i U_i: the solver's c rows, laid out like a wire set
        // the padding slots of mC read row n of c, which must be zero for THIS batch's row stride (an earlier, larger batch had other rows there)
        HIP_CHECK(hipMemsetAsync(ln.d_C.p + domain_n * B, 0, B * sizeof(fe), ln.stream));
        run_msm_g1(ln, mC, ln.d_C.p, 1, B, ln.d_sumC.p);
    }
    if (overlap_q) { HIP_CHECK(hipEventRecord(ln.ev_ws, ln.stream)); HIP_CHECK(hipStreamWaitEvent(ln.stream, ln.ev[2], 0)); }      // the wire sets are through; wait for d and its digits
    run_msm_g1(ln, use_zfew ? mZfew : mZ, ln.d_A.p, 0, B, ln.d_sumZ.p, !latency_call, false, z_digits_ready);
    if (has_commitment) run_msm_g1(ln, mPedSigma, ln.d_W.p, 1, B, ln.d_sumPok.p);      // proof of knowledge of the commitment: same scalars over sigma * Basis
    flush_horner_g1(ln, B, ln.stream);                                           // K, Z, PedSigma: one launch
    if (has_commitment) launch_points_to_affine_be(ln.d_sumPok.p, B, ln.d_cpts.p + 64 * B, ln.d_flags.p, 16, ln.stream);
    HIP_CHECK(hipGetLastError());      // MSM launches
    HIP_CHECK(hipEventRecord(ln.ev[3], ln.stream));
    // 4. assembly
    HIP_CHECK(hipStreamWaitEvent(ln.stream, ln.ev_fs, 0));
    if (early_b2) HIP_CHECK(hipStreamWaitEvent(ln.stream, ln.ev_s2, 0));
    launch_fin_combine(ln.d_sumB2.p, ln.d_sumK.p, ln.d_sumZ.p, eval ? ln.d_sumC.p : nullptr, ln.d_tmp.p, B, ln.d_out.p, ln.d_flags.p, ln.stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(ln.ev[4], ln.stream));
    uint8_t* const h_out = ln.h_out.p; uint8_t* const h_flags = ln.h_flags.p; uint32_t* const h_status = ln.h_status.p;
    HIP_CHECK(hipMemcpyAsync(h_out, ln.d_out.p, 256 * B, hipMemcpyDeviceToHost, ln.stream));
    HIP_CHECK(hipMemcpyAsync(h_flags, ln.d_flags.p, (B + 3) / 4 * 4, hipMemcpyDeviceToHost, ln.stream));
    HIP_CHECK(hipMemcpyAsync(h_status, ln.d_status.p, B * 4, hipMemcpyDeviceToHost, ln.stream));
    if (has_commitment) HIP_CHECK(hipMemcpyAsync(h_cpts, ln.d_cpts.p, 128 * B, hipMemcpyDeviceToHost, ln.stream));      // commitment | its proof of knowledge
    // the small result words, in the lane's pinned block: [0] the resident solver's two sync words, [1] the small-integer path's flag, [2 .. 18) clock stamps
    unsigned long long* const hw = ln.h_words.p; memset(hw, 0, ln.h_words.bytes());
    uint32_t* const h_fsync = reinterpret_cast<uint32_t*>(hw); uint32_t& h_wsflag = *reinterpret_cast<uint32_t*>(hw + 1);
    if (few_solver) HIP_CHECK(hipMemcpyAsync(h_fsync, ln.d_fsync.p, 8, hipMemcpyDeviceToHost, ln.stream));
    if (small_call) HIP_CHECK(hipMemcpyAsync(&h_wsflag, ln.d_wsflag.p, 4, hipMemcpyDeviceToHost, ln.stream));
    wipe_secrets(ln, B, small_call);      // behind the last kernel of the chunk, inside the wait below
    unsigned long long* const h_clk = hw + 2;
    if (!latency_call) { HIP_CHECK(hipMemcpyAsync(h_clk, ln.d_clk.p, (trace ? 44 : 32) * 8, hipMemcpyDeviceToHost, ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_clk.p, 0, 32 * 8, ln.stream)); }
    const auto tc1 = std::chrono::steady_clock::now();
    HIP_CHECK(hipStreamSynchronize(ln.stream));
    const auto tc2 = std::chrono::steady_clock::now();
    if (h_fsync[1]) {      // the resident solver gave up (its workgroups never became resident together: another process's kernel on this device)
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true)) fprintf(stderr, "libprove: the resident witness kernel could not hold the device (shared with another process?); solving level by level\n");
        const uint32_t pen = few_penalty.load();
        few_skip.store(pen); few_penalty.store(pen < 4096 ? pen * 2 : 4096);
        return prove_chunk(ln, reqs, n, results, dbg, false, allow_small);
    }
    if (h_wsflag) {      // a value did not fit the byte plane it was predicted for: the whole chunk again with the generic solver (results never depend on predictions)
        small_fallbacks++;
        return prove_chunk(ln, reqs, n, results, dbg, allow_few_solver, false);
    }
    if (few_solver) few_penalty.store(16);
    for (int k = 0; k < 4; k++) { float ms = 0; (void)hipEventElapsedTime(&ms, ln.ev[k], ln.ev[k + 1]); ln.stage_ms[k] = ms; }
    if (overlap_q) {
        // The quotient ran beside the wire-set MSMs: the stages stay a partition of the call's time, with the shared span charged to the MSMs —
        // msm = (witness end .. wire sets through) + (quotient end .. MSMs through), quotient = what it still ran alone after the wire sets
        float ws = 0, q = ln.stage_ms[1];
        (void)hipEventElapsedTime(&ws, ln.ev[1], ln.ev_ws);
        if (ws > q) ws = q;
        ln.stage_ms[1] = q - ws; ln.stage_ms[2] += ws;
    }
    (void)hipEventElapsedTime(&ln.msm_z_kernel_ms, ln.ev[5], ln.ev[6]); ln.last_batch = B;
    {
        KernelStat& st = ln.stat;
        st.name = latency_call ? (small_call ? "k_wit_chain + k_wit_rows" : few_solver ? (has_commitment ? "k_solver_few + commitment MSM" : "k_solver_few") : "k_solver (one launch per level)") : "k_msm_win<Fp29f>";
        st.ms = ln.msm_z_kernel_ms; st.statements = n; st.columns = B; st.nbases = mZ.nwide; st.nwin = mZ.nwin;
        for (int k = 0; k < 4; k++) st.stage_ms[k] = ln.stage_ms[k];
        // shader clock of the Z launch: (shader-clock ticks) / (100 MHz ticks) over the lives of eight waves spread over the grid, one on each XCD
        // (the XCDs are clocked separately; a pair of stamps from two different waves is useless: the shader-clock counters are not chip-wide —
        // first-start-to-last-end read 1 772 ... 2 323 MHz on launches whose waves all saw 2 010 ... 2 069)
        double ticks = 0, shader = 0;
        if (!latency_call) for (int k = 0; k < 8; k++) { const unsigned long long* c = h_clk + 4 * k; if (c[2] > c[0] && c[3] > c[1]) { ticks += (double)(c[2] - c[0]); shader += (double)(c[3] - c[1]); } }
        st.clock_mhz = ticks > 0 ? (float)(100.0 * shader / ticks) : 0.f;
        std::lock_guard<std::mutex> lk(stat_mu);
        last_stat = st;
    }
    for (size_t i = 0; i < n; i++)
        serialize(h_out + 256 * i, h_flags[i], h_status[i], has_commitment ? h_cpts + 64 * i : nullptr, has_commitment ? h_cpts + 64 * B + 64 * i : nullptr, results[i]);
    if (trace) {
        const auto tc3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "prove_chunk(%zu): enqueue %.2f ms, wait %.2f ms, serialise %.2f ms\n", n, ms(tc0, tc1), ms(tc1, tc2), ms(tc2, tc3));
        auto mhz = [&](int k) { const unsigned long long* c = h_clk + 4 * k; return c[2] > c[0] && c[3] > c[1] ? 100.0 * (double)(c[3] - c[1]) / (double)(c[2] - c[0]) : 0.0; };
        if (!latency_call) fprintf(stderr, "prove_chunk(%zu): shader clock (one workgroup in the middle of each launch): transforms %.0f / %.0f / %.0f MHz, Z kernel %.0f MHz (eight waves, one per XCD: %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f); stages %.2f / %.2f / %.2f / %.2f ms\n", n, mhz(8), mhz(9), mhz(10), (double)ln.stat.clock_mhz, mhz(0), mhz(1), mhz(2), mhz(3), mhz(4), mhz(5), mhz(6), mhz(7),
                                   ln.stage_ms[0], ln.stage_ms[1], ln.stage_ms[2], ln.stage_ms[3]);
    }
}

void AlgorithmImpl::serialize(const uint8_t* o, uint8_t flags, uint32_t status, const uint8_t* commitment_xy, const uint8_t* pok_xy, ProofResult& res) const {
    res.proof_len = 0; res.status = 0;
    if (status != 0xFFFFFFFFu) { res.status = 1; return; }
    if (flags) { res.status = 2; return; }
    uint8_t* p = res.proof;
    auto g1 = [&](const uint8_t* xy, uint8_t* dst) {
        uint8_t y[32]; le_limbs_to_be(xy, dst); le_limbs_to_be(xy + 32, y);
        dst[0] |= be_greater(y, kHalfP) ? 0xC0 : 0x80;
    };
    g1(o, p);
    {   // G2: X.A1 | X.A0, flag from y (A1 unless zero, then A0)
        uint8_t y0[32], y1[32];
        le_limbs_to_be(o + 96, p + 32); le_limbs_to_be(o + 64, p + 64);
        le_limbs_to_be(o + 128, y0); le_limbs_to_be(o + 160, y1);
        const bool large = be_is_zero(y1) ? be_greater(y0, kHalfP) : be_greater(y1, kHalfP);
        p[32] |= large ? 0xC0 : 0x80;
    }
    g1(o + 192, p + 96);
    if (!commitment_xy) {
        p[128] = p[129] = p[130] = p[131] = 0;          // no commitments (ChaCha20-V3)
        memset(p + 132, 0, 32); p[132] = 0x40;          // CommitmentPok = point at infinity
        res.proof_len = 164;
    } else {                                            // one commitment + its proof of knowledge (AES-V2)
        p[128] = p[129] = p[130] = 0; p[131] = 1;
        auto g1be = [&](const uint8_t* xy, uint8_t* dst) { memcpy(dst, xy, 32); dst[0] |= be_greater(xy + 32, kHalfP) ? 0xC0 : 0x80; };
        g1be(commitment_xy, p + 132); g1be(pok_xy, p + 164);
        res.proof_len = 196;
    }
}
}  // namespace gsc
