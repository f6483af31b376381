// The engine behind one algorithm on one device: key tables (engine_tables.hip), the per-batch device pipeline (engine_prove.hip)
// and the lanes that carry batches.  Internal to the library: engine.hpp is the interface capi.cpp sees; engine.hip holds the replica
// dispatch (Algorithm).  Reference counterpart: the per-cipher prover objects of libraries/prover/impl/provers.go:61-77
// (baseProver{r1cs, pk}: SetParams -> tables, Prove -> pipeline).
#pragma once
#include "engine.hpp"
#include "formats.hpp"
#include "kernels.hpp"
#include "wit_small.hpp"
#include <atomic>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace gsc {

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void alloc(size_t count) { if (p) { (void)hipFree(p); p = nullptr; } n = count; if (count) HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T))); }
    void upload(const T* src, size_t count, hipStream_t s) { HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s)); }
    size_t bytes() const { return n * sizeof(T); }
};

// page-locked host memory: the staging buffers of a lane (uploads and downloads are then real asynchronous DMA, not driver-staged copies)
template <class T>
struct PinnedBuf {
    T* p = nullptr; size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete; PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void alloc(size_t count) { if (p) { (void)hipHostFree(p); p = nullptr; } n = count; if (count) { HIP_CHECK(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault)); memset(p, 0, count * sizeof(T)); } }
    size_t bytes() const { return n * sizeof(T); }
};

template <class AffT>
struct MsmSet {                     // one fixed-base MSM of the proving key (kernels.hpp, "multi-scalar multiplication")
    size_t nbases = 0;              // bases of the key in this set
    // windowed part (uniform rows of 2^(c-1) multiples): every base of Z; the wide wires of a wire set when there are many
    DevBuf<AffT> wtable; DevBuf<uint32_t> wrows; size_t nwide = 0; int c = 0, nwin = 0;
    // flat part: [bit groups of eight][narrow wires, own row lengths][window octets of a few wide wires (cv-bit digits)]
    DevBuf<AffT> ftable; DevBuf<uint64_t> rowoff; DevBuf<uint32_t> rowlen; DevBuf<uint32_t> frows; DevBuf<int32_t> octwin;
    size_t nflat = 0, nbit = 0, nexpanded = 0; int cv = 0; DevBuf<AffT> sub; DevBuf<uint8_t> group_ok;
    // latency path: the windowed part once more as a flat set of (base, window) rows of 8-bit digits (no Horner pass behind it)
    std::unique_ptr<MsmSet<AffT>> few_wide;
    bool latency_flat() const { return !nwide || few_wide; }       // calls with a handful of statements need no windowed kernel for this set
};

// The resident solver kernel (k_solver_few) spins at device-wide barriers, so two of them must never share the device: each could
// hold CUs the other's missing workgroups are waiting for.  Launches are therefore chained on the device: a launch first makes its
// stream wait for the previous one's completion event (no host blocking).  Other processes on the same device are not covered —
// there the kernel's bounded polling gives up and the call is solved again with one launch per level (prove_chunk).
struct FewSolverChain { std::mutex m; hipEvent_t last = nullptr; };
FewSolverChain& few_solver_chain(int device);      // one per device, never destroyed (lanes may outlive static destructors): engine_prove.hip

class AlgorithmImpl {
  public:
    Cipher cipher; EngineConfig cfg;
    size_t n_wires = 0, n_public = 0, n_inputs = 0, n_constraints = 0, domain_n = 0; int L = 0;      // n_inputs: public + secret wires (what k_assign_* writes)
    bool has_commitment = false;
    // lanes are handed out one chunk at a time; concurrent calls (and the chunks of one call) take whichever lane is free
    std::mutex pool_mu; std::condition_variable pool_cv; std::vector<uint8_t> lane_busy;
    // a free lane that can hold n statements — the smallest such lane, so that small calls leave the full-capacity lanes to big ones
    size_t acquire_lane(int want = -1, size_t n = 0) {
        std::unique_lock<std::mutex> l(pool_mu);
        size_t got = 0;
        pool_cv.wait(l, [&] {
            bool found = false;
            for (size_t i = 0; i < lane_busy.size(); i++) {
                if (lane_busy[i] || (want >= 0 && (size_t)want != i) || lanes[i]->cap < n) continue;
                if (!found || lanes[i]->cap < lanes[got]->cap) { got = i; found = true; }
            }
            return found;
        });
        lane_busy[got] = 1;
        return got;
    }
    std::atomic<int> calls_in_flight{0};
    int cu_count = 256;                 // compute units of the device: the resident witness kernel needs one per workgroup
    // After a resident launch gave up (CUs held by someone else), the next few_skip calls of this replica go level by level at once
    // instead of spinning through the same timeouts; the penalty doubles up to 4096 calls and is forgotten after a success.
    std::atomic<uint32_t> few_skip{0}; std::atomic<uint32_t> few_penalty{16};
    std::mutex stat_mu; KernelStat last_stat;      // timing of the chunk that finished last on this replica
    void release_lane(size_t i) { { std::lock_guard<std::mutex> l(pool_mu); lane_busy[i] = 0; } pool_cv.notify_all(); }
    hipStream_t stream = nullptr;   // init-time work; proving runs on the lanes' streams
    size_t table_bytes = 0;
    std::vector<uint8_t> row_class;    // per scalar row (wire), predicted by calibrate(): 0 = always 0 or 1, 1 = also -1, else the largest bit length seen (255 = unknown)

    // program
    DevBuf<uint32_t> prog, sched, lookup_coeff; DevBuf<fe> coeff, coeff_inv;
    DevBuf<uint32_t> few_count_ops, few_count_qoff; std::vector<uint32_t> few_count_first;
    DevBuf<uint32_t> few_ops, few_terms, few_lstart;          // the same program laid out for k_solver_few (formats.hpp FewProgram)
    uint32_t n_levels = 0, commit_level = 0; std::vector<uint32_t> level_width; std::vector<uint8_t> level_kind; std::vector<uint32_t> level_long; int has_div = 0;
    // The small-integer witness path (wit_small.hpp): built after calibrate() for circuits that qualify (ChaCha20-V3); small.ok says so.
    // small keeps the sizes and the per-row classes; the item lists live on the device.
    SmallProgram small; std::vector<uint8_t> row_class_a, row_class_b;
    DevBuf<uint32_t> ws_tiny, ws_parts, ws_bits, ws_twire, ws_levels, ws_rtiny, ws_rgen, ws_rtwire; DevBuf<long long> ws_tcoef, ws_rtcoef;
    DevBuf<uint8_t> ws_cls_a, ws_cls_b, ws_cls_c;      // per constraint row: 0 = byte plane, 1 = 32-byte element
    void init_small(const SolverProgram& sp);
    static constexpr size_t OVERLAP_QUOTIENT_BELOW = 4096;      // batch calls smaller than this run the quotient beside the wire-set MSMs (prove_chunk)
    std::atomic<uint64_t> small_fallbacks{0};      // chunks that had to be solved again generically (gsc_describe)
    // NTT
    DevBuf<int32_t> tw_fwd, tw_inv, tw_inv_plain, qr; DevBuf<fe> scale_mid, scale_mid_plain, scale_out, dom;   // dom: omega, omega_inv, g, g_inv, n_inv, 16/n
    // MSM sets
    MsmSet<G1Aff> mA, mB1, mK, mZ, mZfew, mPed, mPedSigma; MsmSet<G2Aff> mB2;     // mPed*: Pedersen commitment bases (AES-V2)
    // Evaluation-form quotient (k_quot_bases.hip, cfg.quotient_eval): mZ then holds the bases V_i (scalars: d on the zeta-coset, launch_compute_d)
    // and mC the bases U_i of the constraint rows (scalars: the solver's c, laid out like a wire set from row_class_c); calls that take the
    // latency layout mZfew keep the coefficient form (it holds the key's own Z).
    bool quotient_eval = false; MsmSet<G1Aff> mC; std::vector<uint8_t> row_class_c;
    // ... and the last quotient kernel writes the digits of d itself (launch_compute_d_digits): mZ's table positions follow quot_digit_index,
    // whole batches skip the recoding pass; the other sets recode into a (small) digit buffer of their own meanwhile (Lane::d_digits_w)
    bool fuse_z_digits = false;
    // batch buffers: one set per lane.  A lane = a HIP stream with its own witness / polynomial / partial-sum buffers; with more than
    // one full lane big batches are cut into chunks that the lanes prove concurrently.  Measured on MI355X: for FULL
    // batches two lanes do not beat one (the MSM kernels fill the chip; chaining the heavy phases so that only the witness stage
    // overlaps was measured in round 3: no gain either), so ChaCha20-V3 has one full lane; calls of a few dozen to a few hundred
    // statements under-fill the chip and do gain from running side by side: the small lanes.
    struct Lane {
        hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;      // side: the assembly's scalar multiplications, beside the MSMs; side2: the B2 sum of a latency-path call
        hipEvent_t ev_ab = nullptr, ev_fs = nullptr, ev_b2 = nullptr, ev_s2 = nullptr;
        hipEvent_t ev_ws = nullptr;     // the wire-set MSMs of a call whose quotient runs beside them are through (stage accounting)
        hipEvent_t ev_few = nullptr;    // completion of this lane's latest k_solver_few launch (FewSolverChain)
        hipEvent_t ev[7] = {};          // 0..4 stage boundaries, 5..6 bracket the dominant kernel (Z-table MSM gather-accumulate)
        float stage_ms[4] = {0, 0, 0, 0}; float msm_z_kernel_ms = 0; size_t last_batch = 0;
        // pinned staging: inputs / randomness / masks up, proof coordinates / flags / status / commitment points and the small result words down
        PinnedBuf<uint8_t> h_in, h_rs, h_mask, h_out, h_flags, h_cpts; PinnedBuf<uint32_t> h_status; PinnedBuf<GlvSplit> h_glv; PinnedBuf<unsigned long long> h_words;
        KernelStat stat;                // of the chunk this lane proved last
        DevBuf<unsigned long long> d_clk;      // clock stamps: [0, 32) eight waves of the Z kernel (MsmWinArgs::clk); [32, 44) the three transform kernels (GSC_TRACE_HOST)
        size_t n_real = 0;              // statements of the chunk being proved (the batch is padded to a multiple of 64)
        size_t cap = 0;
        DevBuf<uint8_t> d_inputs, d_rs, d_out, d_flags, d_mask_in, d_cpts; DevBuf<uint32_t> d_status, d_fsync; DevBuf<GlvSplit> d_glv;
        DevBuf<fe> d_mask, d_commit; DevBuf<G1Xyzz> d_sumD, d_sumPok;
        DevBuf<fe> d_W, d_A, d_B, d_C;
        bool small_active = false;      // the chunk being proved left W, A, B, C as byte planes (the 32-byte matrices then only hold the rows marked wide)
        DevBuf<int8_t> d_W8, d_A8, d_B8, d_C8; DevBuf<uint32_t> d_wsflag;      // byte planes of the small-integer witness path; its "a prediction failed" flag
        DevBuf<G1Xyzz> d_part1a, d_part1b, d_sumA, d_sumB1, d_sumK, d_sumZ, d_sumC, d_tmp; DevBuf<G2Xyzz> d_part2a, d_part2b, d_sumB2;
        DevBuf<uint4> d_digits_s2; DevBuf<uint8_t> d_gok_s2;                                        // side2's digits (its partial sums are the G2 buffers, which nothing else uses)
        DevBuf<uint4> d_digits_s; DevBuf<uint8_t> d_gok_s; DevBuf<G1Xyzz> d_part1c, d_part1d;      // the side stream's MSM scratch (A and B1 of a latency-path call)
        DevBuf<uint4> d_digits;                                                   // signed digits [window][octet][proof]
        DevBuf<uint4> d_digits_w;      // fuse_z_digits: the digits of every set but Z (Z's are written by the last quotient kernel into d_digits and must survive until its sum runs)
        // per-window sums [window][proof] and flat-part sums [proof], one pair per set: the Horner passes of several sets are deferred
        // and run as one launch (MsmHornerJobs), so their inputs must not share storage
        static constexpr int NSETS = 8;      // A, B1, K, Z, Ped, PedSigma, Z (latency layout), C (evaluation-form quotient)
        DevBuf<G1Xyzz> d_sj1[NSETS], d_flat1[NSETS]; DevBuf<G2Xyzz> d_sj2, d_flat2;
        MsmHornerJobs pending1{}, pending2{};
        DevBuf<uint8_t> d_gok;                                                    // bit-group verdicts [group][wave of 64 proofs]
        ~Lane() { if (ev_few) { for (int d = 0; d < 64; d++) { FewSolverChain& c = few_solver_chain(d); std::lock_guard<std::mutex> lk(c.m); if (c.last == ev_few) c.last = nullptr; } (void)hipEventDestroy(ev_few); }
                  for (auto& e : ev) if (e) (void)hipEventDestroy(e); if (ev_ws) (void)hipEventDestroy(ev_ws); if (ev_ab) (void)hipEventDestroy(ev_ab); if (ev_fs) (void)hipEventDestroy(ev_fs); if (ev_b2) (void)hipEventDestroy(ev_b2); if (ev_s2) (void)hipEventDestroy(ev_s2); if (side2) (void)hipStreamDestroy(side2); if (side) (void)hipStreamDestroy(side); if (stream) (void)hipStreamDestroy(stream); }
    };
    std::vector<std::unique_ptr<Lane>> lanes;
    size_t cap = 0;                     // proofs per full lane = the largest chunk
    size_t full_lanes = 0;              // lanes [0, full_lanes) hold `cap` proofs; the rest are small lanes (SMALL_LANE_CAP)
    static constexpr size_t SMALL_LANE_CAP = 1024;     // with full lanes larger than this; half of it otherwise (alloc in engine_tables.hip)

    AlgorithmImpl(Cipher c, const uint8_t* pk, size_t pk_len, const uint8_t* r1cs, size_t r1cs_len, const EngineConfig& cf);
    ~AlgorithmImpl() { lanes.clear(); if (stream) (void)hipStreamDestroy(stream); }

    std::unique_ptr<SolverProgram> init_program(const R1csFile& cs);

    static void pack_inputs(const ProofRequest* reqs, size_t n, size_t B, uint8_t* h_in, uint8_t* h_rs);      // 176 B and 64 B per column

    // Which wires are bits?  Nothing in an R1CS says so, but it is a property of the circuit, not of the statement: solve 64
    // pseudo-random statements once and call a wire a bit when it is 0 or 1 in all of them.  This is only a PREDICTION used to
    // lay out the wire MSMs (bit wires first, in groups of eight with subset-sum tables); k_msm re-checks every group for
    // every wave of proofs and falls back to the digit tables, so a wrong prediction costs time, never correctness.
    void calibrate();

    // decompress `raw` (count points of `sz` bytes) into out[offset...]; returns per-point status
    std::vector<uint8_t> decompress_g1(const std::vector<uint8_t>& raw, G1Aff* out);
    std::vector<uint8_t> decompress_g2(const std::vector<uint8_t>& raw, G2Aff* out);

    static constexpr size_t EXPAND_MAX = 64;      // up to this many wide wires of a set are laid out as window octets of its flat part
    static constexpr int EXPAND_C = 15, NARROW_MAX_BITS = 14;

    // rows of multiples for `n` bases (row i: len[i] entries at off[i]); work is cut into segments of at most 256 multiples
    template <class AffT, class XyzzT>
    void build_rows(const AffT* bases, size_t n, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, AffT* table);
    void launch_build_rows(const G1Aff* b, const MsmRowSeg* sg, size_t n, uint32_t cap, G1Aff* t, G1Xyzz* sc) { launch_build_rows_g1(b, sg, n, cap, t, sc, stream); }
    void launch_build_rows(const G2Aff* b, const MsmRowSeg* sg, size_t n, uint32_t cap, G2Aff* t, G2Xyzz* sc) { launch_build_rows_g2(b, sg, n, cap, t, sc, stream); }
    void launch_shift(const G1Aff* in, const uint32_t* src, const uint32_t* sh, size_t n, G1Aff* out) { launch_shift_bases_g1(in, src, sh, n, out, stream); }
    void launch_shift(const G2Aff* in, const uint32_t* src, const uint32_t* sh, size_t n, G2Aff* out) { launch_shift_bases_g2(in, src, sh, n, out, stream); }

    // Lays out one MSM set and builds its tables.  uniform = true (Z): every base gets a full row, windowed kernel.  Otherwise the
    // bases are sorted by what calibrate() saw on their wires: values in {-1, 0, 1} -> bit groups of eight; values of up to
    // NARROW_MAX_BITS bits (with the margin) -> flat rows of that length; the rest (r, s, the lookup argument's products and inverses)
    // are wide: a few of them become window octets of the flat part, many get the windowed kernel and a Horner pass.
    template <class AffT, class XyzzT, class Decomp>
    // classes: the predicted class per scalar row (default: row_class, the wires); zero_row: a scalar row that is always zero (padding slots; default:
    // row n_wires + 3 of W); latency_layout: also build the (base, window) rows of the windowed part for calls with a handful of statements
    void build_set(MsmSet<AffT>& set, const std::vector<uint8_t>& raw, size_t point_bytes, const std::vector<uint32_t>& rows, int c, const char* what, Decomp decomp, bool uniform, int expand_cv = 0,
                   const std::vector<uint8_t>* classes = nullptr, uint32_t zero_row = 0xFFFFFFFFu, bool latency_layout = true);
    // group tables are built in chunks so that the projective scratch stays below ~2 GiB
    void build_subset(const G1Aff* b, size_t ng, G1Aff* t, uint8_t* ok);
    void build_subset(const G2Aff* b, size_t ng, G2Aff* t, uint8_t* ok);

    void init_key(const R1csFile& cs, const PkFile& key);

    void alloc_lane(Lane& ln, size_t B);

    // Waves of an MSM launch = slices x windows x groups of 64 proofs (windows = 1 for the flat kernel).  Slices of up to `most`
    // bases (256: few partial sums to reduce, a tail of < 2 % at full batches; measured 64 .. 512: kernel time within 1 %, reductions -3 ms); shorter ones when that would leave fewer than ~8k waves,
    // so that a small batch still spreads over the whole chip.
    static size_t msm_slices(size_t nbases, size_t nwin, size_t most, size_t B, size_t& per) {
        const size_t gw = (B / 64) * nwin, want = (8192 + gw - 1) / gw;
        size_t n = (nbases + most - 1) / most; if (n < want) n = want;
        n = (n + 7) & ~(size_t)7;
        per = ((nbases + n - 1) / n + 7) & ~(size_t)7; if (!per) per = 8;
        n = (nbases + per - 1) / per;
        return n ? n : 1;
    }
    size_t WIN_SLICE = 256;      // bases per slice of the windowed kernel at full batches (measured 64 .. 512: kernel time within 1 %)
    bool few_solver_wanted(size_t n, size_t B) const { return n <= (size_t)cfg.few_max && B == 64 && cfg.few_solver; }
    struct MsmCtx { hipStream_t stream; uint4* digits; uint8_t* gok; const int8_t* plane = nullptr; size_t plane_rows = 0, plane_stride = 0; };      // plane: the scalars' byte plane (small-integer witness path)
    // the byte plane that stands for the scalar matrix `scalars` in the chunk this lane is proving (none: the matrix holds every row)
    MsmCtx with_plane(MsmCtx c, const Lane& ln, const fe* scalars) const {
        if (!ln.small_active) return c;
        if (scalars == ln.d_W.p) { c.plane = ln.d_W8.p; c.plane_rows = n_wires; c.plane_stride = small.rows_per_group; }
        else if (scalars == ln.d_C.p) { c.plane = ln.d_C8.p; c.plane_rows = n_constraints; c.plane_stride = n_constraints; }
        return c;
    }
    template <class XyzzT, class LR>
    void reduce_slices(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, XyzzT* out, LR launch_reduce);
    // the same for the first `npr` columns of every row of `stride` (latency path: nobody reads the padding proofs' columns)
    template <class XyzzT, class LRF>
    void reduce_slices_few(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, size_t stride, size_t npr, XyzzT* out, LRF launch_reduce_few);
    // scalars: the wire matrix W (Montgomery; wire sets) or h (canonical; Z)
    // The Horner pass of the windowed part is NOT launched here: it is queued in `pending` and flushed together with those of other
    // sets (flush_horner_*), because each is a serial chain of 254 doublings whose duration does not depend on the batch.
    template <class AffT, class XyzzT, class LF, class LFF, class LW, class LWF, class LR, class LRF>
    void run_msm(Lane& ln, const MsmCtx& ctx, const MsmSet<AffT>& set, const fe* scalars, bool wires, size_t B, size_t n_real, XyzzT* pa, XyzzT* pb, XyzzT* sj, XyzzT* flat, XyzzT* sum, bool timed, bool digits_ready,
                 MsmHornerJobs& pending, LF launch_flat, LFF launch_flat_few, LW launch_win, LWF launch_win_few, LR launch_reduce, LRF launch_reduce_few);
    int set_index(const MsmSet<G1Aff>& set) const { const MsmSet<G1Aff>* all[Lane::NSETS] = {&mA, &mB1, &mK, &mZ, &mPed, &mPedSigma, &mZfew, &mC}; for (int k = 0; k < Lane::NSETS; k++) if (all[k] == &set) return k; return 0; }
    // side = true: on the lane's side stream with scratch buffers of its own (flat sets of calls with a handful of statements only)
    // digits_ready: the windowed part's digits are already in the lane's digit buffer (fuse_z_digits): no recoding pass
    void run_msm_g1(Lane& ln, const MsmSet<G1Aff>& set, const fe* scalars, int mont, size_t B, G1Xyzz* sum, bool timed = false, bool side = false, bool digits_ready = false);
    void run_msm_g2(Lane& ln, const MsmSet<G2Aff>& set, const fe* scalars, int mont, size_t B, G2Xyzz* sum, bool side = false);
    void flush_horner_g1(Lane& ln, size_t B, hipStream_t s) { launch_msm_horner_g1(ln.pending1, B, s); ln.pending1.n = 0; }
    void flush_horner_g2(Lane& ln, size_t B, hipStream_t s) { launch_msm_horner_g2(ln.pending2, B, s); ln.pending2.n = 0; }

    void fetch_column(Lane& ln, const fe* mat, size_t rows, size_t B, size_t col, std::vector<uint8_t>& out);
    // The statement's secrets must not outlive the call in device memory (the witness of these circuits IS a cipher key): enqueued behind a
    // chunk's last kernel, clears the key wires (32-byte rows and byte plane), the rows of r, s, -rs, the raw input records, the prover
    // randomness, its endomorphism split and the commitment mask.  (The reference leaves all of this to Go's garbage collector.)
    void wipe_secrets(Lane& ln, size_t B, bool small_call);
    // TEST HOOK: non-zero bytes left in those areas over all lanes (after draining their streams)
    size_t secret_residue();

    void prove_chunk(Lane& ln, const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* dbg, bool allow_few_solver = true, bool allow_small = true);

    // gnark proof.WriteTo: Ar | Bs | Krs compressed, u32be nbCommitments, commitments, CommitmentPok
    void serialize(const uint8_t* o, uint8_t flags, uint32_t status, const uint8_t* commitment_xy, const uint8_t* pok_xy, ProofResult& res) const;
};

}  // namespace gsc
