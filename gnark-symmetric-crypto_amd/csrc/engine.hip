// GPU prover engine: InitAlgorithm-time upload / table build and the per-batch device pipeline.
// See engine.hpp for the reference interface this mirrors.
#include "engine.hpp"
#include "dispatch.hpp"
#include "formats.hpp"
#include "kernels.hpp"
#include "host_ciphers.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <stdexcept>

namespace gsc {

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); } while (0)

namespace {

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

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }

// Montgomery images of 0, 1, 2, -1, -2 in Fr: gnark puts these at coefficient ids 0..4 of every R1CS
// (SURVEY.md App. A); the solver kernel short-cuts them to additions.
const uint32_t kSmallCoeffs[5][8] = {
    {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u},
    {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u},
    {0x9ffffff6u, 0x592c6838u, 0x3ec19a53u, 0x6df8ed2bu, 0xf0f28c5cu, 0xccdd46deu, 0x340fbe5eu, 0x1c14ef83u},
    {0xa0000006u, 0x974bc177u, 0xda58a367u, 0xf13771b2u, 0x0908122eu, 0x51e1a247u, 0x4729c0fau, 0x2259d6b1u},
    {0x5000000bu, 0xeab58d5bu, 0x3af7d63du, 0xba3afb1du, 0x908ecc00u, 0xeb72fed7u, 0xad21e1cau, 0x144f5eefu},
};
// (p-1)/2, big-endian: a compressed point carries the "larger y" flag iff y > (p-1)/2
const uint8_t kHalfP[32] = {0x18, 0x32, 0x27, 0x39, 0x70, 0x98, 0xd0, 0x14, 0xdc, 0x28, 0x22, 0xdb, 0x40, 0xc0, 0xac, 0x2e,
                            0xcb, 0xc0, 0xb5, 0x48, 0xb4, 0x38, 0xe5, 0x46, 0x9e, 0x10, 0x46, 0x0b, 0x6c, 0x3e, 0x7e, 0xa3};

void le_limbs_to_be(const uint8_t* le, uint8_t* be) { for (int i = 0; i < 32; i++) be[i] = le[31 - i]; }
bool be_greater(const uint8_t* a, const uint8_t* b) { int c = memcmp(a, b, 32); return c > 0; }
bool be_is_zero(const uint8_t* a) { for (int i = 0; i < 32; i++) if (a[i]) return false; return true; }

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

}  // namespace

bool test_hooks_enabled() {
    static const bool on = [] { const char* e = getenv("GSC_ENABLE_TEST_HOOKS"); return e && e[0] == '1' && e[1] == 0; }();
    return on;
}
namespace { const bool g_hooks_read_at_load = test_hooks_enabled(); }      // forces the evaluation when the library is loaded

EngineConfig config_from_env() {
    EngineConfig c;
    c.device = env_int("GSC_DEVICE", 0);
    if (const char* dv = getenv("GSC_DEVICES")) {      // "0,1,2,3": one engine replica per listed device, batches split over them
        for (const char* q = dv; *q;) { while (*q == ',' || *q == ' ') q++; if (!*q) break; char* end = nullptr; const long v = strtol(q, &end, 10); if (end == q) throw std::runtime_error("GSC_DEVICES: expected a comma-separated list of device ordinals"); c.devices.push_back((int)v); q = end; }
    }
    c.max_batch = (size_t)env_int("GSC_MAX_BATCH", 1024);
    c.lanes = env_int("GSC_LANES", 0);
    c.small_lanes = env_int("GSC_SMALL_LANES", -1);
    if (c.small_lanes > 8) throw std::runtime_error("GSC_SMALL_LANES must be at most 8");
    c.min_split = (size_t)env_int("GSC_MIN_SPLIT", 256);
    c.bit_groups = env_int("GSC_BIT_GROUPS", 1);
    c.window_z = env_int("GSC_WINDOW_Z", 0);
    c.window_w = env_int("GSC_WINDOW_W", 0);
    c.z_table_gb = env_int("GSC_Z_TABLE_GB", 48);
    c.w_table_gb = env_int("GSC_W_TABLE_GB", 16);
    c.row_margin_bits = env_int("GSC_ROW_MARGIN_BITS", 1);
    c.few_path = env_int("GSC_FEW_PATH", 1);
    c.few_solver = env_int("GSC_FEW_SOLVER", 1);
    c.few_max = env_int("GSC_FEW_MAX", 0);
    if (c.few_max < 0 || c.few_max > (int)MSM_FEW_PROOFS) throw std::runtime_error("GSC_FEW_MAX must be in [0, 32]");
    c.few_workgroups = env_int("GSC_FEW_WGS", 0);
    c.few_z_gb = env_int("GSC_FEW_Z_GB", 12);
    c.few_wide = env_int("GSC_FEW_WIDE", 1);
    if (c.few_workgroups < 0 || c.few_workgroups > 256) throw std::runtime_error("GSC_FEW_WGS must be in [0, 256]");
    c.trace_host = getenv("GSC_TRACE_HOST") != nullptr;
    if (test_hooks_enabled()) { c.solver_trace = getenv("GSC_SOLVER_TRACE") != nullptr; c.few_test_abort = getenv("GSC_FEW_TEST_ABORT") != nullptr; }
    if (c.max_batch < 64) c.max_batch = 64;
    c.max_batch = (c.max_batch + 63) / 64 * 64;
    if ((c.window_z && (c.window_z < 4 || c.window_z > 16)) || (c.window_w && (c.window_w < 4 || c.window_w > 16))) throw std::runtime_error("GSC_WINDOW_* must be in [4,16]");
    return c;
}

// The resident solver kernel (k_solver_few) spins at device-wide barriers, so two of them must never share the device: each could
// hold CUs the other's missing workgroups are waiting for.  Launches are therefore chained on the device: a launch first makes its
// stream wait for the previous one's completion event (no host blocking).  Other processes on the same device are not covered —
// there the kernel's bounded polling gives up and the call is solved again with one launch per level (prove_chunk).
struct FewSolverChain { std::mutex m; hipEvent_t last = nullptr; };
static FewSolverChain& few_solver_chain(int device) { static FewSolverChain* chains = new FewSolverChain[64]; return chains[device & 63]; }      // never destroyed: lanes may outlive static destructors

// The quotient transforms and the MSMs of a big batch fill the chip on their own (VALU-bound); two of them side by side only thrash each
// other's table gathers (measured in round 2: two lanes no faster than one).  What does overlap is the witness stage — bound by HBM
// traffic and dependent levels, not by VALU issue — with ANOTHER batch's transforms and MSMs.  So the heavy phases of big batches are
// chained per device, in enqueue order, with events (no host blocking), across lanes and algorithms: while one lane computes, the
// other lane's next batch is solved.
struct HeavyChain { std::mutex m; hipEvent_t last = nullptr; };
static HeavyChain& heavy_chain(int device) { static HeavyChain* chains = new HeavyChain[64]; return chains[device & 63]; }
constexpr size_t HEAVY_MIN_BATCH = 1024;      // smaller batches leave the chip under-filled in every stage: they run freely side by side

class AlgorithmImpl {
  public:
    Cipher cipher; EngineConfig cfg;
    size_t n_wires = 0, n_public = 0, n_constraints = 0, domain_n = 0; int L = 0;
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
    // NTT
    DevBuf<int32_t> tw_fwd, tw_inv, qr; DevBuf<fe> scale_mid, scale_out, dom;   // dom: omega, omega_inv, g, g_inv, n_inv, 16/n
    // MSM sets
    MsmSet<G1Aff> mA, mB1, mK, mZ, mZfew, mPed, mPedSigma; MsmSet<G2Aff> mB2;     // mPed*: Pedersen commitment bases (AES-V2)
    // batch buffers: one set per lane.  A lane = a HIP stream with its own witness / polynomial / partial-sum buffers; with more than
    // one lane big batches are cut into chunks that the lanes prove concurrently.  Measured on MI355X: two lanes do
    // NOT beat one lane with the same number of proofs in flight (the MSM kernels already fill the chip and two of them thrash
    // each other's table gathers), so the default is one lane; the option stays for hosts that prefer lower per-call latency.
    struct Lane {
        hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;      // side: the assembly's scalar multiplications, beside the MSMs; side2: the B2 sum of a latency-path call
        hipEvent_t ev_ab = nullptr, ev_fs = nullptr, ev_b2 = nullptr, ev_s2 = nullptr;
        hipEvent_t ev_few = nullptr;    // completion of this lane's latest k_solver_few launch (FewSolverChain)
        hipEvent_t ev_heavy = nullptr;  // completion of this lane's latest transforms + MSMs (HeavyChain)
        hipEvent_t ev[7] = {};          // 0..4 stage boundaries, 5..6 bracket the dominant kernel (Z-table MSM gather-accumulate)
        float stage_ms[4] = {0, 0, 0, 0}; float msm_z_kernel_ms = 0; size_t last_batch = 0;
        size_t n_real = 0;              // statements of the chunk being proved (the batch is padded to a multiple of 64)
        size_t cap = 0;
        DevBuf<uint8_t> d_inputs, d_rs, d_out, d_flags, d_mask_in, d_cpts; DevBuf<uint32_t> d_status, d_fsync; DevBuf<GlvSplit> d_glv;
        DevBuf<fe> d_mask, d_commit; DevBuf<G1Xyzz> d_sumD, d_sumPok;
        DevBuf<fe> d_W, d_A, d_B, d_C;
        DevBuf<G1Xyzz> d_part1a, d_part1b, d_sumA, d_sumB1, d_sumK, d_sumZ, d_tmp; DevBuf<G2Xyzz> d_part2a, d_part2b, d_sumB2;
        DevBuf<uint4> d_digits_s2; DevBuf<uint8_t> d_gok_s2;                                        // side2's digits (its partial sums are the G2 buffers, which nothing else uses)
        DevBuf<uint4> d_digits_s; DevBuf<uint8_t> d_gok_s; DevBuf<G1Xyzz> d_part1c, d_part1d;      // the side stream's MSM scratch (A and B1 of a latency-path call)
        DevBuf<uint4> d_digits;                                                   // signed digits [window][octet][proof]
        // per-window sums [window][proof] and flat-part sums [proof], one pair per set: the Horner passes of several sets are deferred
        // and run as one launch (MsmHornerJobs), so their inputs must not share storage
        static constexpr int NSETS = 7;      // A, B1, K, Z, Ped, PedSigma, Z (latency layout)
        DevBuf<G1Xyzz> d_sj1[NSETS], d_flat1[NSETS]; DevBuf<G2Xyzz> d_sj2, d_flat2;
        MsmHornerJobs pending1{}, pending2{};
        DevBuf<uint8_t> d_gok;                                                    // bit-group verdicts [group][wave of 64 proofs]
        ~Lane() { if (ev_heavy) { for (int d = 0; d < 64; d++) { HeavyChain& c = heavy_chain(d); std::lock_guard<std::mutex> lk(c.m); if (c.last == ev_heavy) c.last = nullptr; } (void)hipEventDestroy(ev_heavy); }
                  if (ev_few) { for (int d = 0; d < 64; d++) { FewSolverChain& c = few_solver_chain(d); std::lock_guard<std::mutex> lk(c.m); if (c.last == ev_few) c.last = nullptr; } (void)hipEventDestroy(ev_few); }
                  for (auto& e : ev) if (e) (void)hipEventDestroy(e); if (ev_ab) (void)hipEventDestroy(ev_ab); if (ev_fs) (void)hipEventDestroy(ev_fs); if (ev_b2) (void)hipEventDestroy(ev_b2); if (ev_s2) (void)hipEventDestroy(ev_s2); if (side2) (void)hipStreamDestroy(side2); if (side) (void)hipStreamDestroy(side); if (stream) (void)hipStreamDestroy(stream); }
    };
    std::vector<std::unique_ptr<Lane>> lanes;
    size_t cap = 0;                     // proofs per full lane = the largest chunk
    size_t full_lanes = 0;              // lanes [0, full_lanes) hold `cap` proofs; the rest are small lanes (SMALL_LANE_CAP)
    static constexpr size_t SMALL_LANE_CAP = 512;

    AlgorithmImpl(Cipher c, const uint8_t* pk, size_t pk_len, const uint8_t* r1cs, size_t r1cs_len, const EngineConfig& cf) : cipher(c), cfg(cf) {
        // measured crossover with the batch kernels (one 64-column batch: 12.9 ms ChaCha20, 43.7 ms AES): 32 statements for ChaCha20 (10.2 ms), ~23 for AES (8.2 ms + 1.6 ms each: 38.4 ms for 20)
        if (!cfg.few_max) cfg.few_max = cipher == CHACHA20 ? 32 : 20;
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
        init_program(cs);
        calibrate();
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
        const size_t small_cap = lane_cap > SMALL_LANE_CAP ? SMALL_LANE_CAP : lane_cap;
        for (int i = 0; i < cfg.small_lanes; i++) { lanes.emplace_back(new Lane); alloc_lane(*lanes.back(), small_cap); }
        lane_busy.assign(lanes.size(), 0);
        cap = lane_cap;
        if (trace) fprintf(stderr, "InitAlgorithm(%d): %zu lane(s) of %zu proofs + %d of %zu, %.0f ms\n", (int)c, nl, lane_cap, cfg.small_lanes, small_cap, ms(t3, now()));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    ~AlgorithmImpl() { lanes.clear(); if (stream) (void)hipStreamDestroy(stream); }

    void init_program(const R1csFile& cs) {
        n_wires = cs.n_wires(); n_public = cs.n_public; n_constraints = cs.n_constraints; has_commitment = cs.has_commitment;
        const size_t expect_in = cipher == CHACHA20 ? 1408 : cipher == AES_128 ? 157 : 173;
        if (cs.n_public - 1 + cs.n_secret != expect_in) throw std::runtime_error("r1cs: witness size does not match the cipher's circuit");
        if (cs.n_coeff() < 5 || memcmp(cs.coeff_limbs.data(), kSmallCoeffs, sizeof kSmallCoeffs)) throw std::runtime_error("r1cs: coefficient ids 0..4 are not 0,1,2,-1,-2");
        SolverProgram sp = build_solver_program(cs);
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
    }

    static void pack_inputs(const ProofRequest* reqs, size_t n, size_t B, std::vector<uint8_t>& h_in, std::vector<uint8_t>& h_rs) {
        h_in.assign(176 * B, 0); h_rs.assign(64 * B, 0);
        for (size_t i = 0; i < B; i++) {
            const ProofRequest& q = reqs[i < n ? i : n - 1];
            uint8_t* rec = h_in.data() + 176 * i;
            memcpy(rec, q.key, q.keylen);
            memcpy(rec + 32, q.nonce, 12);
            rec[44] = (uint8_t)q.counter; rec[45] = (uint8_t)(q.counter >> 8); rec[46] = (uint8_t)(q.counter >> 16); rec[47] = (uint8_t)(q.counter >> 24);
            memcpy(rec + 48, q.plaintext, 64); memcpy(rec + 112, q.ciphertext, 64);
            memcpy(h_rs.data() + 64 * i, q.r, 32); memcpy(h_rs.data() + 64 * i + 32, q.s, 32);
        }
    }

    // Which wires are bits?  Nothing in an R1CS says so, but it is a property of the circuit, not of the statement: solve 64
    // pseudo-random statements once and call a wire a bit when it is 0 or 1 in all of them.  This is only a PREDICTION used to
    // lay out the wire MSMs (bit wires first, in groups of eight with subset-sum tables); k_msm re-checks every group for
    // every wave of proofs and falls back to the digit tables, so a wrong prediction costs time, never correctness.
    void calibrate() {
        row_class.assign(n_wires + 4, 255);
        row_class[n_wires] = row_class[n_wires + 1] = row_class[n_wires + 2] = 254;     // r, s, -rs: uniform scalars
        if (cfg.bit_groups <= 0) return;
        if (cfg.bit_groups >= 2) { std::fill(row_class.begin(), row_class.begin() + n_wires, 0); return; }
        const size_t B = 64;
        std::vector<ProofRequest> reqs(B);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        for (auto& q : reqs) {
            memset(&q, 0, sizeof q);
            q.keylen = cipher == AES_128 ? 16 : 32;
            for (uint32_t i = 0; i < q.keylen; i++) q.key[i] = (uint8_t)next();
            for (auto& b : q.nonce) b = (uint8_t)next();
            for (auto& b : q.plaintext) b = (uint8_t)next();
            q.counter = (uint32_t)(next() & 0xFFFF);
            if (cipher == CHACHA20) chacha20_xor_stream(q.key, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
            else aes_ctr_xor_stream(q.key, q.keylen, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
            q.r[0] = 3; q.s[0] = 5; q.mask[0] = 7;
        }
        std::vector<uint8_t> h_in, h_rs; pack_inputs(reqs.data(), B, B, h_in, h_rs);
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
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    // decompress `raw` (count points of `sz` bytes) into out[offset...]; returns per-point status
    std::vector<uint8_t> decompress_g1(const std::vector<uint8_t>& raw, G1Aff* out) {
        const size_t n = raw.size() / 32; std::vector<uint8_t> st(n);
        if (!n) return st;
        DevBuf<uint8_t> d_raw(raw.size()), d_st(n);
        d_raw.upload(raw.data(), raw.size(), stream);
        launch_decompress_g1(d_raw.p, out, d_st.p, n, stream);
        HIP_CHECK(hipMemcpyAsync(st.data(), d_st.p, n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return st;
    }
    std::vector<uint8_t> decompress_g2(const std::vector<uint8_t>& raw, G2Aff* out) {
        const size_t n = raw.size() / 64; std::vector<uint8_t> st(n);
        if (!n) return st;
        DevBuf<uint8_t> d_raw(raw.size()), d_st(n);
        d_raw.upload(raw.data(), raw.size(), stream);
        launch_decompress_g2(d_raw.p, out, d_st.p, n, stream);
        HIP_CHECK(hipMemcpyAsync(st.data(), d_st.p, n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return st;
    }

    static constexpr size_t EXPAND_MAX = 64;      // up to this many wide wires of a set are laid out as window octets of its flat part
    static constexpr int EXPAND_C = 15, NARROW_MAX_BITS = 14;

    // rows of multiples for `n` bases (row i: len[i] entries at off[i]); work is cut into segments of at most 256 multiples
    template <class AffT, class XyzzT>
    void build_rows(const AffT* bases, size_t n, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len, AffT* table) {
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
    void launch_build_rows(const G1Aff* b, const MsmRowSeg* sg, size_t n, uint32_t cap, G1Aff* t, G1Xyzz* sc) { launch_build_rows_g1(b, sg, n, cap, t, sc, stream); }
    void launch_build_rows(const G2Aff* b, const MsmRowSeg* sg, size_t n, uint32_t cap, G2Aff* t, G2Xyzz* sc) { launch_build_rows_g2(b, sg, n, cap, t, sc, stream); }
    void launch_shift(const G1Aff* in, const uint32_t* src, const uint32_t* sh, size_t n, G1Aff* out) { launch_shift_bases_g1(in, src, sh, n, out, stream); }
    void launch_shift(const G2Aff* in, const uint32_t* src, const uint32_t* sh, size_t n, G2Aff* out) { launch_shift_bases_g2(in, src, sh, n, out, stream); }

    // Lays out one MSM set and builds its tables.  uniform = true (Z): every base gets a full row, windowed kernel.  Otherwise the
    // bases are sorted by what calibrate() saw on their wires: values in {-1, 0, 1} -> bit groups of eight; values of up to
    // NARROW_MAX_BITS bits (with the margin) -> flat rows of that length; the rest (r, s, the lookup argument's products and inverses)
    // are wide: a few of them become window octets of the flat part, many get the windowed kernel and a Horner pass.
    template <class AffT, class XyzzT, class Decomp>
    void build_set(MsmSet<AffT>& set, const std::vector<uint8_t>& raw, size_t point_bytes, const std::vector<uint32_t>& rows, int c, const char* what, Decomp decomp, bool uniform, int expand_cv = 0) {
        const size_t n = raw.size() / point_bytes;
        if (rows.size() != n) throw std::runtime_error(std::string("pk: row map size mismatch for ") + what);
        set.nbases = n; set.c = c; set.nwin = msm_windows(c);
        DevBuf<AffT> bases(n ? n : 1);
        const std::vector<uint8_t> st = decomp(raw, bases.p);
        for (size_t i = 0; i < n; i++) if (st[i] == 1) throw std::runtime_error(std::string("pk: invalid point in ") + what);
        const size_t D = (size_t)1 << (c - 1);
        const uint32_t ROW_ZERO = (uint32_t)(n_wires + 3);
        std::vector<uint32_t> bits, narrow, wide;            // indices into the key's order; the point at infinity contributes nothing: dropped
        std::vector<uint32_t> narrow_len;
        for (size_t i = 0; i < n; i++) {
            if (st[i] == 2) continue;
            const int k = uniform || rows[i] >= row_class.size() ? 255 : row_class[rows[i]];
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
            if (!uniform && cfg.few_path && cfg.few_wide) {      // (the quotient bases have their own budgeted layout: init_key)
                std::vector<uint8_t> raw_w(set.nwide * point_bytes); std::vector<uint32_t> rows_w(set.nwide);
                for (size_t i = 0; i < set.nwide; i++) { memcpy(raw_w.data() + i * point_bytes, raw.data() + (size_t)wide[i] * point_bytes, point_bytes); rows_w[i] = rows[wide[i]]; }
                set.few_wide.reset(new MsmSet<AffT>());
                build_set<AffT, XyzzT>(*set.few_wide, raw_w, point_bytes, rows_w, c, what, decomp, true, 8);
            }
        }
    }
    // group tables are built in chunks so that the projective scratch stays below ~2 GiB
    void build_subset(const G1Aff* b, size_t ng, G1Aff* t, uint8_t* ok) {
        size_t chunk = ((size_t)2 << 30) / (MSM_GROUP_ENTRIES * sizeof(G1Xyzz)); if (chunk > ng) chunk = ng;
        DevBuf<G1Xyzz> sc(chunk * MSM_GROUP_ENTRIES);
        for (size_t g0 = 0; g0 < ng; g0 += chunk) launch_build_subset_g1(b + 8 * g0, ng - g0 < chunk ? ng - g0 : chunk, t + g0 * MSM_GROUP_ENTRIES, sc.p, ok + g0, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    void build_subset(const G2Aff* b, size_t ng, G2Aff* t, uint8_t* ok) {
        size_t chunk = ((size_t)2 << 30) / (MSM_GROUP_ENTRIES * sizeof(G2Xyzz)); if (chunk > ng) chunk = ng;
        DevBuf<G2Xyzz> sc(chunk * MSM_GROUP_ENTRIES);
        for (size_t g0 = 0; g0 < ng; g0 += chunk) launch_build_subset_g2(b + 8 * g0, ng - g0 < chunk ? ng - g0 : chunk, t + g0 * MSM_GROUP_ENTRIES, sc.p, ok + g0, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void init_key(const R1csFile& cs, const PkFile& key) {
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
            tw_fwd.alloc(domain_n / 2 * 12); tw_inv.alloc(domain_n / 2 * 12); scale_mid.alloc(domain_n); scale_out.alloc(domain_n); qr.alloc((2 * NTT_QMAX + 1) * 12);
            DevBuf<uint32_t> d_flag(1); uint32_t flag = 0;
            HIP_CHECK(hipMemsetAsync(d_flag.p, 0, 4, stream));
            launch_ntt_constants(dom.p, dom.p + 1, dom.p + 4, L, tw_fwd.p, tw_inv.p, scale_mid.p, scale_out.p, dom.p + 5, qr.p, d_flag.p, stream);
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
        std::vector<uint32_t> rowsZ(domain_n - 1); for (size_t i = 0; i < rowsZ.size(); i++) rowsZ[i] = (uint32_t)i;
        // Digit widths: explicit (GSC_WINDOW_Z / GSC_WINDOW_W) or the largest that keeps the tables inside the per-algorithm HBM
        // budget (the defaults leave room for all three algorithms of the reference on one 288 GB device: 3 x (48 + 16) GB).
        // Z: uniform rows of 2^(c-1) entries of 64 B: c = 16 is 69 GB for ChaCha20-V3 (2^15 - 1 bases), c = 14 is 69 GB for AES-V2 (2^17 - 1)
        if (!cfg.window_z) { cfg.window_z = 4; for (int c = 16; c >= 4; c--) if ((double)rowsZ.size() * (double)((size_t)1 << (c - 1)) * 64.0 <= cfg.z_table_gb * 1e9) { cfg.window_z = c; break; } }
        if (!cfg.window_w) {      // wire sets: only the wide wires that get the windowed kernel (more than EXPAND_MAX per set) pay for c
            auto wide_of = [&](const std::vector<uint32_t>& rows) {
                size_t k = 0;
                for (uint32_t r : rows) { const int cl = r < row_class.size() ? row_class[r] : 255; if (cfg.bit_groups <= 0 || cl == 255 || cl + cfg.row_margin_bits > NARROW_MAX_BITS) k++; }
                return k > EXPAND_MAX ? (double)k : 0.0;
            };
            const double g1 = wide_of(rowsA) + wide_of(rowsB) + wide_of(rowsK) + 2 * wide_of(cs.commit_private), g2 = wide_of(rowsB2);
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
        timed("G1.Z", [&] { build_set<G1Aff, G1Xyzz>(mZ, key.g1_Z, 32, rowsZ, cfg.window_z, "G1.Z", dec1, true); });      // uniform full-width scalars
        if (cfg.few_path && cfg.few_z_gb > 0) {
            // calls with a handful of statements: the quotient bases once more as (base, window) pairs with their own rows 2^(cv j) d P — more
            // additions per proof than the wide rows above, but no 254-doubling Horner chain behind them (1.4 ms of a 6 ms Prove)
            const size_t nz = key.g1_Z.size() / 32;
            int cv = 0;
            for (int t : {8, 6, 4}) if ((double)nz * msm_windows(t) * (double)((size_t)1 << (t - 1)) * sizeof(G1Aff) <= (double)cfg.few_z_gb * 1e9) { cv = t; break; }
            if (cv) timed("G1.Z (latency layout)", [&] { build_set<G1Aff, G1Xyzz>(mZfew, key.g1_Z, 32, rowsZ, cfg.window_z, "G1.Z", dec1, true, cv); });
        }
        timed("G2.B", [&] { build_set<G2Aff, G2Xyzz>(mB2, cat(key.g2_B, {&key.g2_beta, &key.g2_delta}), 64, rowsB2, cfg.window_w, "G2.B", dec2, false); });
        if (cs.has_commitment) {
            if (cs.n_public_committed) throw std::runtime_error("r1cs: public committed wires are not supported");
            if (key.ped_basis.size() != cs.commit_private.size() * 32) throw std::runtime_error("pk: commitment basis size does not match the r1cs");
            build_set<G1Aff, G1Xyzz>(mPed, key.ped_basis, 32, cs.commit_private, cfg.window_w, "commitment basis", dec1, false);
            build_set<G1Aff, G1Xyzz>(mPedSigma, key.ped_basis_sigma, 32, cs.commit_private, cfg.window_w, "commitment basis (sigma)", dec1, false);
        }
    }

    void alloc_lane(Lane& ln, size_t B) {
        ln.cap = B;
        HIP_CHECK(hipStreamCreate(&ln.stream)); HIP_CHECK(hipStreamCreate(&ln.side)); HIP_CHECK(hipStreamCreate(&ln.side2));
        for (auto& e : ln.ev) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventCreateWithFlags(&ln.ev_few, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_heavy, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ln.ev_ab, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_fs, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_b2, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ln.ev_s2, hipEventDisableTiming));
        ln.d_inputs.alloc(176 * B); ln.d_rs.alloc(64 * B); ln.d_out.alloc(256 * B); ln.d_flags.alloc((B + 3) / 4 * 4); ln.d_status.alloc(B); ln.d_fsync.alloc(2); ln.d_glv.alloc(2 * MSM_FEW_PROOFS);
        ln.d_W.alloc((n_wires + 4) * B); ln.d_A.alloc(domain_n * B); ln.d_B.alloc(domain_n * B); ln.d_C.alloc(domain_n * B);
        // calls with a handful of statements (k_solver_few) write their own columns only: the others must always hold field elements
        // (zero, later whatever an earlier call left there) because the transforms and MSMs run over whole 64-column batches
        HIP_CHECK(hipMemsetAsync(ln.d_W.p, 0, ln.d_W.n * sizeof(fe), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_A.p, 0, ln.d_A.n * sizeof(fe), ln.stream));
        HIP_CHECK(hipMemsetAsync(ln.d_B.p, 0, ln.d_B.n * sizeof(fe), ln.stream)); HIP_CHECK(hipMemsetAsync(ln.d_C.p, 0, ln.d_C.n * sizeof(fe), ln.stream));
        // partial-sum / digit buffers: the largest need over every batch size this context can be asked for
        size_t p1 = 0, p1b = 0, p2 = 0, p2b = 0, dg = 0, sj2 = 0, gk = 0; size_t sj1[Lane::NSETS] = {0, 0, 0, 0, 0, 0, 0};
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
            if (m.nwide) { const size_t bw = b * (size_t)m.nwin; part(m.nwide, msm_slices(m.nwide, (size_t)m.nwin, WIN_SLICE, b, per), bw); if (bw > sj) sj = bw; const size_t d = (size_t)m.nwin * ((m.nwide + 7) / 8) * b; if (d > dg) dg = d; }
        };
        MsmSet<G1Aff>* g1sets[Lane::NSETS] = {&mA, &mB1, &mK, &mZ, &mPed, &mPedSigma, &mZfew};
        for (size_t b = 64; b <= B; b += 64) {
            for (int k = 0; k < Lane::NSETS; k++) if (g1sets[k] != &mZfew || b == 64) need(*g1sets[k], b, p1, p1b, sj1[k]);      // the latency layout only serves 64-column batches
            need(mB2, b, p2, p2b, sj2);
        }
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
        ln.d_sumA.alloc(B); ln.d_sumB1.alloc(B); ln.d_sumK.alloc(B); ln.d_sumZ.alloc(B); ln.d_sumB2.alloc(B); ln.d_tmp.alloc(2 * B);
        if (has_commitment) { ln.d_mask_in.alloc(32 * B); ln.d_mask.alloc(B); ln.d_commit.alloc(B); ln.d_cpts.alloc(128 * B); ln.d_sumD.alloc(B); ln.d_sumPok.alloc(B); }
    }

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
    static constexpr size_t WIN_SLICE = 256;      // bases per slice of the windowed kernel at full batches (measured 64 .. 512: kernel time within 1 %)
    bool few_solver_wanted(size_t n, size_t B) const { return n <= (size_t)cfg.few_max && B == 64 && cfg.few_solver; }
    struct MsmCtx { hipStream_t stream; uint4* digits; uint8_t* gok; };
    template <class XyzzT, class LR>
    void reduce_slices(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, XyzzT* out, LR launch_reduce) {
        XyzzT* src = pa; XyzzT* alt = pb; size_t ns = nslices;
        for (;;) {
            const size_t groups = msm_reduce_groups(ns, cols);
            XyzzT* dst = groups == 1 ? out : alt;
            launch_reduce(src, ns, cols, dst, st);
            if (groups == 1) break;
            XyzzT* t = src; src = dst; alt = t; ns = groups;
        }
    }
    // the same for the first `npr` columns of every row of `stride` (latency path: nobody reads the padding proofs' columns)
    template <class XyzzT, class LRF>
    void reduce_slices_few(hipStream_t st, XyzzT* pa, XyzzT* pb, size_t nslices, size_t cols, size_t stride, size_t npr, XyzzT* out, LRF launch_reduce_few) {
        XyzzT* src = pa; XyzzT* alt = pb; size_t ns = nslices;
        for (;;) {
            const size_t groups = (ns + 63) / 64;
            XyzzT* dst = groups == 1 ? out : alt;
            launch_reduce_few(src, ns, cols, stride, npr, dst, st);
            if (groups == 1) break;
            XyzzT* t = src; src = dst; alt = t; ns = groups;
        }
    }
    // scalars: the wire matrix W (Montgomery; wire sets) or h (canonical; Z)
    // The Horner pass of the windowed part is NOT launched here: it is queued in `pending` and flushed together with those of other
    // sets (flush_horner_*), because each is a serial chain of 254 doublings whose duration does not depend on the batch.
    template <class AffT, class XyzzT, class LF, class LFF, class LW, class LWF, class LR, class LRF>
    void run_msm(Lane& ln, const MsmCtx& ctx, const MsmSet<AffT>& set, const fe* scalars, bool wires, size_t B, size_t n_real, XyzzT* pa, XyzzT* pb, XyzzT* sj, XyzzT* flat, XyzzT* sum, bool timed,
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
                MsmFlatRecodeArgs ra{scalars, m.frows.p, m.octwin.p, m.nflat, B, m.cv, ctx.digits, m.nbit, m.group_ok.p, ctx.gok, wires ? 1 : 0};
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
            MsmFlatRecodeArgs ra{scalars, set.frows.p, set.octwin.p, set.nflat, B, set.cv, ctx.digits, set.nbit, set.group_ok.p, ctx.gok, wires ? 1 : 0};
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
            launch_msm_recode(ra, ctx.stream);
            MsmWinArgs a{set.wtable.p, set.c, set.nwin, set.nwide, ctx.digits, B, nslices, per, pa};
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
    int set_index(const MsmSet<G1Aff>& set) const { const MsmSet<G1Aff>* all[Lane::NSETS] = {&mA, &mB1, &mK, &mZ, &mPed, &mPedSigma, &mZfew}; for (int k = 0; k < Lane::NSETS; k++) if (all[k] == &set) return k; return 0; }
    // side = true: on the lane's side stream with scratch buffers of its own (flat sets of calls with a handful of statements only)
    void run_msm_g1(Lane& ln, const MsmSet<G1Aff>& set, const fe* scalars, int mont, size_t B, G1Xyzz* sum, bool timed = false, bool side = false) {
        const int k = set_index(set);
        if (side) {
            if (!set.latency_flat() || B != 64) throw std::runtime_error("internal: side-stream MSM on a set with a windowed part");
            run_msm(ln, MsmCtx{ln.side, ln.d_digits_s.p, ln.d_gok_s.p}, set, scalars, mont != 0, B, ln.n_real, ln.d_part1c.p, ln.d_part1d.p, ln.d_sj1[k].p, ln.d_flat1[k].p, sum, false, ln.pending1,
                    launch_msm_flat_g1, launch_msm_flat_few_g1, launch_msm_win_g1, launch_msm_win_few_g1, launch_msm_reduce_g1, launch_msm_reduce_few_g1);
            return;
        }
        run_msm(ln, MsmCtx{ln.stream, ln.d_digits.p, ln.d_gok.p}, set, scalars, mont != 0, B, ln.n_real, ln.d_part1a.p, ln.d_part1b.p, ln.d_sj1[k].p, ln.d_flat1[k].p, sum, timed, ln.pending1, launch_msm_flat_g1, launch_msm_flat_few_g1, launch_msm_win_g1, launch_msm_win_few_g1, launch_msm_reduce_g1, launch_msm_reduce_few_g1);
    }
    void run_msm_g2(Lane& ln, const MsmSet<G2Aff>& set, const fe* scalars, int mont, size_t B, G2Xyzz* sum, bool side = false) {
        if (side && (!set.latency_flat() || B != 64)) throw std::runtime_error("internal: side-stream MSM on a set with a windowed part");
        run_msm(ln, side ? MsmCtx{ln.side2, ln.d_digits_s2.p, ln.d_gok_s2.p} : MsmCtx{ln.stream, ln.d_digits.p, ln.d_gok.p}, set, scalars, mont != 0, B, ln.n_real, ln.d_part2a.p, ln.d_part2b.p, ln.d_sj2.p, ln.d_flat2.p, sum, false, ln.pending2, launch_msm_flat_g2, launch_msm_flat_few_g2, launch_msm_win_g2, launch_msm_win_few_g2, launch_msm_reduce_g2, launch_msm_reduce_few_g2);
    }
    void flush_horner_g1(Lane& ln, size_t B, hipStream_t s) { launch_msm_horner_g1(ln.pending1, B, s); ln.pending1.n = 0; }
    void flush_horner_g2(Lane& ln, size_t B, hipStream_t s) { launch_msm_horner_g2(ln.pending2, B, s); ln.pending2.n = 0; }

    void fetch_column(Lane& ln, const fe* mat, size_t rows, size_t B, size_t col, std::vector<uint8_t>& out) {
        out.resize(rows * 32);
        HIP_CHECK(hipMemcpy2DAsync(out.data(), 32, reinterpret_cast<const uint8_t*>(mat) + 32 * col, B * 32, 32, rows, hipMemcpyDeviceToHost, ln.stream));
        HIP_CHECK(hipStreamSynchronize(ln.stream));
    }

    void prove_chunk(Lane& ln, const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* dbg, bool allow_few_solver = true) {
        const size_t B = (n + 63) / 64 * 64;
        ln.n_real = n;
        const bool trace = cfg.trace_host;
        const auto tc0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> h_in, h_rs; pack_inputs(reqs, n, B, h_in, h_rs);
        ln.d_inputs.upload(h_in.data(), h_in.size(), ln.stream);
        ln.d_rs.upload(h_rs.data(), h_rs.size(), ln.stream);
        std::vector<GlvSplit> h_glv;                                     // (lives as long as the other staging vectors of the call)
        if (n <= (size_t)cfg.few_max && cfg.few_path && B == 64) {      // latency path: the two halves of s and r for k_fin_scalarmul_few
            h_glv.resize(2 * n);
            for (size_t i = 0; i < n; i++) for (int role = 0; role < 2; role++) {
                uint32_t w[8]; memcpy(w, h_rs.data() + 64 * i + (role == 0 ? 32 : 0), 32);
                if (!glv_split(w, h_glv[2 * i + role])) throw std::runtime_error("internal: scalar split out of range");
            }
            ln.d_glv.upload(h_glv.data(), h_glv.size(), ln.stream);
        }
        HIP_CHECK(hipMemsetAsync(ln.d_flags.p, 0, ln.d_flags.bytes(), ln.stream));
        HIP_CHECK(hipEventRecord(ln.ev[0], ln.stream));
        // 1. witness
        if (cipher == CHACHA20) launch_assign_chacha(ln.d_inputs.p, ln.d_W.p, B, ln.stream);
        else launch_assign_aes(ln.d_inputs.p, cipher == AES_128 ? 16 : 32, ln.d_W.p, B, ln.stream);
        if (has_commitment) {
            std::vector<uint8_t> h_mask(32 * B);
            for (size_t i = 0; i < B; i++) memcpy(h_mask.data() + 32 * i, reqs[i < n ? i : n - 1].mask, 32);
            ln.d_mask_in.upload(h_mask.data(), h_mask.size(), ln.stream);
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
        bool few_solver = few_solver_wanted(n, B) && allow_few_solver;
        if (few_solver) {      // a recent give-up on this replica: skip the resident kernel for a while (see few_skip)
            uint32_t k = few_skip.load();
            while (k && !few_skip.compare_exchange_weak(k, k - 1)) {}
            if (k) few_solver = false;
        }
        const bool latency_call = n <= (size_t)cfg.few_max && cfg.few_path && B == 64;      // the call takes the latency kernels
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
        std::vector<uint8_t> h_cpts;
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
            h_cpts.resize(128 * B);
            run_levels(commit_level, n_levels);
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
        // 2. quotient polynomial (h overwrites A, canonical, bit-reversed order)
        std::unique_lock<std::mutex> heavy_lock;      // held while the heavy phase is ENQUEUED: the chain's order is the enqueue order
        HeavyChain& hchain = heavy_chain(cfg.device);
        if (B >= HEAVY_MIN_BATCH) {
            heavy_lock = std::unique_lock<std::mutex>(hchain.m);
            if (hchain.last && hchain.last != ln.ev_heavy) HIP_CHECK(hipStreamWaitEvent(ln.stream, hchain.last, 0));
        }
        NttPlan plan{L, tw_fwd.p, tw_inv.p, scale_mid.p, scale_out.p, dom.p + 5, qr.p};
        HIP_CHECK(hipGetLastError());      // witness launches (launch-configuration errors are not sticky: check each group)
        HIP_CHECK(launch_compute_h(plan, ln.d_A.p, ln.d_B.p, ln.d_C.p, n_constraints, B, ln.stream, ln.n_real <= (size_t)cfg.few_max && cfg.few_path ? ln.n_real : 0));      // latency path: the statements' columns only
        HIP_CHECK(hipEventRecord(ln.ev[2], ln.stream));
        if (dbg) fetch_column(ln, ln.d_A.p, domain_n, B, 0, dbg->H);
        // 3. MSMs.  A and B1 first: the two scalar multiplications of the assembly only need those two sums and run on a side stream
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
        run_msm_g1(ln, ln.n_real <= (size_t)cfg.few_max && cfg.few_path && mZfew.nflat ? mZfew : mZ, ln.d_A.p, 0, B, ln.d_sumZ.p, !latency_call);
        if (has_commitment) run_msm_g1(ln, mPedSigma, ln.d_W.p, 1, B, ln.d_sumPok.p);      // proof of knowledge of the commitment: same scalars over sigma * Basis
        flush_horner_g1(ln, B, ln.stream);                                           // K, Z, PedSigma: one launch
        if (has_commitment) launch_points_to_affine_be(ln.d_sumPok.p, B, ln.d_cpts.p + 64 * B, ln.d_flags.p, 16, ln.stream);
        HIP_CHECK(hipGetLastError());      // MSM launches
        HIP_CHECK(hipEventRecord(ln.ev[3], ln.stream));
        if (heavy_lock.owns_lock()) { HIP_CHECK(hipEventRecord(ln.ev_heavy, ln.stream)); hchain.last = ln.ev_heavy; heavy_lock.unlock(); }
        // 4. assembly
        HIP_CHECK(hipStreamWaitEvent(ln.stream, ln.ev_fs, 0));
        if (early_b2) HIP_CHECK(hipStreamWaitEvent(ln.stream, ln.ev_s2, 0));
        launch_fin_combine(ln.d_sumB2.p, ln.d_sumK.p, ln.d_sumZ.p, ln.d_tmp.p, B, ln.d_out.p, ln.d_flags.p, ln.stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(ln.ev[4], ln.stream));
        std::vector<uint8_t> h_out(256 * B), h_flags(ln.d_flags.n); std::vector<uint32_t> h_status(B);
        HIP_CHECK(hipMemcpyAsync(h_out.data(), ln.d_out.p, h_out.size(), hipMemcpyDeviceToHost, ln.stream));
        HIP_CHECK(hipMemcpyAsync(h_flags.data(), ln.d_flags.p, h_flags.size(), hipMemcpyDeviceToHost, ln.stream));
        HIP_CHECK(hipMemcpyAsync(h_status.data(), ln.d_status.p, B * 4, hipMemcpyDeviceToHost, ln.stream));
        if (has_commitment) HIP_CHECK(hipMemcpyAsync(h_cpts.data(), ln.d_cpts.p, 128 * B, hipMemcpyDeviceToHost, ln.stream));      // commitment | its proof of knowledge
        uint32_t h_fsync[2] = {0, 0};
        if (few_solver) HIP_CHECK(hipMemcpyAsync(h_fsync, ln.d_fsync.p, 8, hipMemcpyDeviceToHost, ln.stream));
        const auto tc1 = std::chrono::steady_clock::now();
        HIP_CHECK(hipStreamSynchronize(ln.stream));
        const auto tc2 = std::chrono::steady_clock::now();
        if (h_fsync[1]) {      // the resident solver gave up (its workgroups never became resident together: another process's kernel on this device)
            static std::atomic<bool> warned{false};
            if (!warned.exchange(true)) fprintf(stderr, "libprove: the resident witness kernel could not hold the device (shared with another process?); solving level by level\n");
            const uint32_t pen = few_penalty.load();
            few_skip.store(pen); few_penalty.store(pen < 4096 ? pen * 2 : 4096);
            return prove_chunk(ln, reqs, n, results, dbg, false);
        }
        if (few_solver) few_penalty.store(16);
        for (int k = 0; k < 4; k++) { float ms = 0; (void)hipEventElapsedTime(&ms, ln.ev[k], ln.ev[k + 1]); ln.stage_ms[k] = ms; }
        (void)hipEventElapsedTime(&ln.msm_z_kernel_ms, ln.ev[5], ln.ev[6]); ln.last_batch = B;
        {
            std::lock_guard<std::mutex> lk(stat_mu);
            last_stat.name = latency_call ? (few_solver ? (has_commitment ? "k_solver_few + commitment MSM" : "k_solver_few") : "k_solver (one launch per level)") : "k_msm_win<Fp29f>";
            last_stat.ms = ln.msm_z_kernel_ms; last_stat.statements = n; last_stat.columns = B; last_stat.nbases = mZ.nwide;
            for (int k = 0; k < 4; k++) last_stat.stage_ms[k] = ln.stage_ms[k];
        }
        for (size_t i = 0; i < n; i++)
            serialize(h_out.data() + 256 * i, h_flags[i], h_status[i], has_commitment ? h_cpts.data() + 64 * i : nullptr, has_commitment ? h_cpts.data() + 64 * B + 64 * i : nullptr, results[i]);
        if (trace) {
            const auto tc3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "prove_chunk(%zu): enqueue %.2f ms, wait %.2f ms, serialise %.2f ms\n", n, ms(tc0, tc1), ms(tc1, tc2), ms(tc2, tc3));
        }
    }

    // gnark proof.WriteTo: Ar | Bs | Krs compressed, u32be nbCommitments, commitments, CommitmentPok
    void serialize(const uint8_t* o, uint8_t flags, uint32_t status, const uint8_t* commitment_xy, const uint8_t* pok_xy, ProofResult& res) const {
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
};

void debug_field_ops(int device, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int chain) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available");
    HIP_CHECK(hipSetDevice(device));
    DevBuf<fe> da(n), db(n), dout(n);
    HIP_CHECK(hipMemcpy(da.p, a, 32 * n, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(db.p, b, 32 * n, hipMemcpyHostToDevice));
    launch_field_ops(field, op, da.p, db.p, dout.p, n, chain, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, dout.p, 32 * n, hipMemcpyDeviceToHost));
}

// ---- Algorithm: one replica of the engine per device (GSC_DEVICES), batches split over the replicas ----
Algorithm::Algorithm(Cipher cipher, const uint8_t* pk, size_t pk_len, const uint8_t* r1cs, size_t r1cs_len, const EngineConfig& cfg) {
    std::vector<int> devs = cfg.devices.empty() ? std::vector<int>{cfg.device} : cfg.devices;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available: the GPU prover has no CPU fallback");
    for (int d : devs) if (d < 0 || d >= ndev) throw std::runtime_error("GSC_DEVICES / GSC_DEVICE: device " + std::to_string(d) + " does not exist (" + std::to_string(ndev) + " visible)");
    impls_.resize(devs.size());
    // every replica decodes the key and builds its own tables on its device; the builds run side by side
    std::exception_ptr err; std::mutex err_mu; std::vector<std::thread> th;
    auto make = [&](size_t i) {
        try { EngineConfig c = cfg; c.device = devs[i]; impls_[i].reset(new AlgorithmImpl(cipher, pk, pk_len, r1cs, r1cs_len, c)); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    for (size_t i = 1; i < devs.size(); i++) th.emplace_back(make, i);
    make(0);
    for (auto& t : th) t.join();
    if (err) { impls_.clear(); std::rethrow_exception(err); }
    picker_.reset(new ReplicaPicker(impls_.size()));
}
Algorithm::~Algorithm() = default;
Cipher Algorithm::cipher() const { return impls_[0]->cipher; }
size_t Algorithm::max_batch() const { size_t c = 0; for (auto& i : impls_) c += i->cap; return c; }
size_t Algorithm::devices() const { return impls_.size(); }
size_t Algorithm::lanes() const { return impls_[0]->lanes.size(); }      // full + small: device batches that can be in flight per device
KernelStat Algorithm::last_kernel_stat() const {
    AlgorithmImpl* a = impls_[last_replica_.load() < impls_.size() ? last_replica_.load() : 0].get();
    std::lock_guard<std::mutex> lk(a->stat_mu);
    return a->last_stat;
}
std::string Algorithm::describe() const {
    const AlgorithmImpl* impl_ = impls_[0].get();
    char buf[640];
    snprintf(buf, sizeof buf, "wires=%zu constraints=%zu domain=2^%d max_batch=%zu lanes=%zu small=%zux%zu devices=%zu window_z=%d window_w=%d tables=%.2f GiB bases A=%zu B=%zu K=%zu Z=%zu grouped A=%zu B=%zu K=%zu wide(windowed+expanded) A=%zu+%zu B=%zu+%zu K=%zu+%zu",
             impl_->n_wires, impl_->n_constraints, impl_->L, max_batch(), impl_->full_lanes, impl_->lanes.size() - impl_->full_lanes, impl_->lanes.size() > impl_->full_lanes ? impl_->lanes.back()->cap : (size_t)0, impls_.size(), impl_->cfg.window_z, impl_->cfg.window_w, impl_->table_bytes / 1073741824.0,
             impl_->mA.nbases, impl_->mB1.nbases, impl_->mK.nbases, impl_->mZ.nbases, impl_->mA.nbit, impl_->mB1.nbit, impl_->mK.nbit,
             impl_->mA.nwide, impl_->mA.nexpanded, impl_->mB1.nwide, impl_->mB1.nexpanded, impl_->mK.nwide, impl_->mK.nexpanded);
    // per replica: calls and statements it has served (ReplicaPicker): shows that small calls reach every device
    std::string out = buf;
    out += " served(calls/statements)=";
    const auto sv = picker_->served();
    for (size_t i = 0; i < sv.size(); i++) out += (i ? "," : "") + std::to_string(sv[i].calls) + "/" + std::to_string(sv[i].statements);
    return out;
}
size_t Algorithm::domain_size() const { return impls_[0]->domain_n; }
void Algorithm::debug_compute_h(const uint8_t* abc_be, size_t m, uint8_t* h_out) {
    AlgorithmImpl& a = *impls_[0];
    if (m > a.n_constraints) throw std::runtime_error("debug_compute_h: more rows than constraints");
    HIP_CHECK(hipSetDevice(a.cfg.device));
    struct Hold { AlgorithmImpl& a; size_t i; ~Hold() { a.release_lane(i); } } hold{a, a.acquire_lane(0)};
    AlgorithmImpl::Lane& ln = *a.lanes[0];
    const size_t B = 64, cnt = m * B;
    DevBuf<uint8_t> d_be(3 * cnt * 32 + 32);
    d_be.upload(abc_be, 3 * cnt * 32, ln.stream);
    launch_fr_from_be(d_be.p, ln.d_A.p, cnt, ln.stream);
    launch_fr_from_be(d_be.p + cnt * 32, ln.d_B.p, cnt, ln.stream);
    launch_fr_from_be(d_be.p + 2 * cnt * 32, ln.d_C.p, cnt, ln.stream);
    NttPlan plan{a.L, a.tw_fwd.p, a.tw_inv.p, a.scale_mid.p, a.scale_out.p, a.dom.p + 5, a.qr.p};
    HIP_CHECK(launch_compute_h(plan, ln.d_A.p, ln.d_B.p, ln.d_C.p, m, B, ln.stream));
    HIP_CHECK(hipMemcpyAsync(h_out, ln.d_A.p, a.domain_n * B * 32, hipMemcpyDeviceToHost, ln.stream));
    HIP_CHECK(hipStreamSynchronize(ln.stream));
}
// one replica: cut the request list into chunks (multiples of 64 proofs, at most one lane's capacity) and let worker threads pull
// chunks, each on whichever lane is free.  A call with at least 2 * min_split statements that is alone on the replica is cut into as
// many chunks as there are lanes (the latency-bound witness stage of one chunk hides under the kernels of the other); when other
// calls are in flight it stays whole and the overlap happens between calls instead (measured on AES-128, two callers: +4 %).
static void prove_on_replica(AlgorithmImpl& a, const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* debug_first) {
    if (!n) return;
    const size_t nl = a.full_lanes, lane_cap = a.lanes[0]->cap;      // only full lanes take the chunks of a big call
    size_t nchunks = (n + lane_cap - 1) / lane_cap;
    struct InFlight { std::atomic<int>& c; int seen; explicit InFlight(std::atomic<int>& x) : c(x), seen(x.fetch_add(1) + 1) {} ~InFlight() { c.fetch_sub(1); } } me(a.calls_in_flight);
    if (nl > 1 && me.seen == 1 && n >= 2 * a.cfg.min_split) { const size_t want = (nchunks + nl - 1) / nl * nl; nchunks = want; }
    size_t chunk = ((n + nchunks - 1) / nchunks + 63) / 64 * 64;
    if (chunk > lane_cap) chunk = lane_cap;
    nchunks = (n + chunk - 1) / chunk;
    std::atomic<size_t> next{0};
    std::exception_ptr err; std::mutex err_mu;
    auto work = [&]() {
        try {
            HIP_CHECK(hipSetDevice(a.cfg.device));
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nchunks) break;
                const size_t off = c * chunk, take = n - off < chunk ? n - off : chunk;
                struct Hold { AlgorithmImpl& a; size_t i; ~Hold() { a.release_lane(i); } } hold{a, a.acquire_lane(-1, take)};
                a.prove_chunk(*a.lanes[hold.i], reqs + off, take, results + off, off == 0 ? debug_first : nullptr);
            }
        } catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    const size_t nthreads = nchunks < nl ? nchunks : nl;
    std::vector<std::thread> th;
    for (size_t t = 1; t < nthreads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}
void Algorithm::prove_batch(const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* debug_first) {
    if (!n) return;
    const size_t nd = impls_.size();
    struct Held { ReplicaPicker& p; size_t i, n; std::atomic<size_t>& last; ~Held() { p.release(i, n); last.store(i); } };
    if (nd == 1 || n <= 64) {
        // A call of up to one 64-column batch is not split: it goes, whole, to the least-loaded replica (ReplicaPicker) — concurrent
        // single-proof callers and the micro-batcher's small batches therefore use every GPU of the node, not only the first one.
        Held h{*picker_, picker_->acquire(n), n, last_replica_};
        return prove_on_replica(*impls_[h.i], reqs, n, results, debug_first);
    }
    // Proofs are independent: contiguous shares (multiples of 64) go to the replicas, one host thread per device; nothing is
    // exchanged between devices (the "gather" is the results array the threads fill).
    size_t share = ((n + nd - 1) / nd + 63) / 64 * 64;
    std::exception_ptr err; std::mutex err_mu; std::vector<std::thread> th;
    auto work = [&](size_t d) {
        const size_t off = d * share;
        if (off >= n) return;
        const size_t take = n - off < share ? n - off : share;
        picker_->acquire_on(d, take);
        Held h{*picker_, d, take, last_replica_};
        try { prove_on_replica(*impls_[d], reqs + off, take, results + off, d == 0 ? debug_first : nullptr); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    for (size_t d = 1; d < nd; d++) th.emplace_back(work, d);
    work(0);
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}

}  // namespace gsc
