// Engine front: configuration from the environment, the per-device replicas of an algorithm and the dispatch of calls over them.
// See engine.hpp for the reference interface this mirrors; engine_impl.hpp for the per-device engine.
#include "engine_impl.hpp"
#include "dispatch.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace gsc {

namespace {
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
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
    c.small_lane_cap = env_int("GSC_SMALL_LANE_CAP", 0);
    if (c.small_lane_cap < 0 || c.small_lane_cap % 64) throw std::runtime_error("GSC_SMALL_LANE_CAP must be a multiple of 64");
    // test-only knobs (no product setting, no profile uses a non-default value): honoured with the load-time test-hooks flag only
    const bool hooks = test_hooks_enabled();
    c.min_split = (size_t)(hooks ? env_int("GSC_MIN_SPLIT", 0) : 0);
    c.bit_groups = env_int("GSC_BIT_GROUPS", 1);
    c.window_z = env_int("GSC_WINDOW_Z", 0);
    c.window_w = env_int("GSC_WINDOW_W", 0);
    c.z_table_gb = env_int("GSC_Z_TABLE_GB", 48);
    c.w_table_gb = env_int("GSC_W_TABLE_GB", 16);
    c.row_margin_bits = hooks ? env_int("GSC_ROW_MARGIN_BITS", 1) : 1;
    c.few_path = env_int("GSC_FEW_PATH", 1);
    c.few_solver = env_int("GSC_FEW_SOLVER", 1);
    c.few_max = env_int("GSC_FEW_MAX", 0);
    if (c.few_max < 0 || c.few_max > (int)MSM_FEW_PROOFS) throw std::runtime_error("GSC_FEW_MAX must be in [0, 32]");
    c.few_workgroups = hooks ? env_int("GSC_FEW_WGS", 0) : 0;
    c.few_z_gb = env_int("GSC_FEW_Z_GB", 12);
    c.few_wide = env_int("GSC_FEW_WIDE", 1);
    c.quotient_eval = env_int("GSC_QUOTIENT_EVAL", 1);
    c.fuse_z_digits = env_int("GSC_FUSE_Z_DIGITS", 1);
    c.small_witness = env_int("GSC_SMALL_WITNESS", 1);
    c.small_witness_few = env_int("GSC_SMALL_WITNESS_FEW", 1);
    c.ntt_plain = env_int("GSC_NTT_PLAIN", 1) ? 1 : 0;
    c.overlap_quotient = env_int("GSC_OVERLAP_QUOTIENT", 1);
    if (c.overlap_quotient < 0 || c.overlap_quotient > 2) throw std::runtime_error("GSC_OVERLAP_QUOTIENT must be 0, 1 or 2");
    c.stream_priorities = env_int("GSC_STREAM_PRIORITIES", 1);      // 0 plain; 1 = main normal, side high, third low; 3-digit codes (test hook): one digit per stream, 1 high 2 normal 3 low
    if (c.stream_priorities != 0 && c.stream_priorities != 1 && !(hooks && c.stream_priorities >= 111 && c.stream_priorities <= 333)) throw std::runtime_error("GSC_STREAM_PRIORITIES must be 0 or 1");
    if (test_hooks_enabled()) { c.win_slice = env_int("GSC_WIN_SLICE", 256); if (c.win_slice < 64 || c.win_slice > 4096 || c.win_slice % 8) throw std::runtime_error("GSC_WIN_SLICE must be a multiple of 8 in [64, 4096]"); }
    if (c.small_witness < 0 || c.small_witness > 2 || (c.small_witness == 2 && !test_hooks_enabled())) throw std::runtime_error("GSC_SMALL_WITNESS must be 0 or 1");
    if (c.few_workgroups < 0 || c.few_workgroups > 256) throw std::runtime_error("GSC_FEW_WGS must be in [0, 256]");
    c.trace_host = getenv("GSC_TRACE_HOST") != nullptr;
    if (test_hooks_enabled()) { c.solver_trace = getenv("GSC_SOLVER_TRACE") != nullptr; c.few_test_abort = getenv("GSC_FEW_TEST_ABORT") != nullptr; c.keep_secrets = getenv("GSC_KEEP_SECRETS") != nullptr; c.z_exp_entry_bits = env_int("GSC_Z_EXP_ENTRY_BITS", 0); }
    if (c.max_batch < 64) c.max_batch = 64;
    c.max_batch = (c.max_batch + 63) / 64 * 64;
    if ((c.window_z && (c.window_z < 4 || c.window_z > MSM_MAX_WINDOW)) || (c.window_w && (c.window_w < 4 || c.window_w > 16))) throw std::runtime_error("GSC_WINDOW_Z must be in [4,17], GSC_WINDOW_W in [4,16]");
    return c;
}

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

void debug_clock_trace(int device, uint32_t n, uint32_t interval_us, unsigned long long* out) {
    HIP_CHECK(hipSetDevice(device));
    hipStream_t st; HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    DevBuf<unsigned long long> d(2 * (size_t)n);
    launch_clock_trace(d.p, n, interval_us * 100u, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d.p, 16 * (size_t)n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    HIP_CHECK(e);
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
namespace { thread_local KernelStat t_call_stat; thread_local const Algorithm* t_call_owner = nullptr; }
KernelStat Algorithm::last_kernel_stat() const {
    if (t_call_owner == this) return t_call_stat;      // this thread's own last call
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
    if (impl_->quotient_eval) out += std::string(" quotient=evaluation-form") + (impl_->fuse_z_digits ? "+digits" : "") + (!impl_->fuse_z_digits || !impl_->cfg.overlap_quotient ? "" : impl_->cfg.overlap_quotient == 2 ? "+beside-wire-sets" : "+beside-wire-sets(<4096)") + "(c: " + std::to_string(impl_->mC.nbit) + " grouped + " + std::to_string(impl_->mC.nflat - impl_->mC.nbit) + " flat + " + std::to_string(impl_->mC.nwide) + " windowed)";
    else out += " quotient=coefficient-form";
    out += impl_->small.ok ? " witness=small-integer(" + std::to_string(impl_->small.n_levels) + " chained levels, fallbacks " + std::to_string(impl_->small_fallbacks.load()) + ")" : " witness=generic" + (impl_->small.why.empty() ? std::string() : "(" + impl_->small.why + ")");
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
void Algorithm::debug_compute_d(const uint8_t* ab_be, size_t m, uint8_t* d_out) {
    AlgorithmImpl& a = *impls_[0];
    if (m > a.n_constraints) throw std::runtime_error("debug_compute_d: more rows than constraints");
    HIP_CHECK(hipSetDevice(a.cfg.device));
    struct Hold { AlgorithmImpl& a; size_t i; ~Hold() { a.release_lane(i); } } hold{a, a.acquire_lane(0)};
    AlgorithmImpl::Lane& ln = *a.lanes[0];
    const size_t B = 64, cnt = m * B;
    DevBuf<uint8_t> d_be(2 * cnt * 32 + 32);
    d_be.upload(ab_be, 2 * cnt * 32, ln.stream);
    launch_fr_from_be(d_be.p, ln.d_A.p, cnt, ln.stream);
    launch_fr_from_be(d_be.p + cnt * 32, ln.d_B.p, cnt, ln.stream);
    NttPlan plan{a.L, a.tw_fwd.p, a.tw_inv.p, a.scale_mid.p, a.scale_out.p, a.dom.p + 5, a.qr.p};
    HIP_CHECK(launch_compute_d(plan, ln.d_A.p, ln.d_B.p, m, B, ln.stream));
    HIP_CHECK(hipMemcpyAsync(d_out, ln.d_A.p, a.domain_n * B * 32, hipMemcpyDeviceToHost, ln.stream));
    HIP_CHECK(hipStreamSynchronize(ln.stream));
}
// one replica: cut the request list into chunks (multiples of 64 proofs, at most one lane's capacity) and let worker threads pull
// chunks, each on whichever lane is free.  A call that fits one lane stays whole: cutting a lone 1024-statement AES-128 call over
// the two lanes doubles the latency-bound level launches of the witness stage, and the lanes' streams share a hardware queue more
// often than not, so nothing hides (round 4, profiles/r04_aes128_lanes.txt: 3 255 proofs/s cut, 3 341 - 3 362 whole; round 2 had
// measured +4 % for cutting, on slower MSM kernels).  min_split > 0 (test hook) brings the cut back: the tests use it to cover
// chunked calls with small batches.
static void prove_on_replica(AlgorithmImpl& a, const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* debug_first, KernelStat* stat, std::mutex* stat_mu) {
    if (!n) return;
    const size_t nl = a.full_lanes, lane_cap = a.lanes[0]->cap;      // only full lanes take the chunks of a big call
    size_t nchunks = (n + lane_cap - 1) / lane_cap;
    struct InFlight { std::atomic<int>& c; int seen; explicit InFlight(std::atomic<int>& x) : c(x), seen(x.fetch_add(1) + 1) {} ~InFlight() { c.fetch_sub(1); } } me(a.calls_in_flight);
    if (nl > 1 && a.cfg.min_split && me.seen == 1 && n >= 2 * a.cfg.min_split) { const size_t want = (nchunks + nl - 1) / nl * nl; nchunks = want; }
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
                if (stat) { std::lock_guard<std::mutex> g(*stat_mu); if (a.lanes[hold.i]->stat.ms >= stat->ms) *stat = a.lanes[hold.i]->stat; }      // the call's chunk with the longest dominant kernel
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
size_t Algorithm::debug_secret_residue() { size_t n = 0; for (auto& i : impls_) n += i->secret_residue(); return n; }
void Algorithm::forget_thread_stat() const { if (t_call_owner == this) t_call_owner = nullptr; }
void Algorithm::prove_batch(const ProofRequest* reqs, size_t n, ProofResult* results, DebugVectors* debug_first, bool whole) {
    if (!n) return;
    struct Held { ReplicaPicker& p; size_t i, n; std::atomic<size_t>& last; ~Held() { p.release(i, n); last.store(i); } };
    KernelStat call_stat; std::mutex call_stat_mu;
    struct Publish { const Algorithm* self; KernelStat& st; ~Publish() { t_call_stat = st; t_call_owner = self; } } publish{this, call_stat};
    const std::vector<CallShare> shares = plan_shares(n, impls_.size(), impls_[0]->cap, whole);
    if (shares.size() == 1 && shares[0].pick) {
        // not split: the call goes, whole, to the least-loaded replica (ReplicaPicker) — concurrent single-proof callers and the
        // micro-batcher's batches therefore use every GPU of the node, not only the first one.
        Held h{*picker_, picker_->acquire(n), n, last_replica_};
        return prove_on_replica(*impls_[h.i], reqs, n, results, debug_first, &call_stat, &call_stat_mu);
    }
    // Proofs are independent: contiguous shares (multiples of 64) go to the replicas, one host thread per device; nothing is
    // exchanged between devices (the "gather" is the results array the threads fill).
    std::exception_ptr err; std::mutex err_mu; std::vector<std::thread> th;
    auto work = [&](const CallShare& sh) {
        picker_->acquire_on(sh.replica, sh.n);
        Held h{*picker_, sh.replica, sh.n, last_replica_};
        try { prove_on_replica(*impls_[sh.replica], reqs + sh.off, sh.n, results + sh.off, sh.off == 0 ? debug_first : nullptr, &call_stat, &call_stat_mu); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    for (size_t k = 1; k < shares.size(); k++) th.emplace_back(work, shares[k]);
    work(shares[0]);
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}

}  // namespace gsc
