// C-ABI of libprove (see include/libprove.h) and the JSON host logic behind it.
//
// Mirrors, symbol for symbol, the reference's cgo exports (libraries/prover/libprove.go:17-47) and the
// dispatch / JSON layer they call (libraries/prover/impl/prove_impl.go:65-143, provers.go:53-59, :79-89, :172-182):
// algorithm registry, idempotent InitAlgorithm, Go-encoding/json-compatible input decoding, the
// {"proof":{"proofJson"},"publicSignals"} output, and "panic -> JSON" error reporting.
#include "../../include/libprove.h"
#include "engine.hpp"
#include "dispatch.hpp"
#include "glv.hpp"
#include "host_ciphers.hpp"
#include "json.hpp"
#include "setup.hpp"
#include <sys/random.h>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace gsc;

namespace {

const char* kAlgorithmNames[3] = {"chacha20", "aes-128-ctr", "aes-256-ctr"};   // prove_impl.go:21-25

// Micro-batching: concurrent single-proof Prove() callers (the reference is called from many goroutines / FFI threads:
// libraries/core_test.go:44-111) are gathered into ONE device batch instead of running one proof each, back to back.
// Worker threads — one per (device, lane) of the algorithm's engine — drain the queue; WHEN a worker takes HOW MANY callers is
// csrc/dispatch.hpp's BatchScheduler (tested on CPU with a stub engine): a lone caller on an idle device goes at once; while every
// device has a batch on it the callers that arrive form the next one; the callers of a batch that has just completed are waited for
// as long as they keep coming back (GSC_LINGER_US, default 300 us, sets the windows; 0 = never wait on an idle device); a burst is
// shared out over the free devices and every share goes out as one device batch to the least-loaded engine replica.
class Batcher {
  public:
    explicit Batcher(Algorithm* a) : algo_(a), sched_(a->devices() ? a->devices() : 1, a->max_batch(), linger_from_env()) {
        // one worker per (device, lane): a second batch per device is started when enough callers are queued for it (dispatch.hpp)
        const size_t nw = (a->lanes() ? a->lanes() : 1) * (a->devices() ? a->devices() : 1);
        for (size_t i = 0; i < nw; i++) workers_.emplace_back([this] { run(); });
    }
    ~Batcher() { { std::lock_guard<std::mutex> l(mu_); stop_ = true; } sched_.cv.notify_all(); for (auto& w : workers_) if (w.joinable()) w.join(); }
    // blocks until the proof is done; throws std::runtime_error if the device batch failed
    void submit(const ProofRequest& req, ProofResult& out) {
        Item it{&req, &out, false, std::string(), &it};
        { std::unique_lock<std::mutex> l(mu_); q_.push_back(&it); sched_.arrived(); done_cv_.wait(l, [&] { return it.done; }); }
        if (!it.error.empty()) throw std::runtime_error(it.error);
    }
    // A small call of several statements (ProveBatch / gsc_prove_raw with up to SMALL_CALL statements): its statements queue like single
    // callers, so concurrent small calls share device batches whatever entry point they came through.
    static constexpr size_t SMALL_CALL = 32;
    void submit_many(const ProofRequest* reqs, size_t n, ProofResult* out) {
        std::vector<Item> items(n);
        {
            std::unique_lock<std::mutex> l(mu_);
            for (size_t i = 0; i < n; i++) { items[i] = Item{&reqs[i], &out[i], false, std::string(), items.data()}; q_.push_back(&items[i]); }
            sched_.arrived(n);
            done_cv_.wait(l, [&] { for (const Item& it : items) if (!it.done) return false; return true; });
        }
        for (const Item& it : items) if (!it.error.empty()) throw std::runtime_error(it.error);
    }
  private:
    struct Item { const ProofRequest* req; ProofResult* res; bool done; std::string error; const void* call; };      // call: the same for all statements of one submit_many
    static int linger_from_env() { const char* e = getenv("GSC_LINGER_US"); return e && *e ? atoi(e) : 300; }
    void run() {
        for (;;) {
            std::vector<Item*> take;
            {
                std::unique_lock<std::mutex> l(mu_);
                const size_t want = sched_.wait_for_batch(l, [&] { return q_.size(); }, stop_);
                if (!want) return;                         // stopped, nothing queued
                while (!q_.empty() && take.size() < want) { take.push_back(q_.front()); q_.pop_front(); }
                sched_.started(take.size());
            }
            // Whatever happens below — a device error, an allocation failure, anything thrown — the callers parked on done_cv_ are released
            // and the scheduler hears of the completion: nobody is left waiting on a batch that died.
            std::string err; std::vector<ProofResult> res;
            try {
                std::vector<ProofRequest> reqs(take.size()); res.resize(take.size());
                for (size_t i = 0; i < take.size(); i++) reqs[i] = *take[i]->req;
                algo_->prove_batch(reqs.data(), reqs.size(), res.data(), nullptr, true);      // whole: one device batch on ONE replica (plan_shares)
            } catch (const std::exception& e) { err = e.what(); if (err.empty()) err = "proving failed"; }
            catch (...) { err = "proving failed (unknown error)"; }
            {
                std::lock_guard<std::mutex> l(mu_);
                for (size_t i = 0; i < take.size(); i++) { if (err.empty() && i < res.size()) *take[i]->res = res[i]; take[i]->error = err; take[i]->done = true; }
                size_t calls = 0; const void* prev = nullptr;      // (the statements of a call are queued back to back)
                for (Item* it : take) { if (it->call != prev) calls++; prev = it->call; }
                sched_.completed(take.size(), calls);
            }
            done_cv_.notify_all();
        }
    }
    Algorithm* algo_; BatchScheduler sched_; bool stop_ = false;
    std::mutex mu_; std::condition_variable done_cv_; std::deque<Item*> q_; std::vector<std::thread> workers_;
};

std::mutex g_mu;
std::unique_ptr<Algorithm> g_algo[3];
std::unique_ptr<Batcher> g_batcher[3];     // declared after g_algo: destroyed first
bool g_fixed_rand = false; uint8_t g_r[32], g_s[32], g_mask[32];   // little-endian canonical; written under g_mu, read by fill_randomness under g_mu
DebugVectors g_debug;
// The gsc_debug_* entry points and gsc_set_deterministic_randomness exist for the parity tests only: fixing (r, s, mask) removes
// zero-knowledge for every caller of the process.  They refuse to work unless the process was started with
// GSC_ENABLE_TEST_HOOKS=1; the variable is read ONCE, when the library is loaded, so code running inside the host cannot
// switch them on later.
const bool g_test_hooks = test_hooks_enabled();
bool hooks_refused(const char* what) {
    if (g_test_hooks) return false;
    printf("%s refused: test hooks are disabled (start the process with GSC_ENABLE_TEST_HOOKS=1)\n", what);
    return true;
}

// Fr modulus, big-endian
const uint8_t kFrModBE[32] = {0x30, 0x64, 0x4e, 0x72, 0xe1, 0x31, 0xa0, 0x29, 0xb8, 0x50, 0x45, 0xb6, 0x81, 0x81, 0x58, 0x5d,
                              0x28, 0x33, 0xe8, 0x48, 0x79, 0xb9, 0x70, 0x91, 0x43, 0xe1, 0xf5, 0x93, 0xf0, 0x00, 0x00, 0x01};

// OS CSPRNG bytes, fetched 16 KiB at a time per thread: a batch of 8192 proofs draws ~25 000 scalars, and one getrandom() system
// call per scalar was a measurable part of the step (the GPU idles meanwhile).  The buffer is wiped as it is consumed.
void csprng_bytes(uint8_t* out, size_t n) {
    thread_local uint8_t pool[16384]; thread_local size_t have = 0;
    while (n) {
        if (!have) {
            size_t got = 0;
            while (got < sizeof pool) {
                const ssize_t k = getrandom(pool + got, sizeof pool - got, 0);
                if (k > 0) got += (size_t)k;
                else if (k < 0 && errno == EINTR) continue;
                else throw std::runtime_error(std::string("getrandom failed: ") + strerror(errno));   // e.g. ENOSYS under seccomp: fail the request, never spin or fall back
            }
            have = sizeof pool;
        }
        const size_t take = n < have ? n : have;
        uint8_t* src = pool + (sizeof pool - have);
        memcpy(out, src, take);
        volatile uint8_t* wipe = src; for (size_t i = 0; i < take; i++) wipe[i] = 0;
        out += take; n -= take; have -= take;
    }
}
void random_fr_le(uint8_t out[32]) {   // uniform in [0, r) by rejection (fr.SetRandom in the reference)
    for (;;) {
        uint8_t be[32]; csprng_bytes(be, 32);
        be[0] &= 0x3F;
        if (memcmp(be, kFrModBE, 32) < 0) { for (int i = 0; i < 32; i++) out[i] = be[31 - i]; return; }
    }
}

// A Go panic value, already rendered as the JSON that libprove.go:33-43 would return.
struct GoPanic { std::string json; };
[[noreturn]] void panic_string(const std::string& msg) { throw GoPanic{json_quote(msg)}; }

int find_cipher(const std::string& name) { for (int i = 0; i < 3; i++) if (name == kAlgorithmNames[i]) return i; return -1; }

bool ascii_fold_eq(const std::string& a, const char* b) {
    size_t n = strlen(b); if (a.size() != n) return false;
    for (size_t i = 0; i < n; i++) { char x = a[i], y = b[i]; if (x >= 'A' && x <= 'Z') x += 32; if (y >= 'A' && y <= 'Z') y += 32; if (x != y) return false; }
    return true;
}

struct Decoded {
    std::string cipher; std::vector<uint8_t> key, nonce, input; uint32_t counter = 0;
    Decoded() = default; Decoded(Decoded&&) = default; Decoded& operator=(Decoded&&) = default; Decoded(const Decoded&) = delete; Decoded& operator=(const Decoded&) = delete;
    ~Decoded() { if (!key.empty()) explicit_bzero(key.data(), key.size()); }      // the decoded key does not linger on the heap
};

// encoding/json semantics for InputParams (provers.go:53-59): case-insensitive keys, unknown keys ignored, the first
// type error is remembered while decoding continues, []uint8 from base64 string / null / array of 0..255.
struct TypeError { bool set = false; std::string json; };
void type_error(TypeError& te, const std::string& value, size_t offset, const char* field) {
    if (te.set) return;
    te.set = true;
    te.json = "{\"Value\":" + json_quote(value) + ",\"Type\":{},\"Offset\":" + std::to_string(offset) + ",\"Struct\":\"InputParams\",\"Field\":" + json_quote(field) + "}";
}
size_t go_offset(const JsonValue& v) { return (v.kind == JsonValue::Array || v.kind == JsonValue::Object) ? v.start + 1 : v.offset; }

void decode_bytes(const JsonValue& v, std::vector<uint8_t>& out, TypeError& te, const char* field) {
    switch (v.kind) {
        case JsonValue::Null: out.clear(); return;
        case JsonValue::String: {
            size_t bad = 0; std::vector<uint8_t> tmp;
            if (!base64_decode(v.text, tmp, bad)) throw GoPanic{std::to_string(bad)};   // base64.CorruptInputError is an int64
            out = std::move(tmp); return;
        }
        case JsonValue::Array: {
            std::vector<uint8_t> tmp;
            for (const JsonValue& e : v.items) {
                uint8_t b = 0;
                if (e.kind == JsonValue::Number) {
                    const std::string& t = e.text; bool ok = !t.empty() && t.find_first_not_of("0123456789") == std::string::npos && t.size() <= 3;
                    unsigned val = ok ? (unsigned)atoi(t.c_str()) : 256;
                    if (val > 255) type_error(te, "number " + t, e.offset, field); else b = (uint8_t)val;
                } else if (e.kind != JsonValue::Null) type_error(te, e.go_kind(), go_offset(e), field);
                tmp.push_back(b);
            }
            out = std::move(tmp); return;
        }
        default: type_error(te, v.go_kind(), go_offset(v), field); return;
    }
}

Decoded decode_params(const JsonValue& root) {
    Decoded d; TypeError te;
    if (root.kind == JsonValue::Null) panic_string("runtime error: invalid memory address or nil pointer dereference");
    if (root.kind != JsonValue::Object) { type_error(te, root.go_kind(), go_offset(root), ""); 
        // Go reports Struct/Field empty for a top-level mismatch
        throw GoPanic{"{\"Value\":" + json_quote(root.go_kind()) + ",\"Type\":{},\"Offset\":" + std::to_string(go_offset(root)) + ",\"Struct\":\"\",\"Field\":\"\"}"}; }
    for (const auto& kv : root.members) {
        const JsonValue& v = kv.second;
        if (ascii_fold_eq(kv.first, "cipher")) {
            if (v.kind == JsonValue::String) d.cipher = v.text; else if (v.kind != JsonValue::Null) type_error(te, v.go_kind(), go_offset(v), "cipher");
        } else if (ascii_fold_eq(kv.first, "key")) decode_bytes(v, d.key, te, "key");
        else if (ascii_fold_eq(kv.first, "nonce")) decode_bytes(v, d.nonce, te, "nonce");
        else if (ascii_fold_eq(kv.first, "input")) decode_bytes(v, d.input, te, "input");
        else if (ascii_fold_eq(kv.first, "counter")) {
            if (v.kind == JsonValue::Number) {
                const std::string& t = v.text; bool ok = !t.empty() && t.find_first_not_of("0123456789") == std::string::npos && t.size() <= 10;
                unsigned long long val = ok ? strtoull(t.c_str(), nullptr, 10) : ~0ull;
                if (!ok || val > 0xFFFFFFFFull) type_error(te, "number " + t, v.offset, "counter"); else d.counter = (uint32_t)val;
            } else if (v.kind != JsonValue::Null) type_error(te, v.go_kind(), go_offset(v), "counter");
        }
    }
    if (te.set) throw GoPanic{te.json};
    return d;
}

// length checks of provers.go:81-89 / :174-182 (log.Panicf -> string panic), then the native cipher
ProofRequest make_request(int cipher, const Decoded& d) {
    ProofRequest q{};
    if (cipher == CHACHA20) { if (d.key.size() != 32) panic_string("key length must be 32: " + std::to_string(d.key.size())); }
    else if (d.key.size() != 32 && d.key.size() != 16) panic_string("key length must be 16 or 32: " + std::to_string(d.key.size()));
    if (d.nonce.size() != 12) panic_string("nonce length must be 12: " + std::to_string(d.nonce.size()));
    if (d.input.size() != 64) panic_string("plaintext length must be 64: " + std::to_string(d.input.size()));
    if (cipher != CHACHA20 && d.key.size() != (cipher == AES_128 ? 16u : 32u)) throw GoPanic{"{}"};   // frontend.NewWitness schema mismatch: an error value with no exported fields
    q.keylen = (uint32_t)d.key.size(); memcpy(q.key, d.key.data(), d.key.size());
    memcpy(q.nonce, d.nonce.data(), 12); q.counter = d.counter; memcpy(q.plaintext, d.input.data(), 64);
    if (cipher == CHACHA20) chacha20_xor_stream(q.key, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
    else aes_ctr_xor_stream(q.key, q.keylen, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
    return q;
}
void fill_randomness(ProofRequest& q) {
    if (g_test_hooks) {      // the fixed values can only ever be set in a test process
        std::lock_guard<std::mutex> l(g_mu);
        if (g_fixed_rand) { memcpy(q.r, g_r, 32); memcpy(q.s, g_s, 32); memcpy(q.mask, g_mask, 32); return; }
    }
    random_fr_le(q.r); random_fr_le(q.s); random_fr_le(q.mask);
}

Algorithm* lookup(int cipher) { std::lock_guard<std::mutex> l(g_mu); return g_algo[cipher].get(); }
Batcher* batcher(int cipher) { std::lock_guard<std::mutex> l(g_mu); return g_batcher[cipher].get(); }

std::string success_json(const ProofResult& r, const uint8_t ct[64]) {   // OutputParams, prove_impl.go:45-52
    return "{\"proof\":{\"proofJson\":\"" + base64_encode(r.proof, r.proof_len) + "\"},\"publicSignals\":\"" + base64_encode(ct, 64) + "\"}";
}

// parses + validates one request; throws GoPanic
struct Prepared { int cipher; ProofRequest req; };
Prepared prepare(const JsonValue& v) {
    Decoded d = decode_params(v);
    const int cipher = find_cipher(d.cipher);
    if (cipher < 0) panic_string("could not find prover for" + d.cipher);                       // prove_impl.go:141 (sic: no space)
    if (!lookup(cipher)) panic_string("proving params are not initialized for cipher: " + d.cipher);   // :124-126
    Prepared p{cipher, make_request(cipher, d)};
    fill_randomness(p.req);
    return p;
}

struct Prove_return to_c(const std::string& s) {
    void* buf = malloc(s.size() ? s.size() : 1);           // C.CBytes
    if (!buf) return {nullptr, 0};
    memcpy(buf, s.data(), s.size());
    return {buf, (GoInt)s.size()};
}

std::string prove_one_json(const char* data, size_t len, DebugVectors* dbg) {
    try {
        JsonValue root;
        try { root = json_parse(data, len); } catch (const JsonSyntaxError& e) { throw GoPanic{"{\"Offset\":" + std::to_string(e.offset) + "}"}; }
        Prepared p = prepare(root);
        ProofResult res;
        if (dbg) lookup(p.cipher)->prove_batch(&p.req, 1, &res, dbg);      // test hook: straight to the device, keeps the intermediates
        else batcher(p.cipher)->submit(p.req, res);                       // shares a device batch with concurrent callers
        if (res.status) throw GoPanic{"{}"};     // gnark solver / prover error: no exported fields
        return success_json(res, p.req.ciphertext);
    } catch (const GoPanic& g) {
        printf("%s\n", g.json.c_str());        // libprove.go:35 prints the panic value
        return g.json;
    } catch (const std::exception& e) {
        printf("%s\n", e.what());
        return json_quote(e.what());
    }
}

}  // namespace

extern "C" {

void enforce_binding(void) {}

GoUint8 InitAlgorithm(GoUint8 algorithmID, GoSlice provingKey, GoSlice r1cs) {
    if (algorithmID > 2) return 0;                               // unknown id -> false (prove_impl.go:113)
    std::lock_guard<std::mutex> l(g_mu);
    if (g_algo[algorithmID]) return 1;                           // already initialised (prove_impl.go:74-76)
    try {
        if (!provingKey.data || provingKey.len <= 0 || !r1cs.data || r1cs.len <= 0) throw std::runtime_error("error reading proving key: EOF");
        g_algo[algorithmID].reset(new Algorithm((Cipher)algorithmID, (const uint8_t*)provingKey.data, (size_t)provingKey.len,
                                                (const uint8_t*)r1cs.data, (size_t)r1cs.len, config_from_env()));
        g_batcher[algorithmID].reset(new Batcher(g_algo[algorithmID].get()));
        return 1;
    } catch (const std::exception& e) {
        printf("%s\n", e.what());                                 // fmt.Println(err) in the reference
        return 0;
    }
}

void Free(void* pointer) { free(pointer); }

struct Prove_return Prove(GoSlice params) {
    return to_c(prove_one_json((const char*)params.data, params.len > 0 ? (size_t)params.len : 0, nullptr));
}

struct Prove_return ProveBatch(GoSlice params) {
    std::string out;
    try {
        JsonValue root;
        try { root = json_parse((const char*)params.data, params.len > 0 ? (size_t)params.len : 0); }
        catch (const JsonSyntaxError& e) { throw GoPanic{"{\"Offset\":" + std::to_string(e.offset) + "}"}; }
        if (root.kind != JsonValue::Array) throw GoPanic{json_quote("ProveBatch expects a JSON array")};
        const size_t n = root.items.size();
        std::vector<std::string> results(n);
        std::vector<Prepared> ok; std::vector<size_t> where;
        for (size_t i = 0; i < n; i++) {
            try { ok.push_back(prepare(root.items[i])); where.push_back(i); }
            catch (const GoPanic& g) { results[i] = g.json; }
        }
        // one device batch per algorithm; the algorithms have their own streams and buffers, so their batches run concurrently
        // (the solver levels and the commitment round trip of one hide under the MSMs of another)
        std::exception_ptr err; std::mutex err_mu;
        auto run = [&](int c) {
            try {
                std::vector<ProofRequest> reqs; std::vector<size_t> idx;
                for (size_t k = 0; k < ok.size(); k++) if (ok[k].cipher == c) { reqs.push_back(ok[k].req); idx.push_back(where[k]); }
                if (reqs.empty()) return;
                std::vector<ProofResult> res(reqs.size());
                Batcher* b = reqs.size() <= Batcher::SMALL_CALL ? batcher(c) : nullptr;
                if (b) { if (Algorithm* al = lookup((GoUint8)c)) al->forget_thread_stat(); b->submit_many(reqs.data(), reqs.size(), res.data()); }      // a small group shares device batches with concurrent callers
                else lookup(c)->prove_batch(reqs.data(), reqs.size(), res.data());
                for (size_t k = 0; k < reqs.size(); k++) results[idx[k]] = res[k].status ? std::string("{}") : success_json(res[k], reqs[k].ciphertext);
            } catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
        };
        bool present[3] = {false, false, false};
        for (const Prepared& p : ok) present[p.cipher] = true;
        std::vector<std::thread> th;
        int first = -1;
        for (int c = 0; c < 3; c++) if (present[c]) { if (first < 0) first = c; else th.emplace_back(run, c); }
        if (first >= 0) run(first);
        for (auto& t : th) t.join();
        if (err) std::rethrow_exception(err);
        out = "[";
        for (size_t i = 0; i < n; i++) { if (i) out += ","; out += results[i]; }
        out += "]";
    } catch (const GoPanic& g) { out = g.json; }
    catch (const std::exception& e) { out = json_quote(e.what()); }
    return to_c(out);
}

long long gsc_prove_raw(GoUint8 cipher, const uint8_t* inputs, size_t n, uint8_t* proofs, uint32_t* proof_lens, uint8_t* ciphertexts) {
    if (cipher > 2) return -1;
    Algorithm* a = lookup(cipher);
    if (!a) return -1;
    try {
        static const bool trace = getenv("GSC_TRACE_HOST") != nullptr;      // read once
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<ProofRequest> reqs(n); std::vector<ProofResult> res(n);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* rec = inputs + 112 * i; ProofRequest& q = reqs[i]; q = ProofRequest{};
            q.keylen = cipher == AES_128 ? 16 : 32; memcpy(q.key, rec, q.keylen); memcpy(q.nonce, rec + 32, 12);
            q.counter = (uint32_t)rec[44] | ((uint32_t)rec[45] << 8) | ((uint32_t)rec[46] << 16) | ((uint32_t)rec[47] << 24);
            memcpy(q.plaintext, rec + 48, 64);
            if (cipher == CHACHA20) chacha20_xor_stream(q.key, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
            else aes_ctr_xor_stream(q.key, q.keylen, q.nonce, q.counter, q.plaintext, q.ciphertext, 64);
            fill_randomness(q);
        }
        const auto t1 = std::chrono::steady_clock::now();
        Batcher* b = n && n <= Batcher::SMALL_CALL ? batcher(cipher) : nullptr;
        if (b) { a->forget_thread_stat(); b->submit_many(reqs.data(), n, res.data()); }      // a small call shares device batches with concurrent callers
        else a->prove_batch(reqs.data(), n, res.data());
        const auto t2 = std::chrono::steady_clock::now();
        if (trace) fprintf(stderr, "gsc_prove_raw: prepare %.2f ms, prove_batch %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        long long good = 0;
        for (size_t i = 0; i < n; i++) {
            if (ciphertexts) memcpy(ciphertexts + 64 * i, reqs[i].ciphertext, 64);
            proof_lens[i] = res[i].status ? 0u : (uint32_t)res[i].proof_len;
            memset(proofs + 196 * i, 0, 196);
            if (!res[i].status) { memcpy(proofs + 196 * i, res[i].proof, res[i].proof_len); good++; }
        }
        return good;
    } catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

int gsc_setup(GoSlice r1cs, const uint8_t* seed32, void** pk, size_t* pk_len, void** vk, size_t* vk_len) {
    if (!pk || !pk_len || !vk || !vk_len) return -1;
    *pk = *vk = nullptr; *pk_len = *vk_len = 0;
    if (seed32 && hooks_refused("gsc_setup with a caller-supplied seed")) return -1;      // deterministic toxic waste: test keys only
    try {
        if (!r1cs.data || r1cs.len <= 0) throw std::runtime_error("error reading r1cs: EOF");
        SetupKeys k = groth16_setup((const uint8_t*)r1cs.data, (size_t)r1cs.len, seed32, config_from_env().device);
        void* a = malloc(k.pk.size()); void* b = malloc(k.vk.size());
        if (!a || !b) { free(a); free(b); return -1; }
        memcpy(a, k.pk.data(), k.pk.size()); memcpy(b, k.vk.data(), k.vk.size());
        *pk = a; *pk_len = k.pk.size(); *vk = b; *vk_len = k.vk.size();
        return 0;
    } catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

int gsc_set_deterministic_randomness(const uint8_t* r_be32, const uint8_t* s_be32, const uint8_t* mask_be32) {
    if (hooks_refused("gsc_set_deterministic_randomness")) return -1;
    std::lock_guard<std::mutex> l(g_mu);
    if (!r_be32 || !s_be32) { g_fixed_rand = false; return 0; }
    for (int i = 0; i < 32; i++) { g_r[i] = r_be32[31 - i]; g_s[i] = s_be32[31 - i]; g_mask[i] = mask_be32 ? mask_be32[31 - i] : 0; }
    g_fixed_rand = true;
    return 0;
}

long long gsc_debug_prove(GoSlice params) {
    if (hooks_refused("gsc_debug_prove")) return -1;
    g_debug = DebugVectors();
    std::string r = prove_one_json((const char*)params.data, params.len > 0 ? (size_t)params.len : 0, &g_debug);
    return r.find("\"proof\"") != std::string::npos ? 0 : -1;
}
long long gsc_debug_vector(int which, uint8_t* out, size_t cap) {
    if (hooks_refused("gsc_debug_vector")) return -1;
    const std::vector<uint8_t>* v = which == 0 ? &g_debug.W : which == 1 ? &g_debug.A : which == 2 ? &g_debug.B : which == 3 ? &g_debug.C : which == 4 ? &g_debug.H : nullptr;
    if (!v || v->empty()) return -1;
    if (out) memcpy(out, v->data(), cap < v->size() ? cap : v->size());
    return (long long)(v->size() / 32);
}

int gsc_debug_field_ops(int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int chain) {
    if (hooks_refused("gsc_debug_field_ops")) return -1;
    try { debug_field_ops(config_from_env().device, field, op, a, b, out, n, chain); return 0; }
    catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

int gsc_debug_clock_trace(uint32_t n, uint32_t interval_us, unsigned long long* out) {
    if (hooks_refused("gsc_debug_clock_trace") || !out || !n) return -1;
    try { debug_clock_trace(config_from_env().device, n, interval_us, out); return 0; }
    catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

int gsc_debug_glv_split(const uint8_t* k, uint8_t* out) {
    if (hooks_refused("gsc_debug_glv_split") || !k || !out) return -1;
    uint32_t w[8]; memcpy(w, k, 32);
    GlvSplit s;
    if (!glv_split(w, s)) return -1;
    memcpy(out, s.k1, 20); memcpy(out + 20, s.k2, 20); memcpy(out + 40, &s.neg, 4);
    return 0;
}
long long gsc_debug_compute_h(GoUint8 algorithmID, const uint8_t* abc_be, size_t m, uint8_t* h_out, size_t cap) {
    if (hooks_refused("gsc_debug_compute_h") || algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    if (!h_out) return (long long)a->domain_size();
    if (cap < a->domain_size() * 64 * 32) return -1;
    try { a->debug_compute_h(abc_be, m, h_out); return (long long)a->domain_size(); }
    catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

long long gsc_debug_secret_residue(GoUint8 algorithmID) {
    if (hooks_refused("gsc_debug_secret_residue") || algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    try { return (long long)a->debug_secret_residue(); } catch (const std::exception& e) { fprintf(stderr, "gsc_debug_secret_residue: %s\n", e.what()); return -1; }
}
long long gsc_debug_compute_d(GoUint8 algorithmID, const uint8_t* ab_be, size_t m, uint8_t* d_out, size_t cap) {
    if (hooks_refused("gsc_debug_compute_d") || algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    if (!d_out) return (long long)a->domain_size();
    if (cap < a->domain_size() * 64 * 32) return -1;
    try { a->debug_compute_d(ab_be, m, d_out); return (long long)a->domain_size(); }
    catch (const std::exception& e) { printf("%s\n", e.what()); return -1; }
}

size_t gsc_describe(GoUint8 algorithmID, char* out, size_t cap) {
    if (algorithmID > 2 || !cap) return 0;
    Algorithm* a = lookup(algorithmID);
    std::string s = a ? a->describe() : std::string("not initialised");
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(out, s.data(), n); out[n] = 0; return n;
}
int gsc_last_dominant_kernel(GoUint8 algorithmID, char* name, size_t cap, float* ms, size_t* statements, size_t* columns, size_t* nbases) {
    if (algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    const KernelStat st = a->last_kernel_stat();
    if (name && cap) { const size_t n = strlen(st.name) < cap - 1 ? strlen(st.name) : cap - 1; memcpy(name, st.name, n); name[n] = 0; }
    if (ms) *ms = st.ms;
    if (statements) *statements = st.statements;
    if (columns) *columns = st.columns;
    if (nbases) *nbases = st.nbases;
    return 0;
}
int gsc_last_kernel_clock(GoUint8 algorithmID, float* clock_mhz, int* windows) {
    if (algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    const KernelStat st = a->last_kernel_stat();
    if (clock_mhz) *clock_mhz = st.clock_mhz;
    if (windows) *windows = st.nwin;
    return 0;
}
int gsc_last_stage_ms(GoUint8 algorithmID, float out[4]) {
    if (algorithmID > 2) return -1;
    Algorithm* a = lookup(algorithmID); if (!a) return -1;
    const KernelStat st = a->last_kernel_stat();
    for (int i = 0; i < 4; i++) out[i] = st.stage_ms[i];
    return 0;
}

}  // extern "C"
