// CPU check of csrc/dispatch.hpp with a stub engine (no HIP): concurrent single-statement callers through the batcher policy and
// the replica picker must reach every replica; a replica that holds a long call is skipped; split shares are accounted for.
// Built and run by tests/test_dispatch_policy.py; prints one "key value" line per check and DISPATCH-OK at the end.
#include "../../gnark-symmetric-crypto_amd/csrc/dispatch.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <set>
#include <thread>

using namespace gsc;

#define CHECK(c) do { if (!(c)) { printf("FAILED %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

// capi.cpp's Batcher over a stub engine that "proves" by sleeping: the same BatchScheduler decides when a worker takes how many callers
struct StubNode {
    ReplicaPicker picker; BatchScheduler sched; std::mutex mu; std::condition_variable done; std::deque<int*> q; bool stop = false;
    std::vector<std::thread> workers; std::vector<std::atomic<int>> batches_on; std::vector<size_t> batch_sizes;      // batch_sizes: under mu
    int fixed_us, per_statement_us; size_t cap_; std::atomic<size_t> split_batches{0};
    StubNode(size_t replicas, size_t lanes, size_t cap, int linger_us = 300, int fixed = 1000, int per = 100)
        : picker(replicas), sched(replicas, cap, linger_us), batches_on(replicas), fixed_us(fixed), per_statement_us(per), cap_(cap) {
        for (size_t i = 0; i < replicas * lanes; i++) workers.emplace_back([this] { run(); });
    }
    ~StubNode() { { std::lock_guard<std::mutex> l(mu); stop = true; } sched.cv.notify_all(); for (auto& w : workers) w.join(); }
    void prove_batch(size_t n) {            // Algorithm::prove_batch(..., whole = true): a batcher share goes to ONE replica
        const auto shares = plan_shares(n, picker.size(), cap_, true);
        if (shares.size() != 1 || !shares[0].pick) split_batches++;
        const size_t r = picker.acquire(n);
        batches_on[r]++;
        std::this_thread::sleep_for(std::chrono::microseconds(fixed_us + per_statement_us * (int)n));      // a device batch: a fixed cost plus a share per statement
        picker.release(r, n);
    }
    void run() {
        for (;;) {
            std::vector<int*> take;
            {
                std::unique_lock<std::mutex> l(mu);
                const size_t want = sched.wait_for_batch(l, [&] { return q.size(); }, stop);
                if (!want) return;
                while (!q.empty() && take.size() < want) { take.push_back(q.front()); q.pop_front(); }
                sched.started(take.size()); batch_sizes.push_back(take.size());
            }
            prove_batch(take.size());
            {
                std::lock_guard<std::mutex> l(mu);
                std::set<size_t> calls;           // distinct calls in the batch (capi.cpp Batcher::run counts them the same way)
                for (int* d : take) calls.insert(call_of[d]);
                for (int* d : take) { call_of.erase(d); *d = 1; }
                sched.completed(take.size(), calls.size());
            }
            done.notify_all();
        }
    }
    void submit() { int flag = 0; std::unique_lock<std::mutex> l(mu); q.push_back(&flag); call_of[&flag] = next_call++; sched.arrived(); done.wait(l, [&] { return flag != 0; }); }
    // Batcher::submit_many: n statements of ONE call
    void submit_many(size_t n) {
        std::vector<int> flags(n, 0); std::unique_lock<std::mutex> l(mu);
        const size_t id = next_call++;
        for (auto& f : flags) { q.push_back(&f); call_of[&f] = id; }
        sched.arrived(n);
        done.wait(l, [&] { for (int f : flags) if (!f) return false; return true; });
    }
    std::map<int*, size_t> call_of; size_t next_call = 0;      // the call every queued statement belongs to (under mu)
};

int main() {
    {   // idle node: calls go round-robin
        ReplicaPicker p(4);
        size_t got[4];
        for (int i = 0; i < 4; i++) got[i] = p.acquire(1);
        CHECK(got[0] == 0 && got[1] == 1 && got[2] == 2 && got[3] == 3);
        for (int i = 0; i < 4; i++) p.release(got[i], 1);
        // a replica with a long call in flight is skipped until it is released
        const size_t big = p.acquire(64);
        for (int i = 0; i < 6; i++) { const size_t r = p.acquire(1); CHECK(r != big); p.release(r, 1); }
        p.release(big, 64);
        const auto sv = p.served();
        uint64_t calls = 0, st = 0; for (auto& c : sv) { calls += c.calls; st += c.statements; }
        CHECK(calls == 11 && st == 4 + 64 + 6);
        // shares of a split call are accounted on the replicas they were sent to
        p.acquire_on(2, 128); CHECK(p.acquire(1) != 2); p.release(2, 128);
        printf("picker ok\n");
    }
    {   // how Algorithm::prove_batch spreads a call over 8 replicas (plan_shares): contiguous shares, whole 64-column batches but for the
        // last one, every statement in exactly one share; calls of up to one batch — and batches the micro-batcher took for one device — stay whole
        const size_t cap = 8192;
        for (size_t n : {(size_t)1, (size_t)64, (size_t)65, (size_t)4 * 64 + 37, (size_t)8 * 8192, (size_t)8 * 8192 - 37, (size_t)8191, (size_t)513}) {
            const auto sh = plan_shares(n, 8, cap, false);
            CHECK(!sh.empty() && sh.size() <= 8);
            if (n <= 64) { CHECK(sh.size() == 1 && sh[0].pick && sh[0].off == 0 && sh[0].n == n); continue; }
            size_t at = 0;
            for (size_t k = 0; k < sh.size(); k++) {
                CHECK(!sh[k].pick && sh[k].replica == k && sh[k].off == at && sh[k].n > 0 && sh[k].n <= cap);
                if (k + 1 < sh.size()) CHECK(sh[k].n % 64 == 0);
                at += sh[k].n;
            }
            CHECK(at == n);
            std::vector<int> written(n, 0);
            for (auto& x : sh) for (size_t i = x.off; i < x.off + x.n; i++) written[i]++;
            for (int w : written) CHECK(w == 1);
        }
        CHECK(plan_shares(65, 8, cap, false).size() == 2 && plan_shares(65, 8, cap, false)[1].n == 1);
        CHECK(plan_shares(8 * 8192, 8, cap, false).size() == 8 && plan_shares(8 * 8192, 8, cap, false)[7].n == 8192);
        CHECK(plan_shares(0, 8, cap, false).empty() && plan_shares(70000, 1, cap, false).size() == 1);
        // a batch the micro-batcher took for ONE device is not split again (256 closed-loop callers on 2 GPUs: two batches of 128, not four of 64)
        CHECK(plan_shares(128, 2, 1024, true).size() == 1 && plan_shares(128, 2, 1024, true)[0].pick);
        CHECK(plan_shares(1024, 2, 1024, true).size() == 1 && plan_shares(1025, 2, 1024, true).size() == 2 && plan_shares(128, 2, 1024, false).size() == 2);
        printf("shares ok\n");
    }
    CHECK(batcher_take(0, 3, 64) == 0 && batcher_take(1, 1, 64) == 1 && batcher_take(64, 2, 1024) == 32 && batcher_take(63, 1, 1024) == 63 &&
          batcher_take(5000, 1, 1024) == 1024 && batcher_take(3, 8, 64) == 1 && batcher_take(7, 0, 64) == 7);
    printf("take ok\n");
    {   // 64 concurrent single-statement callers on a two-replica node: both replicas serve, nobody is lost
        StubNode node(2, 1, 1024);
        std::vector<std::thread> callers;
        for (int i = 0; i < 64; i++) callers.emplace_back([&] { node.submit(); });
        for (auto& t : callers) t.join();
        const auto sv = node.picker.served();
        printf("two replicas: calls %llu/%llu statements %llu/%llu\n", (unsigned long long)sv[0].calls, (unsigned long long)sv[1].calls, (unsigned long long)sv[0].statements, (unsigned long long)sv[1].statements);
        CHECK(sv[0].statements + sv[1].statements == 64 && sv[0].calls > 0 && sv[1].calls > 0);
        CHECK(node.split_batches.load() == 0);      // every share the scheduler took went to one replica
    }
    {   // 256 closed-loop callers on two replicas: batches of more than 64 callers still go whole to one replica each
        StubNode node(2, 1, 1024, 300, 2000, 20);
        std::vector<std::thread> callers;
        for (int i = 0; i < 256; i++) callers.emplace_back([&] { for (int k = 0; k < 4; k++) node.submit(); });
        for (auto& t : callers) t.join();
        size_t big = 0; for (size_t b : node.batch_sizes) big += b > 64;
        const auto sv = node.picker.served();
        printf("256 callers on two replicas: %zu batches, %zu of them above 64 callers, calls %llu/%llu\n", node.batch_sizes.size(), big, (unsigned long long)sv[0].calls, (unsigned long long)sv[1].calls);
        CHECK(node.split_batches.load() == 0 && big > 0 && sv[0].calls + sv[1].calls == node.batch_sizes.size());
    }
    {   // sustained load on 8 replicas x 2 lanes: every replica serves, the spread is even within a factor of three
        StubNode node(8, 2, 64);
        std::vector<std::thread> callers;
        for (int i = 0; i < 48; i++) callers.emplace_back([&] { for (int k = 0; k < 20; k++) node.submit(); });
        for (auto& t : callers) t.join();
        const auto sv = node.picker.served();
        uint64_t lo = ~0ull, hi = 0, total = 0;
        for (auto& c : sv) { lo = c.statements < lo ? c.statements : lo; hi = c.statements > hi ? c.statements : hi; total += c.statements; }
        printf("eight replicas: statements min %llu max %llu total %llu\n", (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)total);
        CHECK(total == 48 * 20 && lo > 0 && hi <= 3 * lo + 8);
    }
    {   // a call of 32 statements followed by lone single-statement callers (a host that alternates ProveBatch with Prove): the 32 statements come
        // back as ONE caller, so a lone Prove that arrives after them is not held on the idle device waiting for 31 more arrivals
        StubNode node(1, 3, 1024, 300, 500, 10);
        node.submit_many(32);
        const uint64_t waits0 = node.sched.timed_waits();
        for (int k = 0; k < 5; k++) { node.submit(); node.submit_many(32); }
        printf("mixed small calls and single callers: %llu timed waits on an idle device\n", (unsigned long long)(node.sched.timed_waits() - waits0));
        CHECK(node.sched.timed_waits() == waits0);
        size_t total = 0; for (size_t b : node.batch_sizes) total += b;
        CHECK(total == 32 + 5 * 33);
    }
    {   // closed loop on ONE device (libraries/core_test.go:44-111: every caller issues its next Prove when the previous one has returned):
        // after the first round the callers of a completed batch are waited for and ride ONE batch again — not several small ones
        StubNode node(1, 3, 1024, 300, 2000, 50);
        const int C = 16, rounds = 12;
        std::vector<std::thread> callers;
        for (int i = 0; i < C; i++) callers.emplace_back([&] { for (int k = 0; k < rounds; k++) node.submit(); });
        for (auto& t : callers) t.join();
        size_t total = 0, full = 0;
        for (size_t b : node.batch_sizes) { total += b; if (b >= (size_t)C - 2) full++; }
        printf("closed loop, 16 callers: %zu batches for %zu statements, %zu of them with >= 14 callers\n", node.batch_sizes.size(), total, full);
        CHECK(total == (size_t)C * rounds && node.batch_sizes.size() <= 2 * (size_t)rounds && full >= (size_t)rounds / 2);
    }
    {   // while the device is busy the callers that arrive wait and form ONE next batch
        StubNode node(1, 3, 1024, 300, 150000, 0);
        std::thread first([&] { node.submit(); });
        for (;;) { { std::lock_guard<std::mutex> l(node.mu); if (node.sched.in_flight() == 1) break; } std::this_thread::sleep_for(std::chrono::microseconds(200)); }
        std::vector<std::thread> callers;
        for (int i = 0; i < 10; i++) { callers.emplace_back([&] { node.submit(); }); std::this_thread::sleep_for(std::chrono::microseconds(700)); }
        first.join(); for (auto& t : callers) t.join();
        printf("busy device: batches"); for (size_t b : node.batch_sizes) printf(" %zu", b); printf("\n");
        CHECK(node.batch_sizes.size() == 2 && node.batch_sizes[0] == 1 && node.batch_sizes[1] == 10);
    }
    {   // ... unless enough of them are queued to be worth a second batch on another lane (64, and no fewer than the batch started last)
        StubNode node(1, 3, 1024, 300, 150000, 0);
        std::thread first([&] { node.submit(); });
        for (;;) { { std::lock_guard<std::mutex> l(node.mu); if (node.sched.in_flight() == 1) break; } std::this_thread::sleep_for(std::chrono::microseconds(200)); }
        std::vector<std::thread> callers;
        for (int i = 0; i < 80; i++) callers.emplace_back([&] { node.submit(); });
        first.join(); for (auto& t : callers) t.join();
        size_t total = 0; for (size_t b : node.batch_sizes) total += b;
        printf("busy device, 80 more callers: batches"); for (size_t b : node.batch_sizes) printf(" %zu", b); printf("\n");
        CHECK(total == 81 && node.batch_sizes.size() >= 2 && node.batch_sizes.size() <= 3 && node.batch_sizes[0] == 1 && node.batch_sizes[1] >= 64);
    }
    {   // ... or the oldest of them has waited 20 ms: a caller that arrives under a long batch is not held for all of it
        StubNode node(1, 3, 1024, 300, 400000, 0);
        std::thread first([&] { node.submit(); });
        for (;;) { { std::lock_guard<std::mutex> l(node.mu); if (node.sched.in_flight() == 1) break; } std::this_thread::sleep_for(std::chrono::microseconds(200)); }
        const auto t0 = std::chrono::steady_clock::now();
        double waited_ms = 0;
        std::thread late([&] { node.submit(); waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); });
        late.join();
        first.join();
        printf("late caller under a 400 ms batch: done after %.0f ms (20 ms of waiting + its own 400 ms batch; behind the first batch it would be 800)\n", waited_ms);
        CHECK(node.batch_sizes.size() == 2 && node.batch_sizes[1] == 1 && waited_ms >= 400 + 15 && waited_ms < 400 + 250);
    }
    {   // a lone caller on an idle device is not kept waiting: no worker ever waits on the idle device for it
        StubNode node(1, 3, 1024, 300, 2000, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 50; k++) node.submit();
        const double per_call_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 50;
        uint64_t waits; { std::lock_guard<std::mutex> l(node.mu); waits = node.sched.timed_waits(); }
        printf("lone caller: %.0f us per call (stub batch: 2000 us), %llu waits on the idle device\n", per_call_us, (unsigned long long)waits);
        CHECK(node.batch_sizes.size() == 50 && waits == 0);
    }
    {   // two callers in a closed loop pair up instead of alternating
        StubNode node(1, 3, 1024, 300, 2000, 0);
        std::thread a([&] { for (int k = 0; k < 30; k++) node.submit(); }), b([&] { for (int k = 0; k < 30; k++) node.submit(); });
        a.join(); b.join();
        size_t pairs = 0; for (size_t x : node.batch_sizes) pairs += x == 2;
        printf("two callers: %zu batches, %zu pairs\n", node.batch_sizes.size(), pairs);
        CHECK(pairs >= 15);      // (a caller that the OS keeps off the CPU for longer than the 150 us gap misses its partner)
    }
    printf("DISPATCH-OK\n");
    return 0;
}
