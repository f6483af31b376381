// CPU check of csrc/dispatch.hpp with a stub engine (no HIP): concurrent single-statement callers through the batcher policy and
// the replica picker must reach every replica; a replica that holds a long call is skipped; split shares are accounted for.
// Built and run by tests/test_dispatch_policy.py; prints one "key value" line per check and DISPATCH-OK at the end.
#include "../../gnark-symmetric-crypto_amd/csrc/dispatch.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <thread>

using namespace gsc;

#define CHECK(c) do { if (!(c)) { printf("FAILED %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

// the batcher's worker loop (capi.cpp Batcher::run) over a stub engine that "proves" by sleeping
struct StubNode {
    ReplicaPicker picker; size_t max_batch; std::mutex mu; std::condition_variable cv, done; std::deque<int*> q; int idle = 0; bool stop = false;
    std::vector<std::thread> workers; std::vector<std::atomic<int>> batches_on;
    StubNode(size_t replicas, size_t lanes, size_t cap) : picker(replicas), max_batch(cap), batches_on(replicas) {
        for (size_t i = 0; i < replicas * lanes; i++) workers.emplace_back([this] { run(); });
    }
    ~StubNode() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); for (auto& w : workers) w.join(); }
    void prove_batch(size_t n) {            // Algorithm::prove_batch for a call that is not split
        const size_t r = picker.acquire(n);
        batches_on[r]++;
        std::this_thread::sleep_for(std::chrono::microseconds(1000 + 100 * n));      // a device batch: a fixed cost plus a share per statement
        picker.release(r, n);
    }
    void run() {
        for (;;) {
            std::vector<int*> take;
            {
                std::unique_lock<std::mutex> l(mu);
                idle++;
                cv.wait(l, [&] { return stop || !q.empty(); });
                if (stop && q.empty()) return;
                const size_t want = batcher_take(q.size(), (size_t)idle, max_batch);
                idle--;
                while (!q.empty() && take.size() < want) { take.push_back(q.front()); q.pop_front(); }
            }
            if (take.empty()) continue;
            prove_batch(take.size());
            { std::lock_guard<std::mutex> l(mu); for (int* d : take) *d = 1; }
            done.notify_all();
        }
    }
    void submit() { int flag = 0; std::unique_lock<std::mutex> l(mu); q.push_back(&flag); cv.notify_all(); done.wait(l, [&] { return flag != 0; }); }
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
    printf("DISPATCH-OK\n");
    return 0;
}
