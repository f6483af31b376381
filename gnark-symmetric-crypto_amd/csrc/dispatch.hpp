// Which engine replica (one per device of GSC_DEVICES) serves a call, and how many queued single-proof callers a batcher worker
// takes at a time.  Host-only policy code with no HIP in it: tests/test_dispatch_policy.py compiles it with g++ against a stub
// engine and checks the spread on CPU; engine.hip (Algorithm::prove_batch) and capi.cpp (Batcher) are the users.
//
// The reference's unit of work is ONE statement per Prove call, from any number of concurrent FFI threads
// (libraries/prover/libprove.go:30-47; concurrent callers: libraries/core_test.go:44-111), so small calls must reach every GPU of the
// node, not only the first one.
#pragma once
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace gsc {

// Least-loaded replica for a call that is not split.  Load = statements in flight on the replica; ties go to the replica that has
// served fewer statements so far (so an idle node is filled round-robin, a replica that was handed a long call is skipped until it is
// done, and uneven batch sizes even out over time).  acquire / release bracket the call.
class ReplicaPicker {
  public:
    struct Counters { uint64_t calls = 0, statements = 0; };
    explicit ReplicaPicker(size_t replicas) : load_(replicas, 0), served_(replicas) {}
    size_t size() const { return load_.size(); }
    size_t acquire(size_t statements) {
        std::lock_guard<std::mutex> l(mu_);
        size_t best = 0;
        for (size_t i = 1; i < load_.size(); i++)
            if (load_[i] < load_[best] || (load_[i] == load_[best] && served_[i].statements < served_[best].statements)) best = i;
        take(best, statements);
        return best;
    }
    // a share of a split call goes to a fixed replica (contiguous shares in device order): accounted for, not chosen
    void acquire_on(size_t replica, size_t statements) { std::lock_guard<std::mutex> l(mu_); take(replica, statements); }
    void release(size_t replica, size_t statements) { std::lock_guard<std::mutex> l(mu_); load_[replica] -= statements < load_[replica] ? statements : load_[replica]; }
    std::vector<Counters> served() const { std::lock_guard<std::mutex> l(mu_); return served_; }
  private:
    void take(size_t i, size_t statements) { load_[i] += statements; served_[i].calls++; served_[i].statements += statements; }
    mutable std::mutex mu_;
    std::vector<size_t> load_;
    std::vector<Counters> served_;
};

// How many of `queued` single-proof callers a batcher worker takes when `idle_workers` workers (this one included) have nothing on a
// device: the queue is shared out over the idle workers, so that a burst of callers spreads over every replica and lane instead of
// riding one device batch; a worker that is alone takes everything (one big batch is the most efficient use of a busy node).
inline size_t batcher_take(size_t queued, size_t idle_workers, size_t max_batch) {
    if (!queued) return 0;
    if (idle_workers < 1) idle_workers = 1;
    size_t n = (queued + idle_workers - 1) / idle_workers;
    if (n > max_batch) n = max_batch;
    return n ? n : 1;
}

}  // namespace gsc
