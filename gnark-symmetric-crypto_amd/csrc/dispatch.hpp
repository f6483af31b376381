// Which engine replica (one per device of GSC_DEVICES) serves a call, and when a batcher worker takes how many of the queued
// single-proof callers.  Host-only policy code with no HIP in it: tests/test_dispatch_policy.py compiles it with g++ against a stub
// engine and checks the spread on CPU; engine.hip (Algorithm::prove_batch) and capi.cpp (Batcher) are the users.
//
// The reference's unit of work is ONE statement per Prove call, from any number of concurrent FFI threads
// (libraries/prover/libprove.go:30-47; concurrent callers: libraries/core_test.go:44-111), so small calls must reach every GPU of the
// node, not only the first one.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
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

// How a call of n statements is spread over `replicas` engine replicas (Algorithm::prove_batch):
//   * one replica, a call of up to one 64-column batch, or a batch the micro-batcher took for ONE device (whole = true) that fits a
//     replica's capacity: ONE share, sent whole to the least-loaded replica (pick = true) — the scheduler already shared the queue out
//     per free device, so splitting such a batch again would put several small device batches where one was meant;
//   * anything else: contiguous shares of ceil(n / replicas) statements rounded up to whole 64-column batches, replica d takes
//     [d * share, ...) — the last share may be ragged, replicas beyond the end of the call get nothing; every statement is in exactly one share.
struct CallShare { size_t replica, off, n; bool pick; };
inline std::vector<CallShare> plan_shares(size_t n, size_t replicas, size_t replica_cap, bool whole) {
    std::vector<CallShare> out;
    if (!n) return out;
    if (replicas <= 1 || n <= 64 || (whole && n <= replica_cap)) { out.push_back(CallShare{0, 0, n, true}); return out; }
    const size_t share = ((n + replicas - 1) / replicas + 63) / 64 * 64;
    for (size_t d = 0; d < replicas && d * share < n; d++) out.push_back(CallShare{d, d * share, n - d * share < share ? n - d * share : share, false});
    return out;
}

// How many of `queued` single-proof callers a batcher worker takes when `free_devices` devices (at least one) have no batch of the
// algorithm on them: the queue is shared out over the free devices, so that a burst of callers spreads over every replica; on one
// device everything goes out as ONE batch (the device time per statement falls with the batch: 2.65 ms for one, 0.40 ms at 16,
// 0.20 ms at 64, 0.08 ms at 256 — two overlapped small batches never beat one of twice the size).
inline size_t batcher_take(size_t queued, size_t free_devices, size_t max_batch) {
    if (!queued) return 0;
    if (free_devices < 1) free_devices = 1;
    size_t n = (queued + free_devices - 1) / free_devices;
    if (n > max_batch) n = max_batch;
    return n ? n : 1;
}

// When a batcher worker takes a batch.  Concurrent single-statement callers (libraries/core_test.go:44-111: one Prove per goroutine,
// the next one when the previous has returned) are a closed loop: the callers of a batch that has just completed come back within
// microseconds, and whatever is taken while they are on their way rides a smaller, dearer batch.  Rules, with the queue lock held:
//   * a full lane (queued >= max_batch) goes at once;
//   * every device has a batch of this algorithm on it: wait for a completion (the callers that queue meanwhile form the next batch) —
//     unless the queue holds at least SECOND_BATCH_MIN callers and at least as many as the batch started last: then a second batch
//     per device is worth its overlap (lanes exist for that) — or its oldest caller has waited BUSY_WAIT_MAX_US (a lone caller that
//     arrives under a batch of a thousand is not held for all of it);
//   * a device is free and a batch of k callers completed less than a moment ago: wait until k callers have arrived since, for as long
//     as they keep arriving (gap: linger / 2, in all at most RETURN_MAX_US) — a lone caller's own return satisfies k = 1 at once;
//   * a device is free, several callers are queued and nothing is known about them (a burst out of nowhere): the classic linger window, once;
//   * otherwise (a lone caller, an idle device): go.
// linger_us = 0 switches both waits off; waiting for a busy device is scheduling, not lingering, and stays.
// The class holds only the bookkeeping; the queue, its mutex and the items are the user's (capi.cpp Batcher, tests/native/dispatch_check.cpp).
class BatchScheduler {
  public:
    using Clock = std::chrono::steady_clock;
    static constexpr size_t SECOND_BATCH_MIN = 64;
    static constexpr int RETURN_MAX_US = 2000;
    static constexpr int BUSY_WAIT_MAX_US = 20000;
    BatchScheduler(size_t devices, size_t max_batch, int linger_us) : devices_(devices ? devices : 1), max_batch_(max_batch ? max_batch : 1), linger_us_(linger_us < 0 ? 0 : linger_us) {}
    std::condition_variable cv;             // workers wait here; notified on every arrival, completion and stop
    // caller side, lock held, after the item was queued
    // (n statements of ONE call: a ProveBatch / gsc_prove_raw call of several statements comes back as one caller, not as n)
    void arrived(size_t n = 1) {
        events_++; arrivals_since_done_ += 1;
        const Clock::time_point now = Clock::now();
        for (size_t i = 0; i < n; i++) queued_at_.push_back(now);
        if (now < return_hard_) { return_soft_ = now + std::chrono::microseconds(gap_us()); if (return_soft_ > return_hard_) return_soft_ = return_hard_; }
        cv.notify_all();
    }
    // worker side, lock held: blocks until this worker should take a batch; returns its size (0: `stop` was raised and the queue is empty)
    template <class Queued>
    size_t wait_for_batch(std::unique_lock<std::mutex>& l, Queued queued, const bool& stop) {
        bool lingered = false;
        for (;;) {
            cv.wait(l, [&] { return stop || queued() != 0; });
            if (queued() == 0) return 0;                                      // stop
            const size_t q = queued();
            if (q >= max_batch_ || stop) break;
            if (in_flight_ >= devices_) {
                if (in_flight_ < 2 * devices_) {
                    if (q >= SECOND_BATCH_MIN && q >= last_started_) break;
                    const Clock::time_point limit = (queued_at_.empty() ? Clock::now() : queued_at_.front()) + std::chrono::microseconds(BUSY_WAIT_MAX_US);
                    if (Clock::now() >= limit) break;
                    const uint64_t seen = events_;
                    cv.wait_until(l, limit, [&] { return stop || events_ != seen; });
                    continue;
                }
                const uint64_t seen = events_;
                cv.wait(l, [&] { return stop || events_ != seen; });
                continue;
            }
            if (linger_us_ > 0) {
                const Clock::time_point now = Clock::now();
                if (now < return_soft_ && arrivals_since_done_ < returning_) { timed_waits_++; cv.wait_until(l, return_soft_); continue; }      // re-evaluated on every arrival
                if (q > 1 && !lingered && returning_known(now) == false) {
                    lingered = true; timed_waits_++;
                    cv.wait_for(l, std::chrono::microseconds(linger_us_), [&] { return stop || queued() >= max_batch_; });
                    continue;
                }
            }
            break;
        }
        const size_t free_devices = in_flight_ < devices_ ? devices_ - in_flight_ : 1;
        return batcher_take(queued(), free_devices, max_batch_);
    }
    void started(size_t n) {                                                   // lock held, the batch (the n oldest callers) has left the queue
        in_flight_++; last_started_ = n;
        for (size_t i = 0; i < n && !queued_at_.empty(); i++) queued_at_.pop_front();
    }
    // lock held, before the callers are woken.  n statements of `calls` distinct calls (0: every statement its own caller): the callers that can
    // come back are the CALLS — waiting for as many arrivals as the batch had statements would hold an idle device (up to the gap, re-armed by
    // every arrival) whenever a call of several statements is followed by lone single-statement callers.
    void completed(size_t n, size_t calls = 0) {
        if (in_flight_) in_flight_--;
        events_++; returning_ = calls ? calls : n; arrivals_since_done_ = 0;
        const Clock::time_point now = Clock::now();
        return_soft_ = now + std::chrono::microseconds(gap_us()); return_hard_ = now + std::chrono::microseconds(linger_us_ ? RETURN_MAX_US : 0);
        if (return_soft_ > return_hard_) return_soft_ = return_hard_;
        cv.notify_all();
    }
    size_t in_flight() const { return in_flight_; }
    uint64_t timed_waits() const { return timed_waits_; }     // how often a worker waited on an IDLE device (linger / returning callers): 0 for a lone caller
  private:
    int gap_us() const { return linger_us_ / 2; }
    // the queued callers are (part of) a batch that completed a moment ago: they were waited for already
    bool returning_known(Clock::time_point now) const { return now < return_hard_; }
    size_t devices_, max_batch_; int linger_us_;
    size_t in_flight_ = 0, last_started_ = 0, returning_ = 0, arrivals_since_done_ = 0; uint64_t events_ = 0, timed_waits_ = 0;
    Clock::time_point return_soft_{}, return_hard_{};
    std::deque<Clock::time_point> queued_at_;      // arrival times of the callers still queued, oldest first
};

}  // namespace gsc
