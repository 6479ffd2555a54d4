// slam_pool.h — worker threads for the per-sequence host work of the tracking driver (the sequences of a batch are independent, so every
// host stage is a parallel_for over them).  Product code.
//
// The workers are PROCESS-WIDE: every Pool(threads) adds threads - 1 workers to one shared set, and a parallel_for of any handle is served by
// whichever workers are idle.  With several driver handles per GPU (each stepped by its own host thread) a handle that waits for the GPU leaves
// its share of the cores to the handles that are in a host stage, instead of parking four private workers (measured: see DESIGN.md §9).
// A step of the driver dispatches ~100 short parallel_for calls, so workers and owners first SPIN for a few tens of microseconds (new work usually
// arrives within that) and only then sleep on the condition variable.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdlib.h>
#include <thread>
#include <time.h>
#include <vector>

namespace oslam_drv {

// CPU-time account of one driver handle: the workers add the thread-CPU time of the tasks they run for a batch to the account the batch's owner has
// set for its thread (the owner's own share is in its thread's CPU clock), so a stage's core-seconds can be told from its wall time.
struct CpuAccount { std::atomic<long long> worker_ns{0}; };
inline CpuAccount*& thread_account() { static thread_local CpuAccount* a = nullptr; return a; }
// Binds an account to the calling thread for a scope and restores the previous binding on every exit path (a handle's account must not stay
// bound to a thread after the handle's step returns: the handle may be destroyed while the thread goes on to use the shared workers).
struct AccountScope {
    CpuAccount* prev;
    explicit AccountScope(CpuAccount* a) : prev(thread_account()) { thread_account() = a; }
    ~AccountScope() { thread_account() = prev; }
    AccountScope(const AccountScope&) = delete;
    AccountScope& operator=(const AccountScope&) = delete;
};
inline long long thread_cpu_ns() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

class SharedWorkers {
public:
    struct Batch {
        std::function<void(int)>* fn = nullptr;
        int n = 0;
        std::atomic<int> next{0}, finished{0};
        CpuAccount* acct = nullptr;  // owner's account (may be null)
        std::atomic<int> users{0};   // workers inside work() for this batch (incremented under the pool mutex): the owner may not leave before they have
    };
    // The workers are kept in kMaxShards independent sets (own mutex, queue and threads): the handles of a process are dealt to the sets round robin
    // (pool_shards() of them are in use), so that the ~100 short parallel_for calls per step of many handles do not all meet on one mutex.
    static constexpr int kMaxShards = 16;
    static SharedWorkers& instance(int shard = 0) {
        static SharedWorkers* s = new SharedWorkers[kMaxShards];   // never destroyed: the detached workers may outlive static destructors
        return s[shard < 0 ? 0 : shard % kMaxShards];
    }
    // A Pool asks for k workers while it lives: the set grows to the largest total ever asked for; workers beyond the CURRENT total park, so a later
    // run with fewer or smaller pools does not inherit the thread count of an earlier one.
    void add_workers(int k) {
        std::lock_guard<std::mutex> g(m_);
        wanted_ += k;
        while (nthreads_ < wanted_) { const int id = nthreads_++; std::thread([this, id] { loop(id); }).detach(); }
        epoch_.fetch_add(1, std::memory_order_release);
        cv_.notify_all();
    }
    void remove_workers(int k) {
        std::lock_guard<std::mutex> g(m_);
        wanted_ -= k;
    }
    int threads() {
        std::lock_guard<std::mutex> g(m_);
        return wanted_;
    }
    // runs b.fn(i) for i in [0, n) on the caller and on idle workers; returns when all calls have finished
    void run(Batch& b) {
        b.acct = thread_account();
        {
            std::lock_guard<std::mutex> g(m_);
            active_.push_back(&b);
            epoch_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        work(b);
        auto finished = [&] { return b.finished.load(std::memory_order_acquire) == b.n && b.users.load(std::memory_order_acquire) == 0; };
        for (int spin = 0, ns = spin_iters(); spin < ns && !finished(); spin++) relax();
        std::unique_lock<std::mutex> lk(m_);
        while (!finished()) done_.wait_for(lk, std::chrono::microseconds(100));
        for (size_t i = 0; i < active_.size(); i++)
            if (active_[i] == &b) { active_.erase(active_.begin() + i); break; }
    }

private:
    // polling iterations before a worker (or an owner waiting for its batch) sleeps: ~30-60 us by default; OSLAM_POOL_SPIN overrides (A/B knob)
    static int spin_iters() { static const int n = [] { const char* e = getenv("OSLAM_POOL_SPIN"); const int v = e ? atoi(e) : 4000; return v < 1 ? 1 : v; }(); return n; }
    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void work(Batch& b) {
        for (;;) {
            const int i = b.next.fetch_add(1);
            if (i >= b.n) break;
            (*b.fn)(i);
            b.finished.fetch_add(1, std::memory_order_release);
        }
    }
    Batch* pick_locked() {
        for (Batch* q : active_)
            if (q->next.load() < q->n) { q->users.fetch_add(1); return q; }   // under the lock: the owner waits for users == 0 before the batch (on its stack) goes away
        return nullptr;
    }
    void loop(int id) {
        for (;;) {
            Batch* b = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (id >= wanted_) {   // parked: more workers exist than the live pools asked for
                    cv_.wait_for(lk, std::chrono::milliseconds(20));
                    continue;
                }
                b = pick_locked();
                if (!b) {
                    const unsigned long seen = epoch_.load(std::memory_order_acquire);
                    lk.unlock();
                    bool changed = false;
                    for (int spin = 0, ns = spin_iters(); spin < ns; spin++) {
                        if (epoch_.load(std::memory_order_acquire) != seen) { changed = true; break; }
                        relax();
                    }
                    lk.lock();
                    b = pick_locked();
                    if (!b && !changed) {
                        sleepers_.fetch_add(1);
                        cv_.wait_for(lk, std::chrono::milliseconds(2), [&] { return epoch_.load(std::memory_order_acquire) != seen; });
                        sleepers_.fetch_sub(1);
                        b = pick_locked();
                    }
                }
            }
            if (!b) continue;
            if (b->acct) {
                const long long t0 = thread_cpu_ns();
                work(*b);
                b->acct->worker_ns.fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed);   // before users drops: the owner (and its account) is still there
            } else work(*b);
            b->users.fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<Batch*> active_;
    std::atomic<unsigned long> epoch_{0};
    std::atomic<int> sleepers_{0};
    int nthreads_ = 0, wanted_ = 0;
};

// number of worker sets in use (OSLAM_POOL_SHARDS, default 1 = one process-wide set) and the set of the handle the calling thread is stepping
inline int pool_shards() {
    static const int n = [] { const char* e = getenv("OSLAM_POOL_SHARDS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > SharedWorkers::kMaxShards ? SharedWorkers::kMaxShards : v); }();
    return n;
}
inline int next_pool_shard() { static std::atomic<int> c{0}; return c.fetch_add(1) % pool_shards(); }
inline int& thread_shard() { static thread_local int s = 0; return s; }
struct ShardScope {
    int prev;
    explicit ShardScope(int s) : prev(thread_shard()) { thread_shard() = s; }
    ~ShardScope() { thread_shard() = prev; }
    ShardScope(const ShardScope&) = delete;
    ShardScope& operator=(const ShardScope&) = delete;
};

// parallel_for on the shared workers for library code that has no Pool of its own (runs on the caller alone when no driver handle has added workers)
template <class F>
inline void shared_parallel_for(int n, F&& fn) {
    if (n <= 0) return;
    SharedWorkers& w = SharedWorkers::instance(thread_shard());
    if (n == 1 || w.threads() == 0) { for (int i = 0; i < n; i++) fn(i); return; }
    std::function<void(int)> f = std::ref(fn);
    SharedWorkers::Batch b;
    b.fn = &f; b.n = n;
    w.run(b);
}

class Pool {
public:
    // own_workers = false: use the shared workers without asking for more (the operator table of a handle runs on the handle's thread, between the
    // driver's own parallel sections: the two never need workers at the same time)
    // shard >= 0: the worker set of this pool; -1: the set of whichever handle the calling thread is stepping (thread_shard())
    explicit Pool(int threads, bool own_workers = true, int shard = -1) : threads_(threads < 1 ? 1 : threads), own_(own_workers), shard_(shard) {
        if (threads_ > 1 && own_) SharedWorkers::instance(shard_).add_workers(threads_ - 1);
    }
    ~Pool() {
        if (threads_ > 1 && own_) SharedWorkers::instance(shard_).remove_workers(threads_ - 1);
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    int threads() const { return threads_; }
    // fn(i) for i in [0, n); returns when all calls have finished.  The calling thread takes part.
    template <class F>
    void parallel_for(int n, F&& fn) {
        if (n <= 0) return;
        if (threads_ <= 1 || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::function<void(int)> f = std::ref(fn);
        SharedWorkers::Batch b;
        b.fn = &f; b.n = n;
        SharedWorkers::instance(shard_ >= 0 ? shard_ : thread_shard()).run(b);
    }

private:
    int threads_;
    bool own_;
    int shard_;
};

}  // namespace oslam_drv
