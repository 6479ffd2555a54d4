// slam_pool.h — worker threads for the per-sequence host work of the tracking driver (the sequences of a batch are independent, so every
// host stage is a parallel_for over them).  Product code.
//
// The workers are PROCESS-WIDE: every Pool(threads) adds threads - 1 workers to one shared set, and a parallel_for of any handle is served by
// whichever workers are idle.  With several driver handles per GPU (each stepped by its own host thread) a handle that waits for the GPU leaves
// its share of the cores to the handles that are in a host stage, instead of parking four private workers (measured: see DESIGN.md §9).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace oslam_drv {

class SharedWorkers {
public:
    struct Batch {
        std::function<void(int)>* fn = nullptr;
        int n = 0;
        std::atomic<int> next{0}, finished{0};
        int users = 0;   // workers currently inside work() for this batch (guarded by the pool mutex): the owner may not leave before they have
    };
    static SharedWorkers& instance() {
        static SharedWorkers* s = new SharedWorkers;   // never destroyed: the detached workers may outlive static destructors
        return *s;
    }
    void add_workers(int k) {
        std::lock_guard<std::mutex> g(m_);
        for (int i = 0; i < k; i++) { std::thread([this] { loop(); }).detach(); nthreads_++; }
    }
    int threads() {
        std::lock_guard<std::mutex> g(m_);
        return nthreads_;
    }
    // runs b.fn(i) for i in [0, n) on the caller and on idle workers; returns when all calls have finished
    void run(Batch& b) {
        {
            std::lock_guard<std::mutex> g(m_);
            active_.push_back(&b);
        }
        cv_.notify_all();
        work(b);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return b.finished.load() == b.n && b.users == 0; });
        for (size_t i = 0; i < active_.size(); i++)
            if (active_[i] == &b) { active_.erase(active_.begin() + i); break; }
    }

private:
    void work(Batch& b) {
        for (;;) {
            const int i = b.next.fetch_add(1);
            if (i >= b.n) break;
            (*b.fn)(i);
            if (b.finished.fetch_add(1) + 1 == b.n) {
                std::lock_guard<std::mutex> g(m_);   // the owner checks `finished` under the lock: no lost wake-up
                done_.notify_all();
            }
        }
    }
    void loop() {
        for (;;) {
            Batch* b = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] {
                    for (Batch* q : active_)
                        if (q->next.load() < q->n) { b = q; return true; }
                    return false;
                });
                b->users++;   // under the lock: the owner waits for users == 0 before the batch (on its stack) goes away
            }
            work(*b);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--b->users == 0) done_.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<Batch*> active_;
    int nthreads_ = 0;
};

// parallel_for on the shared workers for library code that has no Pool of its own (runs on the caller alone when no driver handle has added workers)
template <class F>
inline void shared_parallel_for(int n, F&& fn) {
    if (n <= 0) return;
    if (n == 1 || SharedWorkers::instance().threads() == 0) { for (int i = 0; i < n; i++) fn(i); return; }
    std::function<void(int)> f = std::ref(fn);
    SharedWorkers::Batch b;
    b.fn = &f; b.n = n;
    SharedWorkers::instance().run(b);
}

class Pool {
public:
    explicit Pool(int threads) : threads_(threads < 1 ? 1 : threads) {
        if (threads_ > 1) SharedWorkers::instance().add_workers(threads_ - 1);
    }
    int threads() const { return threads_; }
    // fn(i) for i in [0, n); returns when all calls have finished.  The calling thread takes part.
    template <class F>
    void parallel_for(int n, F&& fn) {
        if (n <= 0) return;
        if (threads_ <= 1 || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::function<void(int)> f = std::ref(fn);
        SharedWorkers::Batch b;
        b.fn = &f; b.n = n;
        SharedWorkers::instance().run(b);
    }

private:
    int threads_;
};

}  // namespace oslam_drv
