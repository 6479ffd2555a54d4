// slam_pool.h — a small persistent worker pool for the per-sequence host work of the tracking driver (the sequences of a
// batch are independent, so every host stage is a parallel_for over them).  Product code.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace oslam_drv {

class Pool {
public:
    explicit Pool(int threads) {
        for (int i = 1; i < threads; i++) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int threads() const { return (int)workers_.size() + 1; }
    // fn(i) for i in [0, n); returns when all calls have finished.  The calling thread takes part.
    template <class F>
    void parallel_for(int n, F&& fn) {
        if (n <= 0) return;
        if (workers_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::function<void(int)> f = std::ref(fn);
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &f; n_ = n; next_.store(0); running_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return running_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            work();
            {
                std::lock_guard<std::mutex> g(m_);
                if (--running_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, running_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

}  // namespace oslam_drv
