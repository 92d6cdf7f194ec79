// Fork-join over a fixed set of host threads: ns_batch_prepare's phases (csrc/ns_api.hip) and the host facade's query
// preparation (host/engine.cpp).  Plain C++, no device code.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Fork-join over a fixed set of host threads (ns_batch_prepare's phases).  run(n, fn) calls fn(0..n-1), task i on
// worker i (the calling thread takes task 0), and returns when all are done.  Workers sleep between batches.
class ForkJoin {
public:
    explicit ForkJoin(unsigned width) : width_(std::max(1u, width)) {
        for (unsigned i = 1; i < width_; i++) workers_.emplace_back([this, i]() { loop(i); });
    }
    ~ForkJoin() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
    unsigned width() const { return width_; }
    void run(unsigned n, const std::function<void(unsigned)>& fn) {
        n = std::min(n, width_);
        if (n <= 1) { if (n) fn(0); return; }
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; pending_ = n - 1; gen_++; }
        wake_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void loop(unsigned me) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> l(m_);
                wake_.wait(l, [&]() { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (me < n_) fn = fn_;
            }
            if (fn) {
                (*fn)(me);
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    unsigned width_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned n_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};
