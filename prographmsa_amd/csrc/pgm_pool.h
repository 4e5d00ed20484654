// pgm_pool.h — a small set of persistent host threads for the per-level host work (flattening, staging copies, plans,
// merges).  A guide-tree level issues half a dozen short parallel sections of 0.5-3 ms each; starting and joining 16
// threads for every one of them cost about as much as the sections themselves (0.3-0.5 ms per section).
// Header only: the library (csrc/) and the host mirror (host/) each keep one pool.
#ifndef PGM_POOL_H_
#define PGM_POOL_H_

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pgm_pool {

class Pool {
public:
    explicit Pool(unsigned nthreads) {   // the caller of run() is one of the threads
        for (unsigned t = 0; t + 1 < nthreads; ++t) workers_.emplace_back([this]() { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }

    // fn(i) for every i in [0, n) on at most maxthreads threads, indices handed out one at a time; returns the text of the
    // first exception ("" if none).  A call from inside fn, or while another thread's call is running, runs on the caller alone.
    std::string run(size_t n, unsigned maxthreads, const std::function<void(size_t)> &fn) {
        if (n == 0) return std::string();
        const unsigned helpers = (unsigned)std::min<size_t>(std::min<size_t>(maxthreads, n) - 1, workers_.size());
        std::unique_lock<std::mutex> own(run_mu_, std::defer_lock);
        if (helpers == 0 || in_worker() || !own.try_lock()) {
            std::string err;
            for (size_t i = 0; i < n; ++i) {
                try { fn(i); }
                catch (std::exception &e) { if (err.empty()) err = e.what(); }
            }
            return err;
        }
        // The section's counters live in an object of their own: a worker that wakes up late works on (and finds nothing
        // left in) the section it was woken for, never on the counters of a later one; the caller waits for the indices to
        // be done, not for the workers.
        auto sec = std::make_shared<Section>();
        sec->fn = &fn; sec->n = n; sec->helpers = helpers;
        { std::lock_guard<std::mutex> g(mu_); cur_ = sec; ++gen_; }
        for (unsigned h = 0; h < helpers; ++h) cv_.notify_one();
        work(*sec);
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&]() { return sec->done.load() == sec->n; });
        return sec->err;
    }

private:
    struct Section {
        const std::function<void(size_t)> *fn = nullptr;
        size_t n = 0;
        unsigned helpers = 0;
        std::atomic<size_t> next{0}, done{0};
        std::atomic<unsigned> joined{0};
        std::string err;
    };
    static bool &in_worker() { static thread_local bool f = false; return f; }
    void work(Section &s) {
        for (size_t i; (i = s.next.fetch_add(1)) < s.n;) {
            try { (*s.fn)(i); }
            catch (std::exception &e) { std::lock_guard<std::mutex> g(mu_); if (s.err.empty()) s.err = e.what(); }
            if (s.done.fetch_add(1) + 1 == s.n) { std::lock_guard<std::mutex> g(mu_); done_cv_.notify_all(); }
        }
    }
    void loop() {
        in_worker() = true;
        unsigned long seen = 0;
        for (;;) {
            std::shared_ptr<Section> sec;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&]() { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                sec = cur_;
            }
            if (sec && sec->joined.fetch_add(1) < sec->helpers) work(*sec);
        }
    }

    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    unsigned long gen_ = 0;
    bool stop_ = false;
    std::shared_ptr<Section> cur_;
};

}  // namespace pgm_pool

#endif
