// ctgn_hostpool.hpp — helper threads for the host side of scan-sized calls (plain C++, no HIP: also compiled by the sanitizer tests).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ctgn {

// Helper threads for the HOST side of the calls that move a whole scan (ctgn_transform_points on host views, the full-scan output of
// ctgn_frame_register): gathering the caller's strided records into pinned memory and handing results back are memory-bound loops
// over megabytes that one core runs at 10-20 GB/s — a few cores next to it keep up with the PCIe copies they feed. The threads are
// created on first use, sleep on a condition variable between calls and are joined with the handle. The caller's thread always works too.
struct HostPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    const std::function<void(size_t)> *job = nullptr;
    size_t parts = 0, finished = 0;
    std::atomic<size_t> next{0};
    int inside = 0;                      // helpers currently holding `job`
    uint64_t generation = 0;
    bool stop = false;

    void ensure(int helpers) {
        while ((int) threads.size() < helpers) threads.emplace_back([this] { work(); });
    }
    void work() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_job.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            const std::function<void(size_t)> *f = job;
            const size_t n = parts;
            if (!f) continue;                // woke up after that run had already finished (its caller did the work): nothing to join
            ++inside;
            lk.unlock();
            size_t mine = 0;
            for (size_t i; (i = next.fetch_add(1)) < n; ++mine) (*f)(i);
            lk.lock();
            --inside;
            finished += mine;
            if (finished == parts && inside == 0) cv_done.notify_one();
        }
    }
    // f(0) .. f(parts - 1), each exactly once, on the helpers and the calling thread; returns when all are done
    void run(size_t n, const std::function<void(size_t)> &f) {
        if (threads.empty() || n <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
        {
            // a helper holds (job, parts) from the moment it has read them under the lock until it leaves (`inside`): run() does not return
            // before that, so nothing is published while a helper of an earlier run could still claim an index
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return inside == 0; });
            job = &f; parts = n; finished = 0; next.store(0); ++generation;
        }
        cv_job.notify_all();
        size_t mine = 0;
        for (size_t i; (i = next.fetch_add(1)) < n; ++mine) f(i);
        std::unique_lock<std::mutex> lk(m);
        finished += mine;
        cv_done.wait(lk, [&] { return finished == parts && inside == 0; });
        job = nullptr;
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
    }
};

}  // namespace ctgn
