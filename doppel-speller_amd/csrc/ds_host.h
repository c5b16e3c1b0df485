// Host-side threading helpers of the index build (ds_build.hip, ds_runtime.hip): plain std::thread fan-outs over
// contiguous or dynamically claimed ranges.  Every caller partitions its OUTPUT so that no two threads write the same
// element; results never depend on the number of threads.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace ds {

// DS_HOST_THREADS, else the CPUs this process may run on (std::thread::hardware_concurrency honours the affinity
// mask on Linux), capped at 32: the passes are memory-bound well before that.
inline int host_threads()
{
    if (const char *text = getenv("DS_HOST_THREADS"); text != nullptr) {
        const int wanted = atoi(text);
        if (wanted >= 1) return wanted > 256 ? 256 : wanted;
    }
    const unsigned available = std::thread::hardware_concurrency();
    return available == 0 ? 1 : static_cast<int>(available > 32 ? 32 : available);
}

// fn(thread, begin, end) over `threads` contiguous, ascending ranges of [0, n): thread t's range lies before thread
// t + 1's (the index build relies on it to keep posting lists ascending).
template <typename F>
void parallel_ranges(int64_t n, int threads, F fn)
{
    if (threads <= 1 || n < 2) {
        fn(0, int64_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(static_cast<size_t>(threads));
    for (int t = 0; t < threads; ++t) {
        const int64_t begin = n * t / threads, end = n * (t + 1) / threads;
        pool.emplace_back([=] { fn(t, begin, end); });
    }
    for (std::thread &thread : pool) thread.join();
}

// fn(thread, begin, end) over chunks of `chunk` elements claimed from a shared counter (uneven work per element).
template <typename F>
void parallel_dynamic(int64_t n, int64_t chunk, int threads, F fn)
{
    if (threads <= 1 || n <= chunk) {
        if (n > 0) fn(0, int64_t(0), n);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    pool.reserve(static_cast<size_t>(threads));
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            for (;;) {
                const int64_t begin = next.fetch_add(chunk, std::memory_order_relaxed);
                if (begin >= n) break;
                fn(t, begin, begin + chunk < n ? begin + chunk : n);
            }
        });
    for (std::thread &thread : pool) thread.join();
}

// First failure reported by any worker of a fan-out (workers cannot `return DS_E_ARG` through DS_REQUIRE).
struct FirstError {
    std::atomic<int> raised{0};
    char text[256] = {0};
    template <typename... Args>
    void raise(const char *format, Args... args)
    {
        int expected = 0;
        if (raised.compare_exchange_strong(expected, 1)) {
            snprintf(text, sizeof(text), format, args...);
            raised.store(2, std::memory_order_release);
        }
    }
    bool failed() const { return raised.load(std::memory_order_acquire) != 0; }
    const char *message() const
    {
        while (raised.load(std::memory_order_acquire) == 1) {}
        return text;
    }
};

}  // namespace ds
