// How fast can a 720p frame (2.76 MB) go from pageable memory into a pinned ring slot?
// Variants: threads 1..8, plain memcpy against non-temporal stores.  Host-only measurement.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

static void copy_nt(uint8_t* d, const uint8_t* s, size_t n) {
    // d 64-byte aligned (chunks are multiples of 4096 from an aligned base)
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
        _mm256_stream_si256((__m256i*)(d + i), a);
        _mm256_stream_si256((__m256i*)(d + i + 32), b);
        _mm256_stream_si256((__m256i*)(d + i + 64), c);
        _mm256_stream_si256((__m256i*)(d + i + 96), e);
    }
    if (i < n) memcpy(d + i, s + i, n - i);
    _mm_sfence();
}

struct Pool {
    int workers;
    bool nt;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable go, done;
    const uint8_t* src = nullptr;
    uint8_t* dst = nullptr;
    size_t chunk = 0, bytes = 0;
    unsigned gen = 0;
    int pending = 0;
    bool stop = false;
    void part(int id) {
        const size_t lo = (size_t)id * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
        if (lo >= hi) return;
        if (nt)
            copy_nt(dst + lo, src + lo, hi - lo);
        else
            memcpy(dst + lo, src + lo, hi - lo);
    }
    void run(int id) {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                go.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            part(id + 1);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) done.notify_one();
            }
        }
    }
    Pool(int w, bool nt_) : workers(w), nt(nt_) {
        for (int i = 0; i < w; i++) th.emplace_back([this, i] { run(i); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        go.notify_all();
        for (auto& t : th) t.join();
    }
    void copy(uint8_t* d, const uint8_t* s, size_t n) {
        if (workers) {
            std::lock_guard<std::mutex> lk(m);
            src = s, dst = d, bytes = n;
            chunk = (n / (workers + 1) + 4095) & ~(size_t)4095;
            pending = workers;
            gen++;
        } else {
            src = s, dst = d, bytes = n, chunk = n;
        }
        if (workers) go.notify_all();
        part(0);
        if (workers) {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending == 0; });
        }
    }
};

int main() {
    const size_t fb = (size_t)720 * 1280 * 3;
    const int slots = 192, nsrc = 96;
    uint8_t* ring = nullptr;
    if (hipHostMalloc((void**)&ring, fb * slots, hipHostMallocDefault) != hipSuccess) return 1;
    memset(ring, 0, fb * slots);
    uint8_t* src = (uint8_t*)malloc(fb * nsrc + 64);
    for (size_t i = 0; i < fb * nsrc; i++) src[i] = (uint8_t)(i * 2654435761u >> 24);
    for (int nt = 0; nt < 2; nt++)
        for (int w : {0, 1, 3, 5, 7, 11}) {
            Pool p(w, nt != 0);
            for (int i = 0; i < 64; i++) p.copy(ring + (size_t)(i % slots) * fb, src + (size_t)(i % nsrc) * fb + 16, fb);
            const int n = 1500;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) p.copy(ring + (size_t)(i % slots) * fb, src + (size_t)(i % nsrc) * fb + 16, fb);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            printf("%s threads %2d: %6.1f us/frame  %5.1f GB/s\n", nt ? "stream" : "memcpy", w + 1, us, fb / us / 1e3);
            fflush(stdout);
        }
    return 0;
}
