// Frame ingest, the step before the stitching path (reference main.py:1596-1634: cap.read()
// hands every frame to process_frame as a host array).  A ring of pinned host slots and
// device slots: the caller's decoder writes a frame into a host slot, rtvm_ingest_submit
// copies it to the device on a stream of its own, and whatever stream later reads the frame
// (detection, warp) waits for the copy, not the host.
#include "common.h"

#include <emmintrin.h>

#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// memcpy into a pinned slot with non-temporal stores.  The slot is written once and next read by the copy engine:
// an ordinary store first fetches every destination line into the cache (twice the memory traffic, and the frame
// pushes the caller's data out of the cache for nothing).  Measured for a 720p frame on four threads: 29 us against
// 63 us with memcpy, 62 against 92 us on one.
void copy_streaming(uint8_t* d, const uint8_t* s, size_t n) {
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    if (head >= n) {
        memcpy(d, s, n);
        return;
    }
    memcpy(d, s, head);
    d += head, s += head, n -= head;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 32));
        const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 48), e);
    }
    _mm_sfence();   // the stores are globally visible before the copy engine is told about the slot
    if (i < n) memcpy(d + i, s + i, n - i);
}

// A few threads that copy a frame into a pinned slot side by side with the caller: one thread moves about
// 45 GB/s, a 720p frame (2.8 MB) in 60 us -- more than everything else a call that only hands a frame
// in does.  The workers sleep on a condition variable between frames.
struct CopyPool {
    static constexpr int WORKERS = 3;
    std::thread th[WORKERS];
    std::mutex m;
    std::condition_variable go, done;
    const uint8_t* src = nullptr;
    uint8_t* dst = nullptr;
    size_t chunk = 0, bytes = 0;
    unsigned gen = 0;
    int pending = 0;
    bool stop = false;

    void run(int id) {
        unsigned seen = 0;
        for (;;) {
            const uint8_t* s;
            uint8_t* d;
            size_t lo, hi;
            {
                std::unique_lock<std::mutex> lk(m);
                go.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                s = src, d = dst;
                lo = (size_t)(id + 1) * chunk;
                hi = lo + chunk < bytes ? lo + chunk : bytes;
            }
            if (lo < hi) copy_streaming(d + lo, s + lo, hi - lo);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) done.notify_one();
            }
        }
    }
    CopyPool() {
        for (int i = 0; i < WORKERS; i++) th[i] = std::thread([this, i] { run(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        go.notify_all();
        for (auto& t : th) t.join();
    }
    void copy(uint8_t* d, const uint8_t* s, size_t n) {
        {
            std::lock_guard<std::mutex> lk(m);
            src = s, dst = d, bytes = n;
            chunk = (n / (WORKERS + 1) + 4095) & ~(size_t)4095;
            pending = WORKERS;
            gen++;
        }
        go.notify_all();
        copy_streaming(d, s, chunk < n ? chunk : n);  // the caller's share
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return pending == 0; });
    }
};

}  // namespace

void rtvm_ingest_pool_free(rtvm_ctx* c) {
    delete static_cast<CopyPool*>(c->ingest_pool);
    c->ingest_pool = nullptr;
}

extern "C" int rtvm_ingest_reserve(rtvm_ctx* c, int n_slots) {
    if (!c) return RTVM_E_ARG;
    if (c->fh <= 0 || c->fw <= 0) RTVM_FAIL(c, RTVM_E_STATE, "ingest_reserve: context was created without a frame size");
    if (n_slots < 1 || n_slots > 4096) RTVM_FAIL(c, RTVM_E_ARG, "ingest_reserve: 1..4096 slots");
    if (c->ingest_slots) RTVM_FAIL(c, RTVM_E_STATE, "ingest_reserve: the ring exists already (%d slots)", c->ingest_slots);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->fh * c->fw * 3 * n_slots;
    hipError_t e = hipHostMalloc((void**)&c->ingest_host, bytes, hipHostMallocDefault);
    if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    e = hipMalloc((void**)&c->ingest_dev, bytes);
    if (e != hipSuccess) {
        (void)hipHostFree(c->ingest_host);
        c->ingest_host = nullptr;
        RTVM_FAIL(c, RTVM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    // The pinned pages get their CPU mapping on first touch, at about 0.8 ms per 720p frame: the
    // first trip of a decoder round the ring would run at a ninth of the rate of the later ones.
    // Touch them here, on a few threads.
    {
        const int nthreads = bytes >= ((size_t)64 << 20) ? 4 : 1;
        const size_t chunk = (bytes / nthreads + 4095) & ~(size_t)4095;
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; t++) {
            const size_t lo = (size_t)t * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
            if (lo < hi) th.emplace_back([=] { memset(c->ingest_host + lo, 0, hi - lo); });
        }
        memset(c->ingest_host, 0, chunk < bytes ? chunk : bytes);
        for (auto& t : th) t.join();
    }
    // The copy stream: the stream's packets (the marker of every copy's event) must not sit behind the waits of the
    // canvas half or of the RANSAC streams.  A few candidates, the one that shares a hardware queue with the
    // fewest and least harmful of the context's streams is kept (none: 0; detection only, whose kernels the copy
    // is needed for anyway: 1; a RANSAC stream: 4; the canvas half: 16).
    {
        constexpr int TRIES = 6;
        hipStream_t cand[TRIES];
        int n = 0, best = -1, best_cost = 1 << 30;
        RtvmQueueProbe P;
        const bool probe = c->stream_front != c->stream_main && !getenv("RTVM_NO_QUEUE_PROBE") && rtvm_queue_probe_begin(&P, c->device);
        while (n < TRIES) {
            if (hipStreamCreateWithFlags(&cand[n], hipStreamNonBlocking) != hipSuccess) break;
            n++;
            int cost = 0;
            if (probe) {
                hipStream_t own[5] = {c->stream_main, c->stream_front, c->stream_front2, c->stream_rs, c->stream_lm};
                const int weight[5] = {16, 1, 1, 4, 4};
                for (int i = 0; i < 5 && cost >= 0; i++) {
                    bool seen = false;
                    for (int j = 0; j < i; j++) seen |= own[j] == own[i];
                    if (seen) continue;
                    const int same = rtvm_streams_share_queue(&P, own[i], cand[n - 1]);
                    if (same < 0)
                        cost = -1;
                    else if (same)
                        cost += weight[i];
                }
                if (cost < 0) cost = 1 << 20;   // the probe failed: any stream will do
            }
            if (cost < best_cost) best = n - 1, best_cost = cost;
            if (!probe || cost == 0) break;
        }
        if (probe) rtvm_queue_probe_end(&P);
        if (getenv("RTVM_QUEUE_DEBUG")) fprintf(stderr, "rtvm: copy stream is candidate %d of %d, cost %d\n", best, n, best_cost);
        for (int i = 0; i < n; i++)
            if (i != best) (void)hipStreamDestroy(cand[i]);
        if (best < 0) RTVM_FAIL(c, RTVM_E_HIP, "ingest_reserve: no stream for the copies");
        c->stream_copy = cand[best];
    }
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_ingest, hipEventDisableTiming));
    c->ingest_slots = n_slots;
    return RTVM_OK;
}

extern "C" int rtvm_ingest_host_slot(rtvm_ctx* c, int slot, uint8_t** host) {
    if (!c) return RTVM_E_ARG;
    if (!host) RTVM_FAIL(c, RTVM_E_ARG, "ingest_host_slot: null argument");
    if (slot < 0 || slot >= c->ingest_slots) RTVM_FAIL(c, RTVM_E_ARG, "ingest_host_slot: slot %d of %d", slot, c->ingest_slots);
    *host = c->ingest_host + (size_t)slot * c->fh * c->fw * 3;
    return RTVM_OK;
}

extern "C" int rtvm_ingest_write(rtvm_ctx* c, int slot, const uint8_t* bgr_host) {
    if (!c) return RTVM_E_ARG;
    if (!bgr_host) RTVM_FAIL(c, RTVM_E_ARG, "ingest_write: null argument");
    if (slot < 0 || slot >= c->ingest_slots) RTVM_FAIL(c, RTVM_E_ARG, "ingest_write: slot %d of %d", slot, c->ingest_slots);
    const size_t fb = (size_t)c->fh * c->fw * 3;
    uint8_t* dst = c->ingest_host + (size_t)slot * fb;
    if (fb < ((size_t)1 << 20)) {  // small frames: the hand-over costs more than it saves
        copy_streaming(dst, bgr_host, fb);
        return RTVM_OK;
    }
    if (!c->ingest_pool) c->ingest_pool = new CopyPool();
    static_cast<CopyPool*>(c->ingest_pool)->copy(dst, bgr_host, fb);
    return RTVM_OK;
}

extern "C" int rtvm_ingest_submit(rtvm_ctx* c, int slot, const uint8_t** dev) {
    if (!c) return RTVM_E_ARG;
    if (!dev) RTVM_FAIL(c, RTVM_E_ARG, "ingest_submit: null argument");
    if (slot < 0 || slot >= c->ingest_slots) RTVM_FAIL(c, RTVM_E_ARG, "ingest_submit: slot %d of %d", slot, c->ingest_slots);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t fb = (size_t)c->fh * c->fw * 3;
    HIP_TRY(c, hipMemcpyAsync(c->ingest_dev + slot * fb, c->ingest_host + slot * fb, fb, hipMemcpyHostToDevice, c->stream_copy));
    HIP_TRY(c, hipEventRecord(c->ev_ingest, c->stream_copy));
    c->ingest_used = 1;
    *dev = c->ingest_dev + slot * fb;
    return RTVM_OK;
}

// Consecutive slots in one copy: a 720p frame is 2.8 MB and a copy of that size reaches about half
// the rate of one that moves a whole batch (28 against 56 GB/s measured over PCIe 5.0 x16).
extern "C" int rtvm_ingest_submit_range(rtvm_ctx* c, int first_slot, int count, const uint8_t** dev) {
    if (!c) return RTVM_E_ARG;
    if (!dev) RTVM_FAIL(c, RTVM_E_ARG, "ingest_submit_range: null argument");
    if (count < 1 || first_slot < 0 || first_slot + count > c->ingest_slots)
        RTVM_FAIL(c, RTVM_E_ARG, "ingest_submit_range: slots %d..%d of %d", first_slot, first_slot + count - 1, c->ingest_slots);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t fb = (size_t)c->fh * c->fw * 3;
    HIP_TRY(c, hipMemcpyAsync(c->ingest_dev + first_slot * fb, c->ingest_host + first_slot * fb, fb * count,
                              hipMemcpyHostToDevice, c->stream_copy));
    HIP_TRY(c, hipEventRecord(c->ev_ingest, c->stream_copy));
    c->ingest_used = 1;
    for (int j = 0; j < count; j++) dev[j] = c->ingest_dev + (first_slot + j) * fb;
    return RTVM_OK;
}
