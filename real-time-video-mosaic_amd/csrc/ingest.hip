// Frame ingest, the step before the stitching path (reference main.py:1596-1634: cap.read()
// hands every frame to process_frame as a host array).  A ring of pinned host slots and
// device slots: the caller's decoder writes a frame into a host slot, rtvm_ingest_submit
// copies it to the device on a stream of its own, and whatever stream later reads the frame
// (detection, warp) waits for the copy, not the host.
#include "common.h"

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// A few threads that copy a frame into a pinned slot side by side with the caller: one thread moves about
// 10 GB/s, a 720p frame (2.8 MB) in a quarter of a millisecond -- more than everything else
// a process_frame call does.  The workers sleep on a condition variable between frames.
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
            if (lo < hi) memcpy(d + lo, s + lo, hi - lo);
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
        memcpy(d, s, chunk < n ? chunk : n);  // the caller's share
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
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
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
        memcpy(dst, bgr_host, fb);
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
