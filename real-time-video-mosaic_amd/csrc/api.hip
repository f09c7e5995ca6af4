// C ABI of librtvm.so (declared in include/rtvm.h): context life cycle and the
// host-buffer stage entry points.  Each entry point copies its inputs to the
// context's device buffers, queues the stage's kernels on the context's stream,
// copies the results back and waits.
#include "common.h"

#include <emmintrin.h>

#include <thread>
#include <vector>

#include <cstdlib>

#include <new>

namespace {

template <typename T>
int dev_alloc(rtvm_ctx* c, T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return RTVM_OK;
}

#define ALLOC(c, field, n)                                    \
    do {                                                      \
        int r_ = dev_alloc((c), &(c)->field, (size_t)(n));    \
        if (r_ != RTVM_OK) return r_;                         \
    } while (0)

int alloc_feat(rtvm_ctx* c, FeatSet* f, int n, int desc_bytes) {
    int r;
    if ((r = dev_alloc(c, &f->xy, (size_t)n * 2))) return r;
    if ((r = dev_alloc(c, &f->size, n))) return r;
    if ((r = dev_alloc(c, &f->angle, n))) return r;
    if ((r = dev_alloc(c, &f->response, n))) return r;
    if ((r = dev_alloc(c, &f->octave, n))) return r;
    if ((r = dev_alloc(c, &f->lxy, (size_t)n * 2))) return r;
    if ((r = dev_alloc(c, &f->desc, (size_t)n * desc_bytes))) return r;
    if ((r = dev_alloc(c, &f->count, 4))) return r;
    return RTVM_OK;
}

void free_feat(FeatSet* f) {
    void* ptrs[] = {f->xy, f->size, f->angle, f->response, f->octave, f->lxy, f->desc, f->count};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
}

// Carves one device allocation into the buffers of a frame slot.
struct Carver {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += (n ? n : 1) * sizeof(T);
        return p;
    }
};

void carve_slot(rtvm_ctx* c, Carver& k, FrameSlot& S, int cap, int kind) {
    if (kind == 2) {  // SIFT: features with 128-float descriptors, the L2 matcher's work area
        S.feat.xy = k.take<float>((size_t)cap * 2);
        S.feat.size = k.take<float>(cap);
        S.feat.angle = k.take<float>(cap);
        S.feat.response = k.take<float>(cap);
        S.feat.octave = k.take<int32_t>(cap);
        S.feat.desc = k.take<uint8_t>((size_t)cap * 512);
        S.feat.count = k.take<int>(4);
        const size_t pad = ((size_t)cap + 127) / 128 * 128;
        S.l2_qb = k.take<uint4>(pad * 16);
        S.l2_tb = k.take<uint4>(pad * 16);
        S.l2_qn = k.take<float>(pad);
        S.l2_tn = k.take<float>(pad);
        S.l2_part = k.take<uint32_t>((pad / 128) * pad * 2);
    }
    if (kind == 1) {
        S.pyr = k.take<uint8_t>(c->orb_pyr_bytes);
        S.score = k.take<uint8_t>(c->orb_pyr_bytes);
        S.cand = k.take<uint32_t>(c->orb_cand_total);
        S.cand_count = k.take<int>(RTVM_NLEVELS + RTVM_NLEVELS * 256);
        S.sel_count = k.take<int>(RTVM_NLEVELS);
        S.sel_lxy = k.take<int32_t>((size_t)c->orb_sel_total * 2);
        S.sel_resp = k.take<float>(c->orb_sel_total);
        S.feat.xy = k.take<float>((size_t)cap * 2);
        S.feat.size = k.take<float>(cap);
        S.feat.angle = k.take<float>(cap);
        S.feat.response = k.take<float>(cap);
        S.feat.octave = k.take<int32_t>(cap);
        S.feat.lxy = k.take<int32_t>((size_t)cap * 2);
        S.feat.desc = k.take<uint8_t>((size_t)cap * 32);
        S.feat.count = k.take<int>(4);
    }
    S.qkey = k.take<uint32_t>(cap);
    S.tkey = k.take<uint32_t>(cap);
    S.m_qidx = k.take<int32_t>(cap);
    S.m_tidx = k.take<int32_t>(cap);
    S.m_dist = k.take<int32_t>(cap);
    S.m_count = k.take<int>(4);
    S.rs_src = k.take<float>((size_t)cap * 2);
    S.rs_dst = k.take<float>((size_t)cap * 2);
    S.rs_count = k.take<int>(4);
    S.rs_state = k.take<char>(rtvm_ransac_state_bytes());
    S.rs_idx = k.take<int>(RTVM_RANSAC_MAX_ITERS * 4);
    S.rs_good = k.take<int>(RTVM_RANSAC_MAX_ITERS);
    S.rs_hypH = k.take<double>(RTVM_RANSAC_MAX_ITERS * 9);
    S.rs_csrc = k.take<float>((size_t)cap * 2);
    S.rs_cdst = k.take<float>((size_t)cap * 2);
    S.rs_H = k.take<double>(16);
    S.rs_mask = k.take<uint8_t>(cap);
    S.rs_found = k.take<int>(4);
}

}  // namespace

// Allocates the buffers of one frame slot (one device allocation) and publishes the
// slot to the device copy of the slot array.
int rtvm_slots_alloc(rtvm_ctx* c, int slot, int cap_kp, int kind) {
    FrameSlot& S = c->h_slots[slot];
    memset(&S, 0, sizeof(S));
    Carver dry{nullptr, 0};
    carve_slot(c, dry, S, cap_kp, kind);
    const size_t bytes = dry.off + 256;
    void* arena = nullptr;
    hipError_t e = hipMalloc(&arena, bytes);
    if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "hipMalloc(%zu bytes) for frame slot %d failed: %s", bytes, slot, hipGetErrorString(e));
    c->slot_arena[slot] = arena;
    HIP_TRY(c, hipMemset(arena, 0, bytes));
    Carver k{(char*)arena, 0};
    memset(&S, 0, sizeof(S));
    carve_slot(c, k, S, cap_kp, kind);
    if (slot == RTVM_STAGE_SLOT) {  // host-buffer matcher: descriptors come from the staging buffers
        S.feat.desc = c->d_mq;
        S.feat.count = c->d_mcount;
        S.prev.desc = c->d_mt;
        S.prev.count = c->d_mcount + 1;
    } else if (slot % RTVM_MAX_BATCH == 0) {  // first of a set: bound again whenever a batch is queued
        S.prev = c->feat_prev;
    } else {
        S.prev = c->h_slots[slot - 1].feat;
    }
    HIP_TRY(c, hipMemcpy(&c->d_slots[slot], &S, sizeof(FrameSlot), hipMemcpyHostToDevice));
    return RTVM_OK;
}

// One device allocation per frame of a batch: warped frame, its mask, and the work planes
// of the two distance transforms (kind 0 = new mask, 1 = predicted old mask).
struct SidePlanes {
    uint16_t* S;
    uint32_t* RS;
    float* dist;
    uint32_t *B, *Z, *Z16, *CS;
};

static void carve_side(const rtvm_ctx* c, Carver& k, SideDev& sd, SidePlanes p[2]) {
    const size_t n_px = (size_t)c->ch * c->cw;
    sd.warped = k.take<uint32_t>(n_px);
    sd.mask_new = k.take<uint8_t>(n_px);
    for (int kind = 0; kind < 2; kind++) {
        p[kind].S = k.take<uint16_t>(n_px + 64);
        // 2 * ceil(w / 64) * h words for any h x w the stage entry point accepts (h * w <= ch * cw, h + w <= ch + cw)
        p[kind].RS = k.take<uint32_t>(n_px / 32 + 2 * (size_t)(c->ch + c->cw) + 64);
        p[kind].dist = k.take<float>(n_px);
        p[kind].B = k.take<uint32_t>(c->dt_b_words);
        p[kind].Z = k.take<uint32_t>(c->dt_z_words);
        p[kind].Z16 = k.take<uint32_t>(c->dt_z16_words);
        p[kind].CS = k.take<uint32_t>(c->dt_cs_words);
    }
    sd.dist_new = p[0].dist;
    sd.dist_old = p[1].dist;
    sd.rb = k.take<f32x2>(n_px);
}

void rtvm_dt_job_planes(const rtvm_ctx* c, DtJob* jb, int side, int kind, int pitch, int h, int w) {
    SideDev sd;
    SidePlanes p[2];
    Carver k{(char*)c->side_arena[side], 0};
    carve_side(c, k, sd, p);
    jb->S = p[kind].S;
    jb->RS = p[kind].RS;
    jb->dist = p[kind].dist;
    jb->B = p[kind].B;
    jb->Z = p[kind].Z;
    jb->Z16 = p[kind].Z16;
    jb->CS = p[kind].CS;
    jb->wpitch = (pitch + 31) / 32 + 2;
    jb->znch = (h + 31) / 32;
    jb->zg = (jb->znch + 15) / 16;
    jb->zpitch = w + h;
    jb->cpitch = w;
    jb->ng = (pitch + 63) / 64;
    jb->rpitch = h;
}

// Allocates the planes of frame `side` of a batch and publishes them: d_side[side] and the
// constant part of the frame's two distance-transform jobs.
int rtvm_side_alloc(rtvm_ctx* c, int side) {
    SideDev sd;
    SidePlanes p[2];
    Carver dry{nullptr, 0};
    carve_side(c, dry, sd, p);
    const size_t bytes = dry.off + 256;
    void* arena = nullptr;
    hipError_t e = hipMalloc(&arena, bytes);
    if (e != hipSuccess)
        RTVM_FAIL(c, RTVM_E_NOMEM, "hipMalloc(%zu bytes) for the warp planes of batch frame %d failed: %s", bytes, side,
                  hipGetErrorString(e));
    c->side_arena[side] = arena;
    HIP_TRY(c, hipMemset(arena, 0, bytes));
    Carver k{(char*)arena, 0};
    carve_side(c, k, sd, p);
    c->h_side[side] = sd;
    HIP_TRY(c, hipMemcpy(&c->d_side[side], &sd, sizeof(SideDev), hipMemcpyHostToDevice));
    DtJob jobs[2];
    for (int kind = 0; kind < 2; kind++) {
        DtJob& jb = jobs[kind];
        memset(&jb, 0, sizeof(jb));
        jb.mask = kind ? c->d_mask_old : sd.mask_new;
        jb.pitch = c->cw;
        rtvm_dt_job_planes(c, &jb, side, kind, c->cw, c->ch, c->cw);
    }
    HIP_TRY(c, hipMemcpy(&c->d_jobs[side], &jobs[0], sizeof(DtJob), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(&c->d_jobs[RTVM_MAX_BATCH + side], &jobs[1], sizeof(DtJob), hipMemcpyHostToDevice));
    return RTVM_OK;
}

namespace {

int alloc_all(rtvm_ctx* c) {
    const int mk = RTVM_MAX_MATCH;
    int r;
    // one pinned block: scratch for read-backs (16 KiB), batch reports, batch results, commit homographies
    const size_t rep_bytes = (sizeof(BatchReport) * RTVM_REPORTS + 255) & ~(size_t)255;
    const size_t res_bytes = (sizeof(BatchResults) * RTVM_NSETS + 255) & ~(size_t)255;
    const size_t ch_bytes = sizeof(double) * RTVM_REPORTS * RTVM_MAX_BATCH * 9;
    const size_t pinned_bytes = 16384 + rep_bytes + res_bytes + ch_bytes;
    HIP_TRY(c, hipHostMalloc(&c->h_pinned, pinned_bytes, hipHostMallocDefault));
    memset(c->h_pinned, 0, pinned_bytes);
    c->h_rep = (BatchReport*)((char*)c->h_pinned + 16384);
    c->h_res = (BatchResults*)((char*)c->h_rep + rep_bytes);
    c->h_commit_H = (double*)((char*)c->h_res + res_bytes);
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_rep, c->h_rep, 0));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_res, c->h_res, 0));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_commit_H, c->h_commit_H, 0));
    for (int k = 0; k < RTVM_NSETS; k++) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_feat[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_search[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_front[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_back[k], hipEventDisableTiming));
    }
    for (int k = 0; k < RTVM_REPORTS; k++) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
    c->recs = (BatchRecord*)calloc(RTVM_NSETS, sizeof(BatchRecord));
    if (!c->recs) RTVM_FAIL(c, RTVM_E_NOMEM, "host allocation failed");
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    c->spec_tail = -1;
    c->h_slots = (FrameSlot*)calloc(RTVM_MAX_SLOTS + 1, sizeof(FrameSlot));
    if (!c->h_slots) RTVM_FAIL(c, RTVM_E_NOMEM, "host allocation failed");
    ALLOC(c, d_slots, RTVM_MAX_SLOTS + 1);
    // matcher staging
    ALLOC(c, d_mq, (size_t)mk * 32);
    ALLOC(c, d_mt, (size_t)mk * 32);
    ALLOC(c, d_mcount, 4);
    // frame staging
    const size_t f_px = (size_t)c->fh * c->fw;
    ALLOC(c, d_bgr, f_px * 3);
    // canvas / warp / blend
    const size_t n_px = (size_t)c->ch * c->cw;
    ALLOC(c, d_canvas, n_px * 3);
    ALLOC(c, d_mask_old, n_px);
    {
        // sized for any image rtvm_distance_transform accepts: h < ch + 64, h + w <= ch + cw
        const size_t nch = (size_t)(c->ch + 31) / 32 + 1, lines = (size_t)c->cw + c->ch + 64;
        c->dt_b_words = (size_t)(c->ch + 64) * ((c->cw + c->ch + 31) / 32 + 3);
        c->dt_z_words = 2 * nch * lines;
        c->dt_z16_words = 2 * ((nch + 15) / 16 + 1) * lines;
        c->dt_cs_words = 2 * nch * lines;
    }
    ALLOC(c, d_bold, c->dt_b_words);
    ALLOC(c, d_snap_canvas, n_px * 3);
    ALLOC(c, d_snap_mask, n_px);
    ALLOC(c, d_gk, 32);
    ALLOC(c, d_H, RTVM_MAX_BATCH * 9 + 16);
    ALLOC(c, d_wp, RTVM_MAX_BATCH);
    ALLOC(c, d_cbox, 8);
    ALLOC(c, d_jobs, 2 * RTVM_MAX_BATCH + 2);
    ALLOC(c, d_side, RTVM_MAX_BATCH);
    c->h_side = (SideDev*)calloc(RTVM_MAX_BATCH, sizeof(SideDev));
    if (!c->h_side) RTVM_FAIL(c, RTVM_E_NOMEM, "host allocation failed");
    if ((r = rtvm_side_alloc(c, 0))) return r;
    c->n_side = 1;
    if ((r = rtvm_warp_setup(c))) return r;
    // RANSAC constant table
    c->rs_raw_n = 32768;
    ALLOC(c, d_rs_raw, c->rs_raw_n);
    {
        uint32_t* raw = new (std::nothrow) uint32_t[c->rs_raw_n];
        if (!raw) RTVM_FAIL(c, RTVM_E_NOMEM, "host allocation failed");
        rtvm_ransac_raw_table(raw, c->rs_raw_n);
        hipError_t e = hipMemcpy(c->d_rs_raw, raw, c->rs_raw_n * sizeof(uint32_t), hipMemcpyHostToDevice);
        delete[] raw;
        HIP_TRY(c, e);
    }
    // stage slot (host-buffer matcher / RANSAC entry points)
    if ((r = rtvm_slots_alloc(c, RTVM_STAGE_SLOT, mk, 0))) return r;
    // ORB + stream
    c->desc_bytes = 32;
    if (c->fh > 0 && c->fw > 0 && c->detector == RTVM_DET_SIFT) {
        // the stream path with the reference's default detector
        c->desc_bytes = 512;
        c->orb.max_kp = rtvm_sift_max_keypoints(c->nfeatures);  // the bound every feature array is sized with
        if ((r = rtvm_sift_pipe_create(c))) return r;
        if ((r = alloc_feat(c, &c->feat_prev, c->orb.max_kp, 512))) return r;
        if ((r = rtvm_slots_alloc(c, 0, c->orb.max_kp, 2))) return r;
        c->n_batch = 1;
        if ((r = rtvm_sift_pipe_reserve(c, 1))) return r;
    } else if (c->fh > 0 && c->fw > 0) {
        if ((r = rtvm_orb_plan(c))) return r;
        ALLOC(c, d_xofs, c->orb_xtab_total + 1);
        ALLOC(c, d_xw, c->orb_xtab_total + 1);
        ALLOC(c, d_xsel, c->orb_xtab_total + 4);
        {
            int r_ = dev_alloc(c, (uint8_t**)&c->d_rc_regions, (size_t)(c->orb_rc_total + 1) * 32);
            if (r_ != RTVM_OK) return r_;
        }
        ALLOC(c, d_xwp, c->orb_xtab_total + 4);
        ALLOC(c, d_yofs, c->orb_ytab_total + 1);
        ALLOC(c, d_yw, c->orb_ytab_total + 1);
        if ((r = alloc_feat(c, &c->feat_prev, c->orb.max_kp, 32))) return r;
        if ((r = rtvm_slots_alloc(c, 0, c->orb.max_kp, 1))) return r;
        c->n_batch = 1;
        if ((r = rtvm_orb_setup(c))) return r;
    }
    ALLOC(c, d_state, 1);
    ALLOC(c, d_snap, RTVM_MAX_BATCH);
    ALLOC(c, d_wtab, 32);
    ALLOC(c, d_skip, RTVM_MAX_BATCH);
    if ((r = rtvm_stream_setup(c))) return r;
    return RTVM_OK;
}

void free_all(rtvm_ctx* c) {
    rtvm_sift_pipe_free(c);
    for (int i = 0; i < 8; i++) {
        if (c->sift_buf[i]) (void)hipFree(c->sift_buf[i]);
        c->sift_buf[i] = nullptr;
        c->sift_bytes[i] = 0;
    }
    rtvm_pipe_free(c);
    rtvm_ingest_pool_free(c);
    if (c->stream_copy) {
        (void)hipStreamSynchronize(c->stream_copy);
        (void)hipStreamDestroy(c->stream_copy);
        c->stream_copy = nullptr;
    }
    if (c->ev_ingest) (void)hipEventDestroy(c->ev_ingest);
    c->ev_ingest = nullptr;
    if (c->ingest_host) (void)hipHostFree(c->ingest_host);
    if (c->h_canvas) (void)hipHostFree(c->h_canvas);
    if (c->ingest_dev) (void)hipFree(c->ingest_dev);
    c->ingest_host = c->ingest_dev = nullptr;
    c->ingest_slots = 0;
    void* ptrs[] = {c->d_slots, c->d_mq, c->d_mt, c->d_mcount, c->d_bgr, c->d_canvas,
                    c->d_mask_old, c->d_bold, c->d_snap_canvas, c->d_snap_mask, c->d_side, c->d_snap, c->d_gk, c->d_H,
                    c->d_wp, c->d_cbox, c->d_jobs, c->d_rs_raw, c->d_lq, c->d_lt, c->d_l_idx, c->d_l_dist, c->d_l_work, c->d_xofs, c->d_xw, c->d_xsel, c->d_xwp, c->d_rc_regions, c->d_yofs, c->d_yw, c->d_state, c->d_wtab,
                    c->d_skip};
    free_feat(&c->feat_prev);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int i = 0; i <= RTVM_MAX_SLOTS; i++)
        if (c->slot_arena[i]) (void)hipFree(c->slot_arena[i]);
    for (int i = 0; i < RTVM_MAX_BATCH; i++)
        if (c->side_arena[i]) (void)hipFree(c->side_arena[i]);
    free(c->h_side);
    free(c->h_slots);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (int k = 0; k < RTVM_NSETS; k++) {
        if (c->ev_feat[k]) (void)hipEventDestroy(c->ev_feat[k]);
        if (c->ev_search[k]) (void)hipEventDestroy(c->ev_search[k]);
        if (c->ev_front[k]) (void)hipEventDestroy(c->ev_front[k]);
        if (c->ev_back[k]) (void)hipEventDestroy(c->ev_back[k]);
    }
    for (int k = 0; k < RTVM_REPORTS; k++)
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    free(c->recs);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
}

}  // namespace

extern "C" int rtvm_version(void) { return RTVM_VERSION; }

// ---- streams and the hardware queues behind them -------------------------------------------------
// The runtime gives a process a few hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise) and puts a new
// stream on the queue with the fewest streams, the first such in the order of the queues' addresses: which streams
// end up together depends on what the process created before.  A queue runs its packets in order, so a kernel
// queued behind another stream's wait for an event waits with it.  Measured with the 720p ORB pipeline on four
// queues: 19.7 K frames/s with the streams laid out well, 14.5 K with one stray stream created first; host frames
// through the ingest ring 13 K against 10.5 K.  So the layout is made, not met: candidates are created, a probe
// finds out which of them share a queue, every role takes a candidate from the class it should be in and the
// rest are destroyed.
namespace {

// One lane polls flag[0] until it is set or `ticks` of the 100 MHz wall clock have passed; flag[1] = was it set.
__global__ void k_probe_wait(volatile int* flag, long long ticks) {
    const long long t0 = wall_clock64();
    int seen = 0;
    for (int i = 0; i < (1 << 16) && wall_clock64() - t0 < ticks; i++) {
        if (flag[0]) {
            seen = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    flag[1] = seen;
}
__global__ void k_probe_set(volatile int* flag) { flag[0] = 1; }

}  // namespace

bool rtvm_queue_probe_begin(RtvmQueueProbe* P, int device) {
    P->flag = nullptr;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) return false;
    P->ticks = (long long)khz * 200 / 1000;   // 200 us; a kernel on another queue starts within some ten
    return hipMalloc((void**)&P->flag, 2 * sizeof(int)) == hipSuccess;
}

void rtvm_queue_probe_end(RtvmQueueProbe* P) {
    if (P->flag) (void)hipFree(P->flag);
    P->flag = nullptr;
}

// 1: a kernel on b does not start while one on a runs (the streams share a hardware queue); 0: it does; -1: error.
// "It does" cannot be wrong; "it does not" can (a GPU too busy to start one lane in 200 us), so it is asked twice.
int rtvm_streams_share_queue(RtvmQueueProbe* P, hipStream_t a, hipStream_t b) {
    for (int rep = 0; rep < 2; rep++) {
        int h[2] = {0, 0};
        if (hipMemsetAsync(P->flag, 0, 2 * sizeof(int), a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, P->flag, P->ticks);
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, P->flag);
        if (hipGetLastError() != hipSuccess) return -1;
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        if (hipMemcpyAsync(h, P->flag, sizeof(h), hipMemcpyDeviceToHost, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess)
            return -1;
        if (h[1]) return 0;
    }
    return 1;
}

namespace {

// Creates the working streams of a multi-stream context: roles[r] takes a candidate of the first class of
// prefs[r] (classes in the order they were met) that still has one.  false: nothing was created (the caller
// falls back to plain creation).
struct StreamRole {
    hipStream_t* out;
    int prefs_many[4];   // when there is a class for every role
    int prefs_few[4];    // when there is not: -1 ends the list
};

bool create_streams_by_queue(rtvm_ctx* c, StreamRole* roles, int n_roles) {
    constexpr int MAXC = 10;
    RtvmQueueProbe P;
    if (!rtvm_queue_probe_begin(&P, c->device)) return false;
    hipStream_t cand[MAXC];
    int cls[MAXC], rep[MAXC], n = 0, ncls = 0;
    int pick[8];
    bool ok = true, ideal = false;
    while (ok && !ideal && n < MAXC) {
        if (hipStreamCreate(&cand[n]) != hipSuccess) break;
        int k = -1;
        for (int r = 0; r < ncls && k < 0 && ok; r++) {
            const int same = rtvm_streams_share_queue(&P, cand[rep[r]], cand[n]);
            if (same < 0) ok = false;
            if (same == 1) k = r;
        }
        if (k < 0) {
            k = ncls;
            rep[ncls++] = n;
        }
        cls[n++] = k;
        // the assignment with what there is so far
        bool used[MAXC] = {};
        ideal = true;
        for (int r = 0; r < n_roles; r++) {
            const int* prefs = ncls > n_roles ? roles[r].prefs_many : roles[r].prefs_few;
            pick[r] = -1;
            for (int q = 0; q < 4 && prefs[q] >= 0 && pick[r] < 0; q++) {
                for (int i = 0; i < n && pick[r] < 0; i++)
                    if (!used[i] && cls[i] == prefs[q]) pick[r] = i;
                if (pick[r] < 0) ideal = false;   // the first choice was not to be had
            }
            if (pick[r] < 0)
                for (int i = 0; i < n && pick[r] < 0; i++)
                    if (!used[i]) pick[r] = i;
            if (pick[r] >= 0)
                used[pick[r]] = true;
            else
                ideal = false;
        }
        // with fewer classes than roles the first choices may never come: one candidate per role and class is enough
        if (!ideal && n >= 2 * n_roles) break;
    }
    rtvm_queue_probe_end(&P);
    bool all = ok && n >= n_roles;
    for (int r = 0; r < n_roles && all; r++) all = pick[r] >= 0;
    if (getenv("RTVM_QUEUE_DEBUG")) {
        fprintf(stderr, "rtvm: %d candidate streams, queue classes", n);
        for (int i = 0; i < n; i++) fprintf(stderr, " %d", cls[i]);
        fprintf(stderr, "; roles take");
        for (int r = 0; r < n_roles && all; r++) fprintf(stderr, " %d", pick[r]);
        fprintf(stderr, "%s\n", all ? "" : " nothing (plain creation)");
    }
    bool used[MAXC] = {};
    if (all)
        for (int r = 0; r < n_roles; r++) {
            *roles[r].out = cand[pick[r]];
            used[pick[r]] = true;
        }
    for (int i = 0; i < n; i++)
        if (!used[i]) (void)hipStreamDestroy(cand[i]);
    return all;
}

}  // namespace


static void destroy_streams(rtvm_ctx* c) {
    if (c->stream_front2 != c->stream_front) (void)hipStreamDestroy(c->stream_front2);
    if (c->stream_lm != c->stream_main && c->stream_lm != c->stream_front) (void)hipStreamDestroy(c->stream_lm);
    if (c->stream_rs != c->stream_main && c->stream_rs != c->stream_front) (void)hipStreamDestroy(c->stream_rs);
    if (c->stream_front != c->stream_main) (void)hipStreamDestroy(c->stream_front);
    (void)hipStreamDestroy(c->stream_main);
}

extern "C" int rtvm_create(rtvm_ctx** out, int device, int frame_h, int frame_w, int canvas_h,
                           int canvas_w, int detector, int nfeatures, int flags) {
    if (!out) return RTVM_E_ARG;
    *out = nullptr;
    if (frame_h < 0 || frame_w < 0 || canvas_h < 0 || canvas_w < 0 || nfeatures < 0) return RTVM_E_ARG;
    if (detector != RTVM_DET_ORB && detector != RTVM_DET_SIFT_MATCH_ONLY && detector != RTVM_DET_SIFT) return RTVM_E_ARG;
    if (detector == RTVM_DET_SIFT && frame_h > 0 && (nfeatures < 1 || nfeatures > RTVM_SIFT_STREAM_MAX_KP - 256)) return RTVM_E_ARG;
    // the warp reads the two taps of a row with one 8-byte load that must fit inside the frame
    if ((frame_h > 0 || frame_w > 0) && (frame_h < 2 || frame_w < 2)) return RTVM_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return RTVM_E_HIP;
    rtvm_ctx* c = new (std::nothrow) rtvm_ctx;
    if (!c) return RTVM_E_NOMEM;
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->fh = frame_h;
    c->fw = frame_w;
    c->ch = canvas_h;
    c->cw = canvas_w;
    c->detector = detector;
    c->nfeatures = nfeatures;
    c->flags = flags;
    // A context without a frame size (or with RTVM_FLAG_SINGLE_STREAM) serves the stage entry
    // points only, one call at a time: it gets a single stream.
    // (All streams at the default priority: with the main stream on a high-priority queue the frame
    // rate rose by under 1 %, and a context used next to the single-stream contexts of worker
    // threads stopped making progress.)
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return RTVM_E_HIP;
    }
    const bool multi = frame_h > 0 && !(flags & RTVM_FLAG_SINGLE_STREAM);
    // A SIFT context, whose detection is the longest stream and holds kernels of a few waves, gets three
    // streams: the canvas half, and two that take the batches of the even and of the odd slot sets through
    // detection, matching and RANSAC, so that consecutive batches overlap.  Other contexts: canvas half,
    // detection + matching, RANSAC search, RANSAC polish.
    const bool two_front = detector == RTVM_DET_SIFT && !getenv("RTVM_ONE_DETECT_STREAM");
    bool made = false;
    if (multi && !getenv("RTVM_NO_QUEUE_PROBE")) {
        // Every stream on a queue of its own where the process has queues to spare (and one left for the ingest
        // ring's copies); with four queues the two RANSAC streams, which wait for each other anyway, share one.
        // The canvas half never shares with another stream of the context.
        if (two_front) {
            StreamRole roles[3] = {{&c->stream_main, {0, -1, -1, -1}, {0, -1, -1, -1}},
                                   {&c->stream_front, {1, -1, -1, -1}, {1, 0, -1, -1}},
                                   {&c->stream_front2, {2, -1, -1, -1}, {2, 1, 0, -1}}};
            made = create_streams_by_queue(c, roles, 3);
            if (made) c->stream_rs = c->stream_lm = c->stream_front;   // per batch: the stream of its detection
        } else {
            StreamRole roles[4] = {{&c->stream_main, {0, -1, -1, -1}, {0, -1, -1, -1}},
                                   {&c->stream_front, {1, -1, -1, -1}, {1, 0, -1, -1}},
                                   {&c->stream_rs, {2, -1, -1, -1}, {2, 1, 0, -1}},
                                   {&c->stream_lm, {3, -1, -1, -1}, {2, 3, 1, 0}}};
            made = create_streams_by_queue(c, roles, 4);
            if (made) c->stream_front2 = c->stream_front;
        }
    }
    if (!made) {
        // plain creation: the streams land where the runtime puts them
        if (hipStreamCreate(&c->stream_main) != hipSuccess) {
            delete c;
            return RTVM_E_HIP;
        }
        c->stream_front = c->stream_front2 = c->stream_rs = c->stream_lm = c->stream_main;
        if (multi) {
            hipStream_t* more[4] = {&c->stream_front, two_front ? &c->stream_front2 : nullptr, two_front ? nullptr : &c->stream_rs,
                                    two_front ? nullptr : &c->stream_lm};
            bool ok = true;
            for (hipStream_t* st : more)
                if (st && ok && hipStreamCreate(st) != hipSuccess) {
                    *st = c->stream_main;
                    ok = false;
                }
            if (!two_front || !ok) {
                if (two_front && c->stream_front2 != c->stream_main && c->stream_front2 != c->stream_front)
                    (void)hipStreamDestroy(c->stream_front2);
                c->stream_front2 = c->stream_front;
            }
            if (two_front && ok) c->stream_rs = c->stream_lm = c->stream_front;   // per batch: the stream of its detection
            if (!ok) {
                destroy_streams(c);
                delete c;
                return RTVM_E_HIP;
            }
        }
    }
    c->stream = c->stream_main;
    // The column pass of the blur inside the commit kernel saves 16 bytes of traffic per pixel and frame and costs
    // parallelism (one wave walks the frames of a piece in order): measured +8 % at 2160 x 3840, +5 % at 1080p, +1 %
    // at 720p, -3 % at 360p (1100 waves per piece).  RTVM_FUSED_COMMIT=0/1 overrides (tests run both).
    c->fused_commit = (long long)frame_h * frame_w >= 800000;
    if (const char* e = getenv("RTVM_FUSED_COMMIT")) c->fused_commit = atoi(e) != 0;
    int r = alloc_all(c);
    if (r != RTVM_OK) {
        fprintf(stderr, "rtvm_create: %s\n", c->err);
        free_all(c);
        destroy_streams(c);
        delete c;
        return r;
    }
    *out = c;
    return RTVM_OK;
}

extern "C" void rtvm_destroy(rtvm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream_front);
    (void)hipStreamSynchronize(c->stream_front2);
    (void)hipStreamSynchronize(c->stream_rs);
    (void)hipStreamSynchronize(c->stream_lm);
    (void)hipStreamSynchronize(c->stream_main);
    rtvm_prof_free(c);
    free_all(c);
    destroy_streams(c);
    delete c;
}

extern "C" const char* rtvm_last_error(const rtvm_ctx* c) { return c ? c->err : "null context"; }

extern "C" int rtvm_sync(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    // every front half is followed by a canvas half on the main stream that waits for it
    HIP_TRY(c, hipStreamSynchronize(c->stream_main));
    return RTVM_OK;
}

static int match_hamming_host(rtvm_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt,
                              int32_t* qidx, int32_t* tidx, int32_t* dist, int* m_out, int sorted) {
    if (!c) return RTVM_E_ARG;
    if (!m_out || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t))
        RTVM_FAIL(c, RTVM_E_ARG, "match_hamming: bad argument");
    if (nq > RTVM_MAX_MATCH || nt > RTVM_MAX_MATCH)
        RTVM_FAIL(c, RTVM_E_ARG, "match_hamming: at most %d descriptors per side", RTVM_MAX_MATCH);
    HIP_TRY(c, hipSetDevice(c->device));
    const FrameSlot& S = c->h_slots[RTVM_STAGE_SLOT];
    int* hc = (int*)c->h_pinned;
    hc[0] = nq;
    hc[1] = nt;
    HIP_TRY(c, hipMemcpyAsync(c->d_mcount, hc, 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (nq) HIP_TRY(c, hipMemcpyAsync(c->d_mq, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (nt) HIP_TRY(c, hipMemcpyAsync(c->d_mt, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    int r = rtvm_launch_hamming(c, RTVM_STAGE_SLOT, 1, nq, nt, sorted);
    if (r != RTVM_OK) return r;
    HIP_TRY(c, hipMemcpyAsync(hc + 2, S.m_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int m = hc[2];
    if (m < 0 || m > nq) RTVM_FAIL(c, RTVM_E_HIP, "match_hamming: device returned count %d", m);
    if (m) {
        if (!qidx || !tidx || !dist) RTVM_FAIL(c, RTVM_E_ARG, "match_hamming: null output");
        HIP_TRY(c, hipMemcpyAsync(qidx, S.m_qidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(tidx, S.m_tidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(dist, S.m_dist, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *m_out = m;
    return RTVM_OK;
}

extern "C" int rtvm_match_hamming_xcheck(rtvm_ctx* c, const uint8_t* q, int nq, const uint8_t* t,
                                         int nt, int32_t* qidx, int32_t* tidx, int32_t* dist,
                                         int* m_out) {
    return match_hamming_host(c, q, nq, t, nt, qidx, tidx, dist, m_out, 0);
}

extern "C" int rtvm_match_hamming_xcheck_sorted(rtvm_ctx* c, const uint8_t* q, int nq,
                                                const uint8_t* t, int nt, int32_t* qidx,
                                                int32_t* tidx, int32_t* dist, int* m_out) {
    return match_hamming_host(c, q, nq, t, nt, qidx, tidx, dist, m_out, 1);
}

// ---- L2 knn-2 matcher (SIFT branch) ---------------------------------------------
extern "C" int rtvm_match_l2_knn2(rtvm_ctx* c, const float* q, int nq, const float* t, int nt, int dim,
                                  int32_t* idx, float* dist) {
    if (!c) return RTVM_E_ARG;
    if (nq < 0 || nt < 2 || dim <= 0 || !t || (nq > 0 && (!q || !idx || !dist)))
        RTVM_FAIL(c, RTVM_E_ARG, "match_l2_knn2: bad argument (needs at least 2 train descriptors)");
    if (nq == 0) return RTVM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t need_q = (size_t)nq * dim, need_t = (size_t)nt * dim;
    const size_t need_w = rtvm_l2_work_bytes(nq, nt, dim);
    if (need_q > c->l_cap_q || need_t > c->l_cap_t || need_w > c->l_cap_work) {  // grow the staging buffers
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        void* old[] = {c->d_lq, c->d_lt, c->d_l_idx, c->d_l_dist, c->d_l_work};
        for (void* p : old)
            if (p) (void)hipFree(p);
        c->d_lq = c->d_lt = c->d_l_dist = nullptr;
        c->d_l_idx = nullptr;
        c->d_l_work = nullptr;
        const size_t cq = need_q > c->l_cap_q ? need_q : c->l_cap_q, ct = need_t > c->l_cap_t ? need_t : c->l_cap_t;
        const size_t cw = need_w > c->l_cap_work ? need_w : c->l_cap_work;
        c->l_cap_q = c->l_cap_t = c->l_cap_work = 0;
        ALLOC(c, d_lq, cq);
        ALLOC(c, d_lt, ct);
        ALLOC(c, d_l_idx, (size_t)2 * (cq / dim + 1));
        ALLOC(c, d_l_dist, (size_t)2 * (cq / dim + 1));
        {
            char* w = nullptr;
            int r_ = dev_alloc(c, &w, cw);
            if (r_ != RTVM_OK) return r_;
            c->d_l_work = w;
        }
        c->l_cap_q = cq;
        c->l_cap_t = ct;
        c->l_cap_work = cw;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_lq, q, need_q * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_lt, t, need_t * sizeof(float), hipMemcpyHostToDevice, c->stream));
    int r = rtvm_launch_l2_knn2(c, c->d_lq, nq, c->d_lt, nt, dim, c->d_l_idx, c->d_l_dist, c->d_l_work);
    if (r) return r;
    HIP_TRY(c, hipMemcpyAsync(idx, c->d_l_idx, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dist, c->d_l_dist, (size_t)nq * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

// ---- gray / ORB ---------------------------------------------------------------
extern "C" int rtvm_max_keypoints(const rtvm_ctx* c) { return c ? c->orb.max_kp : 0; }

static int need_frame(rtvm_ctx* c, const char* who, bool orb_only = true) {
    if (c->fh <= 0 || c->fw <= 0) RTVM_FAIL(c, RTVM_E_STATE, "%s: context was created without a frame size", who);
    if (orb_only && c->detector == RTVM_DET_SIFT)
        RTVM_FAIL(c, RTVM_E_STATE, "%s: the context was created with the SIFT detector", who);
    return RTVM_OK;
}

int rtvm_orb_host(rtvm_ctx* c, const uint8_t* bgr, int* n_out, float* xy, float* size, float* angle_deg,
                  float* response, int32_t* octave, uint8_t* desc, uint8_t* gray);  // stream.hip

extern "C" int rtvm_orb(rtvm_ctx* c, const uint8_t* bgr, int* n_out, float* xy, float* size, float* angle_deg,
                        float* response, int32_t* octave, uint8_t* desc) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !n_out) RTVM_FAIL(c, RTVM_E_ARG, "orb: null argument");
    int r = need_frame(c, "orb");
    if (r) return r;
    return rtvm_orb_host(c, bgr, n_out, xy, size, angle_deg, response, octave, desc, nullptr);
}

// gray = level 0 of the pyramid; rtvm_gray runs the ORB front end and copies it out
extern "C" int rtvm_gray(rtvm_ctx* c, const uint8_t* bgr, uint8_t* gray) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !gray) RTVM_FAIL(c, RTVM_E_ARG, "gray: null argument");
    int r = need_frame(c, "gray");
    if (r) return r;
    int n = 0;
    return rtvm_orb_host(c, bgr, &n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gray);
}

extern "C" int rtvm_reserve_batch(rtvm_ctx* c, int n) {
    if (!c) return RTVM_E_ARG;
    if (n < 1 || n > RTVM_MAX_BATCH) RTVM_FAIL(c, RTVM_E_ARG, "reserve_batch: batch size must be 1..%d", RTVM_MAX_BATCH);
    int r = need_frame(c, "reserve_batch", false);
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int kind = c->detector == RTVM_DET_SIFT ? 2 : 1;
    if (kind == 2 && (r = rtvm_sift_pipe_reserve(c, n))) return r;
    for (int s = c->n_batch; s < n; s++) {
        if ((r = rtvm_slots_alloc(c, s, c->orb.max_kp, kind))) return r;
        c->n_batch = s + 1;
    }
    for (int s = c->n_side; s < n; s++) {
        if ((r = rtvm_side_alloc(c, s))) return r;
        c->n_side = s + 1;
    }
    for (int s = c->n_set1; s < n; s++) {
        for (int k = 1; k < RTVM_NSETS; k++)
            if ((r = rtvm_slots_alloc(c, k * RTVM_MAX_BATCH + s, c->orb.max_kp, kind))) return r;
        c->n_set1 = s + 1;
    }
    return RTVM_OK;
}

// ---- RANSAC homography ------------------------------------------------------
extern "C" int rtvm_find_homography_ransac(rtvm_ctx* c, const float* src_xy, const float* dst_xy, int m,
                                           double thresh, int max_iters, double confidence, double H[9],
                                           uint8_t* inlier_mask, int* found) {
    if (!c) return RTVM_E_ARG;
    if (!found || !H || m < 0 || (m > 0 && (!src_xy || !dst_xy)))
        RTVM_FAIL(c, RTVM_E_ARG, "find_homography: bad argument");
    if (m > RTVM_MAX_MATCH) RTVM_FAIL(c, RTVM_E_ARG, "find_homography: at most %d point pairs", RTVM_MAX_MATCH);
    HIP_TRY(c, hipSetDevice(c->device));
    const FrameSlot& S = c->h_slots[RTVM_STAGE_SLOT];
    int* hc = (int*)c->h_pinned;
    hc[0] = m;
    HIP_TRY(c, hipMemcpyAsync(S.rs_count, hc, sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (m) {
        HIP_TRY(c, hipMemcpyAsync(S.rs_src, src_xy, (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(S.rs_dst, dst_xy, (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
    }
    int r = rtvm_launch_ransac(c, RTVM_STAGE_SLOT, 1, thresh, max_iters, confidence);
    if (r) return r;
    double* hH = (double*)c->h_pinned + 8;
    HIP_TRY(c, hipMemcpyAsync(hH, S.rs_H, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hc + 1, S.rs_found, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (inlier_mask && m) HIP_TRY(c, hipMemcpyAsync(inlier_mask, S.rs_mask, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *found = hc[1];
    memcpy(H, hH, 9 * sizeof(double));
    if (!*found && inlier_mask && m) memset(inlier_mask, 0, (size_t)m);
    return RTVM_OK;
}

// ---- canvas / warp / blend ------------------------------------------------
static int need_canvas(rtvm_ctx* c, const char* who) {
    if (c->ch <= 0 || c->cw <= 0 || c->fh <= 0 || c->fw <= 0)
        RTVM_FAIL(c, RTVM_E_STATE, "%s: context was created without frame/canvas size", who);
    return RTVM_OK;
}

extern "C" int rtvm_canvas_init(rtvm_ctx* c, const uint8_t* first_bgr, int row_off, int col_off) {
    if (!c) return RTVM_E_ARG;
    if (!first_bgr) RTVM_FAIL(c, RTVM_E_ARG, "canvas_init: null frame");
    int r = need_canvas(c, "canvas_init");
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, first_bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    r = rtvm_launch_canvas_init(c, c->d_bgr, row_off, col_off);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

static int upload_H(rtvm_ctx* c, const double H[9]) {
    double* hp = (double*)c->h_pinned + 16;
    memcpy(hp, H, 9 * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_H, hp, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_warp_perspective(rtvm_ctx* c, const uint8_t* bgr, const double H[9], uint8_t* out) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !H || !out) RTVM_FAIL(c, RTVM_E_ARG, "warp_perspective: null argument");
    int r = need_canvas(c, "warp_perspective");
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    if ((r = upload_H(c, H))) return r;
    if ((r = rtvm_launch_warp_only(c, c->d_bgr, c->d_H, (uint8_t*)c->h_side[0].warped))) return r;
    HIP_TRY(c, hipMemcpyAsync(out, c->h_side[0].warped, (size_t)c->ch * c->cw * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_distance_transform(rtvm_ctx* c, const uint8_t* mask, int h, int w, float* dist) {
    if (!c) return RTVM_E_ARG;
    if (!mask || !dist || h <= 0 || w <= 0) RTVM_FAIL(c, RTVM_E_ARG, "distance_transform: bad argument");
    if ((size_t)h * w > (size_t)c->ch * c->cw || h + w > c->ch + c->cw || (h + 31) / 32 > (c->ch + 31) / 32 + 1)
        RTVM_FAIL(c, RTVM_E_ARG, "distance_transform: %dx%d exceeds the context's canvas (%dx%d)", h, w, c->ch, c->cw);
    HIP_TRY(c, hipSetDevice(c->device));
    DtJob* jb = (DtJob*)((char*)c->h_pinned + 1024);
    jb->mask = c->h_side[0].mask_new;
    jb->pitch = w;
    jb->x0 = jb->y0 = 0;
    jb->x1 = w;
    jb->y1 = h;
    jb->ox0 = jb->oy0 = 0;
    jb->ox1 = w;
    jb->oy1 = h;
    rtvm_dt_job_planes(c, jb, 0, 0, w, h, w);
    HIP_TRY(c, hipMemcpyAsync(c->d_jobs + RTVM_JOB_STAGE, jb, sizeof(DtJob), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_side[0].mask_new, mask, (size_t)h * w, hipMemcpyHostToDevice, c->stream));
    int r = rtvm_launch_dt(c, c->d_jobs + RTVM_JOB_STAGE, 1, w, h, 1);
    if (r) return r;
    HIP_TRY(c, hipMemcpyAsync(dist, c->h_side[0].dist_new, (size_t)h * w * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_warp_blend(rtvm_ctx* c, const uint8_t* bgr, const double H[9]) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !H) RTVM_FAIL(c, RTVM_E_ARG, "warp_blend: null argument");
    int r = need_canvas(c, "warp_blend");
    if (r) return r;
    if (!c->canvas_ready) RTVM_FAIL(c, RTVM_E_STATE, "warp_blend: call rtvm_canvas_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    if ((r = upload_H(c, H))) return r;
    // The host waits for the two uploads only (the caller's frame and the staging of H may be
    // written again on return); the warp and blend stay queued, and whatever reads the canvas
    // next is ordered behind them on the same stream.
    if (!c->ev_upload) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_upload, c->stream));
    if ((r = rtvm_launch_warp_blend(c, c->d_bgr, c->d_H))) return r;
    HIP_TRY(c, hipEventSynchronize(c->ev_upload));
    return RTVM_OK;
}

// u8 -> float64 with non-temporal stores (the destination is eight times the source and is not read back here)
static void widen_u8_f64(const uint8_t* s, double* d, size_t n) {
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) {
        const __m128i zero = _mm_setzero_si128();
        for (; i + 16 <= n; i += 16) {
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
            const __m128i lo = _mm_unpacklo_epi8(b, zero), hi = _mm_unpackhi_epi8(b, zero);
            const __m128i w[4] = {_mm_unpacklo_epi16(lo, zero), _mm_unpackhi_epi16(lo, zero), _mm_unpacklo_epi16(hi, zero),
                                  _mm_unpackhi_epi16(hi, zero)};
            for (int k = 0; k < 4; k++) {
                _mm_stream_pd(d + i + 4 * k, _mm_cvtepi32_pd(w[k]));
                _mm_stream_pd(d + i + 4 * k + 2, _mm_cvtepi32_pd(_mm_shuffle_epi32(w[k], 0xEE)));
            }
        }
        _mm_sfence();
    }
    for (; i < n; i++) d[i] = (double)s[i];
}

// The mosaic as the reference keeps it (main.py:80-81: np.zeros((h, w, 3)) is float64, and every blend stores
// astype(np.uint8) values into it, main.py:905-910): canvas_h x canvas_w x 3 doubles holding integers 0..255.  The bytes come over
// PCIe into a pinned buffer (a ninth of what the doubles would be) and are widened on the host, on four threads.
extern "C" int rtvm_canvas_read_f64(rtvm_ctx* c, double* out) {
    if (!c) return RTVM_E_ARG;
    if (!out) RTVM_FAIL(c, RTVM_E_ARG, "canvas_read_f64: null output");
    int r = need_canvas(c, "canvas_read_f64");
    if (r) return r;
    if (!c->canvas_ready) RTVM_FAIL(c, RTVM_E_STATE, "canvas_read_f64: call rtvm_canvas_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->ch * c->cw * 3;
    if (!c->h_canvas) {
        hipError_t e = hipHostMalloc((void**)&c->h_canvas, n, hipHostMallocDefault);
        if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_canvas, c->d_canvas, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int nthreads = n >= ((size_t)1 << 20) ? 4 : 1;
    const size_t chunk = ((n / nthreads + 63) & ~(size_t)63);
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) {
        const size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        if (lo < hi) th.emplace_back([=] { widen_u8_f64(c->h_canvas + lo, out + lo, hi - lo); });
    }
    widen_u8_f64(c->h_canvas, out, chunk < n ? chunk : n);
    for (auto& t : th) t.join();
    return RTVM_OK;
}

extern "C" int rtvm_canvas_read(rtvm_ctx* c, uint8_t* out) {
    if (!c) return RTVM_E_ARG;
    if (!out) RTVM_FAIL(c, RTVM_E_ARG, "canvas_read: null output");
    int r = need_canvas(c, "canvas_read");
    if (r) return r;
    if (!c->canvas_ready) RTVM_FAIL(c, RTVM_E_STATE, "canvas_read: call rtvm_canvas_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_canvas, (size_t)c->ch * c->cw * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}
