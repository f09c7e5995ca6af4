// Internal declarations shared by the translation units of librtvm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rtvm.h"

#define RTVM_NLEVELS 8
#define RTVM_MAX_MATCH 16384   // keypoints per side the Hamming matcher (sorted form) accepts
#define RTVM_RANSAC_MAX_ITERS 2000
#define RTVM_SIFT_STREAM_MAX_KP 8192   // keypoints per frame the stream path holds with the SIFT detector

// ---------------------------------------------------------------------------
// ORB geometry and tables, passed to the kernels by value.
struct OrbDev {
    int w[RTVM_NLEVELS], h[RTVM_NLEVELS], pitch[RTVM_NLEVELS];
    unsigned off[RTVM_NLEVELS];        // byte offset of the level plane in a pyramid buffer
    float scale[RTVM_NLEVELS];         // (float)pow(1.2f, level)
    int quota[RTVM_NLEVELS];           // nfeaturesPerLevel
    int cand_off[RTVM_NLEVELS], cand_cap[RTVM_NLEVELS];
    int sel_off[RTVM_NLEVELS];         // per-level selected arrays (capacity quota + 64)
    int xtab_off[RTVM_NLEVELS], ytab_off[RTVM_NLEVELS];
    int tile_off[RTVM_NLEVELS + 1];    // 64x32 tiles of the region inside the 31-pixel border, prefix over levels
    int btile_off[RTVM_NLEVELS + 1];   // 64x32 tiles of the whole level, prefix over levels
    int umax[17];                      // half widths of the circular patch rows
    int gk7[7];                        // 7-tap Gaussian, 8-bit fixed point
    int max_kp;
};

// Feature set of one frame, device resident.
struct FeatSet {
    float* xy;         // max_kp*2
    float* size;       // max_kp
    float* angle;      // max_kp
    float* response;   // max_kp
    int32_t* octave;   // max_kp
    int32_t* lxy;      // max_kp*2 level-pixel coordinates (x,y)
    uint8_t* desc;     // max_kp*32
    int* count;        // 1
};

// Everything the front half of process_frame (detect, match, RANSAC) needs for one
// in-flight frame.  A batch of frames uses slots 0..n-1 and every front-half kernel
// takes the slot array plus a batch index in its grid, so one launch serves the
// whole batch.  slot[0].prev is the persistent "previous frame" feature set,
// slot[j].prev aliases slot[j-1].feat.
struct FrameSlot {
    const uint8_t* bgr;    // frame pixels (device)
    uint8_t* pyr;          // pyramid (un-blurred)
    uint8_t* score;        // FAST score planes, same layout
    uint32_t* cand;        // candidates: packed (score << 24 | y << 12 | x)
    int* cand_count;       // RTVM_NLEVELS counters + RTVM_NLEVELS*256 histogram
    int* sel_count;        // per-level selected counts
    int32_t* sel_lxy;      // per-level selected (x,y)
    float* sel_resp;       // per-level selected Harris response
    FeatSet feat;          // features of this frame (query side)
    FeatSet prev;          // features it is matched against (train side)
    uint32_t* qkey;        // packed (dist << 16 | idx) per query
    uint32_t* tkey;        // per train
    int32_t* m_qidx;       // matches
    int32_t* m_tidx;
    int32_t* m_dist;
    int* m_count;
    float* rs_src;         // matched points, current frame
    float* rs_dst;         // matched points, previous frame
    int* rs_count;
    void* rs_state;        // RsState (ransac.hip)
    int* rs_idx;           // 4 indices per iteration
    int* rs_good;          // inlier count per iteration
    double* rs_hypH;       // 9 doubles per iteration
    float* rs_csrc;        // compacted inliers
    float* rs_cdst;
    double* rs_H;          // 9 doubles
    uint8_t* rs_mask;
    int* rs_found;
    // L2 matcher work area (contexts with the SIFT detector): both descriptor sets as bf16 in
    // matrix-core operand order, their squared norms, per-tile top-2 keys
    uint4* l2_qb;
    uint4* l2_tb;
    float* l2_qn;
    float* l2_tn;
    uint32_t* l2_part;
};

#define RTVM_MAX_BATCH 32
// Four sets of batch slots (set k = slots k*RTVM_MAX_BATCH ...): while the canvas half of
// one batch runs, the RANSAC polish, the RANSAC search and detection + matching of the
// three batches after it work in the other sets.
#define RTVM_NSETS 4
#define RTVM_MAX_SLOTS (RTVM_NSETS * RTVM_MAX_BATCH)
#define RTVM_STAGE_SLOT RTVM_MAX_SLOTS   // slot used by the host-buffer matcher / RANSAC entry points

// One distance-transform job: a rectangle of a mask plane and the planes the
// scans write (S and dist are indexed y*pitch + x like the mask).
struct DtJob {
    const uint8_t* mask;
    int pitch;
    int x0, y0, x1, y1;      // rectangle the transform is computed on (outside = wall)
    int ox0, oy0, ox1, oy1;  // sub-rectangle whose distances are needed
    uint16_t* S;       // steps to the nearest zero along the two diagonals through the pixel
    uint32_t* RS;      // row summaries per 64-column group: [2 directions][ng][rpitch], indexed by the row
    float* dist;
    uint32_t* B;       // one zero bit per pixel: [rows][wpitch] words
    uint32_t* Z;       // zero bits of the diagonal lines per 32-row chunk: [2 orientations][znch][zpitch]
    uint32_t* Z16;     // which chunks of a group of 16 hold a zero: [2][zg][zpitch]
    uint32_t* CS;      // column chunk summaries: [2][znch][cpitch]
    int wpitch, znch, zg, zpitch, cpitch;
    int ng, rpitch;    // 64-column groups of the plane, rows of the plane
};

// Workgroups go to the eight XCDs (each with an L2 of its own) in turn by their linear index.  A
// launch over `n` tiles uses xcd_grid(n) workgroups in x; workgroup i takes tile
// (i % 8) * (grid / 8) + i / 8, so that every XCD works through one contiguous run of tiles and
// the rows and cache lines that neighbouring tiles share are found in its L2.  Returns -1 for the
// padding workgroups.
__device__ __forceinline__ int xcd_tile(int i, int n) {
    const int per = (n + 7) >> 3, j = i >> 3;
    const int t = (i & 7) * per + j;
    return j < per && t < n ? t : -1;  // the grid may be larger than xcd_grid(n)
}
__host__ __device__ inline int xcd_grid(int n) { return ((n + 7) >> 3) << 3; }

// The same for tiles of unequal cost: runs of 16 consecutive tiles go to the XCDs in turn, so that
// neighbours within a run share their cache lines and every XCD still gets tiles from all over
// the image.  Grid: xcd_grid_runs(n) workgroups.
__device__ __forceinline__ int xcd_tile_runs(int i, int n) {
    const int x = i & 7, j = i >> 3;
    const int t = (((j >> 4) << 3) + x) * 16 + (j & 15);
    return t < n ? t : -1;
}
__host__ __device__ inline int xcd_grid_runs(int n) { return ((n + 127) >> 7) << 7; }

// A pointer read from a structure in memory is a generic pointer to the compiler (flat loads,
// 64-bit address arithmetic per lane); every such pointer of this library is device memory:
// as_global says so (global loads, scalar base + lane offset).
template <class T>
using gptr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ gptr<T> as_global(T* p) {
    return (gptr<T>)p;
}

// 16-byte accesses through a device-memory pointer (HIP's uint4 is a class whose members want
// generic pointers).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ uint4 load_u4(gptr<T> p) {
    const u32x4 v = *reinterpret_cast<gptr<const u32x4>>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <class T>
__device__ __forceinline__ void store_u4(gptr<T> p, const uint4& v) {
    *reinterpret_cast<gptr<u32x4>>(p) = u32x4{v.x, v.y, v.z, v.w};
}

// DtJob as the kernels use it: the same fields, the pointers typed as device memory.
struct DtJobG {
    gptr<const uint8_t> mask;
    int pitch;
    int x0, y0, x1, y1;
    int ox0, oy0, ox1, oy1;
    gptr<uint16_t> S;
    gptr<uint32_t> RS;
    gptr<float> dist;
    gptr<uint32_t> B, Z, Z16, CS;
    int wpitch, znch, zg, zpitch, cpitch, ng, rpitch;
    __device__ explicit DtJobG(const DtJob& j)
        : mask(as_global(j.mask)), pitch(j.pitch), x0(j.x0), y0(j.y0), x1(j.x1), y1(j.y1), ox0(j.ox0), oy0(j.oy0),
          ox1(j.ox1), oy1(j.oy1), S(as_global(j.S)), RS(as_global(j.RS)), dist(as_global(j.dist)), B(as_global(j.B)),
          Z(as_global(j.Z)), Z16(as_global(j.Z16)), CS(as_global(j.CS)), wpitch(j.wpitch), znch(j.znch), zg(j.zg),
          zpitch(j.zpitch), cpitch(j.cpitch), ng(j.ng), rpitch(j.rpitch) {}
};

// Between two phases of a workgroup that is ONE wave.  LDS serves the accesses of a wave in order, so only
// the compiler has to be kept from moving them across; __syncthreads() would also wait for every global store
// of the wave to complete.
__device__ __forceinline__ void wave_lds_phase() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two floats handled as one packed operand (v_pk_fma_f32 and friends).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The planes one frame of a batch is warped into and measured on, besides the canvas.
// Every frame of a batch has its own set, so that the warps and the distance transforms
// of a whole batch run as single launches.
struct SideDev {
    uint32_t* warped;    // ch*cw words: the frame in canvas coordinates (ROI only), per pixel B, G, R
                         // and the mask byte (255 when a channel is non-zero)
    uint8_t* mask_new;   // ch*cw: a mask plane for the stage entry points (the batch kernels use the
                         // mask byte of `warped`)
    float* dist_new;     // distance transform of mask_new; after the blur: blend weight of the frame
    float* dist_old;     // distance transform of the canvas mask as it is before this frame; after
                         // the blur: blend weight of the canvas
    f32x2* rb;           // row-blurred blend weights, {of the frame, of the canvas} per pixel
};

// Per-frame warp parameters, written on the device by k_warp_prep.
struct WarpParams {
    double M[9];       // H^-1
    float Mf[9];       // (float)M
    int rx0, ry0, rx1, ry1;   // region outside which the warped frame is zero
    int wx0, wy0, wx1, wy1;   // region where the blend weights are needed (ROI + 15, clipped)
    int bw0;           // block width of the fixed-point coordinate evaluation
    int pad;
};

// Device-side stream state (what VideMosaic keeps between frames).
struct StreamState {
    double H_old[9];
    double H_abs[9];
    double last_valid_H[9];
    double hist[5][9];
    int hist_len;
    int status;        // RTVM_FRAME_* of the last frame
    int n_matches;
    int n_kp;
    int frame_index;
    int abort;         // later frames stay pending: 1 = a canvas mask was mispredicted, 2 = a frame was skipped
    int n_done;        // frames of the last batch that were consumed (stitched or skipped)
    int last_good;     // batch index of the last frame that advanced the state, -1 if none
    int tail_ok;       // the last frame of the last batch advanced the state (a batch whose front
                       // half was matched against that frame ahead of time is valid)
    int bad_frame;     // first frame of the batch whose blend cleared a canvas pixel: the canvas
                       // masks predicted for the later frames of the batch are wrong
    int b_status[RTVM_MAX_BATCH];
    int b_matches[RTVM_MAX_BATCH];
    int b_kp[RTVM_MAX_BATCH];
};

// What the caller reads back about one batch; written by the device into pinned host memory.
struct BatchReport {
    int n_done, abort, tail_ok;
    int reason;   // why n_done < n: RTVM_PENDING_*
    int cleared;  // first frame of the piece whose blend cleared a canvas pixel, or -1
    int status[RTVM_MAX_BATCH];
    int matches[RTVM_MAX_BATCH];
    int kp[RTVM_MAX_BATCH];
};

// What the library remembers about a queued batch, so that its canvas half can be queued
// again for the frames that came back pending.
struct BatchRecord {
    long ticket;     // 0 = empty
    int set, sb, n;  // slot set, first slot, frames
    int ahead;       // its first frame was matched against the frame queued before it (not feat_prev)
    const uint8_t* ptrs[RTVM_MAX_BATCH];
};
#define RTVM_REPORTS 16

// Detection, matching and RANSAC results of a detected batch, for hosts that keep the validate /
// smooth / chain arithmetic on their side (rtvm_batch_results); written by the device into
// pinned host memory, one per slot set.
struct BatchResults {
    int matches[RTVM_MAX_BATCH];
    int found[RTVM_MAX_BATCH];
    int kp[RTVM_MAX_BATCH];
    double H[RTVM_MAX_BATCH][9];
};

// frame pointers of a batch, passed to the binding kernel by value
struct BatchPtrs {
    const uint8_t* p[RTVM_MAX_BATCH];
};

struct rtvm_ctx {
    int device;
    hipStream_t stream;        // stream the launch macros use right now
    hipStream_t stream_main;   // canvas half, host-buffer entry points, read-backs
    hipStream_t stream_front;  // detection + matching of a batch queued ahead of the canvas half
    hipStream_t stream_front2; // the same for the batches in the odd slot sets (contexts with the SIFT detector, whose
                               // detection is the longest stream and holds kernels of a few waves: the detections
                               // of consecutive batches overlap); elsewhere the same stream as stream_front
    hipStream_t stream_rs;     // RANSAC hypothesis search of a batch queued ahead of the canvas half
    hipStream_t stream_lm;     // RANSAC polish (least squares + Levenberg-Marquardt) of such a batch
    hipEvent_t ev_search[RTVM_NSETS]; // RANSAC search of the batch in slot set k is complete
    hipEvent_t ev_feat[RTVM_NSETS];   // detection + matching of the batch in slot set k is complete
    hipEvent_t ev_front[RTVM_NSETS];  // RANSAC polish of the batch in slot set k is complete
    hipEvent_t ev_back[RTVM_NSETS];   // canvas half of the batch that used slot set k is complete
    int* d_cbox;               // [8] bounding box of everything pasted or warped onto the canvas: x0, y0, x1, y1;
                               // then that of the regions of interest of the batch being stitched
    // rtvm_sift's device buffers (pyramids, frame, candidate keys, ...), kept between calls and
    // grown on demand
    void* sift_buf[8];
    size_t sift_bytes[8];
    // frame ingest ring (rtvm_ingest_*): pinned host slots the caller decodes into, device slots
    // they are copied to on a stream of their own
    uint8_t* ingest_host;
    uint8_t* ingest_dev;
    int ingest_slots;
    hipStream_t stream_copy;
    int fused_commit;          // the column pass of the blur runs inside the commit kernel (k_commit_cols)
    uint8_t* h_canvas;         // pinned copy of the canvas for rtvm_canvas_read_f64 (allocated at its first call)
    hipEvent_t ev_ingest;      // the last copy queued
    int ingest_used;           // a copy was queued: streams that read frames wait for ev_ingest
    void* ingest_pool = nullptr;  // threads of rtvm_ingest_write (ingest.hip)
    void* pipe = nullptr;         // state of rtvm_pipe_* (video.hip)
    int rs_alone = 0;   // at most a batch's worth of frames in flight: latency is what counts (ransac.hip)
    hipEvent_t ev_done[RTVM_REPORTS];  // batch number (seq % RTVM_REPORTS) is complete
    hipEvent_t ev_main;
    int back_rec[RTVM_NSETS];  // ev_back[k] has been recorded
    int set_next;              // slot set of the next pipelined batch
    int spec_tail;             // slot of the last frame of the batch queued before, -1 = none:
                               // the next batch is matched against feat_prev after a full wait
    long seq;                  // batches queued since the stream began
    int n_set1;                // allocated slots of each of the sets 1 .. RTVM_NSETS-1
    BatchReport* h_rep;        // [RTVM_REPORTS] pinned, indexed by seq % RTVM_REPORTS
    BatchRecord* recs;         // [RTVM_NSETS]: the batch whose detection results each slot set holds
    long set_owner[RTVM_NSETS]; // ticket of the batch whose detection results the slot set holds
    int set_live[RTVM_NSETS];   // the set holds a batch queued with rtvm_batch_detect and not yet released
    BatchReport* d_rep;        // device view of h_rep
    BatchResults* h_res;       // [RTVM_NSETS] pinned: what rtvm_batch_results hands out
    BatchResults* d_res;       // device view of h_res
    double* h_commit_H;        // [RTVM_REPORTS][RTVM_MAX_BATCH * 9] pinned: homographies of rtvm_batch_commit
    double* d_commit_H;        // device view of h_commit_H
    int fh, fw, ch, cw;
    int detector, nfeatures, flags;
    int desc_bytes;            // bytes of one descriptor of the stream path: 32 (ORB) or 512 (SIFT)
    void* sift_pipe;           // SIFT detection for the frames of a batch (sift.hip)
    char err[512];

    // --- ORB ---
    OrbDev orb;
    size_t orb_pyr_bytes;
    int orb_cand_total, orb_sel_total, orb_xtab_total, orb_ytab_total;
    int orb_chain;         // the pyramid's levels are resized in two chained launches (else one launch per level)
    uint8_t* d_bgr;        // staging frame fh*fw*3
    uint16_t* d_xofs;      // resize tables
    uint16_t* d_xw;
    void* d_rc_regions;    // the chained resize's rectangles per tile and level (RcRegion, orb.hip), both chains
    int orb_rc_total;      // number of them
    uint32_t* d_xsel;      // per output column: byte selector of its two taps within the eight bytes from its quad's first source
    uint32_t* d_xwp;       // per output column: the two horizontal weights as 16-bit halves {256 - w, w}
    uint16_t* d_yofs;
    uint16_t* d_yw;
    FrameSlot* h_slots;    // host mirror of d_slots (RTVM_MAX_SLOTS + 1 entries)
    FrameSlot* d_slots;
    void* slot_arena[RTVM_MAX_SLOTS + 1];  // one device allocation per slot
    int n_batch;           // slots 0..n_batch-1 of set 0 are allocated
    FeatSet feat_prev;     // features of the previous (last accepted) frame

    // --- matcher staging for the host-buffer entry points ---
    uint8_t* d_mq;
    uint8_t* d_mt;
    int* d_mcount;         // [nq, nt]
    // L2 matcher
    float* d_lq;
    float* d_lt;
    int32_t* d_l_idx;
    float* d_l_dist;
    void* d_l_work;            // flag, norms, bf16 operands, per-tile keys (rtvm_l2_work_bytes)
    size_t l_cap_q, l_cap_t, l_cap_work;   // capacities in floats / bytes

    // --- RANSAC ---
    uint32_t* d_rs_raw;    // raw outputs of the fixed-seed RNG
    int rs_raw_n;

    // --- canvas / warp / blend ---
    uint8_t* d_canvas;     // ch*cw*3
    uint8_t* d_mask_old;   // ch*cw: canvas pixel has a non-zero channel
    uint32_t* d_bold;      // zero bit rows of d_mask_old, kept up to date by the commit kernel
    uint8_t* d_snap_canvas;  // canvas and mask as they were before the batch being committed
    uint8_t* d_snap_mask;
    size_t dt_b_words, dt_z_words, dt_z16_words, dt_cs_words;  // per distance-transform job
    SideDev* h_side;       // per frame of a batch (host mirror of d_side)
    SideDev* d_side;
    void* side_arena[RTVM_MAX_BATCH];  // one device allocation per frame of a batch
    int n_side;            // sides 0..n_side-1 are allocated
    float* d_gk;           // 31 Gaussian taps
    double* d_H;           // RTVM_MAX_BATCH x 9 doubles: homographies handed to the warp
    WarpParams* d_wp;      // one per frame of a batch
    DtJob* d_jobs;         // [j] = new mask on the ROI, [RTVM_MAX_BATCH + j] = old mask on the canvas for
                           // frame j of a batch; then RTVM_JOB_STAGE (stage API) and RTVM_JOB_CANVAS
                           // (the whole old mask, for its bit rows)
    int canvas_ready;

    StreamState* d_state;
    StreamState* d_snap;   // state after each frame of the batch (RTVM_MAX_BATCH entries)
    double* d_wtab;        // smoothing weights, row n = linspace(0.5, 1, n) / sum
    int* d_skip;           // RTVM_MAX_BATCH flags: 1 when the frame is not to be stitched
    int stream_ready;
    void* h_pinned;        // small pinned host buffer for read-backs
    int commit_parity;     // which of the two staging slots the last rtvm_frame_commit used
    hipEvent_t ev_upload;  // rtvm_warp_blend: the uploads of the frame and of H have been consumed
    int prof_mode;         // 0 off, 1 every kernel, 2 kernels whose name contains prof_filter
    char prof_filter[64];
    void* prof;            // timing records (prof.hip)
};

// cvtColor(BGR2GRAY) in fixed point: (B*cb + G*cg + R*cr + half) >> shift
struct GrayCoef {
    int cb, cg, cr, shift;
};
static inline GrayCoef rtvm_gray_coef(const rtvm_ctx* c) {
    return (c->flags & RTVM_FLAG_GRAY_LEGACY) ? GrayCoef{1868, 9617, 4899, 14} : GrayCoef{3735, 19235, 9798, 15};
}

#define RTVM_FAIL(ctx, code, ...)                         \
    do {                                                  \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                    \
    } while (0)

#define HIP_TRY(ctx, expr)                                                            \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            RTVM_FAIL(ctx, RTVM_E_HIP, "%s failed: %s (%s:%d)", #expr,                \
                      hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

// ---- launchers implemented by the stage files (all asynchronous on ctx->stream) ----
// match.hip
int rtvm_launch_hamming(rtvm_ctx* c, int sb, int nb, int cap_q, int cap_t, int sorted);
// dt.hip
#define RTVM_JOB_STAGE (2 * RTVM_MAX_BATCH)
#define RTVM_JOB_CANVAS (2 * RTVM_MAX_BATCH + 1)
// with_bits = 0: the zero bit rows of the jobs (DtJob::B) are already up to date
int rtvm_launch_dt(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h, int with_bits);
int rtvm_launch_dt_bits(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h);
// jobs d_jobs[0 .. n_first) and d_jobs[second .. second + n_second) in the same launches
int rtvm_launch_dt2(rtvm_ctx* c, const DtJob* d_jobs, int n_first, int second, int n_second, int max_w, int max_h,
                    int with_bits);
// api.hip: work planes of side `side`, kind 0 (new mask) or 1 (old mask), for an h x w image
void rtvm_dt_job_planes(const rtvm_ctx* c, DtJob* jb, int side, int kind, int pitch, int h, int w);
int rtvm_side_alloc(rtvm_ctx* c, int side);
// warp.hip
int rtvm_warp_setup(rtvm_ctx* c);  // uploads constant tables (after allocation)
int rtvm_launch_canvas_init(rtvm_ctx* c, const uint8_t* d_bgr, int row_off, int col_off);
int rtvm_launch_warp_only(rtvm_ctx* c, const uint8_t* d_bgr, const double* d_H, uint8_t* d_out);
// VideMosaic.warp() for the n frames of a batch (homographies d_H[9*j], optional "do not
// stitch" flags d_skip[j]): warp parameters and distance-transform jobs; warps, masks and
// new-mask distances of all frames; predicted canvas masks and their distances for all
// frames; weights and their 31x31 blur for all frames; blend + commit of all frames, pixel
// by pixel in frame order.
int rtvm_launch_warp_prep(rtvm_ctx* c, const double* d_H, const int* d_skip, int n);
int rtvm_launch_sides(rtvm_ctx* c, const uint8_t* const* d_bgr, int n);
int rtvm_launch_weights(rtvm_ctx* c, int n);
int rtvm_launch_commit(rtvm_ctx* c, int n);
// all of the above for one frame
int rtvm_launch_warp_blend(rtvm_ctx* c, const uint8_t* d_bgr, const double* d_H);
// orb.hip
int rtvm_orb_plan(rtvm_ctx* c);   // host-side geometry (before allocation)
int rtvm_orb_setup(rtvm_ctx* c);  // uploads tables (after allocation)
int rtvm_launch_orb(rtvm_ctx* c, int sb, int nb);
// sift.hip: the same for contexts with the SIFT detector (features into slot.feat; desc = 128 floats)
int rtvm_sift_pipe_create(rtvm_ctx* c);          // geometry (before the slots are allocated)
int rtvm_sift_pipe_reserve(rtvm_ctx* c, int n);  // pyramids and lists for batches of up to n frames
void rtvm_sift_pipe_free(rtvm_ctx* c);
int rtvm_launch_sift(rtvm_ctx* c, int sb, int nb);
// l2match.hip: knn-2 + ratio test + sort by distance (main.py:688-698) of slot.feat against slot.prev
int rtvm_launch_l2_slots(rtvm_ctx* c, int sb, int nb, int cap);
// stream.hip
int rtvm_stream_setup(rtvm_ctx* c);
// api.hip; kind 0: matcher / RANSAC work areas only, 1: plus ORB detection planes and features, 2: plus SIFT features
int rtvm_slots_alloc(rtvm_ctx* c, int slot, int cap_kp, int kind);
// ransac.hip
size_t rtvm_ransac_state_bytes();
void rtvm_ransac_raw_table(uint32_t* raw, int n);
void rtvm_ingest_pool_free(rtvm_ctx* c);
// Which streams share a hardware queue (api.hip): begin allocates the probe's two words on the current device.
struct RtvmQueueProbe {
    int* flag;
    long long ticks;
};
bool rtvm_queue_probe_begin(RtvmQueueProbe* P, int device);
void rtvm_queue_probe_end(RtvmQueueProbe* P);
int rtvm_streams_share_queue(RtvmQueueProbe* P, hipStream_t a, hipStream_t b);   // 1 shared, 0 not, -1 error
// video.hip
void rtvm_pipe_free(rtvm_ctx* c);
int rtvm_launch_ransac(rtvm_ctx* c, int sb, int nb, double thresh, int max_iters, double confidence);
int rtvm_launch_ransac_search(rtvm_ctx* c, int sb, int nb, double thresh, int max_iters, double confidence);
int rtvm_launch_ransac_refine(rtvm_ctx* c, int sb, int nb, double thresh);
size_t rtvm_l2_work_bytes(int nq, int nt, int dim);
int rtvm_launch_l2_knn2(rtvm_ctx* c, const float* dq, int nq, const float* dt, int nt, int dim, int32_t* d_idx,
                        float* d_dist, void* work);

// ---- optional per-kernel timing with HIP events on the context's stream (prof.hip) ----
int rtvm_prof_begin(rtvm_ctx* c, const char* name);
void rtvm_prof_end(rtvm_ctx* c, int rec);
void rtvm_prof_free(rtvm_ctx* c);

#define RTVM_LAUNCH(c, kernel, grid, block, shmem, ...)                          \
    do {                                                                         \
        const int pr_ = (c)->prof_mode ? rtvm_prof_begin((c), #kernel) : -1;     \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (c)->stream, __VA_ARGS__); \
        if (pr_ >= 0) rtvm_prof_end((c), pr_);                                   \
    } while (0)
