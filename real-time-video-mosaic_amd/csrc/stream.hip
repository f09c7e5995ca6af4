// VideMosaic.process_frame (reference main.py:710-759) with every piece of state
// resident on the device: features of the previous frame, H_old, the smoothing
// history and the canvas.
//
// Frames are processed in batches.  The front half of process_frame (detect,
// match against the previous frame, RANSAC) of frame t depends only on frames t and
// t-1, never on the canvas or on H_old, so it runs for a whole batch at once: every
// front-half kernel is launched once with the batch index in its grid.  The back
// half is sequential: one thread chains validate / smooth / H_old @ H for the batch
// in frame order, then each frame is warped and blended into the canvas in turn.
//
// The reference does not advance its state on a skipped frame (fewer than 4 matches,
// or no homography).  Inside a batch frame t+1 was matched against frame t on the
// assumption that t would be accepted; when t is skipped in the middle of a batch,
// the later frames of the batch are left untouched (RTVM_FRAME_PENDING) for the
// caller to resubmit, so the result is the reference's for any batch size.
//
// Four streams: rtvm_process_batch queues detection + matching of a batch on the front
// stream, its RANSAC search and RANSAC polish on two more streams and the back half on the
// main stream, cycling through four sets of frame slots, so detection of batch k+3 and the
// RANSAC stages of batches k+2 and k+1 (small grids, latency bound) overlap the back half
// of batch k (bandwidth bound).  The
// first frame of a batch is then matched against the last frame of the batch before it
// before that frame is known to be accepted; when it was not, the later batch is left
// pending as a whole, exactly like the rest of a batch after a skipped frame.
#include "common.h"

#include <cmath>

namespace {

// points of the matches, in match order: src = current frame, dst = previous frame
// (main.py:850-854)
__global__ __launch_bounds__(256) void k_gather_pts(const FrameSlot* __restrict__ slots, int sb) {
    const FrameSlot& S = slots[sb + blockIdx.y];
    const int m = *S.m_count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *S.rs_count = m;
    if (i >= m) return;
    const int q = S.m_qidx[i], t = S.m_tidx[i];
    S.rs_src[2 * i] = S.feat.xy[2 * q];
    S.rs_src[2 * i + 1] = S.feat.xy[2 * q + 1];
    S.rs_dst[2 * i] = S.prev.xy[2 * t];
    S.rs_dst[2 * i + 1] = S.prev.xy[2 * t + 1];
}

// validate_homography (main.py:761-801).  The 2x2 determinant is evaluated the way
// an LU factorisation with partial pivoting does (numpy.linalg.det); sqrt of a
// negative determinant is NaN, which fails no comparison and therefore passes.
__device__ bool validate_h(const double* H, double trans_thr, double scale_thr) {
    for (int i = 0; i < 9; i++)
        if (isnan(H[i]) || isinf(H[i])) return false;
    const double tx = H[2], ty = H[5];
    const double translation = sqrt(tx * tx + ty * ty);
    const double a = H[0], b = H[1], c = H[3], d = H[4];
    double det;
    if (a == 0. && c == 0.) {
        det = 0.;  // a zero pivot: the factorisation stops, the product of the diagonal is 0
    } else if (fabs(a) >= fabs(c)) {
        const double l = c * (1. / a);
        det = a * (d - l * b);
    } else {
        const double l = a * (1. / c);
        det = -(c * (b - l * d));
    }
    const double scale = sqrt(det);
    if (translation > trans_thr) return false;
    if (fabs(scale - 1.0) > scale_thr) return false;
    if (fabs(H[6]) > 0.001 || fabs(H[7]) > 0.001) return false;
    return true;
}

// validate_homography of one matrix, for the stage entry point: out[0] = 1 when H passes.
__global__ void k_validate(const double* __restrict__ H, double trans_thr, double scale_thr, int* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double h[9];
    for (int i = 0; i < 9; i++) h[i] = H[i];
    out[0] = validate_h(h, trans_thr, scale_thr) ? 1 : 0;
}

// main.py:722-746 and 756 for the n frames of a batch, in order: skip rules,
// validate, smooth (weighted mean of at most five relative homographies),
// H = H_old @ H_smoothed, H_old = H.  Writes the absolute homography and the
// "do not stitch" flag of every frame, and a copy of the state after every frame.
// One wave (the phases are separated for the compiler only, see wave_lds_phase): the state and the
// per-frame inputs are brought into LDS by all lanes; lane e < 9
// carries element e of the 3x3 matrices through the chain (the decisions are taken by all
// lanes alike), lane 0 keeps the scalars, all lanes copy the snapshots out.
__global__ __launch_bounds__(64) void k_batch_glue(const FrameSlot* __restrict__ slots, int sb, int n,
                                                   StreamState* __restrict__ st_g, const double* __restrict__ wtab,
                                                   double* __restrict__ H_out, int* __restrict__ skip,
                                                   double trans_thr, double scale_thr, int ahead,
                                                   StreamState* __restrict__ snap) {
    __shared__ StreamState S;
    __shared__ int s_m[RTVM_MAX_BATCH], s_found[RTVM_MAX_BATCH], s_kp[RTVM_MAX_BATCH], s_skip[RTVM_MAX_BATCH];
    __shared__ double s_H[RTVM_MAX_BATCH][9], s_Habs[RTVM_MAX_BATCH][9];
    __shared__ double s_w[30], s_Hs[9];
    const int lane = threadIdx.x;
    const int e = lane < 9 ? lane : 0;  // the matrix element of this lane
    constexpr int WORDS = (int)(sizeof(StreamState) / sizeof(uint32_t));
    static_assert(sizeof(StreamState) % sizeof(uint32_t) == 0, "StreamState is copied word by word");
    {
        const uint32_t* g = reinterpret_cast<const uint32_t*>(st_g);
        uint32_t* l = reinterpret_cast<uint32_t*>(&S);
        for (int i = lane; i < WORDS; i += 64) l[i] = g[i];
    }
    __shared__ int s_valid[RTVM_MAX_BATCH];
    if (lane < n) {
        const FrameSlot& F = slots[sb + lane];
        s_m[lane] = *F.m_count;
        s_found[lane] = *F.rs_found;
        s_kp[lane] = *F.feat.count;
        double Hr[9];
        for (int i = 0; i < 9; i++) Hr[i] = s_H[lane][i] = F.rs_H[i];
        s_skip[lane] = 1;
        // validate_homography looks at the frame's own matrix only: all frames at once, one per lane,
        // ahead of the chain through the frames
        s_valid[lane] = validate_h(Hr, trans_thr, scale_thr) ? 1 : 0;
    }
    if (lane < 30) s_w[lane] = wtab[lane];
    wave_lds_phase();
    StreamState* st = &S;
    int n_done = 0, last_good = -1;
    // 0 = go on; 1 = a mispredicted canvas mask upstream (detection results stand);
    // 2 = a skipped frame upstream (the frames behind it were matched against the wrong frame)
    int abort = st->abort;
    // matched ahead of time against a frame that did not advance the state
    if (ahead && !st->tail_ok) abort = 2;
    for (int j = 0; j < n; j++) {
        bool snapshot = false;
        if (abort) {
            if (lane == 0) {
                st->b_status[j] = RTVM_FRAME_PENDING;
                st->b_matches[j] = 0;
                st->b_kp[j] = 0;
            }
        } else {
            snapshot = true;
            n_done = j + 1;
            const int m = s_m[j];
            int status = 0;
            if (m < 4)
                status = RTVM_FRAME_FEW_MATCHES;
            else if (!s_found[j])
                status = RTVM_FRAME_NO_HOMOGRAPHY;
            if (lane == 0) {
                st->frame_index += 1;
                st->n_matches = st->b_matches[j] = m;
                st->n_kp = st->b_kp[j] = s_kp[j];
            }
            if (status) {
                if (lane == 0) st->status = st->b_status[j] = status;
                if (j < n - 1) abort = 2;  // the later frames were matched against this one
            } else {
                const bool ok = s_valid[j] != 0;
                if (!ok) status |= RTVM_FRAME_INVALID_H;
                const double Hr_e = ok ? s_H[j][e] : ((e % 4 == 0) ? 1. : 0.);
                // smooth_homography: the history of element e
                int k = st->hist_len;
                double h[5];
                for (int a = 0; a < 5; a++) h[a] = st->hist[a][e];
                const double Ho0 = st->H_old[(e / 3) * 3], Ho1 = st->H_old[(e / 3) * 3 + 1], Ho2 = st->H_old[(e / 3) * 3 + 2];
                wave_lds_phase();  // everything read before anything of this frame is written
                if (k == 5) {
                    for (int a = 0; a < 4; a++) h[a] = h[a + 1];
                    k = 4;
                }
                for (int a = 0; a < 5; a++)
                    if (a == k) h[a] = Hr_e;
                k++;
                double Hs_e = Hr_e;
                if (k >= 2) {
                    double acc = 0.;
                    for (int a = 0; a < 5; a++)
                        if (a < k) acc += s_w[k * 5 + a] * h[a];
                    Hs_e = acc;
                }
                if (lane < 9) {
                    if (ok) st->last_valid_H[e] = Hr_e;
                    for (int a = 0; a < 5; a++)
                        if (a < k) st->hist[a][e] = h[a];
                    s_Hs[e] = Hs_e;
                }
                wave_lds_phase();
                // self.H = np.matmul(self.H_old, H_relative_smoothed): numpy hands a 3x3 float64 product to its
                // BLAS, whose dgemm kernels on a CPU with fused multiply-add sum k = 0, 1, 2 into a zeroed
                // accumulator, one fused multiply-add per term
                const int c2 = e % 3;
                const double H_e = __builtin_fma(Ho2, s_Hs[6 + c2], __builtin_fma(Ho1, s_Hs[3 + c2], Ho0 * s_Hs[c2]));
                if (lane < 9) {
                    st->H_abs[e] = H_e;
                    st->H_old[e] = H_e;
                    s_Habs[j][e] = H_e;
                }
                if (lane == 0) {
                    st->hist_len = k;
                    st->status = st->b_status[j] = status;
                    s_skip[j] = 0;
                }
                last_good = j;
            }
            if (lane == 0) st->last_good = last_good;
        }
        wave_lds_phase();
        if (snapshot) {  // the state to return to when a later frame of the batch cannot be stitched
            const uint32_t* l = reinterpret_cast<const uint32_t*>(&S);
            uint32_t* g = reinterpret_cast<uint32_t*>(snap + j);
            for (int i = lane; i < WORDS; i += 64) g[i] = l[i];
        }
        wave_lds_phase();
    }
    if (lane == 0) {
        st->abort = abort;
        st->n_done = n_done;
        st->last_good = last_good;
        // did the last frame that was consumed advance the state?  (unchanged by a void batch)
        if (n_done > 0) st->tail_ok = (!abort && last_good == n - 1) ? 1 : 0;
    }
    wave_lds_phase();
    {
        const uint32_t* l = reinterpret_cast<const uint32_t*>(&S);
        uint32_t* g = reinterpret_cast<uint32_t*>(st_g);
        for (int i = lane; i < WORDS; i += 64) g[i] = l[i];
    }
    if (lane < n) {
        skip[lane] = s_skip[lane];
        if (!s_skip[lane])
            for (int i = 0; i < 9; i++) H_out[9 * lane + i] = s_Habs[lane][i];
    }
}

// The glue of a canvas half whose absolute homographies come from the host (rtvm_batch_commit):
// the host has applied the skip rules, validate, smooth and chain (main.py:722-746) itself and
// hands over only frames that are to be stitched.  What is left of k_batch_glue is the
// bookkeeping: behind a canvas half that left frames pending (st->abort) the whole piece is
// void; otherwise every frame is stitched with the homography it came with.
__global__ __launch_bounds__(64) void k_batch_host_glue(StreamState* __restrict__ st, const double* __restrict__ H_in,
                                                        int n, double* __restrict__ H_out, int* __restrict__ skip) {
    const int lane = threadIdx.x;
    const int abort = st->abort;
    if (lane < n) {
        skip[lane] = abort ? 1 : 0;
        st->b_status[lane] = abort ? RTVM_FRAME_PENDING : 0;
        st->b_matches[lane] = 0;
        st->b_kp[lane] = 0;
        if (!abort)
            for (int i = 0; i < 9; i++) H_out[9 * lane + i] = H_in[9 * lane + i];
    }
    if (lane == 0) {
        st->n_done = abort ? 0 : n;
        st->last_good = abort ? -1 : n - 1;
        if (!abort) {
            st->tail_ok = 1;
            st->frame_index += n;
            for (int i = 0; i < 9; i++) st->H_old[i] = st->H_abs[i] = H_in[9 * (n - 1) + i];
        }
    }
}

// Per-frame outcome of detection, matching and RANSAC of a batch, gathered into pinned host
// memory for rtvm_batch_results.  Lane j = frame j.
__global__ __launch_bounds__(64) void k_batch_collect(const FrameSlot* __restrict__ slots, int sb, int n,
                                                      BatchResults* __restrict__ out) {
    const int j = threadIdx.x;
    if (j >= n) return;
    const FrameSlot& F = slots[sb + j];
    const int m = *F.m_count;
    out->matches[j] = m;
    out->found[j] = m >= 4 ? *F.rs_found : 0;
    out->kp[j] = *F.feat.count;
    for (int i = 0; i < 9; i++) out->H[j][i] = F.rs_H[i];
}

// End of a batch.  The canvas masks of frames 1..n-1 were predicted (a blend of two
// non-zero pixels stays non-zero); when the blend of frame `bad` broke that, the later
// frames were left untouched: the state goes back to what it was after frame `bad` and
// they are reported pending, like the rest of a batch after a skipped frame.  Then the
// report the host reads.
// host_glue: the homographies came from the host, which keeps the stream state itself; there is
// no device state to go back to.
// lists = false: the caller copies the per-frame lists of the report itself (batch_report_lists, a frame per lane)
__device__ void batch_finish(StreamState* st, const StreamState* snap, int n, BatchReport* rep, int host_glue,
                             bool lists = true) {
    const int bad = st->bad_frame;
    int reason = RTVM_PENDING_NONE;
    bool later = false;
    if (bad >= 0 && bad < n - 1)
        for (int j = bad + 1; j < n; j++) later |= st->b_status[j] != RTVM_FRAME_PENDING;
    if (later) {
        if (!host_glue) *st = snap[bad];
        for (int j = bad + 1; j < n; j++) {
            st->b_status[j] = RTVM_FRAME_PENDING;
            st->b_matches[j] = 0;
            st->b_kp[j] = 0;
        }
        st->abort = 1;  // whatever is queued behind this batch is void
        st->n_done = bad + 1;
        st->last_good = bad;
        st->tail_ok = 1;  // the last frame that was consumed advanced the state
        reason = RTVM_PENDING_MASK;
    } else if (st->n_done < n) {
        reason = st->n_done == 0 ? RTVM_PENDING_VOID : RTVM_PENDING_SKIPPED_FRAME;
    }
    st->bad_frame = 0x7FFFFFFF;
    rep->cleared = bad >= 0 && bad < n ? bad : -1;
    rep->n_done = st->n_done;
    rep->abort = st->abort;
    rep->tail_ok = st->tail_ok;
    rep->reason = reason;
    if (lists)
        for (int j = 0; j < n; j++) {
            rep->status[j] = st->b_status[j];
            rep->matches[j] = st->b_matches[j];
            rep->kp[j] = st->b_kp[j];
        }
}

// the report's per-frame lists, frame j by thread j (after batch_finish and a barrier)
__device__ __forceinline__ void batch_report_lists(const StreamState* st, int n, BatchReport* rep) {
    const int j = threadIdx.x;
    if (j < n) {
        rep->status[j] = st->b_status[j];
        rep->matches[j] = st->b_matches[j];
        rep->kp[j] = st->b_kp[j];
    }
}

// Publishes the frame pointers of a batch and the feature set its first frame is matched
// against to the device copies of the slots.
__global__ void k_batch_bind(FrameSlot* __restrict__ slots, int sb, int n, BatchPtrs ptrs, FeatSet prev) {
    const int j = threadIdx.x;
    if (j < n) slots[sb + j].bgr = ptrs.p[j];
    if (j == 0) slots[sb].prev = prev;
}

// Thread t of the copy: keypoint t / parts, 16-byte words part, part + parts, ... of its descriptor
// (parts = 1 for 32-byte descriptors: one thread per keypoint; 8 for the 512-byte ones).
__device__ __forceinline__ void copy_feat(const FeatSet& cur, const FeatSet& prev, int t, int n, int desc_bytes) {
    const int parts = desc_bytes > 64 ? 8 : 1;
    const int i = t / parts, part = t - i * parts;
    if (t == 0) *prev.count = n;
    if (i >= n) return;
    if (part == 0) {
        prev.xy[2 * i] = cur.xy[2 * i];
        prev.xy[2 * i + 1] = cur.xy[2 * i + 1];
        prev.size[i] = cur.size[i];
        prev.angle[i] = cur.angle[i];
        prev.response[i] = cur.response[i];
        prev.octave[i] = cur.octave[i];
        if (cur.lxy) {
            prev.lxy[2 * i] = cur.lxy[2 * i];
            prev.lxy[2 * i + 1] = cur.lxy[2 * i + 1];
        }
    }
    const uint4* s = reinterpret_cast<const uint4*>(cur.desc + (size_t)i * desc_bytes);
    uint4* d = reinterpret_cast<uint4*>(prev.desc + (size_t)i * desc_bytes);
    for (int w = part; w < desc_bytes / 16; w += parts) d[w] = s[w];
}

__global__ void k_batch_finish(StreamState* __restrict__ st, const StreamState* __restrict__ snap, int n,
                               BatchReport* __restrict__ rep, int host_glue) {
    if (threadIdx.x == 0 && blockIdx.x == 0) batch_finish(st, snap, n, rep, host_glue);
}

// The end of a batch and the hand-over of the features in one workgroup (contexts with 32-byte
// descriptors: a thousand keypoints are 100 KB; the 512-byte ones take k_advance_feats' grid).
__global__ __launch_bounds__(1024) void k_batch_finish_advance(StreamState* st, const StreamState* snap, int n, BatchReport* rep,
                                                               int host_glue, const FrameSlot* __restrict__ slots, int sb,
                                                               FeatSet prev, int max_kp, int desc_bytes) {
    if (threadIdx.x == 0) batch_finish(st, snap, n, rep, host_glue, false);
    __syncthreads();
    batch_report_lists(st, n, rep);
    const int j = st->last_good;
    if (j < 0) return;
    const FeatSet cur = slots[sb + j].feat;
    const int nk = min(*cur.count, max_kp);
    for (int t = threadIdx.x; t < max(nk, 1); t += 1024) copy_feat(cur, prev, t, nk, desc_bytes);
}

// kp_prev, des_prev <- features of the last frame of the batch that advanced the state
// (main.py:757-758); which == -1 reads the index from the stream state, otherwise it is
// the slot to copy.
__global__ __launch_bounds__(256) void k_advance_feats(const FrameSlot* __restrict__ slots, int sb,
                                                       const StreamState* __restrict__ st, int which, FeatSet prev,
                                                       int max_kp, int desc_bytes) {
    const int j = which >= 0 ? which : st->last_good;
    if (j < 0) return;
    const FeatSet cur = slots[sb + j].feat;
    const int n = min(*cur.count, max_kp);
    copy_feat(cur, prev, blockIdx.x * 256 + threadIdx.x, n, desc_bytes);
}

__global__ void k_state_init(StreamState* st, int row_off, int col_off, int* skip) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 9; i++) {
        const double e = (i % 4 == 0) ? 1. : 0.;
        st->H_old[i] = e;
        st->H_abs[i] = e;
        st->last_valid_H[i] = e;
    }
    st->H_old[2] = col_off;
    st->H_old[5] = row_off;
    st->H_abs[2] = col_off;
    st->H_abs[5] = row_off;
    st->hist_len = 0;
    st->status = 0;
    st->n_matches = 0;
    st->n_kp = 0;
    st->frame_index = 0;
    st->abort = 0;
    st->n_done = 0;
    st->last_good = -1;
    st->tail_ok = 1;
    st->bad_frame = 0x7FFFFFFF;
    for (int j = 0; j < RTVM_MAX_BATCH; j++) skip[j] = 0;
}

int need_stream_ctx(rtvm_ctx* c, const char* who) {
    if (c->fh <= 0 || c->fw <= 0 || c->ch <= 0 || c->cw <= 0)
        RTVM_FAIL(c, RTVM_E_STATE, "%s: context was created without frame/canvas size", who);
    if (c->detector != RTVM_DET_ORB && c->detector != RTVM_DET_SIFT)
        RTVM_FAIL(c, RTVM_E_STATE, "%s: the stream path needs a context created with RTVM_DET_ORB or RTVM_DET_SIFT", who);
    return RTVM_OK;
}

// Binds the frames of a batch (and the train side of its first frame) to slots sb .. sb+n-1
// on the current stream.
int bind_batch(rtvm_ctx* c, int sb, const uint8_t* const* ptrs, int n, const FeatSet& prev) {
    BatchPtrs bp;
    memset(&bp, 0, sizeof(bp));
    for (int j = 0; j < n; j++) {
        if (!ptrs[j]) RTVM_FAIL(c, RTVM_E_ARG, "null frame pointer at batch index %d", j);
        bp.p[j] = ptrs[j];
        c->h_slots[sb + j].bgr = ptrs[j];
    }
    // frames that came through the ingest ring: the stream that reads them waits for the copies
    if (c->ingest_used) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_ingest, 0));
    RTVM_LAUNCH(c, k_batch_bind, dim3(1), dim3(64), 0, c->d_slots, sb, n, bp, prev);
    return RTVM_OK;
}

// cv2 detectAndCompute of the context's detector for slots sb..sb+n-1 (main.py:111-112, 717-718)
int launch_detect(rtvm_ctx* c, int sb, int n) {
    return c->detector == RTVM_DET_SIFT ? rtvm_launch_sift(c, sb, n) : rtvm_launch_orb(c, sb, n);
}

// kp_prev, des_prev <- features of slot sb + which (which = -1: of the last frame that advanced the state)
void launch_advance_feats(rtvm_ctx* c, int sb, int which) {
    const int threads = c->orb.max_kp * (c->desc_bytes > 64 ? 8 : 1);
    RTVM_LAUNCH(c, k_advance_feats, dim3((threads + 255) / 256), dim3(256), 0, c->d_slots, sb, c->d_state, which,
                c->feat_prev, c->orb.max_kp, c->desc_bytes);
}

// detect + match against the previous frame + gather points for slots sb..sb+n-1
// prev_feat: the features the first frame is matched against are written on another stream; the event that says
// they are complete (detection itself does not need them)
int queue_features(rtvm_ctx* c, int sb, int n, hipEvent_t prev_feat = nullptr) {
    int r;
    if ((r = launch_detect(c, sb, n))) return r;
    if (prev_feat) HIP_TRY(c, hipStreamWaitEvent(c->stream, prev_feat, 0));
    const int cap = c->orb.max_kp;
    if (c->detector == RTVM_DET_SIFT) {
        if ((r = rtvm_launch_l2_slots(c, sb, n, cap))) return r;
    } else if ((r = rtvm_launch_hamming(c, sb, n, cap, cap, 1))) {
        return r;
    }
    RTVM_LAUNCH(c, k_gather_pts, dim3((cap + 255) / 256, n), dim3(256), 0, c->d_slots, sb);
    return RTVM_OK;
}

// ... + RANSAC: the whole front half on the current stream
int queue_front_half(rtvm_ctx* c, int sb, int n) {
    int r;
    if ((r = queue_features(c, sb, n))) return r;
    return rtvm_launch_ransac(c, sb, n, 2.0, 2000, 0.995);
}

int queue_first_frame(rtvm_ctx* c, const uint8_t* d_bgr, int row_off, int col_off) {
    int r;
    c->spec_tail = -1;
    c->seq = 0;
    for (int k = 0; k < RTVM_NSETS; k++) c->set_live[k] = 0, c->set_owner[k] = 0;
    if ((r = bind_batch(c, 0, &d_bgr, 1, c->feat_prev))) return r;
    if ((r = launch_detect(c, 0, 1))) return r;
    launch_advance_feats(c, 0, 0);
    if ((r = rtvm_launch_canvas_init(c, d_bgr, row_off, col_off))) return r;
    RTVM_LAUNCH(c, k_state_init, dim3(1), dim3(64), 0, c->d_state, row_off, col_off, c->d_skip);
    HIP_TRY(c, hipGetLastError());
    c->stream_ready = 1;
    return RTVM_OK;
}

// glue + warp/blend of every frame + kp_prev/des_prev update, on the main stream
// host_H: device-visible address of n absolute homographies supplied by the host, or null for the
// device-side glue.
int queue_back_half(rtvm_ctx* c, int set, int sb, const uint8_t* const* ptrs, int n, int ahead,
                    const double* host_H = nullptr) {
    int r;
    c->seq += 1;
    const int slot = (int)(c->seq % RTVM_REPORTS);
    // the report slot (and its event) served ticket seq - RTVM_REPORTS: that piece must have run before this one's
    // launches are queued behind the same event
    if (c->seq > RTVM_REPORTS) HIP_TRY(c, hipEventSynchronize(c->ev_done[slot]));
    if (c->set_owner[set] == c->seq) {  // the batch that owns the slot set (not a later piece of it)
        BatchRecord& rec = c->recs[set];
        rec.ticket = c->seq;
        rec.set = set;
        rec.sb = sb;
        rec.n = n;
        rec.ahead = ahead;
        for (int j = 0; j < n; j++) rec.ptrs[j] = ptrs[j];
    }
    if (host_H)
        RTVM_LAUNCH(c, k_batch_host_glue, dim3(1), dim3(64), 0, c->d_state, host_H, n, c->d_H, c->d_skip);
    else
        RTVM_LAUNCH(c, k_batch_glue, dim3(1), dim3(64), 0, c->d_slots, sb, n, c->d_state, c->d_wtab, c->d_H, c->d_skip,
                    50.0, 0.3, ahead, c->d_snap);
    if ((r = rtvm_launch_warp_prep(c, c->d_H, c->d_skip, n))) return r;
    if ((r = rtvm_launch_sides(c, ptrs, n))) return r;
    if ((r = rtvm_launch_weights(c, n))) return r;
    if ((r = rtvm_launch_commit(c, n))) return r;
    if (c->desc_bytes <= 64) {
        RTVM_LAUNCH(c, k_batch_finish_advance, dim3(1), dim3(1024), 0, c->d_state, (const StreamState*)c->d_snap, n, c->d_rep + slot,
                    host_H ? 1 : 0, (const FrameSlot*)c->d_slots, sb, c->feat_prev, c->orb.max_kp, c->desc_bytes);
    } else {
        RTVM_LAUNCH(c, k_batch_finish, dim3(1), dim3(64), 0, c->d_state, c->d_snap, n, c->d_rep + slot, host_H ? 1 : 0);
        launch_advance_feats(c, sb, -1);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_done[slot], c->stream_main));
    return RTVM_OK;
}

// The launch macros use c->stream; this points it at another stream for a scope.
struct StreamScope {
    rtvm_ctx* c;
    StreamScope(rtvm_ctx* ctx, hipStream_t s) : c(ctx) { c->stream = s; }
    ~StreamScope() { c->stream = c->stream_main; }
};

// One batch on the main stream only, in slot set 0, matched against feat_prev.
int queue_batch_serial(rtvm_ctx* c, const uint8_t* const* ptrs, int n) {
    int r;
    if (n < 1 || n > c->n_batch)
        RTVM_FAIL(c, RTVM_E_ARG, "batch of %d frames: reserve it first (rtvm_reserve_batch), at most %d", n,
                  RTVM_MAX_BATCH);
    if (c->set_live[0]) RTVM_FAIL(c, RTVM_E_STATE, "slot set 0 holds a batch that was not released (rtvm_batch_release)");
    c->spec_tail = -1;
    if ((r = bind_batch(c, 0, ptrs, n, c->feat_prev))) return r;
    if ((r = queue_front_half(c, 0, n))) return r;
    c->set_owner[0] = c->seq + 1;
    return queue_back_half(c, 0, 0, ptrs, n, 0);
}

// Detection + matching of a batch on the front stream, the RANSAC search and the RANSAC polish
// on streams of their own, in the next slot set.  Returns the set; *ahead_out tells whether the
// first frame was matched against the last frame queued before (not yet known to be accepted).
int queue_detect(rtvm_ctx* c, const uint8_t* const* ptrs, int n, int* set_out, int* ahead_out) {
    int r;
    if (n < 1 || n > c->n_batch || n > c->n_set1)
        RTVM_FAIL(c, RTVM_E_ARG, "batch of %d frames: reserve it first (rtvm_reserve_batch), at most %d", n,
                  RTVM_MAX_BATCH);
    int set = c->set_next;
    for (int k = 0; k < RTVM_NSETS && c->set_live[set]; k++) set = (set + 1) % RTVM_NSETS;
    if (c->set_live[set])
        RTVM_FAIL(c, RTVM_E_STATE, "all %d slot sets hold batches that were not released (rtvm_batch_release)",
                  RTVM_NSETS);
    const int sb = set * RTVM_MAX_BATCH;
    const int ahead = c->spec_tail >= 0 ? 1 : 0;
    // the batches of the odd slot sets are detected on a stream of their own where there is one
    hipStream_t sf = (set & 1) ? c->stream_front2 : c->stream_front;
    hipEvent_t prev_feat = nullptr;
    if (ahead) {
        // the batch that used this slot set before must have left it
        if (c->back_rec[set]) HIP_TRY(c, hipStreamWaitEvent(sf, c->ev_back[set], 0));
        // the frame the first one is matched against was detected on the other stream
        const int prev_set = c->spec_tail / RTVM_MAX_BATCH;
        if (c->stream_front2 != c->stream_front && ((prev_set ^ set) & 1)) prev_feat = c->ev_feat[prev_set];
    } else {
        // feat_prev is written on the main stream: wait for everything queued there
        HIP_TRY(c, hipEventRecord(c->ev_main, c->stream_main));
        HIP_TRY(c, hipStreamWaitEvent(sf, c->ev_main, 0));
    }
    {
        StreamScope fs(c, sf);
        if ((r = bind_batch(c, sb, ptrs, n, ahead ? c->h_slots[c->spec_tail].feat : c->feat_prev))) return r;
        if ((r = queue_features(c, sb, n, prev_feat))) return r;
        HIP_TRY(c, hipEventRecord(c->ev_feat[set], sf));
    }
    // Few frames in flight (the start of a video, a short burst): latency counts, not work, and the RANSAC
    // search starts with all 64 hypotheses of its first two stages at once (ransac.hip).  (Tried for such
    // bursts and dropped: two half batches, the second with search and polish on the front stream so that
    // the two polishes run side by side -- the search and polish streams share one hardware queue -- : the
    // fixed costs of detection and of a canvas piece come twice, 2 % in all.)
    int live = n;
    for (int k = 0; k < RTVM_NSETS; k++)
        if (c->set_live[k]) live += c->recs[k].n;
    // contexts with two detection streams keep a batch on the stream of its detection all the way through RANSAC
    const bool own = c->stream_front2 != c->stream_front;
    hipStream_t s_rs = own ? sf : c->stream_rs, s_lm = own ? sf : c->stream_lm;
    HIP_TRY(c, hipStreamWaitEvent(s_rs, c->ev_feat[set], 0));
    {
        StreamScope fs(c, s_rs);
        c->rs_alone = live <= RTVM_MAX_BATCH ? 1 : 0;
        r = rtvm_launch_ransac_search(c, sb, n, 2.0, 2000, 0.995);
        c->rs_alone = 0;
        if (r) return r;
        HIP_TRY(c, hipEventRecord(c->ev_search[set], s_rs));
    }
    HIP_TRY(c, hipStreamWaitEvent(s_lm, c->ev_search[set], 0));
    {
        StreamScope fs(c, s_lm);
        if ((r = rtvm_launch_ransac_refine(c, sb, n, 2.0))) return r;
        RTVM_LAUNCH(c, k_batch_collect, dim3(1), dim3(64), 0, c->d_slots, sb, n, c->d_res + set);
        HIP_TRY(c, hipEventRecord(c->ev_front[set], s_lm));
    }
    c->spec_tail = sb + n - 1;
    c->set_next = (set + 1) % RTVM_NSETS;
    *set_out = set;
    *ahead_out = ahead;
    return RTVM_OK;
}

// One batch: detection etc. on the side streams, then the canvas half on the main stream.
int queue_batch_overlapped(rtvm_ctx* c, const uint8_t* const* ptrs, int n) {
    int r, set = 0, ahead = 0;
    if ((r = queue_detect(c, ptrs, n, &set, &ahead))) return r;
    HIP_TRY(c, hipStreamWaitEvent(c->stream_main, c->ev_front[set], 0));
    c->set_owner[set] = c->seq + 1;
    if ((r = queue_back_half(c, set, set * RTVM_MAX_BATCH, ptrs, n, ahead))) return r;
    HIP_TRY(c, hipEventRecord(c->ev_back[set], c->stream_main));
    c->back_rec[set] = 1;
    return RTVM_OK;
}

}  // namespace

// weights of smooth_homography: np.linspace(0.5, 1.0, n) / sum, n = 2..5 (row n of a 6x5 table)
int rtvm_stream_setup(rtvm_ctx* c) {
    double w[6 * 5];
    for (int i = 0; i < 30; i++) w[i] = 0;
    for (int n = 2; n <= 5; n++) {
        const double step = (1.0 - 0.5) / (n - 1);
        double v[5], sum = 0;
        for (int i = 0; i < n; i++) v[i] = i * step + 0.5;
        v[n - 1] = 1.0;
        for (int i = 0; i < n; i++) sum += v[i];
        for (int i = 0; i < n; i++) w[n * 5 + i] = v[i] / sum;
    }
    HIP_TRY(c, hipMemcpy(c->d_wtab, w, sizeof(w), hipMemcpyHostToDevice));
    return RTVM_OK;
}

extern "C" int rtvm_validate_homography(rtvm_ctx* c, const double H[9], double translation_threshold,
                                        double scale_threshold, int* ok) {
    if (!c) return RTVM_E_ARG;
    if (!H || !ok) RTVM_FAIL(c, RTVM_E_ARG, "validate_homography: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    double* hp = (double*)c->h_pinned + 80;
    memcpy(hp, H, 9 * sizeof(double));
    double* dH = c->d_H + RTVM_MAX_BATCH * 9;   // the spare entries behind the batch's homographies
    int* d_ok = (int*)(dH + 12);
    HIP_TRY(c, hipMemcpyAsync(dH, hp, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_validate, dim3(1), dim3(64), 0, dH, translation_threshold, scale_threshold, d_ok);
    HIP_TRY(c, hipGetLastError());
    int* hr = (int*)(hp + 12);
    HIP_TRY(c, hipMemcpyAsync(hr, d_ok, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *ok = *hr;
    return RTVM_OK;
}

extern "C" int rtvm_stream_begin(rtvm_ctx* c, const uint8_t* bgr_dev, int row_off, int col_off) {
    if (!c) return RTVM_E_ARG;
    if (!bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "stream_begin: null frame");
    int r = need_stream_ctx(c, "stream_begin");
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ingest_used) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_ingest, 0));
    return queue_first_frame(c, bgr_dev, row_off, col_off);
}

extern "C" int rtvm_first_frame(rtvm_ctx* c, const uint8_t* bgr, int row_off, int col_off) {
    if (!c) return RTVM_E_ARG;
    if (!bgr) RTVM_FAIL(c, RTVM_E_ARG, "first_frame: null frame");
    int r = need_stream_ctx(c, "first_frame");
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    if ((r = queue_first_frame(c, c->d_bgr, row_off, col_off))) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_process_frame(rtvm_ctx* c, const uint8_t* bgr_dev) {
    if (!c) return RTVM_E_ARG;
    int r = need_stream_ctx(c, "process_frame");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "process_frame: call rtvm_stream_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    return queue_batch_serial(c, &bgr_dev, 1);
}

extern "C" int rtvm_process_batch(rtvm_ctx* c, const uint8_t* const* bgr_dev, int n) {
    if (!c) return RTVM_E_ARG;
    if (!bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "process_batch: null argument");
    int r = need_stream_ctx(c, "process_batch");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "process_batch: call rtvm_stream_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n >= 1 && n <= c->n_set1 && !(c->flags & RTVM_FLAG_SINGLE_STREAM)) return queue_batch_overlapped(c, bgr_dev, n);
    return queue_batch_serial(c, bgr_dev, n);
}

static int read_state(rtvm_ctx* c, StreamState** hs_out) {
    StreamState* hs = (StreamState*)((char*)c->h_pinned + 4096);
    HIP_TRY(c, hipMemcpyAsync(hs, c->d_state, sizeof(StreamState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *hs_out = hs;
    return RTVM_OK;
}

extern "C" long rtvm_batch_ticket(rtvm_ctx* c) { return c ? c->seq : 0; }

extern "C" int rtvm_batch_cleared(rtvm_ctx* c, long ticket, int* first) {
    if (!c) return RTVM_E_ARG;
    if (!first) RTVM_FAIL(c, RTVM_E_ARG, "batch_cleared: null argument");
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "batch_cleared: no stream");
    if (ticket < 1 || ticket > c->seq || c->seq - ticket >= RTVM_REPORTS)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_cleared: ticket %ld is not one of the last %d (%ld queued)", ticket, RTVM_REPORTS,
                  c->seq);
    HIP_TRY(c, hipSetDevice(c->device));
    const int slot = (int)(ticket % RTVM_REPORTS);
    HIP_TRY(c, hipEventSynchronize(c->ev_done[slot]));
    *first = c->h_rep[slot].cleared;
    return RTVM_OK;
}

extern "C" int rtvm_batch_status_of(rtvm_ctx* c, long ticket, int* n_done, int* reason, int* status,
                                    int* n_matches, int* n_keypoints, int cap) {
    if (!c) return RTVM_E_ARG;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "batch_status: no stream");
    if (ticket < 1 || ticket > c->seq || c->seq - ticket >= RTVM_REPORTS)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_status: ticket %ld is not one of the last %d (%ld queued)", ticket,
                  RTVM_REPORTS, c->seq);
    HIP_TRY(c, hipSetDevice(c->device));
    const int slot = (int)(ticket % RTVM_REPORTS);
    HIP_TRY(c, hipEventSynchronize(c->ev_done[slot]));
    const BatchReport* rep = c->h_rep + slot;
    if (n_done) *n_done = rep->n_done;
    if (reason) *reason = rep->reason;
    for (int j = 0; j < cap && j < RTVM_MAX_BATCH; j++) {
        if (status) status[j] = rep->status[j];
        if (n_matches) n_matches[j] = rep->matches[j];
        if (n_keypoints) n_keypoints[j] = rep->kp[j];
    }
    if (rep->abort || !rep->tail_ok) {
        // Frames were left pending, or the last frame did not advance the state: whatever
        // canvas half was queued after this batch is pending as a whole.  Drain it; the
        // caller now knows.
        HIP_TRY(c, hipStreamSynchronize(c->stream_main));
        HIP_TRY(c, hipMemsetAsync(&c->d_state->abort, 0, sizeof(int), c->stream_main));
        // After a skipped frame the frames detected behind it were matched against the wrong
        // frame: what is detected next is matched against the last accepted frame.  A
        // mispredicted mask leaves the detection results intact; a caller that keeps detected
        // batches (rtvm_batch_detect) goes on from them, any other is expected to submit the
        // pending frames again, which are then matched against the last accepted frame too.
        bool live = false;
        for (int k = 0; k < RTVM_NSETS; k++) live |= c->set_live[k] != 0;
        if (!live || rep->abort == 2 || rep->reason == RTVM_PENDING_SKIPPED_FRAME ||
            (rep->n_done > 0 && !rep->tail_ok))
            c->spec_tail = -1;
    }
    return RTVM_OK;
}

// back = 0: the batch queued last; back = 1, 2: the ones before it (waits for that batch
// only, so the caller can keep two batches in flight while it reads the outcome of a third).
extern "C" int rtvm_batch_status_at(rtvm_ctx* c, int back, int* n_done, int* status, int* n_matches,
                                    int* n_keypoints, int cap) {
    if (!c) return RTVM_E_ARG;
    if (back < 0 || back > 2 || c->seq - back < 1)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_status: no batch %d before the last one (%ld queued)", back, c->seq);
    return rtvm_batch_status_of(c, c->seq - back, n_done, nullptr, status, n_matches, n_keypoints, cap);
}

extern "C" int rtvm_batch_status(rtvm_ctx* c, int* n_done, int* status, int* n_matches, int* n_keypoints, int cap) {
    return rtvm_batch_status_at(c, 0, n_done, status, n_matches, n_keypoints, cap);
}

extern "C" int rtvm_batch_retry(rtvm_ctx* c, long ticket, int first, int count) {
    if (!c) return RTVM_E_ARG;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "batch_retry: no stream");
    int owner = -1;
    for (int k = 0; k < RTVM_NSETS; k++)
        if (ticket >= 1 && c->set_owner[k] == ticket && c->recs[k].ticket == ticket) owner = k;
    if (owner < 0)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_retry: the detection results of batch %ld are not held (any more)", ticket);
    const BatchRecord rec = c->recs[owner];
    if (first < 0 || count < 1 || first + count > rec.n)
        RTVM_FAIL(c, RTVM_E_ARG, "batch_retry: frames %d..%d of a batch of %d", first, first + count - 1, rec.n);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamWaitEvent(c->stream_main, c->ev_front[rec.set], 0));  // detection .. RANSAC of the batch
    // The frame before `first` must be the last one that advanced the state (the glue checks
    // it), unless the batch's first frame was matched against feat_prev.
    int r = queue_back_half(c, rec.set, rec.sb + first, rec.ptrs + first, count, first == 0 ? rec.ahead : 1);
    if (r) return r;
    // whoever refills this slot set waits for the retry as well
    HIP_TRY(c, hipEventRecord(c->ev_back[rec.set], c->stream_main));
    c->back_rec[rec.set] = 1;
    return RTVM_OK;
}

extern "C" int rtvm_batch_detect(rtvm_ctx* c, const uint8_t* const* bgr_dev, int n) {
    if (!c) return RTVM_E_ARG;
    if (!bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "batch_detect: null argument");
    int r = need_stream_ctx(c, "batch_detect");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "batch_detect: call rtvm_stream_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    int set = 0, ahead = 0;
    if ((r = queue_detect(c, bgr_dev, n, &set, &ahead))) return r;
    c->seq += 1;
    BatchRecord& rec = c->recs[set];
    rec.ticket = c->seq;
    rec.set = set;
    rec.sb = set * RTVM_MAX_BATCH;
    rec.n = n;
    rec.ahead = ahead;
    for (int j = 0; j < n; j++) rec.ptrs[j] = bgr_dev[j];
    c->set_owner[set] = c->seq;
    c->set_live[set] = 1;
    // no canvas half, no report: the event of this ticket's report slot is simply due now
    HIP_TRY(c, hipEventRecord(c->ev_done[c->seq % RTVM_REPORTS], c->stream_main));
    return RTVM_OK;
}

// the slot set that holds the detection results of batch `ticket`, or -1
static int held_set(const rtvm_ctx* c, long ticket) {
    for (int k = 0; k < RTVM_NSETS; k++)
        if (ticket >= 1 && c->set_owner[k] == ticket && c->recs[k].ticket == ticket) return k;
    return -1;
}

extern "C" int rtvm_batch_results_ready(rtvm_ctx* c, long ticket) {
    if (!c) return RTVM_E_ARG;
    const int set = held_set(c, ticket);
    if (set < 0) RTVM_FAIL(c, RTVM_E_STATE, "batch_results: the detection results of batch %ld are not held (any more)", ticket);
    HIP_TRY(c, hipSetDevice(c->device));
    const hipError_t e = hipEventQuery(c->ev_front[set]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    HIP_TRY(c, e);
    return 0;
}

extern "C" int rtvm_batch_results(rtvm_ctx* c, long ticket, int* n_matches, int* found, int* n_keypoints,
                                  double* H_rel, int cap) {
    if (!c) return RTVM_E_ARG;
    const int set = held_set(c, ticket);
    if (set < 0) RTVM_FAIL(c, RTVM_E_STATE, "batch_results: the detection results of batch %ld are not held (any more)", ticket);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_front[set]));
    const BatchResults* res = c->h_res + set;
    const int n = c->recs[set].n;
    for (int j = 0; j < cap && j < n; j++) {
        if (n_matches) n_matches[j] = res->matches[j];
        if (found) found[j] = res->found[j];
        if (n_keypoints) n_keypoints[j] = res->kp[j];
        if (H_rel) memcpy(H_rel + 9 * j, res->H[j], 9 * sizeof(double));
    }
    return RTVM_OK;
}

extern "C" int rtvm_batch_commit(rtvm_ctx* c, long ticket, int first, int count, const double* H_abs) {
    if (!c) return RTVM_E_ARG;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "batch_commit: no stream");
    if (!H_abs) RTVM_FAIL(c, RTVM_E_ARG, "batch_commit: null homographies");
    const int set = held_set(c, ticket);
    if (set < 0) RTVM_FAIL(c, RTVM_E_STATE, "batch_commit: the detection results of batch %ld are not held (any more)", ticket);
    const BatchRecord rec = c->recs[set];
    if (first < 0 || count < 1 || first + count > rec.n)
        RTVM_FAIL(c, RTVM_E_ARG, "batch_commit: frames %d..%d of a batch of %d", first, first + count - 1, rec.n);
    HIP_TRY(c, hipSetDevice(c->device));
    // the homographies wait in pinned memory of the report slot this piece will get (queue_back_half
    // counts the ticket up); the device reads them from there
    const int slot = (int)((c->seq + 1) % RTVM_REPORTS);
    // ... which the piece RTVM_REPORTS tickets ago used: its glue kernel must have read its homographies first
    if (c->seq + 1 > RTVM_REPORTS) HIP_TRY(c, hipEventSynchronize(c->ev_done[slot]));
    double* hH = c->h_commit_H + (size_t)slot * RTVM_MAX_BATCH * 9;
    memcpy(hH, H_abs, (size_t)count * 9 * sizeof(double));
    HIP_TRY(c, hipStreamWaitEvent(c->stream_main, c->ev_front[rec.set], 0));  // the frames and their features
    int r = queue_back_half(c, rec.set, rec.sb + first, rec.ptrs + first, count, 0,
                            c->d_commit_H + (size_t)slot * RTVM_MAX_BATCH * 9);
    if (r) return r;
    HIP_TRY(c, hipEventRecord(c->ev_back[rec.set], c->stream_main));
    c->back_rec[rec.set] = 1;
    return RTVM_OK;
}

extern "C" int rtvm_batch_done(rtvm_ctx* c, long ticket) {
    if (!c) return RTVM_E_ARG;
    if (ticket < 1 || ticket > c->seq || c->seq - ticket >= RTVM_REPORTS)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_done: ticket %ld is not one of the last %d (%ld queued)", ticket, RTVM_REPORTS, c->seq);
    HIP_TRY(c, hipSetDevice(c->device));
    const hipError_t e = hipEventQuery(c->ev_done[ticket % RTVM_REPORTS]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    HIP_TRY(c, e);
    return 0;
}

extern "C" int rtvm_batch_rebase(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    c->spec_tail = -1;
    return RTVM_OK;
}

extern "C" int rtvm_batch_release(rtvm_ctx* c, long ticket) {
    if (!c) return RTVM_E_ARG;
    for (int k = 0; k < RTVM_NSETS; k++)
        if (c->set_owner[k] == ticket) c->set_live[k] = 0;
    return RTVM_OK;
}

extern "C" int rtvm_frame_status(rtvm_ctx* c, int* status, int* n_keypoints, int* n_matches, double H_abs[9]) {
    if (!c) return RTVM_E_ARG;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_status: no stream");
    HIP_TRY(c, hipSetDevice(c->device));
    StreamState* hs;
    int r = read_state(c, &hs);
    if (r) return r;
    if (status) *status = hs->status;
    if (n_keypoints) *n_keypoints = hs->n_kp;
    if (n_matches) *n_matches = hs->n_matches;
    if (H_abs) memcpy(H_abs, hs->H_abs, 9 * sizeof(double));
    return RTVM_OK;
}

// ---- two-call form used by the Python VideMosaic: the glue between RANSAC and the
// warp (validate, smooth, chain; main.py:733-746) stays in the caller's numpy -------

extern "C" int rtvm_frame_register(rtvm_ctx* c, const uint8_t* bgr, int* n_matches, int* found, double H_rel[9]) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !n_matches || !found || !H_rel) RTVM_FAIL(c, RTVM_E_ARG, "frame_register: null argument");
    int r = need_stream_ctx(c, "frame_register");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_register: call rtvm_first_frame first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    const uint8_t* p = c->d_bgr;
    c->spec_tail = -1;
    if ((r = bind_batch(c, 0, &p, 1, c->feat_prev))) return r;
    if ((r = queue_front_half(c, 0, 1))) return r;
    const FrameSlot& S = c->h_slots[0];
    int* hc = (int*)c->h_pinned;
    double* hH = (double*)c->h_pinned + 8;
    HIP_TRY(c, hipMemcpyAsync(hc, S.m_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hc + 1, S.rs_found, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hH, S.rs_H, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_matches = hc[0];
    *found = hc[0] >= 4 ? hc[1] : 0;
    memcpy(H_rel, hH, 9 * sizeof(double));
    return RTVM_OK;
}

extern "C" int rtvm_stream_set_prev(rtvm_ctx* c, const uint8_t* bgr) {
    if (!c) return RTVM_E_ARG;
    if (!bgr) RTVM_FAIL(c, RTVM_E_ARG, "stream_set_prev: null frame");
    int r = need_stream_ctx(c, "stream_set_prev");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "stream_set_prev: call rtvm_first_frame first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    const uint8_t* p = c->d_bgr;
    c->spec_tail = -1;
    if ((r = bind_batch(c, 0, &p, 1, c->feat_prev))) return r;
    if ((r = launch_detect(c, 0, 1))) return r;
    launch_advance_feats(c, 0, 0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_frame_commit(rtvm_ctx* c, const double H[9]) {
    if (!c) return RTVM_E_ARG;
    if (!H) RTVM_FAIL(c, RTVM_E_ARG, "frame_commit: null homography");
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_commit: no registered frame");
    HIP_TRY(c, hipSetDevice(c->device));
    // No sync: the warp and blend run while the host fetches the next frame.  Two pinned staging
    // slots for H taken in turn are enough, because rtvm_frame_register (which every commit
    // follows) ends with a sync of this stream and so leaves at most one commit in flight.
    double* hp = (double*)c->h_pinned + 32 + 16 * (c->commit_parity ^= 1);
    memcpy(hp, H, 9 * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_H, hp, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int r;
    if ((r = rtvm_launch_warp_blend(c, c->d_bgr, c->d_H))) return r;
    launch_advance_feats(c, 0, 0);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

static int read_feat(rtvm_ctx* c, const FeatSet& F, int* n_out, float* xy, float* size, float* angle_deg,
                     float* response, int32_t* octave, uint8_t* desc) {
    int* hc = (int*)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(hc, F.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n = hc[0];
    if (n < 0 || n > c->orb.max_kp) RTVM_FAIL(c, RTVM_E_HIP, "device returned keypoint count %d", n);
    if (n) {
        if (xy) HIP_TRY(c, hipMemcpyAsync(xy, F.xy, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        if (size) HIP_TRY(c, hipMemcpyAsync(size, F.size, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (angle_deg) HIP_TRY(c, hipMemcpyAsync(angle_deg, F.angle, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (response) HIP_TRY(c, hipMemcpyAsync(response, F.response, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (octave) HIP_TRY(c, hipMemcpyAsync(octave, F.octave, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (desc) HIP_TRY(c, hipMemcpyAsync(desc, F.desc, (size_t)n * c->desc_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return RTVM_OK;
}

// Features of the current (slot 0) or previous (slot 1) frame of the stream.
extern "C" int rtvm_get_features(rtvm_ctx* c, int slot, int* n_out, float* xy, float* size, float* angle_deg,
                                 float* response, int32_t* octave, uint8_t* desc) {
    if (!c) return RTVM_E_ARG;
    if (!n_out || slot < 0 || slot > 1) RTVM_FAIL(c, RTVM_E_ARG, "get_features: bad argument");
    if (c->fh <= 0) RTVM_FAIL(c, RTVM_E_STATE, "get_features: context has no frame size");
    HIP_TRY(c, hipSetDevice(c->device));
    return read_feat(c, slot == 0 ? c->h_slots[0].feat : c->feat_prev, n_out, xy, size, angle_deg, response, octave,
                     desc);
}

extern "C" int rtvm_batch_get_features(rtvm_ctx* c, long ticket, int index, int* n_out, float* xy, float* size,
                                       float* angle_deg, float* response, int32_t* octave, uint8_t* desc) {
    if (!c) return RTVM_E_ARG;
    if (!n_out) RTVM_FAIL(c, RTVM_E_ARG, "batch_get_features: null argument");
    const int set = held_set(c, ticket);
    if (set < 0 || index < 0 || index >= c->recs[set].n)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_get_features: frame %d of batch %ld is not held", index, ticket);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_front[set]));
    return read_feat(c, c->h_slots[c->recs[set].sb + index].feat, n_out, xy, size, angle_deg, response, octave, desc);
}

// cv2.ORB detectAndCompute of a host frame (stage entry point): slot 0's features.
int rtvm_orb_host(rtvm_ctx* c, const uint8_t* bgr, int* n_out, float* xy, float* size, float* angle_deg,
                  float* response, int32_t* octave, uint8_t* desc, uint8_t* gray) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    const uint8_t* p = c->d_bgr;
    int r;
    c->spec_tail = -1;
    if ((r = bind_batch(c, 0, &p, 1, c->feat_prev))) return r;
    if ((r = rtvm_launch_orb(c, 0, 1))) return r;
    if (gray) {
        HIP_TRY(c, hipMemcpy2DAsync(gray, c->fw, c->h_slots[0].pyr + c->orb.off[0], c->orb.pitch[0], c->fw, c->fh,
                                    hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RTVM_OK;
    }
    return read_feat(c, c->h_slots[0].feat, n_out, xy, size, angle_deg, response, octave, desc);
}

static int read_matches(rtvm_ctx* c, const FrameSlot& S, int* m_out, int32_t* qidx, int32_t* tidx, int32_t* dist);

// Matches of the most recent frame against its predecessor, sorted by distance.
extern "C" int rtvm_get_matches(rtvm_ctx* c, int* m_out, int32_t* qidx, int32_t* tidx, int32_t* dist) {
    if (!c) return RTVM_E_ARG;
    if (!m_out) RTVM_FAIL(c, RTVM_E_ARG, "get_matches: null argument");
    if (c->fh <= 0) RTVM_FAIL(c, RTVM_E_STATE, "get_matches: context has no frame size");
    HIP_TRY(c, hipSetDevice(c->device));
    return read_matches(c, c->h_slots[0], m_out, qidx, tidx, dist);
}

extern "C" int rtvm_batch_get_matches(rtvm_ctx* c, long ticket, int index, int* m_out, int32_t* qidx, int32_t* tidx,
                                      int32_t* dist) {
    if (!c) return RTVM_E_ARG;
    if (!m_out) RTVM_FAIL(c, RTVM_E_ARG, "batch_get_matches: null argument");
    const int set = held_set(c, ticket);
    if (set < 0 || index < 0 || index >= c->recs[set].n)
        RTVM_FAIL(c, RTVM_E_STATE, "batch_get_matches: frame %d of batch %ld is not held", index, ticket);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_front[set]));
    return read_matches(c, c->h_slots[c->recs[set].sb + index], m_out, qidx, tidx, dist);
}

static int read_matches(rtvm_ctx* c, const FrameSlot& S, int* m_out, int32_t* qidx, int32_t* tidx, int32_t* dist) {
    int* hc = (int*)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(hc, S.m_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int m = hc[0];
    if (m < 0 || m > RTVM_MAX_MATCH) RTVM_FAIL(c, RTVM_E_HIP, "get_matches: device returned count %d", m);
    if (m) {
        if (qidx) HIP_TRY(c, hipMemcpyAsync(qidx, S.m_qidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (tidx) HIP_TRY(c, hipMemcpyAsync(tidx, S.m_tidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (dist) HIP_TRY(c, hipMemcpyAsync(dist, S.m_dist, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *m_out = m;
    return RTVM_OK;
}
