/*
 * rtvm.h - C ABI of librtvm.so, the MI355X (gfx950) stitching hot path.
 *
 * The reference (PROcessorI/Real-Time-Video-Mosaic) has no FFI of its own: the
 * boundary of its hot path is the Python class VideMosaic (main.py:15-977),
 * whose arithmetic is delegated to OpenCV calls.  Every entry point below
 * replaces one of those calls (or the numpy statements around it); the
 * file:line in each comment is the call site in the reference's main.py that
 * a maintainer would re-point at this library (through ctypes.CDLL).
 *
 * Conventions
 *   - return value 0 = RTVM_OK, negative = error code; no C++ exception
 *     crosses this boundary; the message of the last failure on a context is
 *     returned by rtvm_last_error().
 *   - every array argument is caller-owned, C-contiguous, and sized as the
 *     comment states.  Pointers are host pointers unless the entry point says
 *     "device".  Nothing returned has to be freed by the caller.
 *   - a context is single-threaded (one video stream on one device; it owns four
 *     HIP streams, or one with RTVM_FLAG_SINGLE_STREAM or without a frame size).
 *     Distinct contexts may be used from distinct threads/processes.
 *   - this library contains no CPU fallback: with no usable HIP device
 *     rtvm_create() fails with RTVM_E_HIP.
 */
#ifndef RTVM_H
#define RTVM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTVM_VERSION 100 /* 0.1.0 */

enum {
    RTVM_OK = 0,
    RTVM_E_ARG = -1,        /* bad argument (null pointer, size out of range) */
    RTVM_E_HIP = -2,        /* HIP runtime error, or no device */
    RTVM_E_NOMEM = -3,      /* device or host allocation failed */
    RTVM_E_DEGENERATE = -4, /* geometry is degenerate (singular H, ...) */
    RTVM_E_STATE = -5,      /* call order violated (e.g. warp before canvas_init) */
    RTVM_E_CALLBACK = -6    /* a function the caller supplied (rtvm_glue_fn, rtvm_frame_source) reported a failure */
};

/* rtvm_create: detector */
enum {
    RTVM_DET_ORB = 0,             /* ORB_create(n) + BFMatcher(NORM_HAMMING, crossCheck=True)  main.py:35-37 */
    RTVM_DET_SIFT_MATCH_ONLY = 1, /* stage calls of the SIFT branch only (rtvm_sift, rtvm_match_l2_knn2, ...) */
    RTVM_DET_SIFT = 2             /* SIFT_create(n) + BFMatcher() knn-2 + ratio 0.7  main.py:32-34, 688-692: the
                                     stream path (rtvm_first_frame, rtvm_stream_begin, rtvm_process_*, rtvm_batch_*)
                                     detects, matches and fits with the reference's default detector; features are
                                     128 floats per keypoint and match distances are floats */
};

/* rtvm_create: flags */
enum {
    RTVM_FLAG_WARP_FIXED = 1 << 0, /* warpPerspective arithmetic of OpenCV <= 4.10
                                      (1/32-pixel fixed point); default is the
                                      fp32 arithmetic of OpenCV >= 4.11 */
    RTVM_FLAG_MATCH_LEGACY = 1 << 1, /* cross-check of BFMatcher before OpenCV 4.2 / 3.4.9: a query is matched
                                        with the nearest of the train rows whose nearest query it is, whether or
                                        not that row is the query's own nearest; default is the mutual rule */
    RTVM_FLAG_GRAY_LEGACY = 1 << 3,  /* cvtColor(BGR2GRAY) with the 14-bit coefficients of OpenCV <= 3.x
                                        ((B*1868 + G*9617 + R*4899 + 8192) >> 14); default the 15-bit ones of 4.x */
    RTVM_FLAG_SINGLE_STREAM = 1 << 2 /* the context owns one HIP stream: rtvm_process_batch
                                        queues everything on it (no overlap of consecutive
                                        batches).  Also the choice for contexts that only serve
                                        stage calls from a thread of their own: a process has few
                                        hardware queues and every stream takes a share of one */
};

/* per-frame status bits reported by rtvm_process_frame / rtvm_frame_status */
enum {
    RTVM_FRAME_OK = 0,
    RTVM_FRAME_FEW_MATCHES = 1, /* < 4 matches: frame skipped, state not advanced (main.py:722-724) */
    RTVM_FRAME_NO_HOMOGRAPHY = 2, /* RANSAC found no model: skipped (main.py:729-731) */
    RTVM_FRAME_INVALID_H = 4,   /* validate_homography() failed: identity used (main.py:734-737) */
    RTVM_FRAME_PENDING = 8      /* not processed: an earlier frame of the same batch was skipped */
};

/* why frames of a batch were left RTVM_FRAME_PENDING (rtvm_batch_status_of) */
enum {
    RTVM_PENDING_NONE = 0,
    RTVM_PENDING_SKIPPED_FRAME = 1, /* the frame before them was skipped: they were matched against it */
    RTVM_PENDING_MASK = 2,          /* a blend cleared a canvas pixel: the canvas masks predicted for them are wrong;
                                       their detection, matching and RANSAC results remain valid */
    RTVM_PENDING_VOID = 3           /* the batch was queued behind one that left frames pending */
};

typedef struct rtvm_ctx rtvm_ctx;

/* ---- life cycle ------------------------------------------------------- */

int rtvm_version(void);

/* Replaces VideMosaic.__init__'s detector/matcher/canvas set-up (main.py:29-37, 80-81).
 * frame_h/frame_w: size of every frame of the stream (0 x 0 for a context that only
 * matches or fits; otherwise at least 2 x 2); canvas_h/canvas_w: mosaic
 * canvas (main.py:80-81 computes int(oh*H) x int(ow*W)); nfeatures: ORB_create(n)
 * (main.py:36).  All device memory the context will ever use is allocated here.
 * A context with a frame size (and without RTVM_FLAG_SINGLE_STREAM) owns several HIP streams and chooses them by
 * the hardware queue they sit on: it creates a few candidates, launches pairs of one-lane kernels of at most 200 us
 * to find out which candidates share a queue, keeps one per role and destroys the rest (about 2 ms; environment:
 * RTVM_NO_QUEUE_PROBE=1 takes the streams as the runtime deals them, RTVM_QUEUE_DEBUG=1 reports on stderr). */
int rtvm_create(rtvm_ctx** out, int device, int frame_h, int frame_w, int canvas_h, int canvas_w,
                int detector, int nfeatures, int flags);
void rtvm_destroy(rtvm_ctx* ctx);
const char* rtvm_last_error(const rtvm_ctx* ctx);
/* Waits for all work queued on the context's stream. */
int rtvm_sync(rtvm_ctx* ctx);
/* Upper bound of keypoints rtvm_orb can return (nfeatures plus room for ties at the
 * retainBest cut); size the output arrays of rtvm_orb with it. */
int rtvm_max_keypoints(const rtvm_ctx* ctx);

/* ---- stage entry points (host buffers in, host buffers out) ----------- */

/* cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)  main.py:111,717.
 * bgr: frame_h*frame_w*3 u8; gray: frame_h*frame_w u8. */
int rtvm_gray(rtvm_ctx* ctx, const uint8_t* bgr, uint8_t* gray);

/* cv2.cvtColor + cv2.ORB_create(n).detectAndCompute(gray, None)  main.py:36,111-112,717-718.
 * Outputs (capacity rtvm_max_keypoints()): xy n*2 f32 (level-0 pixel coordinates),
 * size, angle_deg, response n f32, octave n i32, desc n*32 u8.  Keypoints are
 * ordered by pyramid level, then row-major inside a level. */
int rtvm_orb(rtvm_ctx* ctx, const uint8_t* bgr, int* n_out, float* xy, float* size,
             float* angle_deg, float* response, int32_t* octave, uint8_t* desc);

/* cv2.cvtColor + cv2.SIFT_create(nfeatures).detectAndCompute(gray, None)  main.py:33,111-112,717-718
 * (the reference's default detector) on a host frame of h x w x 3 u8, OpenCV's defaults for the
 * other SIFT parameters.  Outputs (capacity rtvm_sift_max_keypoints(nfeatures)): xy n*2 f32,
 * size, angle_deg, response n f32, octave n i32 (OpenCV's packed octave | layer << 8 | ...),
 * desc n*128 f32 (integer-valued 0..255).  Keypoints come sorted by (x, y), duplicates removed,
 * the nfeatures strongest kept (ties included).  Allocates its pyramids on the first call (and
 * again for a larger frame); syncs.  Frames 16..8192 pixels a side; RTVM_E_NOMEM when a frame
 * has more than 262144 scale-space extrema or 65536 keypoints before the filtering. */
int rtvm_sift_max_keypoints(int nfeatures);
int rtvm_sift(rtvm_ctx* ctx, const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy,
              float* size, float* angle_deg, float* response, int32_t* octave, float* desc);

/* cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(q, t)  main.py:37,695.
 * q: nq*32 u8, t: nt*32 u8; outputs have capacity nq; matches are emitted in
 * ascending query index.  nq, nt <= 16384 (RTVM_NOALLOC_MAX_DESCRIPTORS, the limit of every form of the
 * Hamming matcher), more is refused with RTVM_E_ARG. */
int rtvm_match_hamming_xcheck(rtvm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                              int32_t* qidx, int32_t* tidx, int32_t* dist, int* m_out);
/* Same, followed by the stable sort by distance of main.py:698. */
int rtvm_match_hamming_xcheck_sorted(rtvm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t,
                                     int nt, int32_t* qidx, int32_t* tidx, int32_t* dist,
                                     int* m_out);

/* rtvm_match_hamming_xcheck[_sorted] for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference.  It takes no context and calls no hipMalloc / hipFree or any routine that
 * allocates: the work buffers of the matcher (slot descriptor, counts, nearest-neighbour keys of
 * both directions, 129 KiB) are a fixed array in the library's device image, one copy per GPU,
 * held until the process ends.  All pointers are DEVICE pointers on the current device:
 * q_dev nq*32 u8, t_dev nt*32 u8, both 16-byte aligned (hipMalloc's addresses are; an address that is
 * not is refused with RTVM_E_ARG); qidx_dev / tidx_dev / dist_dev have capacity nq; *m_out_dev
 * receives the number of matches.  Matches come in ascending query index, or, with sorted != 0,
 * stably sorted by distance (main.py:698).  nq, nt <= RTVM_NOALLOC_MAX_DESCRIPTORS, more is
 * refused with RTVM_E_ARG.  The kernels are queued on hip_stream (a hipStream_t, NULL = the
 * default stream) and the call returns without waiting.  Every call on a GPU uses that GPU's one
 * array: calls that overlap on one GPU (other streams, other threads) race; order them.
 * rtvm_noalloc_last_error() returns the message of the calling thread's last failure. */
#define RTVM_NOALLOC_MAX_DESCRIPTORS 16384
int rtvm_match_hamming_xcheck_noalloc(const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt,
                                      int32_t* qidx_dev, int32_t* tidx_dev, int32_t* dist_dev,
                                      int32_t* m_out_dev, int sorted, void* hip_stream);
const char* rtvm_noalloc_last_error(void);

/* cv2.BFMatcher().knnMatch(q, t, k=2)  main.py:34,688.
 * q: nq*dim f32, t: nt*dim f32 (nt >= 2); idx: nq*2 i32, dist: nq*2 f32 (ascending). */
int rtvm_match_l2_knn2(rtvm_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim,
                       int32_t* idx, float* dist);

/* cv2.findHomography(src, dst, cv2.RANSAC, ransacReprojThreshold=thresh)  main.py:856-857.
 * src_xy/dst_xy: m*2 f32.  *found = 0 means OpenCV would have returned None.
 * inlier_mask (m u8) may be NULL. */
int rtvm_find_homography_ransac(rtvm_ctx* ctx, const float* src_xy, const float* dst_xy, int m,
                                double thresh, int max_iters, double confidence, double H[9],
                                uint8_t* inlier_mask, int* found);

/* VideMosaic.validate_homography(H)  main.py:761-801 with the thresholds of main.py:100-101
 * (50 px, 0.3): *ok = 1 when H passes.  The arithmetic is the one the device-resident stream
 * path applies to every frame (the 2x2 determinant as an LU factorisation with partial pivoting
 * evaluates it, like numpy.linalg.det; sqrt of a negative determinant is NaN and fails no
 * comparison).  Stage parity only.  Syncs. */
int rtvm_validate_homography(rtvm_ctx* ctx, const double H[9], double translation_threshold,
                             double scale_threshold, int* ok);

/* Canvas zero-fill and paste of the first frame at (row_off, col_off)  main.py:80-90. */
int rtvm_canvas_init(rtvm_ctx* ctx, const uint8_t* first_bgr, int row_off, int col_off);

/* cv2.warpPerspective(frame, H, (canvas_w, canvas_h), flags=cv2.INTER_LINEAR)  main.py:871.
 * out: canvas_h*canvas_w*3 u8.  Stage parity only; rtvm_warp_blend does not call it. */
int rtvm_warp_perspective(rtvm_ctx* ctx, const uint8_t* bgr, const double H[9], uint8_t* out);

/* cv2.distanceTransform(mask, cv2.DIST_L2, 3)  main.py:888-889.
 * mask: h*w u8 (non-zero = inside), dist: h*w f32.  The image must fit the context's
 * work planes: h*w <= canvas_h*canvas_w, h+w <= canvas_h+canvas_w, h <= canvas_h+32. */
int rtvm_distance_transform(rtvm_ctx* ctx, const uint8_t* mask, int h, int w, float* dist);

/* The whole of VideMosaic.warp(): warp, masks, distance transforms, weights,
 * 31x31 blur, blend and commit into the canvas  main.py:871-927.  Returns once the frame and H
 * have been copied (the caller may reuse both); the blend itself is queued on the context's
 * stream and rtvm_canvas_read and the other reading calls wait for it. */
int rtvm_warp_blend(rtvm_ctx* ctx, const uint8_t* bgr, const double H[9]);

/* output_img readers  main.py:1632,1649: canvas_h*canvas_w*3 u8 (the reference's
 * float64 canvas only ever holds these integers).  Syncs the stream. */
int rtvm_canvas_read(rtvm_ctx* ctx, uint8_t* out);
/* The same mosaic as the reference holds it: VideMosaic.output_img is np.zeros((h, w, 3)), float64 (main.py:80-81), into
 * which every blend stores astype(np.uint8) values (main.py:905-910); main.py:1632 hands output_img.copy() to the
 * caller's progress call-back after every frame.  out: canvas_h x canvas_w x 3 doubles.  The bytes cross PCIe (into
 * a pinned buffer the context allocates at the first call) and are widened on the host. */
int rtvm_canvas_read_f64(rtvm_ctx* ctx, double* out);

/* ---- device-resident stream path -------------------------------------- */

/* process_first_frame + canvas set-up (main.py:78-94, 104-112) with a frame that is
 * already in device memory (frame_h*frame_w*3 u8).  Asynchronous. */
int rtvm_stream_begin(rtvm_ctx* ctx, const uint8_t* bgr_dev, int row_off, int col_off);
/* VideMosaic.process_frame (main.py:710-759) for one device-resident frame: ORB,
 * match, RANSAC, validate, smooth, chain, warp+blend, state advance, all queued on
 * the context's stream without a host round trip.  Asynchronous. */
int rtvm_process_frame(rtvm_ctx* ctx, const uint8_t* bgr_dev);
/* Room for batches of up to n frames (1..32) in rtvm_process_batch: four sets of n
 * frame slots, so that three consecutive batches can be in flight.  Allocates. */
int rtvm_reserve_batch(rtvm_ctx* ctx, int n);
/* process_frame for n consecutive device-resident frames (bgr_dev: n device pointers
 * in a host array).  Detection, matching and RANSAC run for the whole batch at once,
 * the warp + blend frame by frame.  When a frame in the middle of the batch is
 * skipped (main.py:722-731) the later frames are left unprocessed with status
 * RTVM_FRAME_PENDING; rtvm_batch_status reports how many frames were consumed and
 * the caller resubmits the rest.  Asynchronous.
 * Consecutive calls overlap on the device: detection + matching and RANSAC of later
 * batches run on HIP streams of their own while an earlier batch is still being warped
 * and blended (up to four batches in flight).  The first frame of such a batch is matched against the last frame of the
 * batch before it; when that frame turns out to be skipped, the whole later batch is
 * left RTVM_FRAME_PENDING (0 frames consumed).  Reading a batch's status ends the
 * overlap when it reports pending frames or a skipped last frame, so whatever is
 * submitted next is matched against the last accepted frame (kp_prev/des_prev of
 * main.py:757-758) and always consumes at least one frame. */
int rtvm_process_batch(rtvm_ctx* ctx, const uint8_t* const* bgr_dev, int n);
/* Of the most recent batch: frames consumed and, per frame (arrays of cap entries,
 * any may be NULL), status bits, match count, keypoint count.  Syncs the stream. */
int rtvm_batch_status(rtvm_ctx* ctx, int* n_done, int* status, int* n_matches, int* n_keypoints, int cap);
/* Same for the most recent batch (back = 0) or the ones queued before it (back = 1, 2),
 * waiting for that batch only: a caller can keep two batches in flight while it reads
 * the outcome of a third.  When the batch read left frames pending or skipped its
 * last frame, the batches queued after it (if any) are pending as a whole; the call
 * then waits for them too. */
int rtvm_batch_status_at(rtvm_ctx* ctx, int back, int* n_done, int* status, int* n_matches,
                         int* n_keypoints, int cap);
/* Ticket of the batch queued last (rtvm_process_batch, rtvm_process_frame, rtvm_batch_detect, rtvm_batch_retry or
 * rtvm_batch_commit); tickets count up from 1 after rtvm_stream_begin.  The library keeps the reports (and, for
 * rtvm_batch_commit, the homographies) of the last 16 tickets: a call that takes a 17th ticket first waits until
 * the piece 16 tickets before it has run. */
long rtvm_batch_ticket(rtvm_ctx* ctx);
/* rtvm_batch_status for one of the last 16 tickets, plus the reason (RTVM_PENDING_*) when
 * frames were left pending.  Waits for that batch only. */
int rtvm_batch_status_of(rtvm_ctx* ctx, long ticket, int* n_done, int* reason, int* status,
                         int* n_matches, int* n_keypoints, int cap);
/* Index (within the piece / batch `ticket`, one of the last 16) of the first frame whose blend
 * turned a non-zero canvas pixel into zero, or -1: the event behind RTVM_PENDING_MASK, reported
 * also when the frame was the last of its piece and nothing had to be left pending.  Such frames
 * come in runs; a caller sizes its next pieces by it.  Waits for that batch only.  No counterpart
 * in the reference (its blend, main.py:905-927, is one frame at a time). */
int rtvm_batch_cleared(rtvm_ctx* ctx, long ticket, int* first);
/* Queues the canvas half again (validate / smooth / chain, warp, blend) for frames first ..
 * first+count-1 of the batch `ticket` (queued with rtvm_process_batch or rtvm_batch_detect and still holding its slot set),
 * reusing its detection, matching and RANSAC results: for frames left pending with
 * RTVM_PENDING_MASK, or for a RTVM_PENDING_VOID batch queued behind such frames, in frame
 * order.  The retry is a batch of its own (rtvm_batch_ticket).  No counterpart in the
 * reference: part of the batching scheme.  Asynchronous. */
int rtvm_batch_retry(rtvm_ctx* ctx, long ticket, int first, int count);
/* The two halves of rtvm_process_batch as separate calls, for callers that keep detection
 * running ahead while they deal with pending frames.  rtvm_batch_detect queues detection,
 * matching and RANSAC of n frames (matched against the frames queued before them, like
 * rtvm_process_batch) and returns at once; the batch gets a ticket (rtvm_batch_ticket) but no
 * status of its own.  rtvm_batch_retry(ticket, first, count) then queues the canvas half of
 * its frames, in frame order across batches.  The batch keeps one of the four slot sets until
 * rtvm_batch_release(ticket); rtvm_batch_detect fails with RTVM_E_STATE when no set is free.
 * After a skipped frame (RTVM_PENDING_SKIPPED_FRAME, or a skipped last frame) the batches
 * detected behind it were matched against the wrong frame: release them and detect again. */
int rtvm_batch_detect(rtvm_ctx* ctx, const uint8_t* const* bgr_dev, int n);
int rtvm_batch_release(rtvm_ctx* ctx, long ticket);
/* ---- detected batches for hosts that keep validate / smooth / chain (main.py:733-746) on their
 * side: the asynchronous form of rtvm_frame_register / rtvm_frame_commit.  The Python VideMosaic
 * runs its headless mode on these: process_frame (main.py:710-759) only enqueues the frame; the
 * numpy statements of the reference decide per frame from the batch's results; the warps and
 * blends follow in pieces.  No counterpart in the reference beyond process_frame itself. ---- */

/* Per-frame outcome of detection, matching and RANSAC (main.py:716-731) of the batch `ticket`
 * queued with rtvm_batch_detect and not yet released: match count, found (0 = fewer than 4
 * matches or no model: the reference skips the frame), keypoint count, H_rel (cap x 9 doubles,
 * frame -> previous frame).  Arrays of cap entries, any may be NULL.  Waits for that batch's
 * RANSAC only.  rtvm_batch_results_ready: 1 when that wait would not block, else 0. */
int rtvm_batch_results(rtvm_ctx* ctx, long ticket, int* n_matches, int* found, int* n_keypoints,
                       double* H_rel, int cap);
int rtvm_batch_results_ready(rtvm_ctx* ctx, long ticket);
/* main.py:748, 756-759 for frames first .. first+count-1 of the detected batch `ticket` with the
 * absolute homographies the host computed (H_abs: count x 9 doubles, copied before the call
 * returns): warp + blend in frame order, then kp_prev/des_prev = features of the last of them.
 * Only frames the host decided to stitch are committed (a skipped frame is never passed).  The
 * piece is a batch of its own (rtvm_batch_ticket, rtvm_batch_status_of): frames behind a blend
 * that cleared a canvas pixel come back RTVM_FRAME_PENDING with reason RTVM_PENDING_MASK and are
 * committed again with the same homographies; pieces queued behind such a piece come back
 * RTVM_PENDING_VOID.  Asynchronous. */
int rtvm_batch_commit(rtvm_ctx* ctx, long ticket, int first, int count, const double* H_abs);
/* 1 when the status of the piece / batch `ticket` (one of the last 16) can be read without
 * waiting, else 0. */
int rtvm_batch_done(rtvm_ctx* ctx, long ticket);
/* After the host skipped a frame (main.py:722-731): what is detected next is matched against the
 * last committed frame (kp_prev/des_prev), not against the frame detected last. */
int rtvm_batch_rebase(rtvm_ctx* ctx);
/* kp_cur/des_cur and self.matches (main.py:718-720) of frame `index` of the held batch `ticket`;
 * outputs as rtvm_get_features / rtvm_get_matches. */
int rtvm_batch_get_features(rtvm_ctx* ctx, long ticket, int index, int* n_out, float* xy, float* size,
                            float* angle_deg, float* response, int32_t* octave, uint8_t* desc);
int rtvm_batch_get_matches(rtvm_ctx* ctx, long ticket, int index, int* m_out, int32_t* qidx, int32_t* tidx,
                           int32_t* dist);

/* ---- the reference's main loop (main.py:1596-1634) as one call --------------------------------------
 * `for frame in video: video_mosaic.process_frame(frame, i)` for n device-resident frames, after
 * rtvm_stream_begin with the first frame: detection, matching and RANSAC of up to four batches (of `batch`
 * frames, 1..32) run ahead of the canvas half, which follows in pieces of canvas_batch frames (0 = half a batch);
 * the skip rules, validate, smooth and chain (main.py:722-746) are applied on the device.  Everything the batch
 * entry points above leave to their caller happens inside: frames left pending behind a blend that cleared a
 * canvas pixel get their canvas half queued again, frames detected behind a skipped frame are detected again.
 * status[i] receives the RTVM_FRAME_* bits of frame i (never RTVM_FRAME_PENDING: every frame is consumed);
 * stats (may be NULL) counts what was queued.  Returns when every frame has been stitched or skipped; H_abs of
 * the last frame: rtvm_frame_status, the mosaic: rtvm_canvas_read. */
typedef struct {
    int frames, batches_queued, frames_queued, batches_with_pending_frames, retries, frames_retried, frames_void,
        frames_detected_again, canvas_pieces, canvas_frames;
} rtvm_video_stats;
int rtvm_process_video(rtvm_ctx* ctx, const uint8_t* const* bgr_dev, int n, int batch, int canvas_batch,
                       int32_t* status, rtvm_video_stats* stats);
/* The same with the frames handed out batch by batch: source(user, first, count, dev) stores the device
 * addresses of frames first .. first+count-1 (count <= batch; e.g. after copying them through the ingest ring)
 * and returns RTVM_OK; it is asked for a range again when detection starts over after a skipped frame.
 * pieces_in_flight: canvas halves queued ahead (0 = 2; 3 suits a source that copies host frames meanwhile).
 * events (may be NULL): one record per canvas half whose outcome was read -- seconds on a steady clock, first
 * video frame, length, frames consumed, RTVM_PENDING_* reason; *n_events counts them all, events_cap are kept. */
typedef int (*rtvm_frame_source)(void* user, int first, int count, const uint8_t** dev);
typedef struct {
    double seconds;
    int first, count, done, reason;
} rtvm_video_event;
int rtvm_process_video_source(rtvm_ctx* ctx, rtvm_frame_source source, void* user, int n, int batch, int canvas_batch,
                              int pieces_in_flight, int32_t* status, rtvm_video_stats* stats, rtvm_video_event* events,
                              int events_cap, int* n_events);

/* ---- process_frame (main.py:710-759) as a call that only takes the frame, for hosts that keep the rules of
 * main.py:722-746 on their side (public thresholds, a history the caller may read or replace: the Python
 * VideMosaic).  rtvm_pipe_put copies a host frame (frame_h x frame_w x 3 u8) into a ring of pinned slots and
 * returns; frames are detected, matched and fitted in batches of `batch`, up to four batches ahead.  When a
 * batch's results are in, glue(user, frame, n_matches, found, H_rel, H_abs) is called for its frames in order,
 * on the thread that is inside rtvm_pipe_put / rtvm_pipe_flush at that time: `frame` counts the put calls from 0,
 * found = 0 means fewer than 4 matches or no model; it returns 1 after writing the absolute homography the frame
 * is to be warped with, 0 when the frame is skipped (main.py:722-731: state not advanced; the frames detected
 * behind it are detected again, matched against the last stitched frame), negative on failure (the call then
 * returns RTVM_E_CALLBACK).  The frames to stitch are warped and blended in pieces of up to `piece` frames
 * (0 = half a batch), three in flight; pieces that come back with frames pending are committed again.
 * rtvm_pipe_flush returns when every frame taken so far has been dealt with.  After a call on the pipe has failed
 * (the caller's function included) the frames in flight are in an undefined state: the pipe accepts nothing more and
 * is to be closed.  One pipe per context; opening it
 * reserves the batch slots and the ingest ring (rtvm_reserve_batch, rtvm_ingest_reserve). */
typedef int (*rtvm_glue_fn)(void* user, long frame, int n_matches, int found, const double H_rel[9], double H_abs[9]);
typedef struct {
    long frames, batches, pieces, pieces_with_pending_frames, pieces_void, frames_detected_again;
} rtvm_pipe_stats;
int rtvm_pipe_open(rtvm_ctx* ctx, int batch, int piece, rtvm_glue_fn glue, void* user);
int rtvm_pipe_put(rtvm_ctx* ctx, const uint8_t* bgr_host);
int rtvm_pipe_flush(rtvm_ctx* ctx);
/* 1 while frames taken by rtvm_pipe_put have not been dealt with, else 0. */
int rtvm_pipe_busy(rtvm_ctx* ctx);
/* After a flush the batch of the newest frame stays held (until the next rtvm_pipe_put detects again) so that
 * kp_cur / des_cur / matches (main.py:718-720) can be read lazily: its ticket and the frame's index in it for
 * rtvm_batch_get_features / rtvm_batch_get_matches; ticket 0 when there is none. */
int rtvm_pipe_last(rtvm_ctx* ctx, long* ticket, int* index);
int rtvm_pipe_get_stats(rtvm_ctx* ctx, rtvm_pipe_stats* out);
/* Drops the pipe and what it still holds. */
int rtvm_pipe_close(rtvm_ctx* ctx);

/* Status bits, match count and the absolute homography of the most recent
 * rtvm_process_frame (any may be NULL).  Syncs the stream. */
int rtvm_frame_status(rtvm_ctx* ctx, int* status, int* n_keypoints, int* n_matches, double H_abs[9]);

/* ---- frame ingest: the step before the path ------------------------------------ */

/* The reference's loop (main.py:1596-1634) gets every frame from cap.read() as a host array.
 * Here a ring of n_slots pinned host frames and as many device frames: the decoder writes
 * frame i into the host slot (rtvm_ingest_host_slot), rtvm_ingest_submit queues its copy to
 * the device on a stream of its own and returns the device address to hand to
 * rtvm_process_batch / rtvm_batch_detect; the streams that read the frame wait for the copy,
 * the host does not.  A slot may be filled again once the frame it held has been consumed
 * (its batch status read with no frame of it pending).  No counterpart in the reference
 * beyond the loop itself.  rtvm_ingest_reserve allocates (once per context). */
int rtvm_ingest_reserve(rtvm_ctx* ctx, int n_slots);
int rtvm_ingest_host_slot(rtvm_ctx* ctx, int slot, uint8_t** host);
/* Copies a frame (frame_h x frame_w x 3 bytes in ordinary host memory, e.g. what cap.read() returned,
 * main.py:1597) into the pinned slot, on four threads for frames of a megabyte and more; returns when the
 * copy is complete.  For callers whose decoder cannot write into the slot itself. */
int rtvm_ingest_write(rtvm_ctx* ctx, int slot, const uint8_t* bgr_host);
int rtvm_ingest_submit(rtvm_ctx* ctx, int slot, const uint8_t** dev);
/* The same for `count` consecutive slots in one copy (a whole batch moves at about twice the
 * rate of frame-sized copies); dev receives the `count` device addresses. */
int rtvm_ingest_submit_range(rtvm_ctx* ctx, int first_slot, int count, const uint8_t** dev);

/* ---- mosaic finalisation: the step after the path, once per video ---------------- */

/* crop_black_areas (main.py:980-1003): rect = {x, y, w, h} of cv2.boundingRect over the pixels
 * whose gray value (cvtColor BGR2GRAY) exceeds `threshold` (cv2.threshold THRESH_BINARY,
 * main.py:991-997) of a host image of h x w x 3 u8; *found = 0 when there is none (the
 * reference then returns the image unchanged).  The margin arithmetic of main.py:999-1002 is
 * the caller's.  Allocates scratch; syncs. */
int rtvm_crop_rect(rtvm_ctx* ctx, const uint8_t* bgr, int h, int w, int threshold, int rect[4], int* found);
/* Same on the context's canvas as it stands (what main.py:1647-1652 passes in), without
 * copying it to the host first. */
int rtvm_canvas_crop_rect(rtvm_ctx* ctx, int threshold, int rect[4], int* found);
/* scale_to_screen's cv2.resize(image, (nw, nh), interpolation=INTER_LINEAR) (main.py:1037) of a
 * host image of h x w x 3 u8 into out (nh x nw x 3).  Allocates scratch; syncs. */
int rtvm_resize_linear(rtvm_ctx* ctx, const uint8_t* bgr, int h, int w, uint8_t* out, int nh, int nw);
/* The same resize of the rectangle x, y, w, h of the context's canvas: crop and scale in one
 * pass, only the scaled image crosses PCIe. */
int rtvm_canvas_read_scaled(rtvm_ctx* ctx, int x, int y, int w, int h, uint8_t* out, int nh, int nw);

/* ---- classical passes on the finished mosaic before path planning (the step after
 * finalisation; reference analyze_for_navigation, main.py:1051-1232) ------------------ */

#define RTVM_MORPH_ERODE 0 /* cv2.MORPH_ERODE .. cv2.MORPH_CLOSE */
#define RTVM_MORPH_DILATE 1
#define RTVM_MORPH_OPEN 2
#define RTVM_MORPH_CLOSE 3
#define RTVM_SHAPE_RECT 0 /* cv2.MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE */
#define RTVM_SHAPE_CROSS 1
#define RTVM_SHAPE_ELLIPSE 2

/* cv2.cvtColor(frame, cv2.COLOR_BGR2HSV) on 8-bit pixels (main.py:471,1065): H in [0,180).
 * Host image h x w x 3 in, h x w x 3 out.  Allocates scratch; syncs. */
int rtvm_bgr2hsv(rtvm_ctx* ctx, const uint8_t* bgr, int h, int w, uint8_t* hsv);
/* cv2.erode / cv2.dilate / cv2.morphologyEx(OPEN, CLOSE) of a mask (0 / non-zero, out is
 * 0 / 255) with cv2.getStructuringElement(shape, (ksize, ksize)), default anchor and border,
 * `iterations` times (main.py:1142-1150,1188-1190,1197-1198,1207-1210,1231-1232 and the masks
 * of main.py:492-498,528-529,647-649).  ksize 1..31.  Allocates scratch; syncs. */
int rtvm_morphology(rtvm_ctx* ctx, const uint8_t* mask, int h, int w, int op, int shape, int ksize,
                    int iterations, uint8_t* out);
/* cv2.Canny(gray, low, high) with the reference's arguments (main.py:489 on the blurred gray
 * image of _detect_buildings_cv2, main.py:1141): aperture 3, L1 gradient.  Host image h x w u8 in,
 * edges h x w of 0 / 255 out.  Allocates scratch; syncs. */
int rtvm_canny(rtvm_ctx* ctx, const uint8_t* gray, int h, int w, double low_thresh, double high_thresh,
               uint8_t* edges);
/* The obstacle masks of analyze_for_navigation (main.py:1065-1232) for a host image h x w x 3:
 * boxes = n_boxes rectangles x1, y1, x2, y2 as main.py:1106 hands them to cv2.rectangle (the
 * detections with their class buffer already applied, both corners included).  Outputs, each
 * h x w u8 of 0 / 255 and each may be NULL: mask_fire and mask_smoke after their morphology
 * (main.py:1150,1190), texture_mask (main.py:1210, the input of findContours), obstacles
 * (main.py:1228, what is_path_clear reads), obstacles_nav (main.py:1232, dilated by the 15x15
 * ellipse).  grid (may be NULL): find_path_astar's occupancy matrix of obstacles_nav at
 * grid_scale (main.py:1346-1359), max(1,h/scale) x max(1,w/scale) bytes of 0 / 1.
 * cv2.Canny of main.py:1141 is not computed here: its edges enter the result only as
 * mask | (dilated_edges & mask), which is mask (rtvm_canny computes them).  Allocates scratch; syncs. */
int rtvm_nav_obstacles(rtvm_ctx* ctx, const uint8_t* bgr, int h, int w, const int32_t* boxes, int n_boxes,
                       uint8_t* mask_fire, uint8_t* mask_smoke, uint8_t* texture_mask, uint8_t* obstacles,
                       uint8_t* obstacles_nav, int grid_scale, uint8_t* grid);
/* find_path_astar's matrix alone (main.py:1346-1359) for any mask h x w: cell (gy, gx) is 1
 * when more than 30 % of scale*scale pixels of its block are non-zero. */
int rtvm_occupancy_grid(rtvm_ctx* ctx, const uint8_t* mask, int h, int w, int scale, uint8_t* grid);

/* ---- two-call form of process_frame for hosts that keep the validate / smooth /
 * chain arithmetic of main.py:733-746 on their side (the Python VideMosaic) ----- */

/* process_first_frame + canvas set-up (main.py:78-94, 104-112), host frame.  Syncs. */
int rtvm_first_frame(rtvm_ctx* ctx, const uint8_t* bgr, int row_off, int col_off);
/* main.py:716-731: upload the frame, detect, match against the previous frame,
 * RANSAC.  *found = 0 when there are fewer than 4 matches or no model.  Syncs. */
int rtvm_frame_register(rtvm_ctx* ctx, const uint8_t* bgr, int* n_matches, int* found, double H_rel[9]);
/* main.py:757-758 alone (frame_prev, kp_prev, des_prev = this frame): detects on the host frame
 * and makes its features the previous ones; canvas, homographies and history are left as they
 * are.  For hosts that stitched some frames through the stage entry points (rtvm_orb,
 * rtvm_match_*, rtvm_find_homography_ransac, rtvm_warp_blend) and go back to the two-call
 * form.  Syncs. */
int rtvm_stream_set_prev(rtvm_ctx* ctx, const uint8_t* bgr);
/* main.py:748, 756-759: warp + blend the registered frame with the absolute
 * homography H, then make it the previous frame.  Asynchronous: the work is queued behind
 * the registration on the context's stream; rtvm_frame_register, rtvm_canvas_read and the
 * other reading calls wait for it. */
int rtvm_frame_commit(rtvm_ctx* ctx, const double H[9]);
/* kp_cur/des_cur (slot 0) or kp_prev/des_prev (slot 1) of the stream; any output may
 * be NULL; capacity rtvm_max_keypoints().  desc: n*32 u8 (ORB) or n*128 f32 (RTVM_DET_SIFT). */
int rtvm_get_features(rtvm_ctx* ctx, int slot, int* n_out, float* xy, float* size, float* angle_deg,
                      float* response, int32_t* octave, uint8_t* desc);
/* self.matches of the most recent frame (sorted by distance); capacity rtvm_max_keypoints().
 * dist: Hamming distances as i32 (ORB) or the bit patterns of the f32 L2 distances (RTVM_DET_SIFT). */
int rtvm_get_matches(rtvm_ctx* ctx, int* m_out, int32_t* qidx, int32_t* tidx, int32_t* dist);

/* ---- measurement aid (no counterpart in the reference) ----------------------- */

/* Times kernels with HIP events on the context's stream: every kernel (filter NULL
 * or "") or those whose name contains filter. */
int rtvm_profile_start(rtvm_ctx* ctx, const char* filter);
/* Syncs, accumulates the intervals, stops timing. */
int rtvm_profile_stop(rtvm_ctx* ctx);
/* Text report, one line per kernel: "name launches total_ms".  Returns bytes needed. */
int rtvm_profile_report(rtvm_ctx* ctx, char* buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* RTVM_H */
