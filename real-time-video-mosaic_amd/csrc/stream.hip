// VideMosaic.process_frame (reference main.py:710-759) with every piece of state
// resident on the device: features of the previous frame, H_old, the smoothing
// history and the canvas.  One call queues ORB, matching, RANSAC, the
// validate / smooth / chain arithmetic and the warp + blend on the context's
// stream; the skip rules of the reference (fewer than 4 matches, no homography)
// become a device-side flag that empties the warp's region of interest and
// suppresses the state advance.
#include "common.h"

#include <cmath>

namespace {

// points of the matches, in match order: src = current frame, dst = previous frame
// (main.py:850-854)
__global__ __launch_bounds__(256) void k_gather_pts(const int* __restrict__ m_count, const int32_t* __restrict__ qidx,
                                                    const int32_t* __restrict__ tidx, const float* __restrict__ xy_cur,
                                                    const float* __restrict__ xy_prev, float* __restrict__ src,
                                                    float* __restrict__ dst, int* __restrict__ rs_count) {
    const int m = *m_count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *rs_count = m;
    if (i >= m) return;
    const int q = qidx[i], t = tidx[i];
    src[2 * i] = xy_cur[2 * q];
    src[2 * i + 1] = xy_cur[2 * q + 1];
    dst[2 * i] = xy_prev[2 * t];
    dst[2 * i + 1] = xy_prev[2 * t + 1];
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
    if (fabs(a) >= fabs(c)) {
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

// main.py:722-746 and 756: skip rules, validate, smooth (weighted mean of at most
// five relative homographies), H = H_old @ H_smoothed, H_old = H.
__global__ void k_frame_update(StreamState* __restrict__ st, const int* __restrict__ m_count,
                               const int* __restrict__ found, const double* __restrict__ H_rel_in,
                               const double* __restrict__ wtab, double* __restrict__ H_out, int* __restrict__ skip,
                               const int* __restrict__ n_kp, double trans_thr, double scale_thr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->frame_index += 1;
    const int m = *m_count;
    st->n_matches = m;
    st->n_kp = *n_kp;
    if (m < 4) {
        st->status = RTVM_FRAME_FEW_MATCHES;
        *skip = 1;
        return;
    }
    if (!*found) {
        st->status = RTVM_FRAME_NO_HOMOGRAPHY;
        *skip = 1;
        return;
    }
    double Hr[9];
    for (int i = 0; i < 9; i++) Hr[i] = H_rel_in[i];
    int status = 0;
    if (!validate_h(Hr, trans_thr, scale_thr)) {
        for (int i = 0; i < 9; i++) Hr[i] = (i % 4 == 0) ? 1. : 0.;
        status |= RTVM_FRAME_INVALID_H;
    } else {
        for (int i = 0; i < 9; i++) st->last_valid_H[i] = Hr[i];
    }
    // smooth_homography
    int n = st->hist_len;
    if (n == 5) {
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 9; i++) st->hist[k][i] = st->hist[k + 1][i];
        n = 4;
    }
    for (int i = 0; i < 9; i++) st->hist[n][i] = Hr[i];
    n++;
    st->hist_len = n;
    double Hs[9];
    if (n < 2) {
        for (int i = 0; i < 9; i++) Hs[i] = Hr[i];
    } else {
        for (int i = 0; i < 9; i++) {
            double s = 0.;
            for (int k = 0; k < n; k++) s += wtab[n * 5 + k] * st->hist[k][i];
            Hs[i] = s;
        }
    }
    // self.H = np.matmul(self.H_old, H_relative_smoothed)
    double H[9];
    for (int r = 0; r < 3; r++)
        for (int c2 = 0; c2 < 3; c2++)
            H[r * 3 + c2] = (st->H_old[r * 3] * Hs[c2] + st->H_old[r * 3 + 1] * Hs[3 + c2]) + st->H_old[r * 3 + 2] * Hs[6 + c2];
    for (int i = 0; i < 9; i++) {
        st->H_abs[i] = H[i];
        st->H_old[i] = H[i];
        H_out[i] = H[i];
    }
    st->status = status;
    *skip = 0;
}

// kp_prev, des_prev <- kp_cur, des_cur (main.py:757-758) unless the frame was skipped
__global__ __launch_bounds__(256) void k_advance_feats(const int* __restrict__ skip, FeatSet cur, FeatSet prev,
                                                       int max_kp) {
    if (skip && *skip) return;
    const int n = *cur.count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *prev.count = n;
    if (i >= n || i >= max_kp) return;
    prev.xy[2 * i] = cur.xy[2 * i];
    prev.xy[2 * i + 1] = cur.xy[2 * i + 1];
    prev.size[i] = cur.size[i];
    prev.angle[i] = cur.angle[i];
    prev.response[i] = cur.response[i];
    prev.octave[i] = cur.octave[i];
    prev.lxy[2 * i] = cur.lxy[2 * i];
    prev.lxy[2 * i + 1] = cur.lxy[2 * i + 1];
    const uint4* s = reinterpret_cast<const uint4*>(cur.desc + (size_t)i * 32);
    uint4* d = reinterpret_cast<uint4*>(prev.desc + (size_t)i * 32);
    d[0] = s[0];
    d[1] = s[1];
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
    *skip = 0;
}

int need_stream_ctx(rtvm_ctx* c, const char* who) {
    if (c->fh <= 0 || c->fw <= 0 || c->ch <= 0 || c->cw <= 0)
        RTVM_FAIL(c, RTVM_E_STATE, "%s: context was created without frame/canvas size", who);
    if (c->detector != RTVM_DET_ORB) RTVM_FAIL(c, RTVM_E_STATE, "%s: the stream path needs the ORB detector", who);
    return RTVM_OK;
}

// detect + match against the previous frame + gather points + RANSAC, all queued
int queue_front_half(rtvm_ctx* c, const uint8_t* d_bgr) {
    int r;
    if ((r = rtvm_launch_orb(c, d_bgr, 0))) return r;
    const int cap = c->orb.max_kp;
    if ((r = rtvm_launch_hamming(c, c->feat[0].desc, c->feat[1].desc, c->feat[0].count, c->feat[1].count, cap, cap, 1)))
        return r;
    RTVM_LAUNCH(c, k_gather_pts, dim3((cap + 255) / 256), dim3(256), 0, c->d_m_count, c->d_m_qidx,
                       c->d_m_tidx, c->feat[0].xy, c->feat[1].xy, c->d_rs_src, c->d_rs_dst, c->d_rs_count);
    return rtvm_launch_ransac(c, c->d_rs_src, c->d_rs_dst, c->d_rs_count, 2.0, 2000, 0.995, c->d_rs_H, c->d_rs_mask,
                              c->d_rs_found);
}

int queue_first_frame(rtvm_ctx* c, const uint8_t* d_bgr, int row_off, int col_off) {
    int r;
    if ((r = rtvm_launch_orb(c, d_bgr, 0))) return r;
    RTVM_LAUNCH(c, k_advance_feats, dim3((c->orb.max_kp + 255) / 256), dim3(256), 0, (const int*)nullptr,
                       c->feat[0], c->feat[1], c->orb.max_kp);
    if ((r = rtvm_launch_canvas_init(c, d_bgr, row_off, col_off))) return r;
    RTVM_LAUNCH(c, k_state_init, dim3(1), dim3(64), 0, c->d_state, row_off, col_off, c->d_skip);
    HIP_TRY(c, hipGetLastError());
    c->stream_ready = 1;
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

extern "C" int rtvm_stream_begin(rtvm_ctx* c, const uint8_t* bgr_dev, int row_off, int col_off) {
    if (!c) return RTVM_E_ARG;
    if (!bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "stream_begin: null frame");
    int r = need_stream_ctx(c, "stream_begin");
    if (r) return r;
    HIP_TRY(c, hipSetDevice(c->device));
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
    if (!bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "process_frame: null frame");
    int r = need_stream_ctx(c, "process_frame");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "process_frame: call rtvm_stream_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((r = queue_front_half(c, bgr_dev))) return r;
    RTVM_LAUNCH(c, k_frame_update, dim3(1), dim3(64), 0, c->d_state, c->d_m_count, c->d_rs_found,
                       c->d_rs_H, c->d_wtab, c->d_H, c->d_skip, c->feat[0].count, 50.0, 0.3);
    if ((r = rtvm_launch_warp_blend(c, bgr_dev, c->d_H, c->d_skip))) return r;
    RTVM_LAUNCH(c, k_advance_feats, dim3((c->orb.max_kp + 255) / 256), dim3(256), 0, c->d_skip,
                       c->feat[0], c->feat[1], c->orb.max_kp);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

extern "C" int rtvm_frame_status(rtvm_ctx* c, int* status, int* n_keypoints, int* n_matches, double H_abs[9]) {
    if (!c) return RTVM_E_ARG;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_status: no stream");
    HIP_TRY(c, hipSetDevice(c->device));
    StreamState* hs = (StreamState*)((char*)c->h_pinned + 4096);
    HIP_TRY(c, hipMemcpyAsync(hs, c->d_state, sizeof(StreamState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (status) *status = hs->status;
    if (n_keypoints) *n_keypoints = hs->n_kp;
    if (n_matches) *n_matches = hs->n_matches;
    if (H_abs) memcpy(H_abs, hs->H_abs, 9 * sizeof(double));
    return RTVM_OK;
}

// ---- two-call form used by the Python VideMosaic: the glue between RANSAC and the
// warp (validate, smooth, chain; main.py:733-746) stays in the caller's numpy -------

// Uploads the frame and queues ORB, matching against the previous frame and RANSAC;
// returns the match count, whether a homography was found and the relative H.
extern "C" int rtvm_frame_register(rtvm_ctx* c, const uint8_t* bgr, int* n_matches, int* found, double H_rel[9]) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !n_matches || !found || !H_rel) RTVM_FAIL(c, RTVM_E_ARG, "frame_register: null argument");
    int r = need_stream_ctx(c, "frame_register");
    if (r) return r;
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_register: call rtvm_first_frame first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_bgr, bgr, (size_t)c->fh * c->fw * 3, hipMemcpyHostToDevice, c->stream));
    if ((r = queue_front_half(c, c->d_bgr))) return r;
    int* hc = (int*)c->h_pinned;
    double* hH = (double*)c->h_pinned + 8;
    HIP_TRY(c, hipMemcpyAsync(hc, c->d_m_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hc + 1, c->d_rs_found, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hH, c->d_rs_H, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_matches = hc[0];
    *found = hc[0] >= 4 ? hc[1] : 0;
    memcpy(H_rel, hH, 9 * sizeof(double));
    return RTVM_OK;
}

// Warps and blends the registered frame with the absolute homography H and makes it
// the previous frame (main.py:748, 756-759).  Asynchronous.
extern "C" int rtvm_frame_commit(rtvm_ctx* c, const double H[9]) {
    if (!c) return RTVM_E_ARG;
    if (!H) RTVM_FAIL(c, RTVM_E_ARG, "frame_commit: null homography");
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "frame_commit: no registered frame");
    HIP_TRY(c, hipSetDevice(c->device));
    double* hp = (double*)c->h_pinned + 32;
    memcpy(hp, H, 9 * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(c->d_H, hp, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int r;
    if ((r = rtvm_launch_warp_blend(c, c->d_bgr, c->d_H, nullptr))) return r;
    RTVM_LAUNCH(c, k_advance_feats, dim3((c->orb.max_kp + 255) / 256), dim3(256), 0, (const int*)nullptr,
                       c->feat[0], c->feat[1], c->orb.max_kp);
    HIP_TRY(c, hipGetLastError());
    // the pinned staging of H is reused by the next call
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

// Features of the current (slot 0) or previous (slot 1) frame of the stream.
extern "C" int rtvm_get_features(rtvm_ctx* c, int slot, int* n_out, float* xy, float* size, float* angle_deg,
                                 float* response, int32_t* octave, uint8_t* desc) {
    if (!c) return RTVM_E_ARG;
    if (!n_out || slot < 0 || slot > 1) RTVM_FAIL(c, RTVM_E_ARG, "get_features: bad argument");
    if (c->fh <= 0) RTVM_FAIL(c, RTVM_E_STATE, "get_features: context has no frame size");
    HIP_TRY(c, hipSetDevice(c->device));
    const FeatSet& F = c->feat[slot];
    int* hc = (int*)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(hc, F.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int n = hc[0];
    if (n < 0 || n > c->orb.max_kp) RTVM_FAIL(c, RTVM_E_HIP, "get_features: device returned count %d", n);
    if (n) {
        if (xy) HIP_TRY(c, hipMemcpyAsync(xy, F.xy, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        if (size) HIP_TRY(c, hipMemcpyAsync(size, F.size, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (angle_deg) HIP_TRY(c, hipMemcpyAsync(angle_deg, F.angle, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (response) HIP_TRY(c, hipMemcpyAsync(response, F.response, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (octave) HIP_TRY(c, hipMemcpyAsync(octave, F.octave, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (desc) HIP_TRY(c, hipMemcpyAsync(desc, F.desc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return RTVM_OK;
}

// Matches of the most recent frame against its predecessor, sorted by distance.
extern "C" int rtvm_get_matches(rtvm_ctx* c, int* m_out, int32_t* qidx, int32_t* tidx, int32_t* dist) {
    if (!c) return RTVM_E_ARG;
    if (!m_out) RTVM_FAIL(c, RTVM_E_ARG, "get_matches: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int* hc = (int*)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(hc, c->d_m_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int m = hc[0];
    if (m < 0 || m > RTVM_MAX_MATCH) RTVM_FAIL(c, RTVM_E_HIP, "get_matches: device returned count %d", m);
    if (m) {
        if (qidx) HIP_TRY(c, hipMemcpyAsync(qidx, c->d_m_qidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (tidx) HIP_TRY(c, hipMemcpyAsync(tidx, c->d_m_tidx, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (dist) HIP_TRY(c, hipMemcpyAsync(dist, c->d_m_dist, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *m_out = m;
    return RTVM_OK;
}
