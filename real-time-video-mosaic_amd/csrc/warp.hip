// VideMosaic.warp() for gfx950 (reference main.py:861-936): perspective warp of
// the frame into canvas coordinates, masks, chamfer distance transforms (dt.hip),
// blend weights, separable 31x31 Gaussian, blend and commit into the canvas.
//
// Data layout in HBM: the canvas is u8 HxWx3 (the reference's float64 canvas
// only ever holds the integers 0..255); every auxiliary plane (warped frame,
// masks, distances, weights) has the canvas' width as pitch and is only touched
// inside the region of interest of the frame being stitched.  mask_old
// (canvas pixel has a non-zero channel) is kept up to date by the commit
// kernel instead of being recomputed from the canvas every frame.
#include "common.h"

#include <cmath>
#include <vector>

namespace {

// BORDER_REFLECT_101 index: fold |p| into one period 2n-2 of the mirror.  The subtraction loop runs
// only for an image narrower than the filter's reach (a modulo here would be evaluated, at a quarter
// of the multiply rate, for every index).
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    p = p < 0 ? -p : p;
    while (p >= period) p -= period;
    return p < n ? p : period - p;
}

// Frame j of the batch: M = H^-1 (cv::invert's 3x3 closed form), the region of the canvas the
// frame can touch, and the two distance-transform jobs of the frame.
__device__ void warp_prep_frame(int j, const double* Hp, WarpParams* wp, DtJob* jobs_new, DtJob* jobs_old, int fh, int fw,
                                int ch, int cw, int force_full, const int* skip) {
    Hp += 9 * j;
    wp += j;
    DtJob* jobs[2] = {jobs_new ? jobs_new + j : nullptr, jobs_old ? jobs_old + j : nullptr};
    if (skip) skip += j;
    if (skip && *skip) {  // frame rejected upstream: nothing to stitch
        wp->rx0 = wp->ry0 = wp->rx1 = wp->ry1 = 0;
        wp->wx0 = wp->wy0 = wp->wx1 = wp->wy1 = 0;
        for (int k = 0; k < 2; k++)
            if (jobs[k]) {
                jobs[k]->x0 = jobs[k]->y0 = jobs[k]->x1 = jobs[k]->y1 = 0;
                jobs[k]->ox0 = jobs[k]->oy0 = jobs[k]->ox1 = jobs[k]->oy1 = 0;
            }
        return;
    }
    double S[9];
    for (int i = 0; i < 9; i++) S[i] = Hp[i];
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    double t[9];
    bool full = force_full != 0;
    if (d == 0.) {
        for (int i = 0; i < 9; i++) t[i] = 0.;
        full = true;
    } else {
        d = 1. / d;
        t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
        t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
        t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
        t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
        t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
        t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
        t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    }
    for (int i = 0; i < 9; i++) {
        wp->M[i] = t[i];
        wp->Mf[i] = (float)t[i];
    }
    // Canvas pixels whose source coordinate can fall inside (-1, fw) x (-1, fh):
    // the image under H of that rectangle grown by half a pixel, plus two pixels.
    int x0 = 0, y0 = 0, x1 = cw, y1 = ch;
    if (!full) {
        const double cx[4] = {-1.5, fw + 0.5, fw + 0.5, -1.5};
        const double cy[4] = {-1.5, -1.5, fh + 0.5, fh + 0.5};
        double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
        bool ok = true;
        for (int k = 0; k < 4; k++) {
            const double w = S[6] * cx[k] + S[7] * cy[k] + S[8];
            const double X = (S[0] * cx[k] + S[1] * cy[k] + S[2]) / w;
            const double Y = (S[3] * cx[k] + S[4] * cy[k] + S[5]) / w;
            if (!(w > 1e-9) || !(fabs(X) < 1e9) || !(fabs(Y) < 1e9)) ok = false;
            mnx = fmin(mnx, X);
            mxx = fmax(mxx, X);
            mny = fmin(mny, Y);
            mxy = fmax(mxy, Y);
        }
        if (ok) {
            x0 = max(0, (int)floor(mnx) - 2);
            y0 = max(0, (int)floor(mny) - 2);
            x1 = min(cw, (int)ceil(mxx) + 3);
            y1 = min(ch, (int)ceil(mxy) + 3);
            if (x1 <= x0 || y1 <= y0) x0 = y0 = x1 = y1 = 0;
        }
    }
    wp->rx0 = x0;
    wp->ry0 = y0;
    wp->rx1 = x1;
    wp->ry1 = y1;
    const bool empty = x1 <= x0;
    wp->wx0 = empty ? 0 : max(0, x0 - 15);
    wp->wy0 = empty ? 0 : max(0, y0 - 15);
    wp->wx1 = empty ? 0 : min(cw, x1 + 15);
    wp->wy1 = empty ? 0 : min(ch, y1 + 15);
    const int bh0 = ch < 16 ? ch : 16;
    wp->bw0 = (1024 / bh0) < cw ? (1024 / bh0) : cw;
    if (jobs[0]) {
        // new mask: transform on the ROI (its rim is zero, so nothing outside can be nearer)
        jobs[0]->x0 = jobs[0]->ox0 = x0;
        jobs[0]->y0 = jobs[0]->oy0 = y0;
        jobs[0]->x1 = jobs[0]->ox1 = x1;
        jobs[0]->y1 = jobs[0]->oy1 = y1;
    }
    if (jobs[1]) {
        // old mask: distances needed where the weights are; the rectangle of the transform (here the
        // whole canvas) is narrowed by k_old_rects
        jobs[1]->x0 = 0;
        jobs[1]->y0 = 0;
        jobs[1]->x1 = empty ? 0 : cw;
        jobs[1]->y1 = empty ? 0 : ch;
        jobs[1]->ox0 = wp->wx0;
        jobs[1]->oy0 = wp->wy0;
        jobs[1]->ox1 = wp->wx1;
        jobs[1]->oy1 = wp->wy1;
    }
}

// The rectangle the distance transform of the canvas mask is computed on, per frame of the batch.
// The reference transforms the whole canvas, but everything outside the bounding box of what
// was ever pasted or warped onto it is zero, and a zero pixel beyond the one-pixel ring around
// that box is never nearer (in 3x3 chamfer steps) than the ring pixel a path to it crosses: the
// transform on box + ring, joined with the window the weights are read in, gives the same
// distances.  cbox = bounding box so far (x0, y0, x1, y1; regions of interest, so a superset of
// the non-zero pixels, which is the safe side); frame j also counts the frames before it in the
// batch.  One block; thread j = frame j.
__device__ void old_rects(const WarpParams* wp, DtJob* jobs_old, int n, int* cbox, int ch, int cw) {
    // the regions of interest of the batch's frames side by side in LDS (x0, y0, x1, y1; an empty one as
    // the neutral element of the joins below): every lane walks the frames before its own
    __shared__ int4 s_roi[RTVM_MAX_BATCH];
    const int j = threadIdx.x;
    int x0 = cbox[0], y0 = cbox[1], x1 = cbox[2], y1 = cbox[3];
    int4 own = make_int4(0x7FFFFFFF, 0x7FFFFFFF, 0, 0);
    bool drawn = false;
    if (j < n) {
        const int4 r = make_int4(wp[j].rx0, wp[j].ry0, wp[j].rx1, wp[j].ry1);
        drawn = r.z > r.x;
        if (drawn) own = r;
        s_roi[j] = own;
    }
    __syncthreads();  // cbox is read by every lane before lane 0 replaces it
    if (drawn) {
        for (int i = 0; i < j; i++) {
            const int4 r = s_roi[i];
            x0 = min(x0, r.x), y0 = min(y0, r.y);
            x1 = max(x1, r.z), y1 = max(y1, r.w);
        }
        // the ring of zeros around the box, then the window of this frame
        x0 = min(max(x0 - 1, 0), wp[j].wx0), y0 = min(max(y0 - 1, 0), wp[j].wy0);
        x1 = max(min(x1 + 1, cw), wp[j].wx1), y1 = max(min(y1 + 1, ch), wp[j].wy1);
        jobs_old[j].x0 = x0;
        jobs_old[j].y0 = y0;
        jobs_old[j].x1 = x1;
        jobs_old[j].y1 = y1;
    }
    if (j == 0) {
        // cbox[4..7]: the box of this batch alone (the commit kernel leaves everything outside it alone)
        int u[4] = {0x7FFFFFFF, 0x7FFFFFFF, 0, 0};
        for (int i = 0; i < n; i++) {
            const int4 r = s_roi[i];
            u[0] = min(u[0], r.x), u[1] = min(u[1], r.y);
            u[2] = max(u[2], r.z), u[3] = max(u[3], r.w);
        }
        const int b[4] = {min(cbox[0], u[0]), min(cbox[1], u[1]), max(cbox[2], u[2]), max(cbox[3], u[3])};
        for (int k = 0; k < 4; k++) cbox[k] = b[k], cbox[4 + k] = u[k];
    }
}

// Stage entry points: the parameters of one frame, no distance-transform jobs.
__global__ void k_warp_prep(const double* __restrict__ Hp, WarpParams* __restrict__ wp, int fh, int fw, int ch, int cw,
                            int force_full) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        warp_prep_frame(0, Hp, wp, nullptr, nullptr, fh, fw, ch, cw, force_full, nullptr);
}

// The frames of a batch: one block, thread j = frame j; then the rectangles of the old-mask jobs.
__global__ __launch_bounds__(64) void k_warp_prep_batch(const double* Hp, WarpParams* wp, DtJob* jobs_new, DtJob* jobs_old,
                                                        int n, int fh, int fw, int ch, int cw, const int* skip,
                                                        StreamState* st, int* cbox) {
    const int j = threadIdx.x;
    if (st && j == 0) st->bad_frame = 0x7FFFFFFF;  // no blend of this batch has cleared a pixel yet
    if (j < n) warp_prep_frame(j, Hp, wp, jobs_new, jobs_old, fh, fw, ch, cw, 0, skip);
    __syncthreads();
    old_rects(wp, jobs_old, n, cbox, ch, cw);
}

// The 2x2 taps at (ix, iy) .. (ix+1, iy+1), 3 channels each; taps outside the frame read 0
// (BORDER_CONSTANT); t[tap][channel], tap = 2*dy + dx.  The two taps of a row are six
// consecutive bytes: one unaligned 8-byte load per row, from an address clamped so that it
// stays inside the frame (at least 8 bytes and 2 columns, checked by the caller).
__device__ __forceinline__ void taps4(const uint8_t* __restrict__ src, int fh, int fw, int iy, int ix, int t[4][3]) {
    const bool vx0 = (unsigned)ix < (unsigned)fw, vx1 = (unsigned)(ix + 1) < (unsigned)fw;
    const bool vy0 = (unsigned)iy < (unsigned)fh, vy1 = (unsigned)(iy + 1) < (unsigned)fh;
    const int xlo = min(max(ix, 0), fw - 2);  // column of the first of the six bytes
    const int y0 = min(max(iy, 0), fh - 1), y1 = min(max(iy + 1, 0), fh - 1);
    // byte offsets in 32 bits (a frame is far below 4 GB), rows and widths below 2^24: full-rate multiplies
    const uint32_t last = (uint32_t)fh * (uint32_t)fw * 3u - 8u;
    const uint32_t a0 = (__umul24((uint32_t)y0, (uint32_t)fw) + (uint32_t)xlo) * 3u;
    const uint32_t a1 = (__umul24((uint32_t)y1, (uint32_t)fw) + (uint32_t)xlo) * 3u;
    const uint32_t b0 = min(a0, last), b1 = min(a1, last);
    uint64_t r0, r1;
    __builtin_memcpy(&r0, src + b0, 8);
    __builtin_memcpy(&r1, src + b1, 8);
    r0 >>= 8 * (int)(a0 - b0);  // at most 2 bytes: the six needed ones stay
    r1 >>= 8 * (int)(a1 - b1);
    // tap ix is pixel (ix - xlo) of the six bytes, tap ix+1 the next one (when inside the frame)
    const int s0 = 24 * min(max(ix - xlo, 0), 2), s1 = 24 * min(max(ix + 1 - xlo, 0), 2);
    const uint32_t p00 = (vy0 && vx0) ? (uint32_t)(r0 >> s0) : 0u, p01 = (vy0 && vx1) ? (uint32_t)(r0 >> s1) : 0u;
    const uint32_t p10 = (vy1 && vx0) ? (uint32_t)(r1 >> s0) : 0u, p11 = (vy1 && vx1) ? (uint32_t)(r1 >> s1) : 0u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        t[0][c] = (p00 >> (8 * c)) & 255;
        t[1][c] = (p01 >> (8 * c)) & 255;
        t[2][c] = (p10 >> (8 * c)) & 255;
        t[3][c] = (p11 >> (8 * c)) & 255;
    }
}

// cv2.warpPerspective(frame, H, (cw, ch), INTER_LINEAR), BORDER_CONSTANT 0, for the
// pixels of the region of interest.  FIXED: 1/32-pixel fixed point (OpenCV <= 4.10),
// else float32 (OpenCV >= 4.11).  Output: 3 bytes per pixel into out3, or, for the frames of
// a batch, one word per pixel into out4: B, G, R and the mask byte (any channel > 0).
// One tile of 64 x 4 canvas pixels at (bx, by), bx a multiple of 64.
template <bool FIXED>
__device__ __forceinline__ void warp_tile(const uint8_t* __restrict__ src, const WarpParams* __restrict__ wp,
                                          gptr<uint8_t> out3, gptr<uint32_t> out4, const DtJob* __restrict__ job_new,
                                          int fh, int fw, int cw, int bx, int by) {
    const int rx0 = wp->rx0, ry0 = wp->ry0, rx1 = wp->rx1, ry1 = wp->ry1;
    const int x = bx + (threadIdx.x & 63), y = by + (threadIdx.x >> 6);
    if (y < ry0 || y >= ry1) return;  // a wave is 64 pixels of one row
    const bool inroi = x >= rx0 && x < rx1;
    int v[3] = {0, 0, 0};
    if (FIXED) {
        const double* M = wp->M;
        const int bw0 = wp->bw0;
        const int xb = (x / bw0) * bw0, x1 = x - xb;
        const double X0 = M[0] * xb + M[1] * y + M[2];
        const double Y0 = M[3] * xb + M[4] * y + M[5];
        const double W0 = M[6] * xb + M[7] * y + M[8];
        double W = W0 + M[6] * x1;
        W = W ? 32. / W : 0;
        const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
        const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
        const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
        int sx = X >> 5, sy = Y >> 5;
        sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;
        sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
        const int ax = X & 31, ay = Y & 31;
        const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
        const int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
        if (inroi && sx >= -1 && sx < fw && sy >= -1 && sy < fh) {
            int t[4][3];
            taps4(src, fh, fw, sy, sx, t);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int s = w00 * t[0][c] + w01 * t[1][c] + w10 * t[2][c] + w11 * t[3][c];
                v[c] = (s + 16384) >> 15;
            }
        }
    } else {
        const float* Mf = wp->Mf;
        const float fx = (float)x, fy = (float)y;
        const float w = (Mf[6] * fx + Mf[7] * fy) + Mf[8];
        const float sx = ((Mf[0] * fx + Mf[1] * fy) + Mf[2]) / w;
        const float sy = ((Mf[3] * fx + Mf[4] * fy) + Mf[5]) / w;
        if (inroi && sx > -1.f && sx < (float)fw && sy > -1.f && sy < (float)fh) {
            const float flx = floorf(sx), fly = floorf(sy);
            const int ix = (int)flx, iy = (int)fly;
            const float a = sx - flx, b = sy - fly;
            int t[4][3];
            taps4(src, fh, fw, iy, ix, t);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float p00 = (float)t[0][c], p01 = (float)t[1][c];
                const float p10 = (float)t[2][c], p11 = (float)t[3][c];
                const float v0 = p00 + (p01 - p00) * a;
                const float v1 = p10 + (p11 - p10) * a;
                const float r = v0 + (v1 - v0) * b;
                const int q = __float2int_rn(r);
                v[c] = q < 0 ? 0 : q > 255 ? 255 : q;
            }
        }
    }
    const bool zero = (v[0] | v[1] | v[2]) == 0;
    if (inroi) {
        const size_t p = (size_t)y * cw + x;
        if (out4) {  // the pixel and its mask byte (main.py:878-879) in one word
            out4[p] = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | (zero ? 0u : 0xFF000000u);
        } else {
            out3[3 * p] = (uint8_t)v[0];
            out3[3 * p + 1] = (uint8_t)v[1];
            out3[3 * p + 2] = (uint8_t)v[2];
        }
    }
    if (job_new) {
        // zero bit rows of the new mask for the distance transform (zero pixels of the ROI)
        const unsigned long long bal = __ballot(inroi && zero);
        if ((threadIdx.x & 63) == 0) {
            const auto B = as_global(job_new->B) + (size_t)y * job_new->wpitch + (bx >> 5);
            B[0] = (uint32_t)bal;
            B[1] = (uint32_t)(bal >> 32);
        }
    }
}

// Where canvas pixel (x, y) lies in the frame: the top-left tap (ix, iy), clamped to [-2, fw + 1] x
// [-2, fh + 1] so that it is an integer whatever the arithmetic produced, and the interpolation fractions
// (FIXED: 0..31 as integers; else the float fractions as bit patterns).  Returns whether the pixel takes
// taps at all (warp_tile's test); then the clamps changed nothing.
template <bool FIXED>
__device__ __forceinline__ bool warp_pos(const WarpParams* __restrict__ wp, int x, int y, int fh, int fw, int& ix, int& iy,
                                         int& qa, int& qb) {
    if (FIXED) {
        const double* M = wp->M;
        const int bw0 = wp->bw0;
        const int xb = (x / bw0) * bw0, x1 = x - xb;
        const double X0 = M[0] * xb + M[1] * y + M[2];
        const double Y0 = M[3] * xb + M[4] * y + M[5];
        const double W0 = M[6] * xb + M[7] * y + M[8];
        double W = W0 + M[6] * x1;
        W = W ? 32. / W : 0;
        const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
        const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
        const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
        int sx = X >> 5, sy = Y >> 5;
        sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;
        sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
        qa = X & 31;
        qb = Y & 31;
        ix = min(max(sx, -2), fw + 1);
        iy = min(max(sy, -2), fh + 1);
        return sx >= -1 && sx < fw && sy >= -1 && sy < fh;
    } else {
        const float* Mf = wp->Mf;
        const float fx = (float)x, fy = (float)y;
        const float w = (Mf[6] * fx + Mf[7] * fy) + Mf[8];
        const float sx = ((Mf[0] * fx + Mf[1] * fy) + Mf[2]) / w;
        const float sy = ((Mf[3] * fx + Mf[4] * fy) + Mf[5]) / w;
        const float flx = floorf(sx), fly = floorf(sy);
        qa = __float_as_int(sx - flx);
        qb = __float_as_int(sy - fly);
        ix = (int)fminf(fmaxf(flx, -2.f), (float)(fw + 1));  // a NaN becomes -2
        iy = (int)fminf(fmaxf(fly, -2.f), (float)(fh + 1));
        return sx > -1.f && sx < (float)fw && sy > -1.f && sy < (float)fh;
    }
}

// The interpolated pixel from its four taps, each a word B | G << 8 | R << 16: warp_tile's arithmetic.
template <bool FIXED>
__device__ __forceinline__ uint32_t warp_interp(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, int qa, int qb) {
    uint32_t out = 0;
    if (FIXED) {
        const int ax = qa, ay = qb;
        const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
        const int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int s = w00 * (int)((p00 >> (8 * c)) & 255) + w01 * (int)((p01 >> (8 * c)) & 255) +
                          w10 * (int)((p10 >> (8 * c)) & 255) + w11 * (int)((p11 >> (8 * c)) & 255);
            out |= (uint32_t)((s + 16384) >> 15) << (8 * c);
        }
    } else {
        const float a = __int_as_float(qa), b = __int_as_float(qb);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float f00 = (float)((p00 >> (8 * c)) & 255), f01 = (float)((p01 >> (8 * c)) & 255);
            const float f10 = (float)((p10 >> (8 * c)) & 255), f11 = (float)((p11 >> (8 * c)) & 255);
            const float v0 = f00 + (f01 - f00) * a;
            const float v1 = f10 + (f11 - f10) * a;
            const float r = v0 + (v1 - v0) * b;
            // round to nearest even, clamp to 0..255 and place the byte: one instruction (v_cvt_pk_u8_f32)
            out = __builtin_amdgcn_cvt_pk_u8_f32(r, c, out);
        }
    }
    return out;
}

constexpr int WT_ROWS = 16;      // canvas rows of a staged tile (64 columns: a wave is 64 pixels of a row)
constexpr int WT_WORDS = 4096;   // frame pixels staged per tile, one word each (16 KB)

struct WarpLds {
    uint32_t px[WT_WORDS];   // the frame's rectangle under the tile, B | G << 8 | R << 16 per pixel, 0 outside the frame
    int corner[8];           // (ix, iy) of the tile's four corner pixels
    int bad[2];              // by tile parity: a pixel's taps lie outside the staged rectangle
};

// One tile of 64 x 16 canvas pixels at (bx, by), four pixels per thread (column bx + lane, rows by + wave + 4 r).
// The frame's rectangle under the tile (from the tile's corners: a homography is monotone along every
// row and column of a tile on which its denominator keeps its sign) goes into LDS by 12-byte loads of four
// pixels, zero outside the frame (BORDER_CONSTANT), and the taps are two LDS reads of two words.  Every
// pixel checks that its taps are inside the rectangle; if one is not, or the rectangle is too large (strong
// minification), the tile is done by warp_tile's gathers from global memory.  `tile` counts the tiles of
// this workgroup.  Same results as warp_tile.
template <bool FIXED>
__device__ __forceinline__ void warp_tile_staged(const uint8_t* __restrict__ src, const WarpParams* __restrict__ wp,
                                                 gptr<uint8_t> out3, gptr<uint32_t> out4, const DtJob* __restrict__ job_new,
                                                 int fh, int fw, int cw, int bx, int by, WarpLds& s, int tile) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rx0 = wp->rx0, ry0 = wp->ry0, rx1 = wp->rx1, ry1 = wp->ry1;
    const int x = bx + lane;
    const bool incol = x >= rx0 && x < rx1;
    int ix[4], iy[4], qa[4], qb[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = by + wv + 4 * r;
        ok[r] = warp_pos<FIXED>(wp, x, y, fh, fw, ix[r], iy[r], qa[r], qb[r]) && incol && y >= ry0 && y < ry1;
    }
    if ((lane == 0 || lane == 63) && (wv == 0 || wv == 3)) {
        const int k = 2 * (lane & 1) + 4 * (wv & 1);
        s.corner[k] = wv ? ix[3] : ix[0];
        s.corner[k + 1] = wv ? iy[3] : iy[0];
    }
    __syncthreads();
    // the next tile's flag (every thread has read it for the tile before this one); this tile's was cleared a tile ago
    if (threadIdx.x == 0) s.bad[(tile + 1) & 1] = 0;
    // the rectangle of taps: the corners' taps and a pixel more on every side, inside [-1, fw] x [-1, fh]
    const int c0x = s.corner[0], c0y = s.corner[1], c1x = s.corner[2], c1y = s.corner[3];
    const int c2x = s.corner[4], c2y = s.corner[5], c3x = s.corner[6], c3y = s.corner[7];
    const int sx0 = max(min(min(c0x, c1x), min(c2x, c3x)) - 1, -1), sx1 = min(max(max(c0x, c1x), max(c2x, c3x)) + 2, fw);
    const int sy0 = max(min(min(c0y, c1y), min(c2y, c3y)) - 1, -1), sy1 = min(max(max(c0y, c1y), max(c2y, c3y)) + 2, fh);
    const int bw = sx1 - sx0 + 1, bh = sy1 - sy0 + 1;
    const int gpr = (bw + 3) >> 2, pw = 4 * gpr;   // groups of four pixels per staged row, words per staged row
    const bool stage = bw > 1 && bh > 1 && pw * bh <= WT_WORDS;
    int at[4];
    bool miss = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int dx = ix[r] - sx0, dy = iy[r] - sy0;
        const bool in = stage && (unsigned)dx < (unsigned)(bw - 1) && (unsigned)dy < (unsigned)(bh - 1);
        miss |= ok[r] && !in;
        at[r] = dy * pw + dx;
    }
    if (miss) s.bad[tile & 1] = 1;
    if (stage) {
        const int ng = gpr * bh;
        const float inv = __builtin_amdgcn_rcpf((float)gpr);
        for (int g = threadIdx.x; g < ng; g += 256) {
            // g / gpr: the quotient is below 1024 / gpr and (g + 0.5) / gpr at least 0.5 / gpr away from an integer
            const int row = (int)(((float)g + 0.5f) * inv), gi = g - row * gpr;
            const int yy = sy0 + row, xx = sx0 + 4 * gi;
            uint4 p = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)yy < (unsigned)fh) {
                const uint32_t off = (__umul24((uint32_t)yy, (uint32_t)fw) + (uint32_t)xx) * 3u;  // of pixel xx when it exists
                if (xx >= 0 && xx + 3 < fw) {
                    uint32_t d[3];
                    __builtin_memcpy(d, src + off, 12);
                    p.x = d[0];
                    p.y = __builtin_amdgcn_alignbit(d[1], d[0], 24);
                    p.z = __builtin_amdgcn_alignbit(d[2], d[1], 16);
                    p.w = d[2] >> 8;
                } else {
                    uint32_t q[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        q[k] = 0;
                        if ((unsigned)(xx + k) < (unsigned)fw) {
                            const uint8_t* b = src + (off + 3u * (uint32_t)k);
                            q[k] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
                        }
                    }
                    p = make_uint4(q[0], q[1], q[2], q[3]);
                }
            }
            *reinterpret_cast<uint4*>(&s.px[row * pw + 4 * gi]) = p;
        }
    }
    __syncthreads();
    if (s.bad[tile & 1]) {  // the same for all threads
#pragma unroll 1
        for (int k = 0; k < WT_ROWS; k += 4) warp_tile<FIXED>(src, wp, out3, out4, job_new, fh, fw, cw, bx, by + k);
        return;
    }
    gptr<uint32_t> bits = nullptr;   // the tile's words of the bit rows: row y at bits + y * bpitch
    uint32_t bpitch = 0;
    if (job_new) {
        bits = as_global(job_new->B) + (bx >> 5);
        bpitch = (uint32_t)job_new->wpitch;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = by + wv + 4 * r;
        if (y < ry0 || y >= ry1) continue;  // the same for the wave
        uint32_t v = 0;
        if (ok[r]) {
            const int i = at[r];
            v = warp_interp<FIXED>(s.px[i], s.px[i + 1], s.px[i + pw], s.px[i + pw + 1], qa[r], qb[r]) & 0xFFFFFFu;
        }
        const bool zero = v == 0;
        if (incol) {
            const uint32_t p = (uint32_t)y * (uint32_t)cw + (uint32_t)x;   // a canvas is far below 2^30 pixels
            if (out4) {  // the pixel and its mask byte (main.py:878-879) in one word
                out4[p] = v | (zero ? 0u : 0xFF000000u);
            } else {
                out3[3 * p] = (uint8_t)v;
                out3[3 * p + 1] = (uint8_t)(v >> 8);
                out3[3 * p + 2] = (uint8_t)(v >> 16);
            }
        }
        if (job_new) {
            // zero bit rows of the new mask for the distance transform (zero pixels of the ROI)
            const unsigned long long bal = __ballot(incol && zero);
            if (lane == 0) {
                const auto B = bits + (uint32_t)y * bpitch;
                B[0] = (uint32_t)bal;
                B[1] = (uint32_t)(bal >> 32);
            }
        }
    }
}

// blockIdx.z = frame of the batch: source srcs.p[z], parameters wp[z], output planes
// sides[z] (or `out_only`, 3 bytes per pixel, for the stage entry point), bit rows of jobs_new[z].
// The grid is sized for a region of interest of about the frame's size and walks the
// tiles of the actual one (tile (0,0) holds the ROI's first pixel).
template <bool FIXED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_warp(BatchPtrs srcs, const WarpParams* __restrict__ wps,
                                              const SideDev* __restrict__ sides, uint8_t* __restrict__ out_only,
                                              const DtJob* __restrict__ jobs_new, int fh, int fw, int ch, int cw) {
    __shared__ __align__(16) WarpLds s;
    const uint8_t* __restrict__ src = srcs.p[blockIdx.z];
    const WarpParams* __restrict__ wp = wps + blockIdx.z;
    const gptr<uint8_t> out3 = as_global(out_only);
    const gptr<uint32_t> out4 = out_only ? nullptr : as_global(sides[blockIdx.z].warped);
    const DtJob* __restrict__ job_new = jobs_new ? jobs_new + blockIdx.z : nullptr;
    const int rx1 = wp->rx1, ry1 = wp->ry1;
    const int ox = wp->rx0 & ~63, oy = wp->ry0;
    if (threadIdx.x == 0) s.bad[0] = 0;
    int tile = 0;
#pragma unroll 1
    for (int by = oy + WT_ROWS * (int)blockIdx.y; by < ry1; by += WT_ROWS * (int)gridDim.y)
#pragma unroll 1
        for (int bx = ox + 64 * (int)blockIdx.x; bx < rx1; bx += 64 * (int)gridDim.x)
            warp_tile_staged<FIXED>(src, wp, out3, out4, job_new, fh, fw, cw, bx, by, s, tile++);
}

// The canvas mask each frame of a batch will meet, as zero bit rows, assuming that a pixel
// of the canvas is non-zero after a frame iff it was non-zero before or the warped frame is
// non-zero there (a blend of two non-zero pixels stays non-zero; the commit kernel reports
// the rare blend that does not).  One thread per word: frame j sees `cur`, then the zero
// bits of frame j's new mask are folded in.  Lets the distance transforms of the old mask
// run for the whole batch at once.
__global__ __launch_bounds__(256) void k_pred_bits(const uint32_t* __restrict__ bold, const WarpParams* __restrict__ wp,
                                                   const DtJob* __restrict__ jobs_new,
                                                   const DtJob* __restrict__ jobs_old, int n, int ch, int cw) {
    const int w = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= ch || w * 32 >= cw) return;
    const int wpitch = jobs_old[0].wpitch;
    uint32_t cur = bold[(size_t)y * wpitch + w];
    const int xa = w * 32;
    for (int j = 0; j < n; j++) {
        jobs_old[j].B[(size_t)y * wpitch + w] = cur;
        const int rx0 = wp[j].rx0, rx1 = wp[j].rx1;
        if (y < wp[j].ry0 || y >= wp[j].ry1 || xa + 32 <= rx0 || xa >= rx1) continue;  // new mask is zero here
        uint32_t roi = 0xFFFFFFFFu;
        if (xa < rx0) roi &= 0xFFFFFFFFu << (rx0 - xa);
        if (xa + 32 > rx1) roi &= 0xFFFFFFFFu >> (xa + 32 - rx1);
        cur &= jobs_new[j].B[(size_t)y * wpitch + w] | ~roi;
    }
}

// weight_new = dn / (dn + do + 1e-6), weight_old = do / (...)  (main.py:892-894),
// evaluated while staging, then the row pass of cv2.GaussianBlur(w, (31,31), 0)
// (main.py:897-898) for both weights: s = k0*x0, then s = fma(x_i, k_i, s) left to right in
// float32, BORDER_REFLECT_101 at the canvas edge.  dn is zero outside the ROI.  The two
// weights of a pixel travel as one pair {new, old}, so every tap is one packed fma.  One
// wave = BR_TILE consecutive pixels of one row staged through LDS, BR_OUT outputs per lane from a
// window of 34 staged pairs; a workgroup takes 4 rows; blockIdx.z = frame of the batch.
// outputs per lane in the row pass of the blur: a wave takes BR_TILE consecutive pixels of a row
constexpr int BR_OUT = 4, BR_TILE = 64 * BR_OUT, BR_STAGE = BR_TILE + 32, BR_LOADS = (BR_TILE + 30 + 63) / 64;
// The staged pairs are read back as 16-byte words at a lane stride of 32 bytes (four pairs).  A 16-byte
// read is served in four groups of 16 lanes (lanes 0-3, 12-15 and 20-27 first), 64 banks of four
// bytes: in pair order the lanes of a group meet two by two on the same banks.  Pairs whose index has
// bit 5 set keep their two 16-byte words in swapped order (pair e lives at e ^ 2 there): every group
// then covers all banks once.
constexpr int BR_STAGE_PAD = BR_STAGE;
__device__ __forceinline__ int br_pad(int e) { return e ^ ((e >> 4) & 2); }

__device__ __forceinline__ void blur_rows_tile(const WarpParams* __restrict__ wp, const float* __restrict__ gk,
                                               const SideDev& side, int cw, int bx, int by, f32x2 (*s_w)[BR_STAGE_PAD]) {
    const int rx0 = wp->rx0, rx1 = wp->rx1, wy0 = wp->wy0, wy1 = wp->wy1;
    const auto dn = as_global<const float>(side.dist_new);
    const auto dold = as_global<const float>(side.dist_old);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = by + wv;
    const bool row_ok = y >= wy0 && y < wy1;
    const size_t row = (size_t)y * cw;
    if (row_ok) {
        const bool row_in_roi = y >= wp->ry0 && y < wp->ry1;
        const int wx0 = wp->wx0, wx1 = wp->wx1;
        // all distances of the lane requested first, the divisions after: one wait, not five
        float a[BR_LOADS], b[BR_LOADS];
        bool ok[BR_LOADS];
#pragma unroll
        for (int q = 0; q < BR_LOADS; q++) {
            const int i = lane + 64 * q;
            const int xs = reflect101(bx - 15 + i, cw);
            // distances exist on [wx0, wx1); columns this row pass never needs may lie outside
            ok[q] = i < BR_TILE + 30 && xs >= wx0 && xs < wx1;
            const int xc = min(max(xs, wx0), wx1 - 1);
            b[q] = dold[row + xc];
            a[q] = (row_in_roi && xs >= rx0 && xs < rx1) ? dn[row + xc] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < BR_LOADS; q++) {
            const int i = lane + 64 * q;
            float wn = 0.f, wo = 0.f;
            if (ok[q]) {
                const float sum = (a[q] + b[q]) + 1e-6f;
                wn = a[q] / sum;
                wo = b[q] / sum;
            }
            if (i < BR_STAGE) s_w[wv][br_pad(i)] = f32x2{wn, wo};
        }
    }
    // a wave reads back only what it staged itself: LDS keeps a wave's accesses in order, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!row_ok) return;
    const int x = bx + BR_OUT * lane;
    if (x + BR_OUT <= rx0 || x >= rx1) return;
    f32x2 v[BR_OUT + 32];
#pragma unroll
    for (int q = 0; q < (BR_OUT + 32) / 2; q++) {
        const float4 a = *reinterpret_cast<const float4*>(&s_w[wv][br_pad(BR_OUT * lane + 2 * q)]);
        v[2 * q] = f32x2{a.x, a.y};
        v[2 * q + 1] = f32x2{a.z, a.w};
    }
    const auto outp = as_global(side.rb) + row + x;
    // The sums of the lane advance tap by tap side by side, so that no packed operation waits for the one
    // before it.  Spelled out as assembly: the scheduler strings the taps of each sum into one chain of
    // dependent packed operations, every one a wait state (s_nop) behind its predecessor, whatever order
    // the source gives them.  The wait state before the first group and after the last covers the
    // neighbouring compiler-generated instructions, whose hazards the compiler cannot see into the assembly.
    static_assert(BR_OUT == 4, "the groups below take four sums");
    f32x2 a[BR_OUT];
    {
        const float k0 = gk[0];
#pragma unroll
        for (int o = 0; o < BR_OUT; o++) a[o] = f32x2{k0, k0} * v[o];
        asm("s_nop 0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    }
#pragma unroll
    for (int i = 1; i < 31; i++) {
        const float ki = gk[i];
        asm("v_pk_fma_f32 %0, %4, %8, %0\n\t"
            "v_pk_fma_f32 %1, %5, %8, %1\n\t"
            "v_pk_fma_f32 %2, %6, %8, %2\n\t"
            "v_pk_fma_f32 %3, %7, %8, %3"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
            : "v"(v[i]), "v"(v[i + 1]), "v"(v[i + 2]), "v"(v[i + 3]), "s"(f32x2{ki, ki}));
    }
    asm("s_nop 0" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    if (x >= rx0 && x + BR_OUT <= rx1) {   // all but the lanes on the region's edges: two 16-byte stores
        typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        *reinterpret_cast<gptr<f32x4a8>>(outp) = f32x4a8{a[0].x, a[0].y, a[1].x, a[1].y};
        *reinterpret_cast<gptr<f32x4a8>>(outp + 2) = f32x4a8{a[2].x, a[2].y, a[3].x, a[3].y};
        return;
    }
#pragma unroll
    for (int o = 0; o < BR_OUT; o++)
        if (x + o >= rx0 && x + o < rx1) outp[o] = a[o];
}

// blockIdx.z = frame of the batch.  Tiles of BR_TILE x 4 pixels of the frame's region (rows of
// the blend window, columns of the ROI), tile (0,0) holding the region's first pixel.  The grid
// (a multiple of 8 workgroups) is sized for a region of about the frame's size: one tile per
// workgroup; a larger region (normally none) is walked with the grid as stride.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void k_blur_rows(const WarpParams* __restrict__ wps, const float* __restrict__ gk,
                                                   const SideDev* __restrict__ sides, int cw) {
    const WarpParams* __restrict__ wp = wps + blockIdx.z;
    const SideDev side = sides[blockIdx.z];
    __shared__ __align__(16) f32x2 s_w[4][BR_STAGE_PAD];
    const int rx1 = wp->rx1, wy1 = wp->wy1;
    const int ox = wp->rx0 & ~3, oy = wp->wy0;
    const int ntx = rx1 > ox ? (rx1 - ox + BR_TILE - 1) / BR_TILE : 0, nty = wy1 > oy ? (wy1 - oy + 3) / 4 : 0;
    const int T = ntx * nty, G = gridDim.x;
    if (T <= G) {  // the usual case: the region's tiles in one contiguous run per XCD
        const int t = xcd_tile(blockIdx.x, T);
        if (t >= 0) blur_rows_tile(wp, gk, side, cw, ox + BR_TILE * (t % ntx), oy + 4 * (t / ntx), s_w);
        return;
    }
    for (int t = blockIdx.x; t < T; t += G) {  // a region larger than the grid was sized for
        blur_rows_tile(wp, gk, side, cw, ox + BR_TILE * (t % ntx), oy + 4 * (t / ntx), s_w);
        // the wave's staging row is reused by its next tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Column pass of the blur (s = k15*r0, then s = fma(r[+t] + r[-t], k[15+t], s) outwards) for
// the pixels of the ROI the warped frame covers: the final blend weights of the frame and of
// the canvas (main.py:897-898), stored over the distance planes.  A lane takes 4 consecutive
// rows of one column: a window of 34 row-blurred weight pairs serves the 4 column sums, both
// weights in one packed operation.  blockIdx.z = frame of the batch.
// rows per lane in the column pass of the blur (a workgroup of four waves covers 64 x 4*BC_ROWS pixels)
constexpr int BC_ROWS = 12;

// The column sums of R consecutive rows (ya ..) of column x from a window of 30 + R row-blurred pairs.
template <bool EDGE, int R>
__device__ __forceinline__ void blur_cols_sums(const float* __restrict__ gk, const SideDev& side, int ch, int cw, int x,
                                               int ya, f32x2 (&a)[R]) {
    const auto rb = as_global<const f32x2>(side.rb);
    f32x2 v[30 + R];
#pragma unroll
    for (int j = 0; j < 30 + R; j++) {
        const int y = EDGE ? reflect101(ya - 15 + j, ch) : ya - 15 + j;
        v[j] = rb[(size_t)y * cw + x];
    }
    {
        const float k0 = gk[15];
#pragma unroll
        for (int i = 0; i < R; i++) a[i] = f32x2{k0, k0} * v[15 + i];
    }
    // Groups of four sums as assembly, the four additions before the four multiply-adds, for the reason given
    // in the row pass: left to the scheduler every sum becomes one chain of add, wait state, multiply-add.
    static_assert(R % 4 == 0, "the groups below take four sums");
#pragma unroll
    for (int i = 0; i < R; i += 4) asm("s_nop 0" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]));
#pragma unroll
    for (int t = 1; t <= 15; t++) {
        const float kt = gk[15 + t];
#pragma unroll
        for (int i = 0; i < R; i += 4) {
            f32x2 t0, t1, t2, t3;
            asm("v_pk_add_f32 %4, %8, %12\n\t"
                "v_pk_add_f32 %5, %9, %13\n\t"
                "v_pk_add_f32 %6, %10, %14\n\t"
                "v_pk_add_f32 %7, %11, %15\n\t"
                "v_pk_fma_f32 %0, %4, %16, %0\n\t"
                "v_pk_fma_f32 %1, %5, %16, %1\n\t"
                "v_pk_fma_f32 %2, %6, %16, %2\n\t"
                "v_pk_fma_f32 %3, %7, %16, %3"
                : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "v"(v[15 + i + t]), "v"(v[16 + i + t]), "v"(v[17 + i + t]), "v"(v[18 + i + t]), "v"(v[15 + i - t]),
                  "v"(v[16 + i - t]), "v"(v[17 + i - t]), "v"(v[18 + i - t]), "s"(f32x2{kt, kt}));
        }
    }
#pragma unroll
    for (int i = 0; i < R; i += 4) asm("s_nop 0" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]));
}

template <bool EDGE>
__device__ __forceinline__ void blur_cols_window(const WarpParams* __restrict__ wp, const float* __restrict__ gk,
                                                 const SideDev& side, int ch, int cw, int x, int ya,
                                                 const bool (&need)[BC_ROWS]) {
    const auto out_n = as_global(side.dist_new), out_o = as_global(side.dist_old);
    f32x2 a[BC_ROWS];
    blur_cols_sums<EDGE, BC_ROWS>(gk, side, ch, cw, x, ya, a);
#pragma unroll
    for (int i = 0; i < BC_ROWS; i++)
        if (need[i]) {
            out_n[(size_t)(ya + i) * cw + x] = a[i].x;
            out_o[(size_t)(ya + i) * cw + x] = a[i].y;
        }
}

__device__ __forceinline__ void blur_cols_tile(const WarpParams* __restrict__ wp, const float* __restrict__ gk,
                                               const SideDev& side, int ch, int cw, int bx, int by) {
    const int rx0 = wp->rx0, ry0 = wp->ry0, rx1 = wp->rx1, ry1 = wp->ry1;
    // a wave is 64 columns of BC_ROWS rows: the row is the same for all its lanes, and saying so keeps
    // the row addresses in scalar registers
    const int x = bx + (threadIdx.x & 63), ya = by + BC_ROWS * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (x < rx0 || x >= rx1 || ya >= ry1 || ya + BC_ROWS <= ry0) return;
    // every pixel of the ROI gets its weights (only those under the frame's mask are ever read):
    // asking the mask first would put one more memory round trip in front of the window loads
    bool need[BC_ROWS];
#pragma unroll
    for (int i = 0; i < BC_ROWS; i++) need[i] = ya + i >= ry0 && ya + i < ry1;
    // only the tiles at the top and bottom of the canvas reflect rows (the same for the whole workgroup)
    if (by - 15 >= 0 && by + 4 * BC_ROWS + 15 <= ch)
        blur_cols_window<false>(wp, gk, side, ch, cw, x, ya, need);
    else
        blur_cols_window<true>(wp, gk, side, ch, cw, x, ya, need);
}

// blockIdx.z = frame of the batch.  Tiles of 64 x 4*BC_ROWS pixels of the ROI; grid as in k_blur_rows.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_blur_cols(const WarpParams* __restrict__ wps, const float* __restrict__ gk,
                                                   const SideDev* __restrict__ sides, int ch, int cw) {
    const WarpParams* __restrict__ wp = wps + blockIdx.z;
    const SideDev side = sides[blockIdx.z];
    const int rx1 = wp->rx1, ry1 = wp->ry1;
    const int ox = wp->rx0 & ~63, oy = wp->ry0;
    const int ntx = rx1 > ox ? (rx1 - ox + 63) / 64 : 0, nty = ry1 > oy ? (ry1 - oy + 4 * BC_ROWS - 1) / (4 * BC_ROWS) : 0;
    const int T = ntx * nty, G = gridDim.x;
    if (T <= G) {  // the usual case: the region's tiles in one contiguous run per XCD
        const int t = xcd_tile(blockIdx.x, T);
        if (t >= 0) blur_cols_tile(wp, gk, side, ch, cw, ox + 64 * (t % ntx), oy + 4 * BC_ROWS * (t / ntx));
        return;
    }
    for (int t = blockIdx.x; t < T; t += G)  // a region larger than the grid was sized for
        blur_cols_tile(wp, gk, side, ch, cw, ox + 64 * (t % ntx), oy + 4 * BC_ROWS * (t / ntx));
}

// Blend and commit of all frames of a batch (main.py:905-927).  With the blend weights
// of every frame already known, what a frame does to a canvas pixel depends on that pixel
// alone, so one thread carries its pixel through the frames in order:
//   new & old : canvas = trunc_u8(canvas * w_old + warped * w_new)
//   new only  : canvas = warped
// and writes canvas, mask_old (pixel has a non-zero channel) and the mask's zero bit rows
// (the seed of the next batch's predicted masks; a wave is 64 pixels of a row and rewrites
// their two bit words) once.
//
// The weights of frames 1..n-1 rest on predicted canvas masks, and the prediction fails
// when two non-zero pixels blend to zero: the first such frame goes into st->bad_frame.
// REDO = false is the optimistic pass: it applies every frame and saves what the canvas
// held before into the snapshot planes.  REDO = true runs after it, does nothing unless a
// frame before the last went bad, and then rebuilds the touched pixels from the snapshot
// with the frames up to the bad one only (the later ones come back pending).
template <bool REDO>
__global__ __launch_bounds__(256) void k_commit(const WarpParams* __restrict__ wps, const SideDev* __restrict__ sides,
                                                int n, uint8_t* __restrict__ mask_old, uint8_t* __restrict__ canvas,
                                                uint8_t* __restrict__ snap_mask, uint8_t* __restrict__ snap_canvas,
                                                uint32_t* __restrict__ bold, int wpitch, int ch, int cw,
                                                StreamState* __restrict__ st, const int* __restrict__ ubox) {
    int last = n - 1;
    if (REDO) {
        last = st->bad_frame;
        if (last >= n - 1) return;  // every frame was applied with the right weights
    }
    const int bx = blockIdx.x * 64, by = blockIdx.y * 4;
    // outside the box of the batch's regions of interest nothing happens (most of a large canvas)
    if (bx >= ubox[2] || bx + 64 <= ubox[0] || by >= ubox[3] || by + 4 <= ubox[1]) return;
    const int x = bx + (threadIdx.x & 63), y = by + (threadIdx.x >> 6);
    // does any frame of the batch touch this wave's 64 pixels?
    bool touched = false;
    for (int j = 0; j < n; j++) {
        const WarpParams* __restrict__ wp = wps + j;
        touched |= y >= wp->ry0 && y < wp->ry1 && bx < wp->rx1 && bx + 64 > wp->rx0;
    }
    if (!touched) return;
    const bool oncanvas = x < cw;  // y < ch: inside some ROI
    const size_t p = (size_t)y * cw + (oncanvas ? x : 0);
    int c0 = 0, c1 = 0, c2 = 0, mo = 255;
    if (oncanvas) {
        const uint8_t* __restrict__ src_c = REDO ? snap_canvas : canvas;
        const uint8_t* __restrict__ src_m = REDO ? snap_mask : mask_old;
        c0 = src_c[3 * p], c1 = src_c[3 * p + 1], c2 = src_c[3 * p + 2];
        mo = src_m[p];
        if (!REDO) {
            snap_canvas[3 * p] = (uint8_t)c0;
            snap_canvas[3 * p + 1] = (uint8_t)c1;
            snap_canvas[3 * p + 2] = (uint8_t)c2;
            snap_mask[p] = (uint8_t)mo;
        }
    }
    bool changed = REDO;
    // four frames at a time: their pixels and weights are requested together, then blended in order
    constexpr int G = 4;
    for (int j0 = 0; j0 <= last; j0 += G) {
        uint32_t px[G];
        float wa[G], wb[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int j = j0 + u;
            px[u] = 0;
            wa[u] = wb[u] = 0.f;
            if (j > last) continue;
            const WarpParams* __restrict__ wp = wps + j;
            if (y < wp->ry0 || y >= wp->ry1 || bx >= wp->rx1 || bx + 64 <= wp->rx0) continue;  // wave-uniform
            if (!oncanvas || x < wp->rx0 || x >= wp->rx1) continue;
            px[u] = as_global<const uint32_t>(sides[j].warped)[p];
            wa[u] = as_global<const float>(sides[j].dist_new)[p];
            wb[u] = as_global<const float>(sides[j].dist_old)[p];
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            if (!(px[u] >> 24)) continue;  // outside the frame's mask (or its region)
            const int w0 = px[u] & 255, w1 = (px[u] >> 8) & 255, w2 = (px[u] >> 16) & 255;
            changed = true;
            if (!mo) {
                c0 = w0, c1 = w1, c2 = w2;
                mo = 255;
            } else {
                const float a = wa[u], b = wb[u];
                c0 = (int)((float)c0 * b + (float)w0 * a) & 255;  // astype(uint8) truncates
                c1 = (int)((float)c1 * b + (float)w1 * a) & 255;
                c2 = (int)((float)c2 * b + (float)w2 * a) & 255;
                mo = (c0 | c1 | c2) ? 255 : 0;
                // two non-zero pixels blended to zero: the masks predicted for the later frames of
                // the batch do not hold
                if (!REDO && !mo) atomicMin(&st->bad_frame, j0 + u);
            }
        }
    }
    if (oncanvas && changed) {
        canvas[3 * p] = (uint8_t)c0;
        canvas[3 * p + 1] = (uint8_t)c1;
        canvas[3 * p + 2] = (uint8_t)c2;
        mask_old[p] = (uint8_t)mo;
    }
    const unsigned long long bal = __ballot(oncanvas && mo == 0);
    if ((threadIdx.x & 63) == 0) {
        uint32_t* B = bold + (size_t)y * wpitch + (bx >> 5);
        B[0] = (uint32_t)bal;
        B[1] = (uint32_t)(bal >> 32);
    }
}

// k_commit with four pixels of a row per lane, for canvas widths that are a multiple of 4 (the rows of every plane
// then start on 16-byte boundaries): the canvas' twelve bytes as three words, its mask bytes as one, the frames'
// pixels and weights as 16-byte loads - a quarter of the memory instructions, none of them a byte wide.  A wave is
// 256 pixels of a row; the zero bits of eight lanes make a word of the bit rows.  Same results as k_commit.
template <bool REDO>
__global__ __launch_bounds__(256) void k_commit4(const WarpParams* __restrict__ wps, const SideDev* __restrict__ sides,
                                                 int n, uint8_t* __restrict__ mask_old, uint8_t* __restrict__ canvas,
                                                 uint8_t* __restrict__ snap_mask, uint8_t* __restrict__ snap_canvas,
                                                 uint32_t* __restrict__ bold, int wpitch, int ch, int cw,
                                                 StreamState* __restrict__ st, const int* __restrict__ ubox) {
    int last = n - 1;
    if (REDO) {
        last = st->bad_frame;
        if (last >= n - 1) return;  // every frame was applied with the right weights
    }
    const int bx = blockIdx.x * 256, by = blockIdx.y * 4;
    if (bx >= ubox[2] || bx + 256 <= ubox[0] || by >= ubox[3] || by + 4 <= ubox[1]) return;
    const int lane = threadIdx.x & 63;
    const int x = bx + 4 * lane, y = by + (threadIdx.x >> 6);
    bool touched = false;
    for (int j = 0; j < n; j++) {
        const WarpParams* __restrict__ wp = wps + j;
        touched |= y >= wp->ry0 && y < wp->ry1 && bx < wp->rx1 && bx + 256 > wp->rx0;
    }
    if (!touched) return;
    const bool oncanvas = x < cw;  // then all four pixels are; y < ch: inside some ROI
    const uint32_t p = (uint32_t)y * (uint32_t)cw + (uint32_t)(oncanvas ? x : 0);   // a canvas is far below 2^32 / 3 pixels
    int cv[4][3], mo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cv[k][0] = cv[k][1] = cv[k][2] = 0, mo[k] = 255;
    if (oncanvas) {
        const uint8_t* __restrict__ src_c = REDO ? snap_canvas : canvas;
        const uint8_t* __restrict__ src_m = REDO ? snap_mask : mask_old;
        uint32_t w[3];
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = reinterpret_cast<const uint32_t*>(src_c + 3u * p)[i];
        const uint32_t mw = *reinterpret_cast<const uint32_t*>(src_m + p);
        if (!REDO) {
#pragma unroll
            for (int i = 0; i < 3; i++) reinterpret_cast<uint32_t*>(snap_canvas + 3u * p)[i] = w[i];
            *reinterpret_cast<uint32_t*>(snap_mask + p) = mw;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) cv[k][c] = (w[(3 * k + c) >> 2] >> (8 * ((3 * k + c) & 3))) & 255;
            mo[k] = (mw >> (8 * k)) & 255;
        }
    }
    bool changed = REDO;
    // two frames at a time: their pixels and weights are requested together, then blended in order
    constexpr int G = 2;
    for (int j0 = 0; j0 <= last; j0 += G) {
        uint4 px[G];
        float4 wa[G], wb[G];
        int lo[G], hi[G];   // the lane's pixels lo <= k < hi lie in the frame's region
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int j = j0 + u;
            px[u] = make_uint4(0u, 0u, 0u, 0u);
            wa[u] = wb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            lo[u] = hi[u] = 0;
            if (j > last) continue;
            const WarpParams* __restrict__ wp = wps + j;
            if (y < wp->ry0 || y >= wp->ry1 || bx >= wp->rx1 || bx + 256 <= wp->rx0) continue;  // wave-uniform
            if (!oncanvas || x + 4 <= wp->rx0 || x >= wp->rx1) continue;
            lo[u] = max(wp->rx0 - x, 0);
            hi[u] = min(wp->rx1 - x, 4);
            px[u] = load_u4(as_global<const uint32_t>(sides[j].warped) + p);
            const uint4 a = load_u4(as_global<const float>(sides[j].dist_new) + p);
            const uint4 b = load_u4(as_global<const float>(sides[j].dist_old) + p);
            wa[u] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
            wb[u] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t pxs[4] = {px[u].x, px[u].y, px[u].z, px[u].w};
            const float was[4] = {wa[u].x, wa[u].y, wa[u].z, wa[u].w}, wbs[4] = {wb[u].x, wb[u].y, wb[u].z, wb[u].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // outside the frame's region the plane holds another frame's pixels; outside its mask nothing is blended
                if (k < lo[u] || k >= hi[u] || !(pxs[k] >> 24)) continue;
                const int w0 = pxs[k] & 255, w1 = (pxs[k] >> 8) & 255, w2 = (pxs[k] >> 16) & 255;
                changed = true;
                if (!mo[k]) {
                    cv[k][0] = w0, cv[k][1] = w1, cv[k][2] = w2;
                    mo[k] = 255;
                } else {
                    const float a = was[k], b = wbs[k];
                    cv[k][0] = (int)((float)cv[k][0] * b + (float)w0 * a) & 255;  // astype(uint8) truncates
                    cv[k][1] = (int)((float)cv[k][1] * b + (float)w1 * a) & 255;
                    cv[k][2] = (int)((float)cv[k][2] * b + (float)w2 * a) & 255;
                    mo[k] = (cv[k][0] | cv[k][1] | cv[k][2]) ? 255 : 0;
                    // two non-zero pixels blended to zero: the masks predicted for the later frames of
                    // the batch do not hold
                    if (!REDO && !mo[k]) atomicMin(&st->bad_frame, j0 + u);
                }
            }
        }
    }
    if (oncanvas && changed) {
        uint32_t w[3] = {0u, 0u, 0u}, mw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) w[(3 * k + c) >> 2] |= (uint32_t)cv[k][c] << (8 * ((3 * k + c) & 3));
            mw |= (uint32_t)mo[k] << (8 * k);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) reinterpret_cast<uint32_t*>(canvas + 3u * p)[i] = w[i];
        *reinterpret_cast<uint32_t*>(mask_old + p) = mw;
    }
    // zero bits: the lane's four at 4 * (lane % 8) of the word its eight lanes share
    uint32_t z = 0;
    if (oncanvas) {
#pragma unroll
        for (int k = 0; k < 4; k++) z |= (mo[k] == 0 ? 1u : 0u) << k;
    }
    z <<= 4 * (lane & 7);
    z |= __shfl_xor(z, 1);
    z |= __shfl_xor(z, 2);
    z |= __shfl_xor(z, 4);
    if ((lane & 7) == 0 && bx + 4 * lane < cw) bold[(size_t)y * wpitch + (bx >> 5) + (lane >> 3)] = z;
}

// Column pass of the blur and commit in one kernel: the final blend weights never go to memory.  A wave is 64
// columns x CC_ROWS rows of the canvas (lane = column, the rows' pixels in registers as B, G, R and the mask byte of
// one word each) and carries them through the frames of the batch in order, as k_commit does.  For a frame that
// covers some of its pixels it fetches the frame's warped pixels; if any of them meets a non-zero canvas pixel it
// fetches the window of row-blurred weight pairs and runs the column sums of blur_cols_sums (the same operations in
// the same order as k_blur_cols: same weights), then blends.  Same results as k_blur_cols + k_commit.
// What it saves: 8 bytes per pixel and frame written by the column pass and read back by the commit; and the
// column sums where a frame lies on empty canvas.  blockIdx = tile of 64 x CC_WAVES*CC_ROWS canvas pixels.
// rows per lane of k_commit_cols: a window of 38 pairs for 8 sums (12 rows from 42 pairs as in k_blur_cols: 184
// registers, two waves per SIMD and a third fewer waves: 51 against 37 us per 4K frame; 4 rows from 34: 51 us)
constexpr int CC_ROWS = 8;
// waves per workgroup (no wave needs another: one-wave workgroups spread most evenly over the compute units)
constexpr int CC_WAVES = 1;

template <bool REDO>
__device__ __forceinline__ void commit_cols_tile(const WarpParams* __restrict__ wps, const SideDev* __restrict__ sides,
                                                 const float* __restrict__ gk, int n, int last, uint8_t* __restrict__ mask_old,
                                                 uint8_t* __restrict__ canvas, uint8_t* __restrict__ snap_mask,
                                                 uint8_t* __restrict__ snap_canvas, uint32_t* __restrict__ bold,
                                                 int wpitch, int ch, int cw, StreamState* __restrict__ st, int bx, int by) {
    const int lane = threadIdx.x & 63;
    const int x = bx + lane, ya = by + CC_ROWS * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (ya >= ch) return;
    // the frames of the batch that touch this wave's pixels, one bit each (a batch has at most 32 frames)
    uint32_t frames = 0;
    for (int j = 0; j < n; j++) {
        const WarpParams* __restrict__ wp = wps + j;
        if (ya < wp->ry1 && ya + CC_ROWS > wp->ry0 && bx < wp->rx1 && bx + 64 > wp->rx0) frames |= 1u << j;
    }
    if (!frames) return;
    const bool oncanvas = x < cw;
    const int rows = min(CC_ROWS, ch - ya);   // the same for the whole wave
    const uint32_t p0 = (uint32_t)ya * (uint32_t)cw + (uint32_t)(oncanvas ? x : 0);
    uint32_t cv[CC_ROWS];   // B | G << 8 | R << 16 | mask << 24
#pragma unroll
    for (int i = 0; i < CC_ROWS; i++) {
        cv[i] = 0xFF000000u;
        if (oncanvas && i < rows) {
            const uint32_t p = p0 + (uint32_t)i * (uint32_t)cw;
            const uint8_t* __restrict__ src_c = REDO ? snap_canvas : canvas;
            const uint8_t* __restrict__ src_m = REDO ? snap_mask : mask_old;
            const uint32_t c0 = src_c[3u * p], c1 = src_c[3u * p + 1], c2 = src_c[3u * p + 2], mo = src_m[p];
            cv[i] = c0 | (c1 << 8) | (c2 << 16) | (mo << 24);
            if (!REDO) {
                snap_canvas[3u * p] = (uint8_t)c0;
                snap_canvas[3u * p + 1] = (uint8_t)c1;
                snap_canvas[3u * p + 2] = (uint8_t)c2;
                snap_mask[p] = (uint8_t)mo;
            }
        }
    }
    uint32_t changed = REDO ? 0xFFFFFFFFu : 0u;   // bit i: row i was written to
    const bool edge = !(ya - 15 >= 0 && ya + CC_ROWS + 15 <= ch);   // wave-uniform
    // The warped pixels of a frame are requested while the frame before it is being blended: a wave walks its
    // frames one after the other, and with few waves per SIMD nothing else hides the round trips.
    auto fetch = [&](int j, uint32_t (&px)[CC_ROWS], bool& col_in) {
        const WarpParams* __restrict__ wp = wps + j;
        const int rx0 = wp->rx0, rx1 = wp->rx1, ry0 = wp->ry0, ry1 = wp->ry1;
        col_in = oncanvas && x >= rx0 && x < rx1;
#pragma unroll
        for (int i = 0; i < CC_ROWS; i++) {
            px[i] = 0;
            if (col_in && ya + i >= ry0 && ya + i < ry1)
                px[i] = as_global<const uint32_t>(sides[j].warped)[p0 + (uint32_t)i * (uint32_t)cw];
        }
    };
    if (last < 31) frames &= (2u << last) - 1u;
    uint32_t pxn[CC_ROWS];
    bool col_in_n = false;
    if (frames) fetch(__builtin_ctz(frames), pxn, col_in_n);
    while (frames) {
        const int j = __builtin_ctz(frames);
        frames &= frames - 1u;
        uint32_t px[CC_ROWS];
        const bool col_in = col_in_n;
#pragma unroll
        for (int i = 0; i < CC_ROWS; i++) px[i] = pxn[i];
        if (frames) fetch(__builtin_ctz(frames), pxn, col_in_n);
        bool blend = false;
#pragma unroll
        for (int i = 0; i < CC_ROWS; i++) blend |= (px[i] >> 24) && (cv[i] >> 24);
        f32x2 a[CC_ROWS];
#pragma unroll
        for (int i = 0; i < CC_ROWS; i++) a[i] = f32x2{0.f, 0.f};
        if (__any(blend)) {
            // the lanes outside the frame's columns hold no pixel of it: their sums are never looked at
            if (col_in) {
                if (edge)
                    blur_cols_sums<true, CC_ROWS>(gk, sides[j], ch, cw, x, ya, a);
                else
                    blur_cols_sums<false, CC_ROWS>(gk, sides[j], ch, cw, x, ya, a);
            }
        }
#pragma unroll
        for (int i = 0; i < CC_ROWS; i++) {
            if (!(px[i] >> 24)) continue;  // outside the frame's mask (or its region)
            const int w0 = px[i] & 255, w1 = (px[i] >> 8) & 255, w2 = (px[i] >> 16) & 255;
            changed |= 1u << i;
            if (!(cv[i] >> 24)) {
                cv[i] = (px[i] & 0xFFFFFFu) | 0xFF000000u;
            } else {
                const float wa = a[i].x, wb = a[i].y;
                const int c0 = cv[i] & 255, c1 = (cv[i] >> 8) & 255, c2 = (cv[i] >> 16) & 255;
                const int r0 = (int)((float)c0 * wb + (float)w0 * wa) & 255;  // astype(uint8) truncates
                const int r1 = (int)((float)c1 * wb + (float)w1 * wa) & 255;
                const int r2 = (int)((float)c2 * wb + (float)w2 * wa) & 255;
                const uint32_t mo = (r0 | r1 | r2) ? 255u : 0u;
                cv[i] = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | (mo << 24);
                // two non-zero pixels blended to zero: the masks predicted for the later frames of
                // the batch do not hold
                if (!REDO && !mo) atomicMin(&st->bad_frame, j);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CC_ROWS; i++) {
        if (i >= rows) continue;
        const uint32_t p = p0 + (uint32_t)i * (uint32_t)cw;
        if (oncanvas && ((changed >> i) & 1u)) {
            canvas[3u * p] = (uint8_t)cv[i];
            canvas[3u * p + 1] = (uint8_t)(cv[i] >> 8);
            canvas[3u * p + 2] = (uint8_t)(cv[i] >> 16);
            mask_old[p] = (uint8_t)(cv[i] >> 24);
        }
        const unsigned long long bal = __ballot(oncanvas && (cv[i] >> 24) == 0);
        if (lane == 0) {
            uint32_t* B = bold + (size_t)(ya + i) * wpitch + (bx >> 5);
            B[0] = (uint32_t)bal;
            B[1] = (uint32_t)(bal >> 32);
        }
    }
}

// blockIdx = tile of the canvas.  REDO runs after the optimistic pass, as a small grid that walks the tiles of the
// batch's box only if a frame went bad (its workgroups leave at once otherwise, which is the usual case).
template <bool REDO>
__global__ __launch_bounds__(64 * CC_WAVES) void k_commit_cols(const WarpParams* __restrict__ wps, const SideDev* __restrict__ sides,
                                                     const float* __restrict__ gk, int n, uint8_t* __restrict__ mask_old,
                                                     uint8_t* __restrict__ canvas, uint8_t* __restrict__ snap_mask,
                                                     uint8_t* __restrict__ snap_canvas, uint32_t* __restrict__ bold,
                                                     int wpitch, int ch, int cw, StreamState* __restrict__ st,
                                                     const int* __restrict__ ubox) {
    constexpr int TH = CC_WAVES * CC_ROWS;
    if (!REDO) {
        const int bx = blockIdx.x * 64, by = blockIdx.y * TH;
        if (bx >= ubox[2] || bx + 64 <= ubox[0] || by >= ubox[3] || by + TH <= ubox[1]) return;
        commit_cols_tile<false>(wps, sides, gk, n, n - 1, mask_old, canvas, snap_mask, snap_canvas, bold, wpitch, ch, cw, st, bx, by);
        return;
    }
    const int last = st->bad_frame;
    if (last >= n - 1) return;  // every frame was applied with the right weights
    const int tx0 = max(ubox[0], 0) / 64, tx1 = (min(ubox[2], cw) + 63) / 64;
    const int ty0 = max(ubox[1], 0) / TH, ty1 = (min(ubox[3], ch) + TH - 1) / TH;
    const int ntx = tx1 - tx0, T = ntx * (ty1 - ty0);
    for (int t = blockIdx.x; t < T; t += gridDim.x)
        commit_cols_tile<true>(wps, sides, gk, n, last, mask_old, canvas, snap_mask, snap_canvas, bold, wpitch, ch, cw, st,
                               (tx0 + t % ntx) * 64, (ty0 + t / ntx) * TH);
}

__global__ __launch_bounds__(256) void k_canvas_paste(const uint8_t* __restrict__ src, uint8_t* __restrict__ canvas,
                                                      uint8_t* __restrict__ mask_old, int fh, int fw, int ch, int cw,
                                                      int row_off, int col_off, int* __restrict__ cbox) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {  // bounding box of the canvas content
        cbox[0] = min(max(col_off, 0), cw), cbox[1] = min(max(row_off, 0), ch);
        cbox[2] = max(min(col_off + fw, cw), 0), cbox[3] = max(min(row_off + fh, ch), 0);
    }
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= fw || y >= fh) return;
    const int cy = y + row_off, cx = x + col_off;
    if (cy < 0 || cy >= ch || cx < 0 || cx >= cw) return;
    const size_t s = ((size_t)y * fw + x) * 3, p = (size_t)cy * cw + cx;
    const uint8_t b = src[s], g = src[s + 1], r = src[s + 2];
    canvas[3 * p] = b;
    canvas[3 * p + 1] = g;
    canvas[3 * p + 2] = r;
    mask_old[p] = (b | g | r) ? 255 : 0;
}

}  // namespace

// Grid extent for kernels that walk the tiles of a frame's region of interest: enough
// tiles for a region of `expect` pixels (plus one for the alignment of the first tile),
// never more than the canvas has.  Larger regions are covered by the kernels' loops.
static unsigned roi_tiles(int canvas, int expect, int tile) {
    const int all = (canvas + tile - 1) / tile + 1, want = (expect + tile - 1) / tile + 1;
    return (unsigned)(want < all ? want : all);
}

// cv::getGaussianKernel(31, 0, CV_32F): sigma = 0.3*((31-1)*0.5 - 1) + 0.8
int rtvm_warp_setup(rtvm_ctx* c) {
    const int n = 31;
    const double sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.5 / (sigma * sigma);
    double t[31], sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        t[i] = std::exp(scale2 * x * x);
        sum += t[i];
    }
    sum = 1. / sum;
    float k[31];
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    HIP_TRY(c, hipMemcpy(c->d_gk, k, sizeof(k), hipMemcpyHostToDevice));
    // the whole old mask as a distance-transform job: only its zero bit rows are ever built
    // (once per canvas; the commit kernel keeps them up to date)
    DtJob jb;
    memset(&jb, 0, sizeof(jb));
    rtvm_dt_job_planes(c, &jb, 0, 1, c->cw, c->ch, c->cw);
    jb.mask = c->d_mask_old;
    jb.pitch = c->cw;
    jb.x1 = jb.ox1 = c->cw;
    jb.y1 = jb.oy1 = c->ch;
    jb.B = c->d_bold;
    HIP_TRY(c, hipMemcpy(c->d_jobs + RTVM_JOB_CANVAS, &jb, sizeof(DtJob), hipMemcpyHostToDevice));
    return RTVM_OK;
}

int rtvm_launch_canvas_init(rtvm_ctx* c, const uint8_t* d_bgr, int row_off, int col_off) {
    const size_t n_px = (size_t)c->ch * c->cw;
    HIP_TRY(c, hipMemsetAsync(c->d_canvas, 0, n_px * 3, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_mask_old, 0, n_px, c->stream));
    RTVM_LAUNCH(c, k_canvas_paste, dim3((c->fw + 63) / 64, (c->fh + 3) / 4), dim3(256), 0, d_bgr,
                       c->d_canvas, c->d_mask_old, c->fh, c->fw, c->ch, c->cw, row_off, col_off, c->d_cbox);
    HIP_TRY(c, hipGetLastError());
    int r = rtvm_launch_dt_bits(c, c->d_jobs + RTVM_JOB_CANVAS, 1, c->cw, c->ch);
    if (r != RTVM_OK) return r;
    c->canvas_ready = 1;
    return RTVM_OK;
}

// Full-canvas warp into d_out (stage parity of cv2.warpPerspective alone).
int rtvm_launch_warp_only(rtvm_ctx* c, const uint8_t* d_bgr, const double* d_H, uint8_t* d_out) {
    RTVM_LAUNCH(c, k_warp_prep, dim3(1), dim3(64), 0, d_H, c->d_wp, c->fh, c->fw, c->ch, c->cw, 1);
    BatchPtrs bp;
    memset(&bp, 0, sizeof(bp));
    bp.p[0] = d_bgr;
    const dim3 grid((c->cw + 63) / 64, (c->ch + WT_ROWS - 1) / WT_ROWS, 1);
    if (c->flags & RTVM_FLAG_WARP_FIXED)
        RTVM_LAUNCH(c, k_warp<true>, grid, dim3(256), 0, bp, c->d_wp, (const SideDev*)nullptr, d_out,
                    (const DtJob*)nullptr, c->fh, c->fw, c->ch, c->cw);
    else
        RTVM_LAUNCH(c, k_warp<false>, grid, dim3(256), 0, bp, c->d_wp, (const SideDev*)nullptr, d_out,
                    (const DtJob*)nullptr, c->fh, c->fw, c->ch, c->cw);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

int rtvm_launch_warp_prep(rtvm_ctx* c, const double* d_H, const int* d_skip, int n) {
    if (n < 1 || n > c->n_side) RTVM_FAIL(c, RTVM_E_STATE, "warp: %d frames, planes for %d", n, c->n_side);
    RTVM_LAUNCH(c, k_warp_prep_batch, dim3(1), dim3(64), 0, (const double*)d_H, c->d_wp, c->d_jobs, c->d_jobs + RTVM_MAX_BATCH,
                n, c->fh, c->fw, c->ch, c->cw, d_skip, c->d_state, c->d_cbox);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// For the n frames of the batch rtvm_launch_warp_prep prepared: warps, masks and zero
// bits; the canvas masks the frames are predicted to meet; the distance transforms of all
// 2n masks.
int rtvm_launch_sides(rtvm_ctx* c, const uint8_t* const* d_bgr, int n) {
    BatchPtrs bp;
    memset(&bp, 0, sizeof(bp));
    for (int j = 0; j < n; j++) bp.p[j] = d_bgr[j];
    const dim3 grid(roi_tiles(c->cw, c->fw + c->fh / 2, 64), roi_tiles(c->ch, c->fh + c->fw / 2, WT_ROWS), n);
    if (c->flags & RTVM_FLAG_WARP_FIXED)
        RTVM_LAUNCH(c, k_warp<true>, grid, dim3(256), 0, bp, c->d_wp, c->d_side, (uint8_t*)nullptr, c->d_jobs, c->fh,
                    c->fw, c->ch, c->cw);
    else
        RTVM_LAUNCH(c, k_warp<false>, grid, dim3(256), 0, bp, c->d_wp, c->d_side, (uint8_t*)nullptr, c->d_jobs, c->fh,
                    c->fw, c->ch, c->cw);
    RTVM_LAUNCH(c, k_pred_bits, dim3(((c->cw + 31) / 32 + 63) / 64, (c->ch + 3) / 4), dim3(256), 0, c->d_bold, c->d_wp,
                c->d_jobs, c->d_jobs + RTVM_MAX_BATCH, n, c->ch, c->cw);
    return rtvm_launch_dt2(c, c->d_jobs, n, RTVM_MAX_BATCH, n, c->cw, c->ch, 0);
}

// Blend weights of the n frames: dn / (dn + do + eps) and its 31x31 Gaussian blur.
int rtvm_launch_weights(rtvm_ctx* c, int n) {
    const int bw = c->fw + c->fh / 2, bh = c->fh + c->fw / 2;  // a region of about the frame's size
    {
        const unsigned ex = roi_tiles(c->cw, bw, BR_TILE), ey = roi_tiles(c->ch, bh + 30, 4);
        RTVM_LAUNCH(c, k_blur_rows, dim3(xcd_grid(ex * ey), 1, n), dim3(256), 0, c->d_wp, c->d_gk, c->d_side, c->cw);
    }
    if (!c->fused_commit) {
        const unsigned ex = roi_tiles(c->cw, bw, 64), ey = roi_tiles(c->ch, bh, 4 * BC_ROWS);
        RTVM_LAUNCH(c, k_blur_cols, dim3(xcd_grid(ex * ey), 1, n), dim3(256), 0, c->d_wp, c->d_gk, c->d_side, c->ch, c->cw);
    }
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// Blend and commit of the n frames of the batch, in order.
int rtvm_launch_commit(rtvm_ctx* c, int n) {
    const int wpitch = (c->cw + 31) / 32 + 2;
    if (c->fused_commit) {   // the column pass of the blur runs inside the commit
        const dim3 grid((c->cw + 63) / 64, (c->ch + CC_WAVES * CC_ROWS - 1) / (CC_WAVES * CC_ROWS));
        RTVM_LAUNCH(c, k_commit_cols<false>, grid, dim3(64 * CC_WAVES), 0, c->d_wp, c->d_side, c->d_gk, n, c->d_mask_old, c->d_canvas,
                    c->d_snap_mask, c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state, (const int*)(c->d_cbox + 4));
        if (n > 1)
            RTVM_LAUNCH(c, k_commit_cols<true>, dim3(1024), dim3(64 * CC_WAVES), 0, c->d_wp, c->d_side, c->d_gk, n, c->d_mask_old, c->d_canvas,
                        c->d_snap_mask, c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state,
                        (const int*)(c->d_cbox + 4));
        HIP_TRY(c, hipGetLastError());
        return RTVM_OK;
    }
    if (c->cw % 4 == 0) {
        const dim3 grid((c->cw + 255) / 256, (c->ch + 3) / 4);
        RTVM_LAUNCH(c, k_commit4<false>, grid, dim3(256), 0, c->d_wp, c->d_side, n, c->d_mask_old, c->d_canvas, c->d_snap_mask,
                    c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state, (const int*)(c->d_cbox + 4));
        if (n > 1)
            RTVM_LAUNCH(c, k_commit4<true>, grid, dim3(256), 0, c->d_wp, c->d_side, n, c->d_mask_old, c->d_canvas,
                        c->d_snap_mask, c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state,
                        (const int*)(c->d_cbox + 4));
        HIP_TRY(c, hipGetLastError());
        return RTVM_OK;
    }
    const dim3 grid((c->cw + 63) / 64, (c->ch + 3) / 4);
    RTVM_LAUNCH(c, k_commit<false>, grid, dim3(256), 0, c->d_wp, c->d_side, n, c->d_mask_old, c->d_canvas, c->d_snap_mask,
                c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state, (const int*)(c->d_cbox + 4));
    if (n > 1)
        RTVM_LAUNCH(c, k_commit<true>, grid, dim3(256), 0, c->d_wp, c->d_side, n, c->d_mask_old, c->d_canvas,
                    c->d_snap_mask, c->d_snap_canvas, c->d_bold, wpitch, c->ch, c->cw, c->d_state,
                    (const int*)(c->d_cbox + 4));
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// The whole of VideMosaic.warp() for one frame, H read from device memory.
int rtvm_launch_warp_blend(rtvm_ctx* c, const uint8_t* d_bgr, const double* d_H) {
    int r;
    if ((r = rtvm_launch_warp_prep(c, d_H, nullptr, 1))) return r;
    if ((r = rtvm_launch_sides(c, &d_bgr, 1))) return r;
    if ((r = rtvm_launch_weights(c, 1))) return r;
    return rtvm_launch_commit(c, 1);
}
