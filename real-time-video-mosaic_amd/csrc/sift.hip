// cv2.SIFT_create(n).detectAndCompute(gray, None) for gfx950 (reference main.py:33, 111-112,
// 717-718; OpenCV defaults nOctaveLayers = 3, contrastThreshold = 0.04, edgeThreshold = 10,
// sigma = 1.6).  Stage entry point rtvm_sift: host frame in, keypoints and 128-float
// descriptors out.
//
//   k_sift_base      gray (15-bit fixed point) as float, doubled with the arithmetic of
//                    cv::resize(INTER_LINEAR) on floats
//   k_sift_blur_rows / k_sift_blur_cols
//                    cv::GaussianBlur(Size(), sigma) on floats: row pass s = k0*x0,
//                    fma(x_i, k_i, s); column pass centre then fma(r[+i] + r[-i], k_i, s); the
//                    column pass also writes the difference of Gaussians to the layer below
//   k_sift_decimate  next octave = every second pixel of layer 3
//   k_sift_extrema   26-neighbour extrema above the contrast floor -> candidate keys (every
//                    octave and layer in one launch)
//   k_sift_refine    one wave per candidate (keys sorted on the host: octave, layer, row,
//                    column): adjustLocalExtrema, orientation histogram, peaks
//   k_sift_describe  one wave per keypoint: calcSIFTDescriptor (samples in parallel, sums in order)
// Sums that OpenCV accumulates sequentially (orientation and descriptor histograms) are
// accumulated sequentially by one thread here as well; exp, atan2, cos, sin and pow are
// evaluated in double precision and rounded to float.  Duplicate removal and
// retainBest(nfeatures) run on the host between the last two kernels.
#include "common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace {

constexpr int SL = 3;           // nOctaveLayers
constexpr int SBORDER = 5;      // SIFT_IMG_BORDER
constexpr int ORI_BINS = 36;
constexpr int MAX_PEAKS = 8;
constexpr int SD = 4, SNB = 8;
constexpr float SSIGMA = 1.6f;
constexpr float SCONTRAST = 0.04f, SEDGE = 10.f;
constexpr int MAX_OCT = 16;
constexpr int MAX_TAPS = 33;
constexpr int SMALL_OCTAVE_PX = 4000;  // octaves of at most this many pixels share one launch (2 x 4 bytes each in LDS);
                                       // measured: one workgroup is slower than eleven launches from 160 x 90 up

struct SImg {
    float* px;
    int h, w;
};

struct SKp {
    float x, y, size, angle, response;
    int octave;
};

struct SPyr {
    SImg gp[MAX_OCT * (SL + 3)];
    SImg dog[MAX_OCT * (SL + 2)];
};

__device__ __forceinline__ int reflect101s(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_sift_base(const uint8_t* __restrict__ bgr, int h, int w, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= 2 * w || y >= 2 * h) return;
    float fx = (float)(((double)x + 0.5) * 0.5 - 0.5), fy = (float)(((double)y + 0.5) * 0.5 - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= sx;
    fy -= sy;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= w - 1) fx = 0, sx = w - 1;
    if (sy < 0) fy = 0, sy = 0;
    if (sy >= h - 1) fy = 0, sy = h - 1;
    const int x1 = min(sx + 1, w - 1), y1 = min(sy + 1, h - 1);
    auto gray = [&](int yy, int xx) {
        const uint8_t* p = bgr + ((size_t)yy * w + xx) * 3;
        return (float)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15);
    };
    const float a1 = fx, a0 = 1.f - a1, b1 = fy, b0 = 1.f - b1;
    const float r0 = gray(sy, sx) * a0 + gray(sy, x1) * a1;
    const float r1 = gray(y1, sx) * a0 + gray(y1, x1) * a1;
    out[(size_t)y * 2 * w + x] = r0 * b0 + r1 * b1;
}

constexpr int MAX_HALF = MAX_TAPS / 2;

// k[0..n-1]: the kernel in row order; c[q] = k[n/2 + q]: the same taps counted from the centre
struct Taps {
    float k[MAX_TAPS];
    float c[MAX_HALF + 1];
    int n;
};

// The tap loops are unrolled to the maximum length with an early exit, so that the taps sit in
// scalar registers loaded once per kernel instead of one scalar load (and wait) per tap.
__device__ __forceinline__ float row_taps(const Taps& t, const float* p) {  // p: leftmost sample of the window
    float a = t.k[0] * p[0];
#pragma unroll
    for (int q = 1; q < MAX_TAPS; q++) {
        if (q >= t.n) break;
        a = __builtin_fmaf(p[q], t.k[q], a);
    }
    return a;
}

__device__ __forceinline__ float col_taps(const Taps& t, const float* p, int pitch) {  // p: the centre sample
    const int half = t.n / 2;
    float a = t.c[0] * p[0];
#pragma unroll
    for (int q = 1; q <= MAX_HALF; q++) {
        if (q > half) break;
        a = __builtin_fmaf(p[q * pitch] + p[-q * pitch], t.c[q], a);
    }
    return a;
}

// The same sums where the window crosses the border (BORDER_REFLECT_101); rare, plain loops.
__device__ __forceinline__ float row_taps_border(const Taps& t, const float* row, int x, int w) {
    const int half = t.n / 2;
    float a = t.k[0] * row[reflect101s(x - half, w)];
    for (int q = 1; q < t.n; q++) a = __builtin_fmaf(row[reflect101s(x - half + q, w)], t.k[q], a);
    return a;
}

__device__ __forceinline__ float col_taps_border(const Taps& t, const float* col, int y, int h, int pitch) {
    const int half = t.n / 2;
    float a = t.c[0] * col[(size_t)y * pitch];
    for (int q = 1; q <= half; q++)
        a = __builtin_fmaf(col[(size_t)reflect101s(y + q, h) * pitch] + col[(size_t)reflect101s(y - q, h) * pitch], t.c[q], a);
    return a;
}

__global__ __launch_bounds__(256) void k_sift_blur_rows(const float* __restrict__ src, float* __restrict__ dst, int h, int w,
                                                        Taps t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int half = t.n / 2;
    const float* __restrict__ row = src + (size_t)y * w;
    float a;
    if (x >= half && x + half < w) {  // no border in reach: the same sum without the index folding
        a = row_taps(t, row + (x - half));
    } else {
        a = row_taps_border(t, row, x, w);
    }
    dst[(size_t)y * w + x] = a;
}

// Column pass; with `below` (the Gaussian layer under the one being made) it also writes the
// difference of the two layers, dog = this - below, the value a separate subtraction would give.
__global__ __launch_bounds__(256) void k_sift_blur_cols(const float* __restrict__ src, float* __restrict__ dst, int h, int w,
                                                        Taps t, const float* __restrict__ below, float* __restrict__ dog) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int half = t.n / 2;
    const float* __restrict__ c0 = src + (size_t)y * w + x;
    float a;
    if (y >= half && y + half < h) {  // wave-uniform: a wave is one row
        a = col_taps(t, c0, w);
    } else {
        a = col_taps_border(t, src + x, y, h, w);
    }
    dst[(size_t)y * w + x] = a;
    if (below) dog[(size_t)y * w + x] = a - below[(size_t)y * w + x];
}

__global__ __launch_bounds__(256) void k_sift_decimate(const float* __restrict__ src, int sw, float* __restrict__ dst, int h,
                                                       int w) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    dst[(size_t)y * w + x] = src[(size_t)(2 * y) * sw + 2 * x];
}

// The small octaves (a few thousand pixels each) are launch-latency bound as eleven kernels per
// octave: here one workgroup builds all of them, octave after octave and layer after layer, with
// the arithmetic of k_sift_decimate / k_sift_blur_rows / k_sift_blur_cols.  The layer being
// blurred and its row pass live in LDS (2 x h x w floats of the first of these octaves, dynamic);
// the finished layers and differences go to the pyramid in global memory.
struct TapsAll {
    Taps t[SL + 3];
};

// Layer i of octave o from the layer below it in `cur` (LDS): row pass into rowp, column pass
// into the pyramid, the difference of the two layers, and `cur` replaced by the new layer.
__device__ __forceinline__ void small_layer(const SPyr* P, int o, int i, const Taps& t, float* cur, float* rowp, int h, int w) {
    const int tid = threadIdx.x, npx = h * w, half = t.n / 2;
    float* dst = P->gp[o * (SL + 3) + i].px;
    float* dog = P->dog[o * (SL + 2) + i - 1].px;
    for (int p = tid; p < npx; p += 1024) {
        const int y = p / w, x = p - y * w;
        const float* row = cur + y * w;
        rowp[p] = (x >= half && x + half < w) ? row_taps(t, row + (x - half)) : row_taps_border(t, row, x, w);
    }
    __syncthreads();
    for (int p = tid; p < npx; p += 1024) {
        const int y = p / w, x = p - y * w;
        const float a = (y >= half && y + half < h) ? col_taps(t, rowp + p, w) : col_taps_border(t, rowp + x, y, h, w);
        dst[p] = a;
        dog[p] = a - cur[p];
        cur[p] = a;  // only this thread reads cur[p] in this pass
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sift_small_octaves(const SPyr* P, int o0, int o1, TapsAll T, int lds_px) {
    extern __shared__ float s_small[];
    float* cur = s_small;            // the layer below the one being made
    float* rowp = s_small + lds_px;  // its row pass
    const int tid = threadIdx.x;
    for (int o = o0; o < o1; o++) {
        const int h = P->gp[o * (SL + 3)].h, w = P->gp[o * (SL + 3)].w, npx = h * w;
        {
            // global memory written by this workgroup is visible to it after the barrier below
            const SImg src = P->gp[(o - 1) * (SL + 3) + SL];
            float* dst = P->gp[o * (SL + 3)].px;
            for (int p = tid; p < npx; p += 1024) {
                const int y = p / w, x = p - y * w;
                const float v = src.px[(size_t)(2 * y) * src.w + 2 * x];
                dst[p] = v;
                cur[p] = v;
            }
        }
        __syncthreads();
        small_layer(P, o, 1, T.t[1], cur, rowp, h, w);
        small_layer(P, o, 2, T.t[2], cur, rowp, h, w);
        small_layer(P, o, 3, T.t[3], cur, rowp, h, w);
        small_layer(P, o, 4, T.t[4], cur, rowp, h, w);
        small_layer(P, o, 5, T.t[5], cur, rowp, h, w);
        static_assert(SL + 3 == 6, "one call per layer above");
    }
}

// Tiles of 64x4 positions of every octave in one launch: tile_start[o] is the first linear tile
// of octave o, blockIdx.y + 1 the layer.
struct ExtremaTiles {
    int n_oct;
    int tile_start[MAX_OCT + 1];
    int tiles_x[MAX_OCT];
};

__global__ __launch_bounds__(256) void k_sift_extrema(const SPyr* __restrict__ P, ExtremaTiles T, float threshold,
                                                      unsigned long long* __restrict__ keys, int* __restrict__ count, int cap) {
    int octave = 0;
    while (octave + 1 < T.n_oct && (int)blockIdx.x >= T.tile_start[octave + 1]) octave++;
    const int tile = blockIdx.x - T.tile_start[octave], layer = blockIdx.y + 1;
    const int ty = tile / T.tiles_x[octave], tx = tile - ty * T.tiles_x[octave];
    const SImg img = P->dog[octave * (SL + 2) + layer];
    const float* __restrict__ prev = P->dog[octave * (SL + 2) + layer - 1].px;
    const float* __restrict__ next = P->dog[octave * (SL + 2) + layer + 1].px;
    const int c = SBORDER + tx * 64 + (threadIdx.x & 63), r = SBORDER + ty * 4 + (threadIdx.x >> 6);
    if (c >= img.w - SBORDER || r >= img.h - SBORDER) return;
    const float val = img.px[(size_t)r * img.w + c];
    if (!(fabsf(val) > threshold)) return;
    bool mx = val > 0, mn = val < 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const size_t p = (size_t)(r + dy) * img.w + c + dx;
            const float v0 = img.px[p], v1 = prev[p], v2 = next[p];
            mx &= val >= v0 && val >= v1 && val >= v2;
            mn &= val <= v0 && val <= v1 && val <= v2;
        }
    if (!(mx || mn)) return;
    const int pos = atomicAdd(count, 1);
    if (pos < cap)
        keys[pos] = ((unsigned long long)octave << 56) | ((unsigned long long)layer << 48) | ((unsigned long long)r << 24) |
                    (unsigned long long)c;
}

__device__ bool solve3(const float a[3][3], const float b[3], float x[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
              a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (d == 0) return false;
    d = 1.f / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
    return true;
}

#define AT(im, y, x) ((im).px[(size_t)(y) * (im).w + (x)])

__device__ bool adjust_extremum(const SPyr* __restrict__ P, int octv, int& layer, int& r, int& c, SKp& kp) {
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < 5; i++) {
        const SImg img = P->dog[octv * (SL + 2) + layer];
        const SImg prev = P->dog[octv * (SL + 2) + layer - 1];
        const SImg next = P->dog[octv * (SL + 2) + layer + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float v2 = AT(img, r, c) * 2;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second;
        const float dss = (AT(next, r, c) + AT(prev, r, c) - v2) * second;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross;
        const float dxs = (AT(next, r, c + 1) - AT(next, r, c - 1) - AT(prev, r, c + 1) + AT(prev, r, c - 1)) * cross;
        const float dys = (AT(next, r + 1, c) - AT(next, r - 1, c) - AT(prev, r + 1, c) + AT(prev, r - 1, c)) * cross;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        if (!solve3(H, dD, X)) X[0] = X[1] = X[2] = 0;
        xi = -X[2], xr = -X[1], xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT32_MAX / 3) || fabsf(xr) > (float)(INT32_MAX / 3) || fabsf(xc) > (float)(INT32_MAX / 3))
            return false;
        c += __float2int_rn(xc);
        r += __float2int_rn(xr);
        layer += __float2int_rn(xi);
        if (layer < 1 || layer > SL || c < SBORDER || c >= img.w - SBORDER || r < SBORDER || r >= img.h - SBORDER) return false;
    }
    if (i >= 5) return false;
    const SImg img = P->dog[octv * (SL + 2) + layer];
    const SImg prev = P->dog[octv * (SL + 2) + layer - 1];
    const SImg next = P->dog[octv * (SL + 2) + layer + 1];
    const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                         (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                         (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
    const float t = (dD[0] * xc + dD[1] * xr) + dD[2] * xi;
    const float contr = AT(img, r, c) * img_scale + t * 0.5f;
    if (fabsf(contr) * SL < SCONTRAST) return false;
    const float v2 = AT(img, r, c) * 2.f;
    const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second;
    const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second;
    const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * SEDGE >= (SEDGE + 1) * (SEDGE + 1) * det) return false;
    kp.x = (c + xc) * (float)(1 << octv);
    kp.y = (r + xr) * (float)(1 << octv);
    kp.octave = octv + (layer << 8) + (__float2int_rn((xi + 0.5f) * 255) << 16);
    kp.size = SSIGMA * (float)pow(2.0, (double)((layer + xi) / SL)) * (float)(1 << octv) * 2;
    kp.response = fabsf(contr);
    return true;
}

__device__ __forceinline__ float angle_deg(float y, float x) {
    float a = (float)(atan2((double)y, (double)x) * (180.0 / 3.14159265358979323846));
    if (a < 0) a += 360.f;
    if (a >= 360.f) a -= 360.f;
    return a;
}

// calcOrientationHist by one wave: 64 window positions at a time (gradient, exp, atan2, bin),
// lane b adds the weighted magnitudes that fall into bin b in scan order.  Returns the largest
// smoothed bin; hist (LDS) holds the smoothed histogram.
__device__ float orientation_hist(const SImg img, int px, int py, int radius, float sigma, float* temphist, float* hist,
                                  float* s_val, int* s_bin) {
    const int n = ORI_BINS, lane = threadIdx.x;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float acc = 0.f;  // bin `lane` of the raw histogram
    const int side = 2 * radius + 1;
    const unsigned total = (unsigned)side * (unsigned)side;  // side <= 2 * 23170 + 1: fits 32 bits unsigned
    __syncthreads();
    for (unsigned base = 0; base < total; base += 64) {
        const unsigned t = base + lane;
        int bin = -1;
        if (t < total) {
            const int i = (int)(t / (unsigned)side) - radius, j = (int)(t % (unsigned)side) - radius;
            const int y = py + i, x = px + j;
            if (!(y <= 0 || y >= img.h - 1 || x <= 0 || x >= img.w - 1)) {
                const float dx = AT(img, y, x + 1) - AT(img, y, x - 1);
                const float dy = AT(img, y - 1, x) - AT(img, y + 1, x);
                const float W = (float)exp((double)((float)(i * i + j * j) * expf_scale));
                const float ori = angle_deg(dy, dx);
                const float mag = sqrtf(dx * dx + dy * dy);
                bin = __float2int_rn((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                s_val[lane] = W * mag;
            }
        }
        s_bin[lane] = bin;
        __syncthreads();
        // lane b owns bin b in a register and takes the samples in scan order: every bin receives its
        // terms in the order of the sequential loop
#pragma unroll
        for (int l0 = 0; l0 < 64; l0 += 8) {
            const int4 ba = *reinterpret_cast<const int4*>(&s_bin[l0]), bb = *reinterpret_cast<const int4*>(&s_bin[l0 + 4]);
            const float4 va = *reinterpret_cast<const float4*>(&s_val[l0]), vb = *reinterpret_cast<const float4*>(&s_val[l0 + 4]);
            const int b[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
            const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int u = 0; u < 8; u++) acc = b[u] == lane ? acc + v[u] : acc;
        }
        __syncthreads();
    }
    if (lane < n) temphist[lane] = acc;
    __syncthreads();
    if (lane < n) {
        const int i = lane;
        const float m2 = temphist[(i + n - 2) % n], m1 = temphist[(i + n - 1) % n];
        const float p1 = temphist[(i + 1) % n], p2 = temphist[(i + 2) % n];
        hist[i] = (m2 + p2) * (1.f / 16.f) + (m1 + p1) * (4.f / 16.f) + temphist[i] * (6.f / 16.f);
    }
    __syncthreads();
    float maxval = hist[0];
    for (int i = 1; i < n; i++) maxval = fmaxf(maxval, hist[i]);
    return maxval;
}

// One wave per candidate; out: per candidate MAX_PEAKS keypoints, npeaks[candidate] of them valid.
// One wave per candidate, a fixed grid striding over the candidates the extrema kernel counted
// (the host does not know the count yet).  Keypoints go to `list` in whatever order the waves
// finish: the host sorts them (KeyPointsFilter::removeDuplicatedSorted), so the order of the
// candidates and of the list does not reach the result.
__global__ __launch_bounds__(64) void k_sift_refine(const SPyr* __restrict__ P, const unsigned long long* __restrict__ keys,
                                                    const int* __restrict__ count, int cap, SKp* __restrict__ list,
                                                    int* __restrict__ list_count, int list_cap) {
    __shared__ float temphist[ORI_BINS], hist[ORI_BINS];
    __shared__ __align__(16) float s_val[64];
    __shared__ __align__(16) int s_bin[64];
    const int lane = threadIdx.x;
    const int n = min(*count, cap);
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
        const unsigned long long key = keys[q];
        const int o = (int)(key >> 56), c0 = (int)(key & 0xFFFFFF), r0 = (int)((key >> 24) & 0xFFFFFF);
        int layer = (int)((key >> 48) & 0xFF), r = r0, c = c0;
        SKp kp;
        if (!adjust_extremum(P, o, layer, r, c, kp)) continue;  // the same in every lane
        const float scl_octv = kp.size * 0.5f / (float)(1 << o);
        const float omax = orientation_hist(P->gp[o * (SL + 3) + layer], c, r, __float2int_rn(4.5f * scl_octv),
                                            1.5f * scl_octv, temphist, hist, s_val, s_bin);
        if (lane == 0) {
            const float mag_thr = omax * 0.8f;
            int peaks = 0;
            for (int j = 0; j < ORI_BINS && peaks < MAX_PEAKS; j++) {
                const int l = j > 0 ? j - 1 : ORI_BINS - 1, r2 = j < ORI_BINS - 1 ? j + 1 : 0;
                if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                    float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                    bin = bin < 0 ? ORI_BINS + bin : bin >= ORI_BINS ? bin - ORI_BINS : bin;
                    kp.angle = 360.f - (360.f / ORI_BINS) * bin;
                    if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
                    const int pos = atomicAdd(list_count, 1);
                    if (pos < list_cap) list[pos] = kp;
                    peaks++;
                }
            }
        }
        __syncthreads();  // the histograms are reused by the next candidate
    }
}

// One wave per keypoint.  The lanes evaluate 64 consecutive sample positions of the window at a
// time (gradient, exp, atan2, the eight trilinear shares and the bin index); lane 0 then adds
// the shares to the histogram in LDS in scan order, which is the order of OpenCV's sequential
// loop, so the float sums come out the same.
__global__ __launch_bounds__(64) void k_sift_describe(const SPyr* __restrict__ P, const SKp* __restrict__ kps, int n,
                                                      float* __restrict__ desc) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const SKp kp = kps[q];
    const int oct = kp.octave & 255, layer = (kp.octave >> 8) & 255;
    const float scale_oct = 1.f / (float)(1 << oct);
    float ori = 360.f - kp.angle;
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const SImg img = P->gp[oct * (SL + 3) + layer];
    const float ptx = kp.x * scale_oct, pty = kp.y * scale_oct, scl = kp.size * scale_oct * 0.5f;
    const int d = SD, nb = SNB;
    const int px = __float2int_rn(ptx), py = __float2int_rn(pty);
    float cos_t = (float)cos((double)(ori * (float)(3.14159265358979323846 / 180.)));
    float sin_t = (float)sin((double)(ori * (float)(3.14159265358979323846 / 180.)));
    const float bins_per_rad = nb / 360.f;
    const float exp_scale = -1.f / (d * d * 0.5f);
    const float hist_width = 3.f * scl;
    int radius = __float2int_rn(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = min(radius, (int)sqrt((double)img.w * img.w + (double)img.h * img.h));
    cos_t /= hist_width;
    sin_t /= hist_width;
    constexpr int HLEN = (SD + 2) * (SD + 2) * (SNB + 2);
    __shared__ float hist[HLEN];
    __shared__ float s_v[64][8];
    __shared__ __align__(16) int s_idx[64];
    for (int i = lane; i < HLEN; i += 64) hist[i] = 0.f;
    const int side = 2 * radius + 1;
    const unsigned total = (unsigned)side * (unsigned)side;  // side <= 2 * 23170 + 1: fits 32 bits unsigned
    for (unsigned base = 0; base < total; base += 64) {
        const unsigned t = base + lane;
        int idx = -1;
        if (t < total) {
            const int i = (int)(t / (unsigned)side) - radius, j = (int)(t % (unsigned)side) - radius;
            const float c_rot = j * cos_t - i * sin_t;
            const float r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f;
            float cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img.h - 1 && c > 0 && c < img.w - 1) {
                const float dx = AT(img, r, c + 1) - AT(img, r, c - 1);
                const float dy = AT(img, r - 1, c) - AT(img, r + 1, c);
                const float W = (float)exp((double)((c_rot * c_rot + r_rot * r_rot) * exp_scale));
                const float Ori = angle_deg(dy, dx);
                const float Mag = sqrtf(dx * dx + dy * dy);
                float obin = (Ori - ori) * bins_per_rad;
                const float mag = Mag * W;
                const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
                int o0 = (int)floorf(obin);
                rbin -= r0;
                cbin -= c0;
                obin -= o0;
                if (o0 < 0) o0 += nb;
                if (o0 >= nb) o0 -= nb;
                const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
                const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
                const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
                const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
                const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
                idx = ((r0 + 1) * (d + 2) + c0 + 1) * (nb + 2) + o0;
                s_v[lane][0] = v_rco000, s_v[lane][1] = v_rco001, s_v[lane][2] = v_rco010, s_v[lane][3] = v_rco011;
                s_v[lane][4] = v_rco100, s_v[lane][5] = v_rco101, s_v[lane][6] = v_rco110, s_v[lane][7] = v_rco111;
            }
        }
        s_idx[lane] = idx;
        __syncthreads();
        // the eight shares of a sample go to eight different bins: lane e < 8 adds share e of every
        // sample, in sample order, so every bin still receives its terms in scan order
        if (lane < 8) {
            const int ofs = (lane & 1) + ((lane >> 1) & 1) * (nb + 2) + (lane >> 2) * (d + 2) * (nb + 2);
            for (int l0 = 0; l0 < 64; l0 += 8) {
                // indices and shares of eight samples first, then the additions one after the other
                const int4 ka = *reinterpret_cast<const int4*>(&s_idx[l0]);
                const int4 kb = *reinterpret_cast<const int4*>(&s_idx[l0 + 4]);
                const int k[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = s_v[l0 + u][lane];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (k[u] >= 0) hist[k[u] + ofs] += v[u];
            }
        }
        __syncthreads();
    }
    if (lane != 0) return;
    float* __restrict__ dst = desc + (size_t)q * 128;
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (nb + 2);
            hist[idx] += hist[idx + nb];
            hist[idx + 1] += hist[idx + nb + 1];
            for (int k = 0; k < nb; k++) dst[(i * d + j) * nb + k] = hist[idx + k];
        }
    const int len = d * d * nb;
    float nrm2 = 0;
    for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0;
    for (int k = 0; k < len; k++) {
        const float val = fminf(dst[k], thr);
        dst[k] = val;
        nrm2 += val * val;
    }
    nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    for (int k = 0; k < len; k++) {
        const int v = __float2int_rn(dst[k] * nrm2);
        dst[k] = (float)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

// ---------------------------------------------------------------------------- host
int cv_round_d(double v) { return (int)std::lrint(v); }

Taps gauss_taps(double sigma) {
    Taps t;
    t.n = cv_round_d(sigma * 8 + 1) | 1;
    std::vector<double> e(t.n);
    double sum = 0;
    const double scale2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < t.n; i++) {
        const double x = i - (t.n - 1) * 0.5;
        e[i] = std::exp(scale2 * x * x);
        sum += e[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < MAX_TAPS; i++) t.k[i] = i < t.n ? (float)(e[i] * sum) : 0.f;
    for (int q = 0; q <= MAX_HALF; q++) t.c[q] = t.n / 2 + q < t.n ? t.k[t.n / 2 + q] : 0.f;
    return t;
}

bool kp_less(const SKp& a, const SKp& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    return a.octave > b.octave;
}

// Buffer `slot` of the context with room for `bytes`: kept between calls, replaced by a larger
// one when needed (the stream is idle then: every call ends with a sync).
int sift_buffer(rtvm_ctx* c, int slot, size_t bytes, void** out) {
    if (c->sift_bytes[slot] < bytes) {
        if (c->sift_buf[slot]) (void)hipFree(c->sift_buf[slot]);
        c->sift_buf[slot] = nullptr;
        c->sift_bytes[slot] = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&c->sift_buf[slot], want);
        if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        c->sift_bytes[slot] = want;
    }
    *out = c->sift_buf[slot];
    return RTVM_OK;
}
struct Dev {
    void* p = nullptr;
};
#define SIFT_BUF(c, slot, dev, bytes)                                   \
    do {                                                                \
        int r_ = sift_buffer((c), (slot), (size_t)(bytes), &(dev).p);   \
        if (r_ != RTVM_OK) return r_;                                   \
    } while (0)

dim3 grid2(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

}  // namespace

extern "C" int rtvm_sift_max_keypoints(int nfeatures) { return nfeatures + 256; }

extern "C" int rtvm_sift(rtvm_ctx* c, const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy, float* size,
                         float* angle_deg_out, float* response, int32_t* octave, float* desc) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !n_out || !xy || !size || !angle_deg_out || !response || !octave || !desc)
        RTVM_FAIL(c, RTVM_E_ARG, "sift: null argument");
    if (h < 16 || w < 16 || h > 8192 || w > 8192 || nfeatures < 1) RTVM_FAIL(c, RTVM_E_ARG, "sift: frame %dx%d, %d features", h, w, nfeatures);
    HIP_TRY(c, hipSetDevice(c->device));
    const int bh = 2 * h, bw = 2 * w;
    const int n_oct = cv_round_d(std::log((double)std::min(bw, bh)) / std::log(2.) - 2) + 1;
    if (n_oct < 1 || n_oct > MAX_OCT) RTVM_FAIL(c, RTVM_E_ARG, "sift: %d octaves", n_oct);
    double sig[SL + 3];
    sig[0] = SSIGMA;
    const double kf = std::pow(2., 1. / SL);
    for (int i = 1; i < SL + 3; i++) {
        const double sp = std::pow(kf, (double)(i - 1)) * SSIGMA, st = sp * kf;
        sig[i] = std::sqrt(st * st - sp * sp);
    }
    // one arena: frame, row-pass scratch, Gaussian and DoG pyramids
    SPyr P;
    memset(&P, 0, sizeof(P));
    size_t total = 0;
    std::vector<size_t> off_gp((size_t)n_oct * (SL + 3)), off_dog((size_t)n_oct * (SL + 2));
    {
        int oh = bh, ow = bw;
        for (int o = 0; o < n_oct; o++) {
            for (int i = 0; i < SL + 3; i++) {
                off_gp[o * (SL + 3) + i] = total;
                P.gp[o * (SL + 3) + i].h = oh, P.gp[o * (SL + 3) + i].w = ow;
                total += (size_t)oh * ow;
            }
            for (int i = 0; i < SL + 2; i++) {
                off_dog[o * (SL + 2) + i] = total;
                P.dog[o * (SL + 2) + i].h = oh, P.dog[o * (SL + 2) + i].w = ow;
                total += (size_t)oh * ow;
            }
            oh /= 2, ow /= 2;
            if (oh < 1 || ow < 1) RTVM_FAIL(c, RTVM_E_ARG, "sift: frame too small for %d octaves", n_oct);
        }
    }
    const size_t scratch_off = total;
    total += (size_t)bh * bw;
    const int cand_cap = 1 << 18;
    Dev arena, frame, keys, count, pyr, cand_out, kp_dev, desc_dev;
    SIFT_BUF(c, 0, arena, total * sizeof(float));
    SIFT_BUF(c, 1, frame, (size_t)h * w * 3);
    SIFT_BUF(c, 2, keys, (size_t)cand_cap * 8);
    SIFT_BUF(c, 3, count, 2 * sizeof(int));  // extrema found, keypoints listed
    SIFT_BUF(c, 4, pyr, sizeof(SPyr));
    float* A = (float*)arena.p;
    for (size_t i = 0; i < off_gp.size(); i++) P.gp[i].px = A + off_gp[i];
    for (size_t i = 0; i < off_dog.size(); i++) P.dog[i].px = A + off_dog[i];
    float* scratch = A + scratch_off;
    HIP_TRY(c, hipMemcpyAsync(frame.p, bgr, (size_t)h * w * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(pyr.p, &P, sizeof(SPyr), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(count.p, 0, 2 * sizeof(int), c->stream));
    // base image: doubled, blurred by sqrt(sigma^2 - 1) (into gp[0] via the last DoG plane as a temporary)
    {
        float* dbl = P.dog[SL + 1].px;  // free until the DoG of octave 0 is formed
        RTVM_LAUNCH(c, k_sift_base, grid2(bw, bh), dim3(256), 0, (const uint8_t*)frame.p, h, w, dbl);
        const float sig_diff = std::sqrt(std::max(SSIGMA * SSIGMA - 1.f, 0.01f));
        const Taps t = gauss_taps((double)sig_diff);
        RTVM_LAUNCH(c, k_sift_blur_rows, grid2(bw, bh), dim3(256), 0, (const float*)dbl, scratch, bh, bw, t);
        RTVM_LAUNCH(c, k_sift_blur_cols, grid2(bw, bh), dim3(256), 0, (const float*)scratch, P.gp[0].px, bh, bw, t,
                    (const float*)nullptr, (float*)nullptr);
    }
    Taps taps[SL + 3];
    for (int i = 1; i < SL + 3; i++) taps[i] = gauss_taps(sig[i]);
    const float threshold = (float)(int)std::floor(0.5 * SCONTRAST / SL * 255);
    // octaves from o_small on are built by one workgroup in one launch
    int o_small = n_oct;
    for (int o = n_oct - 1; o >= 1 && P.gp[o * (SL + 3)].h * P.gp[o * (SL + 3)].w <= SMALL_OCTAVE_PX; o--) o_small = o;
    for (int o = 0; o < o_small; o++) {
        const int oh = P.gp[o * (SL + 3)].h, ow = P.gp[o * (SL + 3)].w;
        for (int i = 0; i < SL + 3; i++) {
            SImg& dst = P.gp[o * (SL + 3) + i];
            if (i == 0) {
                if (o == 0) continue;
                const SImg& src = P.gp[(o - 1) * (SL + 3) + SL];
                RTVM_LAUNCH(c, k_sift_decimate, grid2(ow, oh), dim3(256), 0, (const float*)src.px, src.w, dst.px, oh, ow);
            } else {
                const SImg& src = P.gp[o * (SL + 3) + i - 1];
                RTVM_LAUNCH(c, k_sift_blur_rows, grid2(ow, oh), dim3(256), 0, (const float*)src.px, scratch, oh, ow, taps[i]);
                RTVM_LAUNCH(c, k_sift_blur_cols, grid2(ow, oh), dim3(256), 0, (const float*)scratch, dst.px, oh, ow, taps[i],
                            (const float*)src.px, P.dog[o * (SL + 2) + i - 1].px);
            }
        }
    }
    if (o_small < n_oct) {
        TapsAll ta;
        memset(&ta, 0, sizeof(ta));
        for (int i = 1; i < SL + 3; i++) ta.t[i] = taps[i];
        const int lds_px = P.gp[o_small * (SL + 3)].h * P.gp[o_small * (SL + 3)].w;
        const size_t lds_bytes = (size_t)2 * lds_px * sizeof(float);
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sift_small_octaves),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        RTVM_LAUNCH(c, k_sift_small_octaves, dim3(1), dim3(1024), lds_bytes, (const SPyr*)pyr.p, o_small, n_oct, ta, lds_px);
    }
    {
        ExtremaTiles T;
        memset(&T, 0, sizeof(T));
        T.n_oct = 0;
        for (int o = 0; o < n_oct; o++) {
            const int oh = P.gp[o * (SL + 3)].h, ow = P.gp[o * (SL + 3)].w;
            if (!(oh > 2 * SBORDER && ow > 2 * SBORDER)) break;  // the octaves only get smaller
            const dim3 g = grid2(ow - 2 * SBORDER, oh - 2 * SBORDER);
            T.tiles_x[o] = (int)g.x;
            T.tile_start[o + 1] = T.tile_start[o] + (int)(g.x * g.y);
            T.n_oct = o + 1;
        }
        if (T.n_oct > 0)
            RTVM_LAUNCH(c, k_sift_extrema, dim3(T.tile_start[T.n_oct], SL), dim3(256), 0, (const SPyr*)pyr.p, T, threshold,
                        (unsigned long long*)keys.p, (int*)count.p, cand_cap);
    }
    HIP_TRY(c, hipGetLastError());
    // refinement of every candidate, queued behind the extrema without a host round trip
    const int list_cap = 1 << 16, list_first = 1 << 13;
    SIFT_BUF(c, 5, cand_out, (size_t)list_cap * sizeof(SKp));
    int* d_list_count = (int*)count.p + 1;
    RTVM_LAUNCH(c, k_sift_refine, dim3(4096), dim3(64), 0, (const SPyr*)pyr.p, (const unsigned long long*)keys.p,
                (const int*)count.p, cand_cap, (SKp*)cand_out.p, d_list_count, list_cap);
    HIP_TRY(c, hipGetLastError());
    int counts[2] = {0, 0};
    std::vector<SKp> all(list_first);
    HIP_TRY(c, hipMemcpyAsync(counts, count.p, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(all.data(), cand_out.p, (size_t)list_first * sizeof(SKp), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (counts[0] > cand_cap) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: %d scale-space extrema, room for %d", counts[0], cand_cap);
    if (counts[1] > list_cap) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: %d keypoints before filtering, room for %d", counts[1], list_cap);
    if (counts[1] > list_first) {
        all.resize(counts[1]);
        HIP_TRY(c, hipMemcpyAsync(all.data() + list_first, (const SKp*)cand_out.p + list_first,
                                  (size_t)(counts[1] - list_first) * sizeof(SKp), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    all.resize(counts[1]);
    // KeyPointsFilter::removeDuplicatedSorted, retainBest(nfeatures); keypoints that compare equal
    // are equal in every field, so the order they arrived in does not matter
    std::sort(all.begin(), all.end(), kp_less);
    std::vector<SKp> uniq;
    for (const SKp& q : all)
        if (uniq.empty() || uniq.back().x != q.x || uniq.back().y != q.y || uniq.back().size != q.size ||
            uniq.back().angle != q.angle)
            uniq.push_back(q);
    if ((int)uniq.size() > nfeatures) {
        std::vector<float> resp;
        for (const SKp& q : uniq) resp.push_back(q.response);
        std::nth_element(resp.begin(), resp.begin() + (nfeatures - 1), resp.end(), std::greater<float>());
        const float cut = resp[nfeatures - 1];
        std::vector<SKp> kept;
        for (const SKp& q : uniq)
            if (q.response >= cut) kept.push_back(q);
        uniq.swap(kept);
    }
    const int n = std::min((int)uniq.size(), rtvm_sift_max_keypoints(nfeatures));
    *n_out = n;
    if (n == 0) return RTVM_OK;
    SIFT_BUF(c, 7, kp_dev, (size_t)n * (sizeof(SKp) + 128 * sizeof(float)) + 256);
    desc_dev.p = (char*)kp_dev.p + (((size_t)n * sizeof(SKp) + 255) & ~(size_t)255);
    HIP_TRY(c, hipMemcpyAsync(kp_dev.p, uniq.data(), (size_t)n * sizeof(SKp), hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_sift_describe, dim3(n), dim3(64), 0, (const SPyr*)pyr.p, (const SKp*)kp_dev.p, n,
                (float*)desc_dev.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(desc, desc_dev.p, (size_t)n * 128 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < n; q++) {
        const SKp& kp = uniq[q];
        xy[2 * q] = kp.x * 0.5f;
        xy[2 * q + 1] = kp.y * 0.5f;
        size[q] = kp.size * 0.5f;
        angle_deg_out[q] = kp.angle;
        response[q] = kp.response;
        octave[q] = (kp.octave & ~255) | ((kp.octave - 1) & 255);
    }
    return RTVM_OK;
}
