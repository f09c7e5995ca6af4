// Classical per-pixel passes the reference runs on the finished mosaic before path planning
// (analyze_for_navigation, reference main.py:1051-1232; occupancy grid of find_path_astar,
// main.py:1344-1359): BGR2HSV, colour tests for fire and smoke, local standard deviation,
// fixed-point Gaussian texture measure, binary morphology, the dilated obstacle mask and its
// down-sampled grid.  Byte and integer work on an image of at most a few megapixels, run once per
// video: every kernel is a tiled pass bounded by LDS and integer issue, not by HBM.
#include "common.h"

#include <cmath>
#include <vector>

namespace {

constexpr int NAV_TW = 64, NAV_TH = 16;  // output tile of the tiled kernels, 256 threads

__device__ __forceinline__ int nav_reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// round-half-even of num/den for positive integers: what cvRound gives for the double quotient
__device__ __forceinline__ int div_round_even(int num, int den) {
    const int q = num / den, r = num - q * den;
    return 2 * r > den ? q + 1 : 2 * r == den ? q + (q & 1) : q;
}

// cv2.cvtColor(BGR2HSV), 8 bit: H in [0,180), S and H through 12-bit reciprocal tables
// (sdiv[i] = round(255*4096/i), hdiv[i] = round(180*4096/(6 i)), both 0 at i = 0).
__device__ __forceinline__ void hsv_tables(int* s_sdiv, int* s_hdiv) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        s_sdiv[i] = i ? div_round_even(255 << 12, i) : 0;
        s_hdiv[i] = i ? div_round_even((180 << 12) / 6, i) : 0;
    }
}

__device__ __forceinline__ void hsv_px(const int* s_sdiv, const int* s_hdiv, int b, int g, int r, int& H, int& S, int& V) {
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    S = (diff * s_sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * s_hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    H = h < 0 ? 0 : h > 255 ? 255 : h;
    V = v;
}

__global__ __launch_bounds__(256) void k_bgr2hsv(const uint8_t* __restrict__ bgr, size_t n, uint8_t* __restrict__ hsv) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    hsv_tables(s_sdiv, s_hdiv);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int H, S, V;
    hsv_px(s_sdiv, s_hdiv, bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], H, S, V);
    hsv[3 * i] = (uint8_t)H;
    hsv[3 * i + 1] = (uint8_t)S;
    hsv[3 * i + 2] = (uint8_t)V;
}

__device__ __forceinline__ bool in_range(int H, int S, int V, int l0, int l1, int l2, int u0, int u1, int u2) {
    return H >= l0 && H <= u0 && S >= l1 && S <= u1 && V >= l2 && V <= u2;
}

// main.py:1065-1138, 1155-1174, 1196 per pixel: gray, the fire mask before morphology, the smoke
// mask before the texture test, gray > 15, and the filled detection rectangles (main.py:1106).
// numpy's `g + 25` on uint8 planes wraps at 256 and so does this.  Canny (main.py:1141-1144)
// enters as mask | (edges & mask) = mask and is not computed.
__global__ __launch_bounds__(256) void k_nav_classify(const uint8_t* __restrict__ bgr, int h, int w,
                                                      const int* __restrict__ boxes, int n_boxes,
                                                      uint8_t* __restrict__ gray, uint8_t* __restrict__ fire,
                                                      uint8_t* __restrict__ smoke, uint8_t* __restrict__ valid,
                                                      uint8_t* __restrict__ boxed) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    hsv_tables(s_sdiv, s_hdiv);
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    const int gr = (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15;
    int H, S, V;
    hsv_px(s_sdiv, s_hdiv, b, g, r, H, S, V);
    const bool f_hsv = in_range(H, S, V, 0, 80, 120, 15, 255, 255) || in_range(H, S, V, 160, 80, 120, 180, 255, 255) ||
                       in_range(H, S, V, 15, 70, 130, 35, 255, 255);
    const bool f1 = r > ((g + 25) & 255) && r > ((b + 35) & 255) && r > 130;
    const bool f2 = r > 130 && g > 110 && r > ((b + 25) & 255) && g > ((b + 15) & 255) && abs(r - g) < 60;
    const bool s_hsv = in_range(H, S, V, 0, 0, 90, 180, 45, 245) || in_range(H, S, V, 0, 0, 60, 180, 55, 140);
    const bool s_bgr = abs(r - g) < 25 && abs(g - b) < 25 && r > 60 && g > 60 && b > 60;
    bool in_box = false;
    for (int k = 0; k < n_boxes; k++)
        in_box |= x >= boxes[4 * k] && x <= boxes[4 * k + 2] && y >= boxes[4 * k + 1] && y <= boxes[4 * k + 3];
    gray[i] = (uint8_t)gr;
    fire[i] = f_hsv && (f1 || f2) ? 255 : 0;
    smoke[i] = s_hsv && s_bgr && gr > 70 ? 255 : 0;
    valid[i] = gr > 15 ? 255 : 0;
    boxed[i] = in_box ? 255 : 0;
}

// main.py:1178-1185: cv2.filter2D with ones(11,11)/121 of gray and of gray^2 as float, local
// deviation sqrt(max(E[x^2] - E[x]^2, 0)) < 40 kept in the smoke mask.  Taps in row-major order,
// one fused multiply-add each; BORDER_REFLECT_101.  Tile of 64x16 outputs, four per thread.
__global__ __launch_bounds__(256) void k_nav_low_texture(const uint8_t* __restrict__ gray, int h, int w,
                                                         uint8_t* __restrict__ smoke) {
    constexpr int TW = NAV_TW + 10, TH = NAV_TH + 10;
    __shared__ float s_g[TH][TW + 1];
    const int x0 = blockIdx.x * NAV_TW, y0 = blockIdx.y * NAV_TH;
    for (int t = threadIdx.x; t < TW * TH; t += 256) {
        const int ty = t / TW, tx = t - ty * TW;
        s_g[ty][tx] = (float)gray[(size_t)nav_reflect101(y0 + ty - 5, h) * w + nav_reflect101(x0 + tx - 5, w)];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float kf = 1.0f / 121.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int oy = ly + 4 * q;
        const int x = x0 + lx, y = y0 + oy;
        float m = 0.f, sq = 0.f;
        for (int i = 0; i < 11; i++)
#pragma unroll
            for (int j = 0; j < 11; j++) {
                const float v = s_g[oy + i][lx + j];
                m = __builtin_fmaf(kf, v, m);
                sq = __builtin_fmaf(kf, v * v, sq);
            }
        if (x < w && y < h) {
            const float m2 = m * m;
            float d = sq - m2;
            d = d > 0.f ? d : 0.f;
            if (!(sqrtf(d) < 40.f)) smoke[(size_t)y * w + x] = 0;
        }
    }
}

// main.py:1201-1204: cv2.GaussianBlur(gray, (11,11), 0) in OpenCV's 8.8 fixed point (rows into
// 16 bits, columns into 32 with 16 fractional bits, rounded), |gray - blurred| > 6 inside the
// eroded valid area.  The kernel taps arrive by value.
struct GaussTaps {
    int k[11];
};

__global__ __launch_bounds__(256) void k_nav_texture(const uint8_t* __restrict__ gray, const uint8_t* __restrict__ valid,
                                                     int h, int w, GaussTaps taps, uint8_t* __restrict__ tex) {
    constexpr int TW = NAV_TW + 10, TH = NAV_TH + 10;
    __shared__ uint8_t s_g[TH][TW + 2];
    __shared__ uint16_t s_r[TH][NAV_TW];
    const int x0 = blockIdx.x * NAV_TW, y0 = blockIdx.y * NAV_TH;
    for (int t = threadIdx.x; t < TW * TH; t += 256) {
        const int ty = t / TW, tx = t - ty * TW;
        s_g[ty][tx] = gray[(size_t)nav_reflect101(y0 + ty - 5, h) * w + nav_reflect101(x0 + tx - 5, w)];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NAV_TW * TH; t += 256) {
        const int ty = t / NAV_TW, tx = t - ty * NAV_TW;
        int s = 0;
#pragma unroll
        for (int j = 0; j < 11; j++) s += taps.k[j] * s_g[ty][tx + j];
        s_r[ty][tx] = (uint16_t)s;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int oy = ly + 4 * q, x = x0 + lx, y = y0 + oy;
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < 11; i++) s += (uint32_t)taps.k[i] * s_r[oy + i][lx];
        uint32_t v = (s + (1u << 15)) >> 16;
        v = v > 255 ? 255 : v;
        if (x < w && y < h) {
            const size_t p = (size_t)y * w + x;
            tex[p] = abs((int)s_g[oy + 5][lx + 5] - (int)v) > 6 && valid[p] ? 255 : 0;
        }
    }
}

// One iteration of cv2.erode / cv2.dilate on a 0/255 mask with a structuring element given as
// the span [lo, hi) of each of its rows relative to the anchor.  Pixels outside the image never
// decide the result (OpenCV's default border value).  The tile's rows are prefix-summed in LDS,
// so each element row costs two reads: set pixels in the span = P[hi] - P[lo].
struct MorphElement {
    int rows, ay, reach;  // reach: largest |offset| of any span end
    signed char lo[32], hi[32];
};

__global__ __launch_bounds__(256) void k_nav_morph(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int h,
                                                   int w, MorphElement e, int dilate) {
    constexpr int MAXR = 16;
    __shared__ uint8_t s_m[NAV_TH + 2 * MAXR][NAV_TW + 2 * MAXR];
    __shared__ unsigned short s_p[NAV_TH + 2 * MAXR][NAV_TW + 2 * MAXR + 1];
    const int R = e.reach, TW = NAV_TW + 2 * R, TH = NAV_TH + e.rows - 1;
    const int x0 = blockIdx.x * NAV_TW - R, y0 = blockIdx.y * NAV_TH - e.ay;
    const uint8_t neutral = dilate ? 0 : 1;
    for (int t = threadIdx.x; t < TW * TH; t += 256) {
        const int ty = t / TW, tx = t - ty * TW;
        const int x = x0 + tx, y = y0 + ty;
        s_m[ty][tx] = x >= 0 && x < w && y >= 0 && y < h ? (src[(size_t)y * w + x] != 0) : neutral;
    }
    __syncthreads();
    if ((int)threadIdx.x < TH) {
        unsigned short acc = 0;
        s_p[threadIdx.x][0] = 0;
        for (int tx = 0; tx < TW; tx++) {
            acc += s_m[threadIdx.x][tx];
            s_p[threadIdx.x][tx + 1] = acc;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    bool any[4] = {false, false, false, false}, all[4] = {true, true, true, true};
    // unrolled to the largest element with an early exit: the spans are then read from the kernel
    // arguments at fixed offsets (scalar registers) instead of one scalar load per row and output
#pragma unroll
    for (int i = 0; i < 31; i++) {
        if (i >= e.rows) break;
        const int lo = e.lo[i], hi = e.hi[i];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int oy = ly + 4 * q;
            const int cnt = (int)s_p[oy + i][lx + R + hi] - (int)s_p[oy + i][lx + R + lo];
            any[q] |= cnt > 0;
            all[q] &= cnt == hi - lo;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = blockIdx.x * NAV_TW + lx, y = blockIdx.y * NAV_TH + ly + 4 * q;
        if (x < w && y < h) dst[(size_t)y * w + x] = (dilate ? any[q] : all[q]) ? 255 : 0;
    }
}

// main.py:1151,1191,1228: obstacles = rectangles | fire | smoke | texture
__global__ __launch_bounds__(256) void k_nav_union(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                   const uint8_t* __restrict__ c, const uint8_t* __restrict__ d, size_t n,
                                                   uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] | b[i] | c[i] | d[i];
}

// main.py:1349-1359: one thread per grid cell counts the set pixels of its scale x scale block
// (clipped to the image); blocked when count > 0.3 * scale^2, compared in double as Python does.
__global__ __launch_bounds__(256) void k_occupancy(const uint8_t* __restrict__ mask, int h, int w, int scale, int gh,
                                                   int gw, uint8_t* __restrict__ grid) {
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63), gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= gw || gy >= gh) return;
    int count = 0;
    const int y1 = min(h, gy * scale + scale), x1 = min(w, gx * scale + scale);
    for (int y = gy * scale; y < y1; y++)
        for (int x = gx * scale; x < x1; x++) count += mask[(size_t)y * w + x] != 0;
    grid[(size_t)gy * gw + gx] = (double)count > (double)(scale * scale) * 0.3 ? 1 : 0;
}

// ---- cv2.Canny (main.py:489,1141): aperture 3, L1 gradient ----
// Candidate map of one tile of 64 x 16 pixels: the gray tile with two pixels of replicated border in
// LDS, Sobel 3x3 and |dx| + |dy| for the tile and a one-pixel rim (zero outside the image), then the
// maximum test along the gradient direction.  map: 0 candidate (above `low`), 1 nothing, 2 edge
// (candidate above `high`).
constexpr int CN_TW = 64, CN_TH = 16;
__global__ __launch_bounds__(256) void k_canny_nms(const uint8_t* __restrict__ gray, int h, int w, int low, int high,
                                                   uint8_t* __restrict__ map) {
    __shared__ uint8_t s_g[CN_TH + 4][CN_TW + 4];
    __shared__ int s_m[CN_TH + 2][CN_TW + 2];
    __shared__ short s_dx[CN_TH + 2][CN_TW + 2], s_dy[CN_TH + 2][CN_TW + 2];
    const int x0 = blockIdx.x * CN_TW, y0 = blockIdx.y * CN_TH;
    for (int i = threadIdx.x; i < (CN_TH + 4) * (CN_TW + 4); i += 256) {
        const int r = i / (CN_TW + 4), c = i % (CN_TW + 4);
        const int y = min(max(y0 - 2 + r, 0), h - 1), x = min(max(x0 - 2 + c, 0), w - 1);
        s_g[r][c] = gray[(size_t)y * w + x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (CN_TH + 2) * (CN_TW + 2); i += 256) {
        const int r = i / (CN_TW + 2), c = i % (CN_TW + 2);
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        int gx = 0, gy = 0, m = 0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            // pixel (y, x) is s_g[r + 1][c + 1]
            const int a = s_g[r][c], b = s_g[r][c + 1], cc = s_g[r][c + 2];
            const int d = s_g[r + 1][c], f = s_g[r + 1][c + 2];
            const int g = s_g[r + 2][c], hh = s_g[r + 2][c + 1], k = s_g[r + 2][c + 2];
            gx = (cc + 2 * f + k) - (a + 2 * d + g);
            gy = (g + 2 * hh + k) - (a + 2 * b + cc);
            m = abs(gx) + abs(gy);
        }
        s_m[r][c] = m;
        s_dx[r][c] = (short)gx;
        s_dy[r][c] = (short)gy;
    }
    __syncthreads();
    const int TG22 = 13573;  // tan 22.5 degrees in 15-bit fixed point
    for (int i = threadIdx.x; i < CN_TH * CN_TW; i += 256) {
        const int r = i / CN_TW + 1, c = i % CN_TW + 1;
        const int y = y0 + r - 1, x = x0 + c - 1;
        if (y >= h || x >= w) continue;
        const int m = s_m[r][c];
        uint8_t out = 1;
        if (m > low) {
            const int xs = s_dx[r][c], ys = s_dy[r][c];
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * TG22;
            bool keep;
            if (ay < tg22x) {
                keep = m > s_m[r][c - 1] && m >= s_m[r][c + 1];
            } else if (ay > tg22x + (ax << 16)) {
                keep = m > s_m[r - 1][c] && m >= s_m[r + 1][c];
            } else {
                const int sgn = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > s_m[r - 1][c - sgn] && m > s_m[r + 1][c + sgn];
            }
            if (keep) out = m > high ? 2 : 0;
        }
        map[(size_t)y * w + x] = out;
    }
}

// Hysteresis: a candidate with an edge among its 8 neighbours becomes an edge.  A workgroup takes a
// tile of 64 x 16 pixels with a one-pixel rim and repeats the step inside the tile until nothing
// changes there; *changed says whether any tile changed, and the host launches again until none did.
__global__ __launch_bounds__(256) void k_canny_hyst(uint8_t* __restrict__ map, int h, int w, int* __restrict__ changed) {
    __shared__ uint8_t s[CN_TH + 2][CN_TW + 2];
    __shared__ int s_again, s_any;
    const int x0 = blockIdx.x * CN_TW, y0 = blockIdx.y * CN_TH;
    for (int i = threadIdx.x; i < (CN_TH + 2) * (CN_TW + 2); i += 256) {
        const int r = i / (CN_TW + 2), c = i % (CN_TW + 2);
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        s[r][c] = (y >= 0 && y < h && x >= 0 && x < w) ? map[(size_t)y * w + x] : (uint8_t)1;
    }
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) s_again = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < CN_TH * CN_TW; i += 256) {
            const int r = i / CN_TW + 1, c = i % CN_TW + 1;
            if (s[r][c] != 0) continue;
            const bool near = s[r - 1][c - 1] == 2 || s[r - 1][c] == 2 || s[r - 1][c + 1] == 2 || s[r][c - 1] == 2 ||
                              s[r][c + 1] == 2 || s[r + 1][c - 1] == 2 || s[r + 1][c] == 2 || s[r + 1][c + 1] == 2;
            if (near) {
                s[r][c] = 2;  // only ever 0 -> 2: a neighbour reading the old value finds the new one next round
                s_again = 1;
            }
        }
        __syncthreads();
        const int again = s_again;
        __syncthreads();
        if (!again) break;
        if (threadIdx.x == 0) s_any = 1;
    }
    __syncthreads();
    if (!s_any) return;
    for (int i = threadIdx.x; i < CN_TH * CN_TW; i += 256) {
        const int r = i / CN_TW + 1, c = i % CN_TW + 1;
        const int y = y0 + r - 1, x = x0 + c - 1;
        if (y < h && x < w && s[r][c] == 2) map[(size_t)y * w + x] = 2;
    }
    if (threadIdx.x == 0) *changed = 1;
}

__global__ __launch_bounds__(256) void k_canny_out(const uint8_t* __restrict__ map, size_t n, uint8_t* __restrict__ edges) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) edges[i] = map[i] == 2 ? 255 : 0;
}

// ---- host side ----

int cv_round(double v) { return (int)std::lrint(v); }

// cv2.getStructuringElement(shape, (k, k)) as row spans; shape 0 rect, 1 cross, 2 ellipse
bool make_element(int shape, int k, MorphElement& e) {
    if (k < 1 || k > 31 || shape < 0 || shape > 2) return false;
    e.rows = k;
    e.ay = k / 2;
    e.reach = 0;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < 32; i++) e.lo[i] = e.hi[i] = 0;
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0) {
            j2 = k;
        } else if (shape == 1) {
            if (i == r) j2 = k;
            else j1 = c, j2 = c + 1;
        } else {
            const int dy = i - r;
            if (std::abs(dy) <= r) {
                const int dx = cv_round(c * std::sqrt((r * r - dy * dy) * inv_r2));
                j1 = std::max(c - dx, 0);
                j2 = std::min(c + dx + 1, k);
            }
        }
        e.lo[i] = (signed char)(j1 - c);
        e.hi[i] = (signed char)(j2 - c);
        if (j2 > j1) e.reach = std::max(e.reach, std::max(std::abs(j1 - c), std::abs(j2 - c)));
    }
    return true;
}

// OpenCV's 8.8 fixed-point Gaussian kernel for sigma derived from the size: rounding error
// carried from tap to tap, the centre takes what is left of 256.
void gauss_taps(int n, int* k) {
    const double sigma = 0.3 * ((n - 1) * 0.5 - 1) + 0.8;
    std::vector<double> t(n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        t[i] = std::exp(-0.5 / (sigma * sigma) * x * x);
        sum += t[i];
    }
    double err = 0;
    int acc = 0;
    for (int i = 0; i < n / 2; i++) {
        const double adj = t[i] / sum * 256. + err;
        const int v = cv_round(adj);
        err = adj - v;
        k[i] = k[n - 1 - i] = v;
        acc += v;
    }
    k[n / 2] = 256 - 2 * acc;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    uint8_t* u8() const { return (uint8_t*)p; }
};

dim3 tiles(int h, int w) { return dim3((w + NAV_TW - 1) / NAV_TW, (h + NAV_TH - 1) / NAV_TH); }

// op 0 erode, 1 dilate, 2 open, 3 close; cv2.morphologyEx runs every iteration of the first
// operation before the second one's.  The result ends in `m`; `t` is scratch of the same size.
int morphology_dev(rtvm_ctx* c, uint8_t*& m, uint8_t*& t, int h, int w, int op, int shape, int k, int iterations) {
    MorphElement e;
    if (!make_element(shape, k, e) || op < 0 || op > 3) RTVM_FAIL(c, RTVM_E_ARG, "morphology: bad operation or element");
    if (k == 1 || iterations < 1) return RTVM_OK;
    const bool first = op == 1 || op == 3;
    for (int pass = 0; pass < (op >= 2 ? 2 : 1); pass++)
        for (int it = 0; it < iterations; it++) {
            RTVM_LAUNCH(c, k_nav_morph, tiles(h, w), dim3(256), 0, (const uint8_t*)m, t, h, w, e,
                        (int)(pass == 0 ? first : !first));
            std::swap(m, t);
        }
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

}  // namespace

extern "C" int rtvm_bgr2hsv(rtvm_ctx* c, const uint8_t* bgr, int h, int w, uint8_t* hsv) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !hsv || h < 1 || w < 1) RTVM_FAIL(c, RTVM_E_ARG, "bgr2hsv: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)h * w;
    DevBuf in, out;
    HIP_TRY(c, hipMalloc(&in.p, n * 3));
    HIP_TRY(c, hipMalloc(&out.p, n * 3));
    HIP_TRY(c, hipMemcpyAsync(in.p, bgr, n * 3, hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_bgr2hsv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, in.u8(), n, out.u8());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(hsv, out.p, n * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_morphology(rtvm_ctx* c, const uint8_t* mask, int h, int w, int op, int shape, int ksize,
                               int iterations, uint8_t* out) {
    if (!c) return RTVM_E_ARG;
    if (!mask || !out || h < 1 || w < 1 || iterations < 0) RTVM_FAIL(c, RTVM_E_ARG, "morphology: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)h * w;
    DevBuf a, b;
    HIP_TRY(c, hipMalloc(&a.p, n));
    HIP_TRY(c, hipMalloc(&b.p, n));
    HIP_TRY(c, hipMemcpyAsync(a.p, mask, n, hipMemcpyHostToDevice, c->stream));
    uint8_t *m = a.u8(), *t = b.u8();
    MorphElement e;
    if (!make_element(shape, ksize, e) || op < 0 || op > 3)
        RTVM_FAIL(c, RTVM_E_ARG, "morphology: operation %d, shape %d, size %d not supported", op, shape, ksize);
    if (ksize == 1 || iterations == 0) {
        // cv2 copies the source; values other than 0 still come back as 255
        MorphElement one;
        make_element(0, 1, one);
        RTVM_LAUNCH(c, k_nav_morph, tiles(h, w), dim3(256), 0, (const uint8_t*)m, t, h, w, one, 1);
        HIP_TRY(c, hipGetLastError());
        std::swap(m, t);
    } else {
        const int r = morphology_dev(c, m, t, h, w, op, shape, ksize, iterations);
        if (r) return r;
    }
    HIP_TRY(c, hipMemcpyAsync(out, m, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_canny(rtvm_ctx* c, const uint8_t* gray, int h, int w, double low_thresh, double high_thresh,
                          uint8_t* edges) {
    if (!c) return RTVM_E_ARG;
    if (!gray || !edges || h < 1 || w < 1 || !(low_thresh == low_thresh) || !(high_thresh == high_thresh))
        RTVM_FAIL(c, RTVM_E_ARG, "canny: bad argument");
    if (low_thresh > high_thresh) std::swap(low_thresh, high_thresh);
    const double lim = 1 << 20;  // magnitudes stay below 2040
    const int low = (int)std::floor(std::min(std::max(low_thresh, -lim), lim));
    const int high = (int)std::floor(std::min(std::max(high_thresh, -lim), lim));
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)h * w;
    DevBuf in, map, out, flag;
    HIP_TRY(c, hipMalloc(&in.p, n));
    HIP_TRY(c, hipMalloc(&map.p, n));
    HIP_TRY(c, hipMalloc(&out.p, n));
    HIP_TRY(c, hipMalloc(&flag.p, sizeof(int)));
    HIP_TRY(c, hipMemcpyAsync(in.p, gray, n, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((w + CN_TW - 1) / CN_TW, (h + CN_TH - 1) / CN_TH);
    RTVM_LAUNCH(c, k_canny_nms, grid, dim3(256), 0, (const uint8_t*)in.u8(), h, w, low, high, map.u8());
    // Hysteresis inside every tile until nothing changes there, launched again until no tile changed.  A round only
    // turns "weak" pixels into "edge" pixels, so the rounds end; an edge that weaves across a tile border needs one
    // round per crossing, which no function of the tile grid bounds.
    int* h_flag = (int*)c->h_pinned;
    for (;;) {
        HIP_TRY(c, hipMemsetAsync(flag.p, 0, sizeof(int), c->stream));
        RTVM_LAUNCH(c, k_canny_hyst, grid, dim3(256), 0, map.u8(), h, w, (int*)flag.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(h_flag, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!*h_flag) break;
    }
    RTVM_LAUNCH(c, k_canny_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const uint8_t*)map.u8(), n, out.u8());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(edges, out.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_nav_obstacles(rtvm_ctx* c, const uint8_t* bgr, int h, int w, const int32_t* boxes, int n_boxes,
                                  uint8_t* mask_fire, uint8_t* mask_smoke, uint8_t* texture_mask, uint8_t* obstacles,
                                  uint8_t* obstacles_nav, int grid_scale, uint8_t* grid) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || h < 1 || w < 1 || n_boxes < 0 || (n_boxes && !boxes) || (grid && grid_scale < 1))
        RTVM_FAIL(c, RTVM_E_ARG, "nav_obstacles: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)h * w;
    DevBuf img, dbox, gray, fire, smoke, valid, boxed, tex, tmp, obs, dgrid;
    HIP_TRY(c, hipMalloc(&img.p, n * 3));
    for (DevBuf* b : {&gray, &fire, &smoke, &valid, &boxed, &tex, &tmp, &obs}) HIP_TRY(c, hipMalloc(&b->p, n));
    HIP_TRY(c, hipMemcpyAsync(img.p, bgr, n * 3, hipMemcpyHostToDevice, c->stream));
    // rectangles as cv2.rectangle(..., -1) fills them: both corners included, in either order
    std::vector<int> norm(4 * (size_t)std::max(n_boxes, 1));
    for (int k = 0; k < n_boxes; k++) {
        norm[4 * k] = std::min(boxes[4 * k], boxes[4 * k + 2]);
        norm[4 * k + 2] = std::max(boxes[4 * k], boxes[4 * k + 2]);
        norm[4 * k + 1] = std::min(boxes[4 * k + 1], boxes[4 * k + 3]);
        norm[4 * k + 3] = std::max(boxes[4 * k + 1], boxes[4 * k + 3]);
    }
    HIP_TRY(c, hipMalloc(&dbox.p, norm.size() * sizeof(int)));
    HIP_TRY(c, hipMemcpyAsync(dbox.p, norm.data(), norm.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));

    RTVM_LAUNCH(c, k_nav_classify, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, img.u8(), h, w, (const int*)dbox.p,
                n_boxes, gray.u8(), fire.u8(), smoke.u8(), valid.u8(), boxed.u8());
    uint8_t *f = fire.u8(), *s = smoke.u8(), *v = valid.u8(), *x = tex.u8(), *t = tmp.u8();
    int r;
    // main.py:1147-1150
    if ((r = morphology_dev(c, f, t, h, w, 3, 2, 7, 2))) return r;
    if ((r = morphology_dev(c, f, t, h, w, 2, 2, 3, 1))) return r;
    // main.py:1178-1190
    RTVM_LAUNCH(c, k_nav_low_texture, tiles(h, w), dim3(256), 0, (const uint8_t*)gray.u8(), h, w, s);
    if ((r = morphology_dev(c, s, t, h, w, 3, 2, 11, 3))) return r;
    if ((r = morphology_dev(c, s, t, h, w, 2, 2, 3, 1))) return r;
    // main.py:1196-1210
    if ((r = morphology_dev(c, v, t, h, w, 0, 0, 10, 1))) return r;
    GaussTaps taps;
    gauss_taps(11, taps.k);
    RTVM_LAUNCH(c, k_nav_texture, tiles(h, w), dim3(256), 0, (const uint8_t*)gray.u8(), (const uint8_t*)v, h, w, taps, x);
    if ((r = morphology_dev(c, x, t, h, w, 3, 2, 5, 1))) return r;
    if ((r = morphology_dev(c, x, t, h, w, 2, 2, 3, 1))) return r;
    // main.py:1151,1191,1228 and 1231-1232
    RTVM_LAUNCH(c, k_nav_union, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const uint8_t*)boxed.u8(),
                (const uint8_t*)f, (const uint8_t*)s, (const uint8_t*)x, n, obs.u8());
    MorphElement e15;
    make_element(2, 15, e15);
    uint8_t* nav = gray.u8();  // the gray plane is no longer needed
    RTVM_LAUNCH(c, k_nav_morph, tiles(h, w), dim3(256), 0, (const uint8_t*)obs.u8(), nav, h, w, e15, 1);
    HIP_TRY(c, hipGetLastError());
    int gh = 0, gw = 0;
    if (grid) {
        gh = std::max(1, h / grid_scale);
        gw = std::max(1, w / grid_scale);
        HIP_TRY(c, hipMalloc(&dgrid.p, (size_t)gh * gw));
        RTVM_LAUNCH(c, k_occupancy, dim3((gw + 63) / 64, (gh + 3) / 4), dim3(256), 0, (const uint8_t*)nav, h, w,
                    grid_scale, gh, gw, dgrid.u8());
        HIP_TRY(c, hipGetLastError());
    }
    if (mask_fire) HIP_TRY(c, hipMemcpyAsync(mask_fire, f, n, hipMemcpyDeviceToHost, c->stream));
    if (mask_smoke) HIP_TRY(c, hipMemcpyAsync(mask_smoke, s, n, hipMemcpyDeviceToHost, c->stream));
    if (texture_mask) HIP_TRY(c, hipMemcpyAsync(texture_mask, x, n, hipMemcpyDeviceToHost, c->stream));
    if (obstacles) HIP_TRY(c, hipMemcpyAsync(obstacles, obs.p, n, hipMemcpyDeviceToHost, c->stream));
    if (obstacles_nav) HIP_TRY(c, hipMemcpyAsync(obstacles_nav, nav, n, hipMemcpyDeviceToHost, c->stream));
    if (grid) HIP_TRY(c, hipMemcpyAsync(grid, dgrid.p, (size_t)gh * gw, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}

extern "C" int rtvm_occupancy_grid(rtvm_ctx* c, const uint8_t* mask, int h, int w, int scale, uint8_t* grid) {
    if (!c) return RTVM_E_ARG;
    if (!mask || !grid || h < 1 || w < 1 || scale < 1) RTVM_FAIL(c, RTVM_E_ARG, "occupancy_grid: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const int gh = std::max(1, h / scale), gw = std::max(1, w / scale);
    DevBuf in, out;
    HIP_TRY(c, hipMalloc(&in.p, (size_t)h * w));
    HIP_TRY(c, hipMalloc(&out.p, (size_t)gh * gw));
    HIP_TRY(c, hipMemcpyAsync(in.p, mask, (size_t)h * w, hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_occupancy, dim3((gw + 63) / 64, (gh + 3) / 4), dim3(256), 0, (const uint8_t*)in.u8(), h, w, scale, gh,
                gw, out.u8());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(grid, out.p, (size_t)gh * gw, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}
