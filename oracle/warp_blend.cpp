// CPU restatement of VideMosaic.warp() (reference main.py:861-936): the OpenCV
// calls (warpPerspective, distanceTransform, GaussianBlur) restated from their
// published 4.x algorithms, the numpy statements around them restated literally.
// Test infrastructure only (see oracle.h).  Parity unpinned.
#include "oracle.h"

#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline int cv_round(double v) { return (int)std::nearbyint(v); }  // round half to even

// cv::invert() of a 3x3 double matrix (closed form through cofactors).  A
// singular matrix gives the zero matrix, as cv::invert does.
void invert3x3(const double* S, double* D) {
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.) {
        for (int i = 0; i < 9; i++) D[i] = 0.;
        return;
    }
    d = 1. / d;
    double t[9];
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    for (int i = 0; i < 9; i++) D[i] = t[i];
}

inline int tap(const uint8_t* src, int fh, int fw, int y, int x, int c) {
    return ((unsigned)x < (unsigned)fw && (unsigned)y < (unsigned)fh) ? src[((size_t)y * fw + x) * 3 + c] : 0;
}

}  // namespace

// main.py:871: cv2.warpPerspective(frame, H, (cw, ch), flags=INTER_LINEAR),
// BORDER_CONSTANT 0.  M = H^-1 in double.
//   fixed != 0 (OpenCV <= 4.10): source coordinates rounded to 1/32 pixel,
//     computed per 64-pixel block of the destination row as OpenCV's
//     warpPerspectiveInvoker does, integer bilinear weights summing to 32768.
//   fixed == 0 (OpenCV >= 4.11 for 8UC3): coordinates and interpolation in
//     float32, horizontal then vertical lerp, round half to even.
extern "C" int rtvm_oracle_warp_perspective(const uint8_t* bgr, int fh, int fw, const double H[9],
                                            int ch, int cw, int fixed, uint8_t* out) {
    if (fh <= 0 || fw <= 0 || ch <= 0 || cw <= 0) return -1;
    double M[9];
    invert3x3(H, M);
    if (fixed) {
        const int bh0 = ch < 16 ? ch : 16;
        const int bw0 = (1024 / bh0) < cw ? (1024 / bh0) : cw;
    #pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) {
                const int xb = (x / bw0) * bw0, x1 = x - xb;
                const double X0 = M[0] * xb + M[1] * y + M[2];
                const double Y0 = M[3] * xb + M[4] * y + M[5];
                const double W0 = M[6] * xb + M[7] * y + M[8];
                double W = W0 + M[6] * x1;
                W = W ? 32. / W : 0;
                const double fX = std::fmax((double)INT_MIN, std::fmin((double)INT_MAX, (X0 + M[0] * x1) * W));
                const double fY = std::fmax((double)INT_MIN, std::fmin((double)INT_MAX, (Y0 + M[3] * x1) * W));
                const int X = cv_round(fX), Y = cv_round(fY);
                int sx = X >> 5, sy = Y >> 5;
                sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;
                sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
                const int ax = X & 31, ay = Y & 31;
                const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
                const int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
                uint8_t* o = out + ((size_t)y * cw + x) * 3;
                for (int c = 0; c < 3; c++) {
                    const int v = w00 * tap(bgr, fh, fw, sy, sx, c) + w01 * tap(bgr, fh, fw, sy, sx + 1, c) +
                                  w10 * tap(bgr, fh, fw, sy + 1, sx, c) + w11 * tap(bgr, fh, fw, sy + 1, sx + 1, c);
                    o[c] = (uint8_t)((v + 16384) >> 15);
                }
            }
    } else {
        float Mf[9];
        for (int i = 0; i < 9; i++) Mf[i] = (float)M[i];
    #pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) {
                const float fx = (float)x, fy = (float)y;
                const float w = (Mf[6] * fx + Mf[7] * fy) + Mf[8];
                const float sx = ((Mf[0] * fx + Mf[1] * fy) + Mf[2]) / w;
                const float sy = ((Mf[3] * fx + Mf[4] * fy) + Mf[5]) / w;
                uint8_t* o = out + ((size_t)y * cw + x) * 3;
                if (!(sx > -1.f && sx < (float)fw && sy > -1.f && sy < (float)fh)) {
                    o[0] = o[1] = o[2] = 0;
                    continue;
                }
                const float flx = std::floor(sx), fly = std::floor(sy);
                const int ix = (int)flx, iy = (int)fly;
                const float a = sx - flx, b = sy - fly;
                for (int c = 0; c < 3; c++) {
                    const float p00 = (float)tap(bgr, fh, fw, iy, ix, c), p01 = (float)tap(bgr, fh, fw, iy, ix + 1, c);
                    const float p10 = (float)tap(bgr, fh, fw, iy + 1, ix, c), p11 = (float)tap(bgr, fh, fw, iy + 1, ix + 1, c);
                    const float v0 = p00 + (p01 - p00) * a;
                    const float v1 = p10 + (p11 - p10) * a;
                    const float v = v0 + (v1 - v0) * b;
                    int r = (int)std::nearbyintf(v);
                    o[c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                }
            }
    }
    return 0;
}

// main.py:888-889: cv2.distanceTransform(mask, DIST_L2, 3): 3x3 chamfer,
// a = 0.955 (edge), b = 1.3693 (diagonal) in Q16, forward then backward raster
// pass over a buffer with a one-pixel border of DIST_MAX, result t / 65536.
extern "C" void rtvm_oracle_distance_transform(const uint8_t* mask, int h, int w, float* dist) {
    const unsigned HV = (unsigned)cv_round(0.955 * 65536.0);
    const unsigned DG = (unsigned)cv_round(1.3693 * 65536.0);
    const unsigned DMAX = UINT_MAX - DG;
    const float scale = 1.f / 65536.f;
    const int step = w + 2;
    std::vector<unsigned> buf((size_t)(h + 2) * step, DMAX);
    for (int i = 0; i < h; i++) {
        unsigned* tmp = &buf[(size_t)(i + 1) * step + 1];
        const uint8_t* s = mask + (size_t)i * w;
        for (int j = 0; j < w; j++) {
            if (!s[j])
                tmp[j] = 0;
            else {
                unsigned t0 = tmp[j - step - 1] + DG;
                unsigned t = tmp[j - step] + HV;
                if (t0 > t) t0 = t;
                t = tmp[j - step + 1] + DG;
                if (t0 > t) t0 = t;
                t = tmp[j - 1] + HV;
                if (t0 > t) t0 = t;
                tmp[j] = (t0 > DMAX) ? DMAX : t0;
            }
        }
    }
    for (int i = h - 1; i >= 0; i--) {
        unsigned* tmp = &buf[(size_t)(i + 1) * step + 1];
        float* d = dist + (size_t)i * w;
        for (int j = w - 1; j >= 0; j--) {
            unsigned t0 = tmp[j];
            if (t0 > HV) {
                unsigned t = tmp[j + step + 1] + DG;
                if (t0 > t) t0 = t;
                t = tmp[j + step] + HV;
                if (t0 > t) t0 = t;
                t = tmp[j + step - 1] + DG;
                if (t0 > t) t0 = t;
                t = tmp[j + 1] + HV;
                if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
            t0 = (t0 > DMAX) ? DMAX : t0;
            d[j] = (float)t0 * scale;
        }
    }
}

namespace {
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// cv::getGaussianKernel(31, 0, CV_32F): sigma = 0.3*((31-1)*0.5 - 1) + 0.8 = 5
void gauss31(float* k) {
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
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
}
}  // namespace

// main.py:897-898: cv2.GaussianBlur(w.astype(float32), (31,31), 0), separable,
// BORDER_REFLECT_101, float32 accumulation with fused multiply-adds, as OpenCV's
// vector filters run on a CPU with FMA3 (its AVX2 code path): the generic row filter
// s = k0*x0, then s = fma(x_i, k_i, s) left to right; the symmetric column filter
// s = k15*r0, then s = fma(r[+i] + r[-i], k[15+i], s) outwards.
// One clone of the function is compiled for FMA3 hardware; the other calls libm's fmaf,
// which gives the same correctly rounded result.
extern "C" __attribute__((target_clones("fma", "default"))) void rtvm_oracle_gaussian_blur31(const float* src, int h,
                                                                                              int w, float* dst) {
    float k[31];
    gauss31(k);
    std::vector<float> rows((size_t)h * w);
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = k[0] * src[(size_t)y * w + reflect101(x - 15, w)];
            for (int i = 1; i < 31; i++) s = __builtin_fmaf(src[(size_t)y * w + reflect101(x - 15 + i, w)], k[i], s);
            rows[(size_t)y * w + x] = s;
        }
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = k[15] * rows[(size_t)y * w + x];
            for (int i = 1; i <= 15; i++)
                s = __builtin_fmaf(rows[(size_t)reflect101(y + i, h) * w + x] + rows[(size_t)reflect101(y - i, h) * w + x],
                                   k[15 + i], s);
            dst[(size_t)y * w + x] = s;
        }
}

// main.py:871-927 literally, on a u8 canvas: the reference's float64 canvas
// only ever holds the integers 0..255 (first frame u8, warped u8, blended.astype(uint8)).
extern "C" int rtvm_oracle_warp_blend(uint8_t* canvas, int ch, int cw, const uint8_t* bgr, int fh,
                                      int fw, const double H[9], int fixed) {
    const size_t n = (size_t)ch * cw;
    std::vector<uint8_t> warped(n * 3), mnew(n), mold(n), overlap(n);
    if (rtvm_oracle_warp_perspective(bgr, fh, fw, H, ch, cw, fixed, warped.data())) return -1;
    int any = 0;
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads) reduction(| : any)
    for (size_t p = 0; p < n; p++) {  // :878-882
        mnew[p] = (warped[3 * p] | warped[3 * p + 1] | warped[3 * p + 2]) ? 255 : 0;
        mold[p] = (canvas[3 * p] | canvas[3 * p + 1] | canvas[3 * p + 2]) ? 255 : 0;
        overlap[p] = mnew[p] & mold[p];
        any |= overlap[p] != 0;
    }
    if (any) {  // :885
        std::vector<float> dn(n), dold(n), wn(n), wo(n), bn(n), bo(n);
        // the two raster scans are sequential inside; with threads they run side by side
#pragma omp parallel sections num_threads(rtvm_oracle_nthreads > 1 ? 2 : 1)
        {
#pragma omp section
            rtvm_oracle_distance_transform(mnew.data(), ch, cw, dn.data());  // :888
#pragma omp section
            rtvm_oracle_distance_transform(mold.data(), ch, cw, dold.data());  // :889
        }
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
        for (size_t p = 0; p < n; p++) {  // :892-894
            const float sum = (dn[p] + dold[p]) + 1e-6f;
            wn[p] = dn[p] / sum;
            wo[p] = dold[p] / sum;
        }
        rtvm_oracle_gaussian_blur31(wn.data(), ch, cw, bn.data());  // :897
        rtvm_oracle_gaussian_blur31(wo.data(), ch, cw, bo.data());  // :898
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
        for (size_t p = 0; p < n; p++) {
            if (overlap[p]) {  // :905-910, astype(uint8) truncates
                for (int c = 0; c < 3; c++) {
                    const float v = (float)canvas[3 * p + c] * bo[p] + (float)warped[3 * p + c] * bn[p];
                    canvas[3 * p + c] = (uint8_t)(int)v;
                }
            } else if (mnew[p]) {  // :922-924
                for (int c = 0; c < 3; c++) canvas[3 * p + c] = warped[3 * p + c];
            }
        }
    } else {  // :927
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
        for (size_t p = 0; p < 3 * n; p++)
            if (warped[p] > 0) canvas[p] = warped[p];
    }
    return 0;
}
