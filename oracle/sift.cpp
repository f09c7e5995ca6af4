// CPU restatement of cv2.SIFT_create(n).detectAndCompute(gray, None) (reference main.py:33,
// 111-112, 717-718; OpenCV 4.x features2d/src/sift.dispatch.cpp, sift.simd.hpp) with OpenCV's
// defaults nOctaveLayers = 3, contrastThreshold = 0.04, edgeThreshold = 10, sigma = 1.6.
// Test infrastructure only; parity with OpenCV itself is unpinned (see oracle.h).
//
// Where OpenCV calls its vectorised approximations (hal::exp32f, fastAtan2, cosf/sinf) this
// restatement evaluates the function in double precision and rounds to float, which the GPU
// reproduces bit for bit; where OpenCV's thread scheduling decides the order of the
// keypoints, the order here is (octave, layer, row, column, histogram peak).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

constexpr int L = 3;             // nOctaveLayers
constexpr int BORDER = 5;        // SIFT_IMG_BORDER
constexpr int ORI_BINS = 36;     // SIFT_ORI_HIST_BINS
constexpr int MAX_PEAKS = 8;     // orientation peaks kept per extremum (OpenCV keeps all; more than 8 do not occur)
constexpr int D = 4, NB = 8;     // SIFT_DESCR_WIDTH, SIFT_DESCR_HIST_BINS
constexpr float SIGMA = 1.6f;
constexpr float CONTRAST = 0.04f, EDGE = 10.f;

struct ImgF {
    int h = 0, w = 0;
    std::vector<float> px;
    float at(int y, int x) const { return px[(size_t)y * w + x]; }
};

inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int cv_round_d(double v) { return (int)std::lrint(v); }

// cv::GaussianBlur(src, dst, Size(), sigma, sigma) on CV_32F: ksize = cvRound(sigma*8 + 1) | 1,
// kernel from double exp, row pass s = k0*x0, fma(x_i, k_i, s); column pass symmetric, fma.
void gauss_blur(const ImgF& s, double sigma, ImgF& d) {
    const int n = cv_round_d(sigma * 8 + 1) | 1, half = n / 2;
    std::vector<float> k(n);
    {
        std::vector<double> t(n);
        double sum = 0;
        const double scale2 = -0.5 / (sigma * sigma);
        for (int i = 0; i < n; i++) {
            const double x = i - (n - 1) * 0.5;
            t[i] = std::exp(scale2 * x * x);
            sum += t[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    }
    ImgF rows;
    rows.h = s.h, rows.w = s.w, rows.px.resize(s.px.size());
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < s.h; y++)
        for (int x = 0; x < s.w; x++) {
            float a = k[0] * s.at(y, reflect101(x - half, s.w));
            for (int i = 1; i < n; i++) a = __builtin_fmaf(s.at(y, reflect101(x - half + i, s.w)), k[i], a);
            rows.px[(size_t)y * s.w + x] = a;
        }
    d.h = s.h, d.w = s.w, d.px.resize(s.px.size());
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < s.h; y++)
        for (int x = 0; x < s.w; x++) {
            float a = k[half] * rows.at(y, x);
            for (int i = 1; i <= half; i++)
                a = __builtin_fmaf(rows.at(reflect101(y + i, s.h), x) + rows.at(reflect101(y - i, s.h), x), k[half + i], a);
            d.px[(size_t)y * s.w + x] = a;
        }
}

// createInitialImage: gray as float, doubled with cv::resize(INTER_LINEAR) in float
// (sample position (d + 0.5)/2 - 0.5, clamped, S[x]*a0 + S[x+1]*a1 then rows likewise),
// blurred by sqrt(sigma^2 - (2*0.5)^2).
void initial_image(const uint8_t* gray, int h, int w, ImgF& base) {
    ImgF dbl;
    dbl.h = 2 * h, dbl.w = 2 * w, dbl.px.resize((size_t)4 * h * w);
    std::vector<int> xo(2 * w), yo(2 * h);
    std::vector<float> xa(2 * w), ya(2 * h);
    auto table = [](int n, std::vector<int>& o, std::vector<float>& a) {
        for (int d = 0; d < 2 * n; d++) {
            float f = (float)((d + 0.5) * 0.5 - 0.5);
            int s = (int)std::floor(f);
            f -= s;
            if (s < 0) f = 0, s = 0;
            if (s >= n - 1) f = 0, s = n - 1;
            o[d] = s, a[d] = f;
        }
    };
    table(w, xo, xa);
    table(h, yo, ya);
    for (int y = 0; y < 2 * h; y++) {
        const int y0 = yo[y], y1 = std::min(y0 + 1, h - 1);
        const float b1 = ya[y], b0 = 1.f - b1;
        for (int x = 0; x < 2 * w; x++) {
            const int x0 = xo[x], x1 = std::min(x0 + 1, w - 1);
            const float a1 = xa[x], a0 = 1.f - a1;
            const float r0 = (float)gray[(size_t)y0 * w + x0] * a0 + (float)gray[(size_t)y0 * w + x1] * a1;
            const float r1 = (float)gray[(size_t)y1 * w + x0] * a0 + (float)gray[(size_t)y1 * w + x1] * a1;
            dbl.px[(size_t)y * 2 * w + x] = r0 * b0 + r1 * b1;
        }
    }
    const float sig_diff = std::sqrt(std::max(SIGMA * SIGMA - 1.f, 0.01f));
    gauss_blur(dbl, (double)sig_diff, base);
}

struct Kp {
    float x, y, size, angle, response;
    int octave;
};

// 3x3 solve the way cv::Matx33f::solve(b, DECOMP_LU) does it: Cramer's rule.
bool solve3(const float a[3][3], const float b[3], float x[3]) {
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

bool adjust_extremum(const std::vector<ImgF>& dog, int octv, int& layer, int& r, int& c, Kp& kp, float& xi_out) {
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < 5; i++) {
        const ImgF& img = dog[octv * (L + 2) + layer];
        const ImgF& prev = dog[octv * (L + 2) + layer - 1];
        const ImgF& next = dog[octv * (L + 2) + layer + 1];
        const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale,
                             (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                             (next.at(r, c) - prev.at(r, c)) * deriv_scale};
        const float v2 = img.at(r, c) * 2;
        const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second;
        const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second;
        const float dss = (next.at(r, c) + prev.at(r, c) - v2) * second;
        const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross;
        const float dxs = (next.at(r, c + 1) - next.at(r, c - 1) - prev.at(r, c + 1) + prev.at(r, c - 1)) * cross;
        const float dys = (next.at(r + 1, c) - next.at(r - 1, c) - prev.at(r + 1, c) + prev.at(r - 1, c)) * cross;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        if (!solve3(H, dD, X)) X[0] = X[1] = X[2] = 0;  // Matx::solve returns zeros for a singular matrix
        xi = -X[2], xr = -X[1], xc = -X[0];
        if (std::fabs(xi) < 0.5f && std::fabs(xr) < 0.5f && std::fabs(xc) < 0.5f) break;
        if (std::fabs(xi) > (float)(INT32_MAX / 3) || std::fabs(xr) > (float)(INT32_MAX / 3) ||
            std::fabs(xc) > (float)(INT32_MAX / 3))
            return false;
        c += cv_round(xc);
        r += cv_round(xr);
        layer += cv_round(xi);
        if (layer < 1 || layer > L || c < BORDER || c >= img.w - BORDER || r < BORDER || r >= img.h - BORDER) return false;
    }
    if (i >= 5) return false;
    {
        const ImgF& img = dog[octv * (L + 2) + layer];
        const ImgF& prev = dog[octv * (L + 2) + layer - 1];
        const ImgF& next = dog[octv * (L + 2) + layer + 1];
        const float dD[3] = {(img.at(r, c + 1) - img.at(r, c - 1)) * deriv_scale,
                             (img.at(r + 1, c) - img.at(r - 1, c)) * deriv_scale,
                             (next.at(r, c) - prev.at(r, c)) * deriv_scale};
        const float t = (dD[0] * xc + dD[1] * xr) + dD[2] * xi;
        const float contr = img.at(r, c) * img_scale + t * 0.5f;
        if (std::fabs(contr) * L < CONTRAST) return false;
        const float v2 = img.at(r, c) * 2.f;
        const float dxx = (img.at(r, c + 1) + img.at(r, c - 1) - v2) * second;
        const float dyy = (img.at(r + 1, c) + img.at(r - 1, c) - v2) * second;
        const float dxy = (img.at(r + 1, c + 1) - img.at(r + 1, c - 1) - img.at(r - 1, c + 1) + img.at(r - 1, c - 1)) * cross;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * EDGE >= (EDGE + 1) * (EDGE + 1) * det) return false;
        kp.x = (c + xc) * (float)(1 << octv);
        kp.y = (r + xr) * (float)(1 << octv);
        kp.octave = octv + (layer << 8) + (cv_round((xi + 0.5f) * 255) << 16);
        kp.size = SIGMA * (float)std::pow(2.0, (double)((layer + xi) / L)) * (float)(1 << octv) * 2;
        kp.response = std::fabs(contr);
    }
    xi_out = xi;
    return true;
}

// degrees in [0, 360), atan2 in double rounded to float
inline float angle_deg(float y, float x) {
    float a = (float)(std::atan2((double)y, (double)x) * (180.0 / 3.14159265358979323846));
    if (a < 0) a += 360.f;
    if (a >= 360.f) a -= 360.f;
    return a;
}

float orientation_hist(const ImgF& img, int px, int py, int radius, float sigma, float* hist) {
    const int n = ORI_BINS;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float temphist[ORI_BINS];
    for (int i = 0; i < n; i++) temphist[i] = 0.f;
    for (int i = -radius; i <= radius; i++) {
        const int y = py + i;
        if (y <= 0 || y >= img.h - 1) continue;
        for (int j = -radius; j <= radius; j++) {
            const int x = px + j;
            if (x <= 0 || x >= img.w - 1) continue;
            const float dx = img.at(y, x + 1) - img.at(y, x - 1);
            const float dy = img.at(y - 1, x) - img.at(y + 1, x);
            const float W = (float)std::exp((double)((float)(i * i + j * j) * expf_scale));
            const float ori = angle_deg(dy, dx);
            const float mag = std::sqrt(dx * dx + dy * dy);
            int bin = cv_round((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            temphist[bin] += W * mag;
        }
    }
    float maxval = 0;
    for (int i = 0; i < n; i++) {
        const float m2 = temphist[(i + n - 2) % n], m1 = temphist[(i + n - 1) % n];
        const float p1 = temphist[(i + 1) % n], p2 = temphist[(i + 2) % n];
        hist[i] = (m2 + p2) * (1.f / 16.f) + (m1 + p1) * (4.f / 16.f) + temphist[i] * (6.f / 16.f);
        maxval = i == 0 ? hist[0] : std::max(maxval, hist[i]);
    }
    return maxval;
}

void descriptor(const ImgF& img, float ptx, float pty, float ori, float scl, float* dst) {
    const int d = D, n = NB;
    const int px = cv_round(ptx), py = cv_round(pty);
    float cos_t = (float)std::cos((double)(ori * (float)(3.14159265358979323846 / 180.)));
    float sin_t = (float)std::sin((double)(ori * (float)(3.14159265358979323846 / 180.)));
    const float bins_per_rad = n / 360.f;
    const float exp_scale = -1.f / (d * d * 0.5f);
    const float hist_width = 3.f * scl;
    int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = std::min(radius, (int)std::sqrt((double)img.w * img.w + (double)img.h * img.h));
    cos_t /= hist_width;
    sin_t /= hist_width;
    float hist[(D + 2) * (D + 2) * (NB + 2)];
    for (float& v : hist) v = 0.f;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = j * cos_t - i * sin_t;
            const float r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f;
            float cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img.h - 1 && c > 0 && c < img.w - 1))
                continue;
            const float dx = img.at(r, c + 1) - img.at(r, c - 1);
            const float dy = img.at(r - 1, c) - img.at(r + 1, c);
            const float W = (float)std::exp((double)((c_rot * c_rot + r_rot * r_rot) * exp_scale));
            const float Ori = angle_deg(dy, dx);
            const float Mag = std::sqrt(dx * dx + dy * dy);
            float obin = (Ori - ori) * bins_per_rad;
            const float mag = Mag * W;
            const int r0 = (int)std::floor(rbin), c0 = (int)std::floor(cbin);
            int o0 = (int)std::floor(obin);
            rbin -= r0;
            cbin -= c0;
            obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
            const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
            const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v_rco000;
            hist[idx + 1] += v_rco001;
            hist[idx + (n + 2)] += v_rco010;
            hist[idx + (n + 3)] += v_rco011;
            hist[idx + (d + 2) * (n + 2)] += v_rco100;
            hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
            hist[idx + (d + 3) * (n + 2)] += v_rco110;
            hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
        }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; k++) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
    const float thr = std::sqrt(nrm2) * 0.2f;
    nrm2 = 0;
    for (int k = 0; k < len; k++) {
        const float val = std::min(dst[k], thr);
        dst[k] = val;
        nrm2 += val * val;
    }
    nrm2 = 512.f / std::max(std::sqrt(nrm2), FLT_EPSILON);
    for (int k = 0; k < len; k++) {
        const int v = cv_round(dst[k] * nrm2);
        dst[k] = (float)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

bool kp_less(const Kp& a, const Kp& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    return a.octave > b.octave;
}

}  // namespace

extern "C" int rtvm_oracle_sift_max_keypoints(int nfeatures) { return nfeatures + 256; }

extern "C" int rtvm_oracle_sift(const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy, float* size,
                                float* angle_deg_out, float* response, int32_t* octave, float* desc) {
    if (h < 16 || w < 16) return -1;
    std::vector<uint8_t> gray((size_t)h * w);
    rtvm_oracle_gray(bgr, h, w, gray.data());
    ImgF base;
    initial_image(gray.data(), h, w, base);
    const int n_oct = cv_round_d(std::log((double)std::min(base.w, base.h)) / std::log(2.) - 2) + 1;
    double sig[L + 3];
    sig[0] = SIGMA;
    const double k = std::pow(2., 1. / L);
    for (int i = 1; i < L + 3; i++) {
        const double sp = std::pow(k, (double)(i - 1)) * SIGMA, st = sp * k;
        sig[i] = std::sqrt(st * st - sp * sp);
    }
    std::vector<ImgF> gp((size_t)n_oct * (L + 3)), dog((size_t)n_oct * (L + 2));
    for (int o = 0; o < n_oct; o++)
        for (int i = 0; i < L + 3; i++) {
            ImgF& dst = gp[o * (L + 3) + i];
            if (o == 0 && i == 0) {
                dst = base;
            } else if (i == 0) {
                const ImgF& src = gp[(o - 1) * (L + 3) + L];
                dst.h = src.h / 2, dst.w = src.w / 2, dst.px.resize((size_t)dst.h * dst.w);
                for (int y = 0; y < dst.h; y++)
                    for (int x = 0; x < dst.w; x++) dst.px[(size_t)y * dst.w + x] = src.at(2 * y, 2 * x);
            } else {
                gauss_blur(gp[o * (L + 3) + i - 1], sig[i], dst);
            }
        }
    for (int o = 0; o < n_oct; o++)
        for (int i = 0; i < L + 2; i++) {
            const ImgF &a = gp[o * (L + 3) + i], &b = gp[o * (L + 3) + i + 1];
            ImgF& d = dog[o * (L + 2) + i];
            d.h = a.h, d.w = a.w, d.px.resize(a.px.size());
            for (size_t p = 0; p < a.px.size(); p++) d.px[p] = b.px[p] - a.px[p];
        }
    const int threshold = (int)std::floor(0.5 * CONTRAST / L * 255);
    std::vector<Kp> kps;
    for (int o = 0; o < n_oct; o++)
        for (int i = 1; i <= L; i++) {
            const ImgF& img = dog[o * (L + 2) + i];
            const ImgF& prev = dog[o * (L + 2) + i - 1];
            const ImgF& next = dog[o * (L + 2) + i + 1];
            for (int r = BORDER; r < img.h - BORDER; r++)
                for (int c = BORDER; c < img.w - BORDER; c++) {
                    const float val = img.at(r, c);
                    if (!(std::fabs(val) > threshold)) continue;
                    bool mx = val > 0, mn = val < 0;
                    for (int dy = -1; dy <= 1 && (mx || mn); dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const float v[3] = {img.at(r + dy, c + dx), prev.at(r + dy, c + dx), next.at(r + dy, c + dx)};
                            for (int q = 0; q < 3; q++) {
                                mx &= val >= v[q];
                                mn &= val <= v[q];
                            }
                        }
                    if (!(mx || mn)) continue;
                    int r1 = r, c1 = c, layer = i;
                    Kp kp;
                    float xi;
                    if (!adjust_extremum(dog, o, layer, r1, c1, kp, xi)) continue;
                    const float scl_octv = kp.size * 0.5f / (float)(1 << o);
                    float hist[ORI_BINS];
                    const float omax = orientation_hist(gp[o * (L + 3) + layer], c1, r1, cv_round(4.5f * scl_octv),
                                                        1.5f * scl_octv, hist);
                    const float mag_thr = omax * 0.8f;
                    int peaks = 0;
                    for (int j = 0; j < ORI_BINS && peaks < MAX_PEAKS; j++) {
                        const int l = j > 0 ? j - 1 : ORI_BINS - 1, r2 = j < ORI_BINS - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? ORI_BINS + bin : bin >= ORI_BINS ? bin - ORI_BINS : bin;
                            kp.angle = 360.f - (360.f / ORI_BINS) * bin;
                            if (std::fabs(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
                            kps.push_back(kp);
                            peaks++;
                        }
                    }
                }
        }
    // KeyPointsFilter::removeDuplicatedSorted, retainBest(nfeatures), then back to the input image's scale
    std::stable_sort(kps.begin(), kps.end(), kp_less);
    std::vector<Kp> uniq;
    for (const Kp& q : kps)
        if (uniq.empty() || uniq.back().x != q.x || uniq.back().y != q.y || uniq.back().size != q.size ||
            uniq.back().angle != q.angle)
            uniq.push_back(q);
    if (nfeatures > 0 && (int)uniq.size() > nfeatures) {
        std::vector<float> resp;
        for (const Kp& q : uniq) resp.push_back(q.response);
        std::nth_element(resp.begin(), resp.begin() + (nfeatures - 1), resp.end(), std::greater<float>());
        const float cut = resp[nfeatures - 1];
        std::vector<Kp> kept;
        for (const Kp& q : uniq)
            if (q.response >= cut) kept.push_back(q);
        uniq.swap(kept);
    }
    const int cap = rtvm_oracle_sift_max_keypoints(nfeatures);
    const int n = std::min((int)uniq.size(), cap);
    *n_out = n;
#pragma omp parallel for schedule(dynamic, 8) num_threads(rtvm_oracle_nthreads)
    for (int q = 0; q < n; q++) {
        Kp kp = uniq[q];
        // descriptor on the pyramid image the keypoint was found in
        int oct = kp.octave & 255, layer = (kp.octave >> 8) & 255;
        const float scale_oct = 1.f / (float)(1 << oct);  // octave index counted from the doubled image
        float ang = 360.f - kp.angle;
        if (std::fabs(ang - 360.f) < FLT_EPSILON) ang = 0.f;
        descriptor(gp[oct * (L + 3) + layer], kp.x * scale_oct, kp.y * scale_oct, ang, kp.size * scale_oct * 0.5f,
                   desc + (size_t)q * 128);
        // firstOctave = -1: coordinates and size back to the input image, octave field re-based
        kp.octave = (kp.octave & ~255) | ((kp.octave - 1) & 255);
        xy[2 * q] = kp.x * 0.5f;
        xy[2 * q + 1] = kp.y * 0.5f;
        size[q] = kp.size * 0.5f;
        angle_deg_out[q] = kp.angle;
        response[q] = kp.response;
        octave[q] = kp.octave;
    }
    return 0;
}
