// CPU restatement of cv2.cvtColor(BGR2GRAY) and cv2.ORB_create(n).detectAndCompute
// (reference main.py:36,111-112,717-718) from OpenCV 4.x's published algorithm:
// 8-level pyramid (factor 1.2, INTER_LINEAR_EXACT), FAST-9/16 (threshold 20) with
// non-maximum suppression, 31-pixel border removal, retainBest by FAST score then
// by Harris response, intensity-centroid orientation, 7x7 sigma-2 Gaussian blur,
// 256-bit steered BRIEF.  Test infrastructure only (see oracle.h).  Parity unpinned.
//
// OpenCV leaves the keypoint order after KeyPointsFilter::retainBest
// implementation-defined (std::nth_element); here the survivors keep their
// detection order (level, then row-major), which is the order every
// implementation checked against this file must produce.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

const int NLEVELS = 8, EDGE = 31, PATCH = 31, HALF_PATCH = 15, FAST_T = 20;
const float HARRIS_K = 0.04f;

inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_roundf(float v) { return (int)std::nearbyintf(v); }
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

struct Img {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

// cv::resize(src, dst, INTER_LINEAR_EXACT) for 8UC1: 8.8 fixed-point coefficients
// from scale*(i+0.5)-0.5 evaluated in double, horizontal pass to 8.8 values,
// vertical pass to 16.16, rounded to nearest.
void lin_coeffs(int src, int dst, std::vector<int>& ofs, std::vector<int>& c1) {
    const double inv_scale = (double)dst / src;
    const double scale = 1.0 / inv_scale;
    ofs.resize(dst);
    c1.resize(dst);
    for (int v = 0; v < dst; v++) {
        const double f = scale * ((double)v + 0.5) - 0.5;
        const int i = (int)std::floor(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                ofs[v] = i;
                c1[v] = cv_round((f - (double)i) * 256.0);
            } else {
                ofs[v] = src - 1;  // clamped to the last sample
                c1[v] = -1;
            }
        } else {
            ofs[v] = 0;
            c1[v] = -1;
        }
    }
}

void resize_exact(const Img& s, Img& d, int dw, int dh) {
    d.w = dw;
    d.h = dh;
    d.px.assign((size_t)dw * dh, 0);
    std::vector<int> xo, xc, yo, yc;
    lin_coeffs(s.w, dw, xo, xc);
    lin_coeffs(s.h, dh, yo, yc);
    auto hline = [&](int sy, int x) -> unsigned {
        if (xc[x] < 0) return (unsigned)s.at(sy, xo[x]) << 8;
        return (unsigned)(256 - xc[x]) * s.at(sy, xo[x]) + (unsigned)xc[x] * s.at(sy, xo[x] + 1);
    };
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            unsigned v;
            if (yc[y] < 0)
                v = hline(yo[y], x) << 8;
            else
                v = (unsigned)(256 - yc[y]) * hline(yo[y], x) + (unsigned)yc[y] * hline(yo[y] + 1, x);
            v = (v + 32768u) >> 16;
            d.px[(size_t)y * dw + x] = (uint8_t)(v > 255 ? 255 : v);
        }
}

// cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on 8UC1 as it
// runs inside ORB (the level is a sub-matrix, so the separable-filter engine is
// used): float kernel -> 8-bit fixed point (cvRound(k*256)), exact integer row and
// column sums, result rounded to nearest even and saturated.
void blur7(const Img& s, Img& d) {
    double t[7], sum = 0;
    const double sigma = 2.0, scale2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < 7; i++) {
        const double x = i - 3.0;
        t[i] = std::exp(scale2 * x * x);
        sum += t[i];
    }
    sum = 1. / sum;
    int k[7];
    for (int i = 0; i < 7; i++) k[i] = cv_roundf((float)(t[i] * sum) * 256.f);
    d.w = s.w;
    d.h = s.h;
    d.px.assign(s.px.size(), 0);
    std::vector<int> rows((size_t)s.w * s.h);
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < s.h; y++)
        for (int x = 0; x < s.w; x++) {
            int a = 0;
            for (int i = 0; i < 7; i++) a += k[i] * s.at(y, reflect101(x + i - 3, s.w));
            rows[(size_t)y * s.w + x] = a;
        }
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 0; y < s.h; y++)
        for (int x = 0; x < s.w; x++) {
            unsigned a = 0;
            for (int i = 0; i < 7; i++) a += (unsigned)k[i] * (unsigned)rows[(size_t)reflect101(y + i - 3, s.h) * s.w + x];
            unsigned q = a >> 16, r = a & 0xFFFFu;
            if (r > 32768u || (r == 32768u && (q & 1))) q++;
            d.px[(size_t)y * s.w + x] = (uint8_t)(q > 255 ? 255 : q);
        }
}

struct KP {
    int x, y, level;
    float response, angle;
};

// cv::FAST(img, threshold=20, nonmaxSuppression=true), type 9_16: corners in
// row-major order with the cornerScore<16> response.
void fast9(const Img& im, std::vector<KP>& out, int level) {
    static const int offs[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    const int w = im.w, h = im.h;
    std::vector<uint8_t> score((size_t)w * h, 0);
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int v = im.at(y, x);
            int d[25];
            for (int k = 0; k < 25; k++) d[k] = v - im.at(y + offs[k & 15][1], x + offs[k & 15][0]);
            // corner test: 9 contiguous pixels all brighter than v+t or all darker than v-t
            bool corner = false;
            for (int sgn = 0; sgn < 2 && !corner; sgn++) {
                int run = 0;
                for (int k = 0; k < 25; k++) {
                    const bool on = sgn ? d[k] > FAST_T : d[k] < -FAST_T;
                    run = on ? run + 1 : 0;
                    if (run >= 9) {
                        corner = true;
                        break;
                    }
                }
            }
            if (!corner) continue;
            // cornerScore<16>: largest threshold for which the pixel is still a corner
            int a0 = FAST_T;
            for (int k = 0; k < 16; k += 2) {
                int a = std::min(d[k + 1], d[k + 2]);
                a = std::min(a, d[k + 3]);
                if (a <= a0) continue;
                a = std::min(a, d[k + 4]);
                a = std::min(a, d[k + 5]);
                a = std::min(a, d[k + 6]);
                a = std::min(a, d[k + 7]);
                a = std::min(a, d[k + 8]);
                a0 = std::max(a0, std::min(a, d[k]));
                a0 = std::max(a0, std::min(a, d[k + 9]));
            }
            int b0 = -a0;
            for (int k = 0; k < 16; k += 2) {
                int b = std::max(d[k + 1], d[k + 2]);
                b = std::max(b, d[k + 3]);
                b = std::max(b, d[k + 4]);
                b = std::max(b, d[k + 5]);
                if (b >= b0) continue;
                b = std::max(b, d[k + 6]);
                b = std::max(b, d[k + 7]);
                b = std::max(b, d[k + 8]);
                b0 = std::min(b0, std::max(b, d[k]));
                b0 = std::min(b0, std::max(b, d[k + 9]));
            }
            score[(size_t)y * w + x] = (uint8_t)(-b0 - 1);
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int s = score[(size_t)y * w + x];
            if (!s) continue;
            bool mx = true;
            for (int dy = -1; dy <= 1 && mx; dy++)
                for (int dx = -1; dx <= 1; dx++)
                    if ((dx || dy) && s <= score[(size_t)(y + dy) * w + x + dx]) {
                        mx = false;
                        break;
                    }
            if (mx) out.push_back({x, y, level, (float)s, -1.f});
        }
}

// KeyPointsFilter::retainBest: everything with response >= the n-th largest
// response (ties at the cut are all kept); detection order is preserved.
void retain_best(std::vector<KP>& kps, int n) {
    if (n < 0 || (int)kps.size() <= n) return;
    if (n == 0) {
        kps.clear();
        return;
    }
    std::vector<float> r;
    for (auto& k : kps) r.push_back(k.response);
    std::nth_element(r.begin(), r.begin() + n - 1, r.end(), std::greater<float>());
    const float cut = r[n - 1];
    std::vector<KP> keep;
    for (auto& k : kps)
        if (k.response >= cut) keep.push_back(k);
    kps.swap(keep);
}

float harris(const Img& im, int x0, int y0) {
    int a = 0, b = 0, c = 0;
    for (int i = -3; i <= 3; i++)
        for (int j = -3; j <= 3; j++) {
            const int y = y0 + i, x = x0 + j;
            const int Ix = (im.at(y, x + 1) - im.at(y, x - 1)) * 2 + (im.at(y - 1, x + 1) - im.at(y - 1, x - 1)) +
                           (im.at(y + 1, x + 1) - im.at(y + 1, x - 1));
            const int Iy = (im.at(y + 1, x) - im.at(y - 1, x)) * 2 + (im.at(y + 1, x - 1) - im.at(y - 1, x - 1)) +
                           (im.at(y + 1, x + 1) - im.at(y - 1, x + 1));
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * b - (float)c * c - HARRIS_K * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void make_umax(int* umax) {
    int v, v0, vmax = (int)std::floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HALF_PATCH * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

float ic_angle(const Img& im, int x0, int y0, const int* umax) {
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * im.at(y0, x0 + u);
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = im.at(y0 + v, x0 + u), val_minus = im.at(y0 - v, x0 + u);
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

int g_pattern[256 * 4];
bool g_pattern_ok = false;

bool load_pattern() {
    if (g_pattern_ok) return true;
    // the table lives next to the product sources as a data file
    const char* env = getenv("RTVM_ORB_PATTERN");
    std::string path = env ? env : "";
    if (path.empty()) return false;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[256];
    int n = 0;
    while (n < 256 && fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        if (sscanf(line, "%d %d %d %d", &g_pattern[4 * n], &g_pattern[4 * n + 1], &g_pattern[4 * n + 2], &g_pattern[4 * n + 3]) == 4) n++;
    }
    fclose(f);
    g_pattern_ok = n == 256;
    return g_pattern_ok;
}

void level_geometry(int w, int h, int* lw, int* lh, float* scale) {
    for (int l = 0; l < NLEVELS; l++) {
        scale[l] = (float)std::pow((double)1.2f, (double)l);
        const float inv = 1.0f / scale[l];
        lw[l] = cv_round(w * inv);
        lh[l] = cv_round(h * inv);
    }
}

void build_pyramid(const uint8_t* gray, int h, int w, std::vector<Img>& pyr, float* scale) {
    int lw[NLEVELS], lh[NLEVELS];
    level_geometry(w, h, lw, lh, scale);
    pyr.resize(NLEVELS);
    pyr[0].w = w;
    pyr[0].h = h;
    pyr[0].px.assign(gray, gray + (size_t)w * h);
    for (int l = 1; l < NLEVELS; l++) resize_exact(pyr[l - 1], pyr[l], lw[l], lh[l]);
}

}  // namespace

extern "C" void rtvm_oracle_set_orb_pattern(const int32_t* pattern1024) {
    for (int i = 0; i < 1024; i++) g_pattern[i] = pattern1024[i];
    g_pattern_ok = true;
}

// cvtColor(BGR2GRAY), 8-bit: (B*3735 + G*19235 + R*9798 + 16384) >> 15 (OpenCV 4.x), or with the
// 14-bit coefficients of OpenCV <= 3.x when rtvm_oracle_set_gray_legacy(1) was called:
// (B*1868 + G*9617 + R*4899 + 8192) >> 14.
int rtvm_oracle_gray_legacy = 0;
extern "C" int rtvm_oracle_set_gray_legacy(int on) {
    const int old = rtvm_oracle_gray_legacy;
    rtvm_oracle_gray_legacy = on ? 1 : 0;
    return old;
}
extern "C" void rtvm_oracle_gray(const uint8_t* bgr, int h, int w, uint8_t* gray) {
    const int cb = rtvm_oracle_gray_legacy ? 1868 : 3735, cg = rtvm_oracle_gray_legacy ? 9617 : 19235;
    const int cr = rtvm_oracle_gray_legacy ? 4899 : 9798, sh = rtvm_oracle_gray_legacy ? 14 : 15;
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (size_t p = 0; p < (size_t)h * w; p++)
        gray[p] = (uint8_t)((bgr[3 * p] * cb + bgr[3 * p + 1] * cg + bgr[3 * p + 2] * cr + (1 << (sh - 1))) >> sh);
}

extern "C" int rtvm_oracle_max_keypoints(int nfeatures) { return nfeatures + 64 * NLEVELS; }

extern "C" int rtvm_oracle_orb_pyramid(const uint8_t* gray, int h, int w, int* level_h, int* level_w, uint8_t* pyr_out,
                                       uint8_t* blur_out) {
    std::vector<Img> pyr;
    float scale[NLEVELS];
    build_pyramid(gray, h, w, pyr, scale);
    size_t off = 0;
    for (int l = 0; l < NLEVELS; l++) {
        level_h[l] = pyr[l].h;
        level_w[l] = pyr[l].w;
        if (pyr_out) memcpy(pyr_out + off, pyr[l].px.data(), pyr[l].px.size());
        if (blur_out) {
            Img b;
            blur7(pyr[l], b);
            memcpy(blur_out + off, b.px.data(), b.px.size());
        }
        off += pyr[l].px.size();
    }
    return 0;
}

extern "C" int rtvm_oracle_orb(const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy, float* size,
                               float* angle_deg, float* response, int32_t* octave, uint8_t* desc) {
    if (!g_pattern_ok && !load_pattern()) return -2;
    std::vector<uint8_t> gray((size_t)h * w);
    rtvm_oracle_gray(bgr, h, w, gray.data());
    std::vector<Img> pyr;
    float scale[NLEVELS];
    build_pyramid(gray.data(), h, w, pyr, scale);
    // features per level (computeKeyPoints)
    int quota[NLEVELS];
    {
        const float factor = (float)(1.0 / (double)1.2f);
        float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)NLEVELS));
        int sum = 0;
        for (int l = 0; l < NLEVELS - 1; l++) {
            quota[l] = cv_roundf(nd);
            sum += quota[l];
            nd *= factor;
        }
        quota[NLEVELS - 1] = std::max(nfeatures - sum, 0);
    }
    int umax[HALF_PATCH + 2];
    make_umax(umax);
    std::vector<KP> all;
    for (int l = 0; l < NLEVELS; l++) {
        const Img& im = pyr[l];
        std::vector<KP> kps;
        fast9(im, kps, l);
        // KeyPointsFilter::runByImageBorder(edgeThreshold)
        std::vector<KP> in;
        if (im.w > 2 * EDGE && im.h > 2 * EDGE)
            for (auto& k : kps)
                if (k.x >= EDGE && k.x < im.w - EDGE && k.y >= EDGE && k.y < im.h - EDGE) in.push_back(k);
        retain_best(in, 2 * quota[l]);
        for (auto& k : in) k.response = harris(im, k.x, k.y);
        retain_best(in, quota[l]);
        for (auto& k : in) k.angle = ic_angle(im, k.x, k.y, umax);
        all.insert(all.end(), in.begin(), in.end());
    }
    const int cap = rtvm_oracle_max_keypoints(nfeatures);
    if ((int)all.size() > cap) return -3;
    std::vector<Img> blur(NLEVELS);
    for (int l = 0; l < NLEVELS; l++) blur7(pyr[l], blur[l]);
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (size_t i = 0; i < all.size(); i++) {
        const KP& k = all[i];
        const float sf = scale[k.level];
        xy[2 * i] = (float)k.x * sf;
        xy[2 * i + 1] = (float)k.y * sf;
        size[i] = PATCH * sf;
        angle_deg[i] = k.angle;
        response[i] = k.response;
        octave[i] = k.level;
        // computeOrbDescriptors, WTA_K = 2
        float ang = k.angle;
        ang *= (float)(M_PI / 180.f);
        const float a = (float)std::cos(ang), b = (float)std::sin(ang);
        const float inv = 1.f / sf;
        const int cx = cv_roundf(xy[2 * i] * inv), cy = cv_roundf(xy[2 * i + 1] * inv);
        const Img& im = blur[k.level];
        for (int byte = 0; byte < 32; byte++) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const int* p = &g_pattern[(byte * 8 + bit) * 4];
                const float x1 = p[0] * a - p[1] * b, y1 = p[0] * b + p[1] * a;
                const float x2 = p[2] * a - p[3] * b, y2 = p[2] * b + p[3] * a;
                const int t0 = im.at(cy + cv_roundf(y1), cx + cv_roundf(x1));
                const int t1 = im.at(cy + cv_roundf(y2), cx + cv_roundf(x2));
                val |= (t0 < t1) << bit;
            }
            desc[32 * i + byte] = (uint8_t)val;
        }
    }
    *n_out = (int)all.size();
    return 0;
}
