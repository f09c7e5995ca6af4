// CPU restatement of the classical per-pixel passes the reference runs on the finished mosaic
// before path planning (analyze_for_navigation, main.py:1051-1232, and the occupancy grid of
// find_path_astar, main.py:1344-1359): the OpenCV calls are restated from their published
// algorithms.  Test infrastructure only; parity with OpenCV itself is unpinned (see oracle.h).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// cv2.cvtColor(BGR2HSV) on 8-bit pixels: H in [0,180), 12-bit fixed-point division tables.
struct HsvTables {
    int sdiv[256], hdiv[256];
    HsvTables() {
        sdiv[0] = hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            sdiv[i] = (int)std::lrint((255 << 12) / (1. * i));
            hdiv[i] = (int)std::lrint((180 << 12) / (6. * i));
        }
    }
};
const HsvTables g_hsv;

inline void hsv_px(int b, int g, int r, uint8_t out[3]) {
    int v = std::max(b, std::max(g, r)), vmin = std::min(b, std::min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * g_hsv.sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * g_hsv.hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    out[0] = (uint8_t)(h < 0 ? 0 : h > 255 ? 255 : h);
    out[1] = (uint8_t)s;
    out[2] = (uint8_t)v;
}

inline bool in_range(const uint8_t p[3], int l0, int l1, int l2, int u0, int u1, int u2) {
    return p[0] >= l0 && p[0] <= u0 && p[1] >= l1 && p[1] <= u1 && p[2] >= l2 && p[2] <= u2;
}

// cv2.getStructuringElement: the span [lo, hi) of every row, relative to the anchor (the centre).
struct Element {
    int rows, ay;
    int lo[32], hi[32];
};

bool make_element(int shape, int k, Element& e) {
    if (k < 1 || k > 31) return false;
    e.rows = k;
    e.ay = k / 2;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0) {  // MORPH_RECT
            j2 = k;
        } else if (shape == 1) {  // MORPH_CROSS
            if (i == r) j2 = k;
            else j1 = c, j2 = c + 1;
        } else {  // MORPH_ELLIPSE
            const int dy = i - r;
            if (std::abs(dy) <= r) {
                const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
                j1 = std::max(c - dx, 0);
                j2 = std::min(c + dx + 1, k);
            }
        }
        e.lo[i] = j1 - c;
        e.hi[i] = j2 - c;
    }
    return true;
}

// cv2.erode / cv2.dilate of a 0/255 mask, one iteration, default border (pixels outside the
// image never decide the result).
void morph_once(const std::vector<uint8_t>& src, std::vector<uint8_t>& dst, int h, int w, const Element& e, bool dilate) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            bool any = false, all = true;
            for (int i = 0; i < e.rows; i++) {
                const int yy = y + i - e.ay;
                if (yy < 0 || yy >= h) continue;
                for (int dx = e.lo[i]; dx < e.hi[i]; dx++) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= w) continue;
                    const bool s = src[(size_t)yy * w + xx] != 0;
                    any |= s;
                    all &= s;
                }
            }
            dst[(size_t)y * w + x] = (dilate ? any : all) ? 255 : 0;
        }
}

// op: 0 erode, 1 dilate, 2 open, 3 close (cv2.MORPH_*); cv2.morphologyEx runs all the
// iterations of the first operation before the second one's.
bool morphology(std::vector<uint8_t>& m, int h, int w, int op, int shape, int k, int iterations) {
    Element e;
    if (!make_element(shape, k, e) || op < 0 || op > 3) return false;
    std::vector<uint8_t> t(m.size());
    const bool first = op == 1 || op == 3;  // dilate first for dilate / close
    for (int pass = 0; pass < (op >= 2 ? 2 : 1); pass++)
        for (int it = 0; it < iterations; it++) {
            morph_once(m, t, h, w, e, pass == 0 ? first : !first);
            m.swap(t);
        }
    return true;
}

// cv2.filter2D(f32, -1, ones((11,11), f32)/121): correlation in row-major kernel order, one
// fused multiply-add per tap (the vector filter of OpenCV's AVX2 build), BORDER_REFLECT_101.
__attribute__((target_clones("fma", "default"))) void box11_fma(const std::vector<float>& src, std::vector<float>& dst,
                                                                int h, int w) {
    const float kf = 1.0f / 121.0f;
    std::vector<int> xs((size_t)w + 10);
    for (int x = -5; x < w + 5; x++) xs[x + 5] = reflect101(x, w);
    for (int y = 0; y < h; y++) {
        const float* rows[11];
        for (int i = 0; i < 11; i++) rows[i] = &src[(size_t)reflect101(y + i - 5, h) * w];
        for (int x = 0; x < w; x++) {
            float s = 0.f;
            for (int i = 0; i < 11; i++)
                for (int j = 0; j < 11; j++) s = __builtin_fmaf(kf, rows[i][xs[x + j]], s);
            dst[(size_t)y * w + x] = s;
        }
    }
}

// cv2.GaussianBlur(u8, (11,11), 0): sigma = 0.3*((11-1)*0.5-1)+0.8 = 2, kernel in 8.8 fixed
// point with the rounding error carried from tap to tap and the centre taking what is left of
// 256; rows then columns, the column sum rounded from 16 fractional bits.
void gauss_kernel_fixed(int n, int* k) {
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
        const int v = (int)std::lrint(adj);
        err = adj - v;
        k[i] = k[n - 1 - i] = v;
        acc += v;
    }
    k[n / 2] = 256 - 2 * acc;
}

void gauss11_u8(const std::vector<uint8_t>& src, std::vector<uint8_t>& dst, int h, int w) {
    int k[11];
    gauss_kernel_fixed(11, k);
    std::vector<uint16_t> rows((size_t)h * w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = 0; j < 11; j++) s += k[j] * src[(size_t)y * w + reflect101(x + j - 5, w)];
            rows[(size_t)y * w + x] = (uint16_t)s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = 0; i < 11; i++) s += (uint32_t)k[i] * rows[(size_t)reflect101(y + i - 5, h) * w + x];
            const uint32_t v = (s + (1u << 15)) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
        }
}

}  // namespace

extern "C" int rtvm_oracle_bgr2hsv(const uint8_t* bgr, int h, int w, uint8_t* hsv) {
    for (size_t i = 0; i < (size_t)h * w; i++) hsv_px(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], hsv + 3 * i);
    return 0;
}

extern "C" int rtvm_oracle_gauss11_u8(const uint8_t* src, int h, int w, uint8_t* dst) {
    std::vector<uint8_t> in(src, src + (size_t)h * w), out((size_t)h * w);
    gauss11_u8(in, out, h, w);
    std::memcpy(dst, out.data(), out.size());
    return 0;
}

extern "C" int rtvm_oracle_morphology(const uint8_t* mask, int h, int w, int op, int shape, int ksize, int iterations,
                                      uint8_t* out) {
    if (op < 0 || op > 3 || shape < 0 || shape > 2 || ksize < 1 || ksize > 31 || iterations < 0) return -1;
    std::vector<uint8_t> m((size_t)h * w);
    for (size_t i = 0; i < m.size(); i++) m[i] = mask[i] ? 255 : 0;
    if (iterations > 0 && ksize > 1 && !morphology(m, h, w, op, shape, ksize, iterations)) return -1;
    std::memcpy(out, m.data(), m.size());
    return 0;
}

extern "C" int rtvm_oracle_nav_obstacles(const uint8_t* bgr, int h, int w, const int32_t* boxes, int n_boxes,
                                         uint8_t* mask_fire, uint8_t* mask_smoke, uint8_t* texture_mask,
                                         uint8_t* obstacles, uint8_t* obstacles_nav) {
    if (h < 1 || w < 1 || n_boxes < 0) return -1;
    const size_t n = (size_t)h * w;
    std::vector<uint8_t> gray(n), fire(n), smoke(n), valid(n), tex(n), obs(n, 0);
    rtvm_oracle_gray(bgr, h, w, gray.data());  // main.py:1112

    // main.py:1079-1107: filled rectangles, both corners included, clipped to the image
    for (int k = 0; k < n_boxes; k++) {
        const int x1 = std::min(boxes[4 * k], boxes[4 * k + 2]), x2 = std::max(boxes[4 * k], boxes[4 * k + 2]);
        const int y1 = std::min(boxes[4 * k + 1], boxes[4 * k + 3]), y2 = std::max(boxes[4 * k + 1], boxes[4 * k + 3]);
        for (int y = std::max(y1, 0); y <= std::min(y2, h - 1); y++)
            for (int x = std::max(x1, 0); x <= std::min(x2, w - 1); x++) obs[(size_t)y * w + x] = 255;
    }

    // main.py:1116-1138 fire and 1155-1174 smoke: HSV ranges and BGR tests.  `g + 25` and its
    // like are uint8 sums in numpy and wrap at 256; Canny's edges (main.py:1141-1144) enter as
    // mask | (edges & mask), which is the mask itself.
    for (size_t i = 0; i < n; i++) {
        const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        uint8_t p[3];
        hsv_px(b, g, r, p);
        const bool f_hsv = in_range(p, 0, 80, 120, 15, 255, 255) || in_range(p, 160, 80, 120, 180, 255, 255) ||
                           in_range(p, 15, 70, 130, 35, 255, 255);
        const bool f1 = r > ((g + 25) & 255) && r > ((b + 35) & 255) && r > 130;
        const bool f2 = r > 130 && g > 110 && r > ((b + 25) & 255) && g > ((b + 15) & 255) && std::abs(r - g) < 60;
        fire[i] = f_hsv && (f1 || f2) ? 255 : 0;
        const bool s_hsv = in_range(p, 0, 0, 90, 180, 45, 245) || in_range(p, 0, 0, 60, 180, 55, 140);
        const bool s_bgr = std::abs(r - g) < 25 && std::abs(g - b) < 25 && r > 60 && g > 60 && b > 60;
        smoke[i] = s_hsv && s_bgr && gray[i] > 70 ? 255 : 0;
        valid[i] = gray[i] > 15 ? 255 : 0;  // main.py:1196
    }
    // main.py:1147-1150
    morphology(fire, h, w, 3, 2, 7, 2);
    morphology(fire, h, w, 2, 2, 3, 1);

    // main.py:1178-1185: local standard deviation over 11x11 below 40
    {
        std::vector<float> gf(n), gf2(n), mean(n), sq(n);
        for (size_t i = 0; i < n; i++) gf[i] = (float)gray[i], gf2[i] = gf[i] * gf[i];
        box11_fma(gf, mean, h, w);
        box11_fma(gf2, sq, h, w);
        for (size_t i = 0; i < n; i++) {
            const float m2 = mean[i] * mean[i];
            float d = sq[i] - m2;
            d = d > 0.f ? d : 0.f;
            if (!(std::sqrt(d) < 40.f)) smoke[i] = 0;
        }
    }
    // main.py:1188-1190
    morphology(smoke, h, w, 3, 2, 11, 3);
    morphology(smoke, h, w, 2, 2, 3, 1);

    // main.py:1196-1210 texture: |gray - GaussianBlur(gray, 11x11)| > 6 inside the eroded valid area
    morphology(valid, h, w, 0, 0, 10, 1);
    {
        std::vector<uint8_t> blurred(n);
        gauss11_u8(gray, blurred, h, w);
        for (size_t i = 0; i < n; i++) tex[i] = std::abs((int)gray[i] - (int)blurred[i]) > 6 && valid[i] ? 255 : 0;
    }
    morphology(tex, h, w, 3, 2, 5, 1);
    morphology(tex, h, w, 2, 2, 3, 1);

    // main.py:1151,1191,1228,1231-1232
    for (size_t i = 0; i < n; i++) obs[i] = obs[i] | fire[i] | smoke[i] | tex[i];
    std::vector<uint8_t> nav = obs;
    morphology(nav, h, w, 1, 2, 15, 1);

    if (mask_fire) std::memcpy(mask_fire, fire.data(), n);
    if (mask_smoke) std::memcpy(mask_smoke, smoke.data(), n);
    if (texture_mask) std::memcpy(texture_mask, tex.data(), n);
    if (obstacles) std::memcpy(obstacles, obs.data(), n);
    if (obstacles_nav) std::memcpy(obstacles_nav, nav.data(), n);
    return 0;
}

// main.py:1346-1359: cell (gy, gx) of the h/scale x w/scale grid is blocked when more than
// 30 % of scale*scale pixels of its block are set.
// cv2.Canny(gray, low, high) with the defaults the reference uses (main.py:489,1141: aperture 3,
// L1 gradient).  [OpenCV-recall]  Sobel 3x3 in 16-bit with replicated borders; magnitude
// |dx| + |dy| with a ring of zeros around the image; a pixel above `low` is a candidate when it is a
// maximum along its gradient direction (sectors cut at tan 22.5 and tan 67.5 in 15-bit fixed point;
// `>` towards the left / upper neighbour, `>=` towards the right / lower one on the axes, `>` on both
// diagonals); candidates above `high` are edges, and so is every candidate connected to one through
// candidates (8 neighbours).  OpenCV's stack order and its "neighbour already pushed" shortcuts do not
// change that set.
extern "C" int rtvm_oracle_canny(const uint8_t* gray, int h, int w, double low_thresh, double high_thresh,
                                 uint8_t* edges) {
    if (h < 1 || w < 1) return -1;
    if (low_thresh > high_thresh) std::swap(low_thresh, high_thresh);
    const int low = (int)std::floor(low_thresh), high = (int)std::floor(high_thresh);
    auto px = [&](int y, int x) -> int {
        y = std::min(std::max(y, 0), h - 1);
        x = std::min(std::max(x, 0), w - 1);
        return gray[(size_t)y * w + x];
    };
    std::vector<int16_t> dx((size_t)h * w), dy((size_t)h * w);
    const int mw = w + 2;
    std::vector<int> mag((size_t)(h + 2) * mw, 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int gx = (px(y - 1, x + 1) + 2 * px(y, x + 1) + px(y + 1, x + 1)) -
                           (px(y - 1, x - 1) + 2 * px(y, x - 1) + px(y + 1, x - 1));
            const int gy = (px(y + 1, x - 1) + 2 * px(y + 1, x) + px(y + 1, x + 1)) -
                           (px(y - 1, x - 1) + 2 * px(y - 1, x) + px(y - 1, x + 1));
            dx[(size_t)y * w + x] = (int16_t)gx;
            dy[(size_t)y * w + x] = (int16_t)gy;
            mag[(size_t)(y + 1) * mw + x + 1] = std::abs(gx) + std::abs(gy);
        }
    const int TG22 = 13573;  // (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    std::vector<uint8_t> map((size_t)h * w, 1);  // 0 candidate, 1 no edge, 2 edge
    std::vector<int> stack;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* m0 = &mag[(size_t)(y + 1) * mw + x + 1];
            const int m = m0[0];
            if (m <= low) continue;
            const int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
            const int ax = std::abs(xs), ay = std::abs(ys) << 15;
            const int tg22x = ax * TG22;
            bool keep;
            if (ay < tg22x) {
                keep = m > m0[-1] && m >= m0[1];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    keep = m > m0[-mw] && m >= m0[mw];
                } else {
                    const int sgn = (xs ^ ys) < 0 ? -1 : 1;
                    keep = m > m0[-mw - sgn] && m > m0[mw + sgn];
                }
            }
            if (!keep) continue;
            if (m > high) {
                map[(size_t)y * w + x] = 2;
                stack.push_back(y * w + x);
            } else {
                map[(size_t)y * w + x] = 0;
            }
        }
    while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int y = p / w, x = p % w;
        for (int j = -1; j <= 1; j++)
            for (int i = -1; i <= 1; i++) {
                const int yy = y + j, xx = x + i;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                uint8_t& q = map[(size_t)yy * w + xx];
                if (q == 0) {
                    q = 2;
                    stack.push_back(yy * w + xx);
                }
            }
    }
    for (size_t i = 0; i < map.size(); i++) edges[i] = map[i] == 2 ? 255 : 0;
    return 0;
}

extern "C" int rtvm_oracle_occupancy_grid(const uint8_t* mask, int h, int w, int scale, uint8_t* grid) {
    if (h < 1 || w < 1 || scale < 1) return -1;
    const int gh = std::max(1, h / scale), gw = std::max(1, w / scale);
    for (int gy = 0; gy < gh; gy++)
        for (int gx = 0; gx < gw; gx++) {
            int count = 0;
            for (int y = gy * scale; y < std::min(h, gy * scale + scale); y++)
                for (int x = gx * scale; x < std::min(w, gx * scale + scale); x++) count += mask[(size_t)y * w + x] > 0;
            grid[(size_t)gy * gw + gx] = count > scale * scale * 0.3 ? 1 : 0;
        }
    return 0;
}
