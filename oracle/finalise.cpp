// CPU restatement of the mosaic finalisation the reference runs once per video on the
// finished canvas (main.py:980-1038): crop_black_areas' bounding rectangle and
// scale_to_screen's cv2.resize(INTER_LINEAR).  Test infrastructure only; parity with
// OpenCV itself is unpinned (see oracle.h).
#include <cmath>
#include <cstdint>
#include <vector>

#include "oracle.h"

// main.py:991-997: gray = cvtColor(BGR2GRAY); threshold(gray, t, 255, THRESH_BINARY) keeps
// gray > t; boundingRect(findNonZero(...)) = {min x, min y, max x - min x + 1, max y - min y + 1}.
// Returns 0 when no pixel passes (findNonZero gives None, main.py:995).
extern "C" int rtvm_oracle_crop_rect(const uint8_t* bgr, int h, int w, int threshold, int rect[4]) {
    std::vector<uint8_t> gray((size_t)h * w);
    rtvm_oracle_gray(bgr, h, w, gray.data());
    int x0 = w, y0 = h, x1 = -1, y1 = -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (gray[(size_t)y * w + x] > threshold) {
                x0 = x < x0 ? x : x0;
                x1 = x > x1 ? x : x1;
                y0 = y < y0 ? y : y0;
                y1 = y > y1 ? y : y1;
            }
    if (x1 < 0) return 0;
    rect[0] = x0;
    rect[1] = y0;
    rect[2] = x1 - x0 + 1;
    rect[3] = y1 - y0 + 1;
    return 1;
}

namespace {
inline int cv_floor(float v) { return (int)std::floor(v); }
inline short sat_short_round(float v) {
    const long r = std::lrintf(v);  // round half to even, as cvRound
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}
}  // namespace

// main.py:1037: cv2.resize(image, (nw, nh), interpolation=INTER_LINEAR) on 8UC3, OpenCV's
// fixed-point path: coefficients in 11 bits (saturate_cast<short>(c * 2048)), horizontal
// pass into int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
// Source x is clamped with its weight set to 0 at the borders, source rows are clipped.
extern "C" int rtvm_oracle_resize_linear(const uint8_t* bgr, int h, int w, uint8_t* out, int nh, int nw) {
    if (h < 1 || w < 1 || nh < 1 || nw < 1) return -1;
    const double inv_x = (double)nw / w, inv_y = (double)nh / h;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    std::vector<int> xofs(nw), yofs(nh);
    std::vector<short> ia(2 * (size_t)nw), ib(2 * (size_t)nh);
    for (int dx = 0; dx < nw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) fx = 0, sx = 0;
        if (sx >= w - 1) fx = 0, sx = w - 1;
        xofs[dx] = sx;
        ia[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        ia[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    for (int dy = 0; dy < nh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ib[2 * dy] = sat_short_round((1.f - fy) * 2048.f);
        ib[2 * dy + 1] = sat_short_round(fy * 2048.f);
    }
    for (int dy = 0; dy < nh; dy++) {
        int r0 = yofs[dy], r1 = yofs[dy] + 1;
        r0 = r0 < 0 ? 0 : r0 >= h ? h - 1 : r0;
        r1 = r1 < 0 ? 0 : r1 >= h ? h - 1 : r1;
        const int b0 = ib[2 * dy], b1 = ib[2 * dy + 1];
        for (int dx = 0; dx < nw; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < w ? sx + 1 : sx;
            const int a0 = ia[2 * dx], a1 = ia[2 * dx + 1];
            for (int c = 0; c < 3; c++) {
                const int S0 = bgr[((size_t)r0 * w + sx) * 3 + c] * a0 + bgr[((size_t)r0 * w + sx1) * 3 + c] * a1;
                const int S1 = bgr[((size_t)r1 * w + sx) * 3 + c] * a0 + bgr[((size_t)r1 * w + sx1) * 3 + c] * a1;
                const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                out[((size_t)dy * nw + dx) * 3 + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
    return 0;
}
