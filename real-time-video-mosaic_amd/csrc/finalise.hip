// Mosaic finalisation, the step after the stitching path (reference main.py:980-1038, run
// once per video on the finished canvas): the bounding rectangle crop_black_areas cuts to,
// and scale_to_screen's cv2.resize(INTER_LINEAR).  Both work on a host image or directly on
// the context's device canvas.
#include "common.h"

#include <cmath>
#include <vector>

namespace {

// main.py:991-997: gray > threshold (cvtColor BGR2GRAY, 15-bit fixed point), bounding box of
// those pixels.  A wave is 64 pixels of a row; it folds its hits into {min x, min y, -max x,
// -max y} with four atomics (box[] starts at INT_MAX).
__global__ __launch_bounds__(256) void k_crop_rect(const uint8_t* __restrict__ img, int pitch_px, int x0, int y0, int w,
                                                   int h, int threshold, int* __restrict__ box) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool hit = false;
    if (x < w && y < h) {
        const uint8_t* p = img + ((size_t)(y0 + y) * pitch_px + x0 + x) * 3;
        const int g = (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
        hit = g > threshold;
    }
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) {
        const int bx = blockIdx.x * 64;
        atomicMin(&box[0], bx + __ffsll((long long)m) - 1);
        atomicMin(&box[1], y);
        atomicMin(&box[2], -(bx + 63 - __clzll((long long)m)));
        atomicMin(&box[3], -y);
    }
}

// cv2.resize(INTER_LINEAR) on 8UC3 in OpenCV's fixed-point arithmetic: 11-bit coefficients,
// horizontal pass in int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
// Offsets and coefficients come from the host (float/double arithmetic of the table set-up).
// One thread per output pixel.
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* __restrict__ img, int pitch_px, int x0, int y0,
                                                       int w, int h, const int* __restrict__ xofs,
                                                       const short* __restrict__ ia, const int* __restrict__ yofs,
                                                       const short* __restrict__ ib, uint8_t* __restrict__ out, int nw,
                                                       int nh) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= nw || dy >= nh) return;
    int r0 = yofs[dy], r1 = r0 + 1;
    r0 = r0 < 0 ? 0 : r0 >= h ? h - 1 : r0;
    r1 = r1 < 0 ? 0 : r1 >= h ? h - 1 : r1;
    const int b0 = ib[2 * dy], b1 = ib[2 * dy + 1];
    const int sx = xofs[dx], sx1 = sx + 1 < w ? sx + 1 : sx;
    const int a0 = ia[2 * dx], a1 = ia[2 * dx + 1];
    const uint8_t* p00 = img + ((size_t)(y0 + r0) * pitch_px + x0 + sx) * 3;
    const uint8_t* p01 = img + ((size_t)(y0 + r0) * pitch_px + x0 + sx1) * 3;
    const uint8_t* p10 = img + ((size_t)(y0 + r1) * pitch_px + x0 + sx) * 3;
    const uint8_t* p11 = img + ((size_t)(y0 + r1) * pitch_px + x0 + sx1) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int S0 = p00[c] * a0 + p01[c] * a1;
        const int S1 = p10[c] * a0 + p11[c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        out[((size_t)dy * nw + dx) * 3 + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

short sat_short_round(float v) {
    const long r = lrintf(v);
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

// Source offsets and 11-bit coefficient pairs of one axis (n source samples -> nn outputs);
// clamp_weights: OpenCV clamps the x offsets and zeroes the weight there, the rows are clipped
// when they are read instead.
void axis_tables(int n, int nn, bool clamp_weights, std::vector<int>& ofs, std::vector<short>& coef) {
    const double scale = 1. / ((double)nn / n);
    ofs.resize(nn);
    coef.resize(2 * (size_t)nn);
    for (int d = 0; d < nn; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (clamp_weights) {
            if (s < 0) f = 0, s = 0;
            if (s >= n - 1) f = 0, s = n - 1;
        }
        ofs[d] = s;
        coef[2 * d] = sat_short_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = sat_short_round(f * 2048.f);
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

int crop_rect_dev(rtvm_ctx* c, const uint8_t* d_img, int pitch_px, int x0, int y0, int w, int h, int threshold,
                  int rect[4], int* found) {
    DevBuf box;
    HIP_TRY(c, hipMalloc(&box.p, 4 * sizeof(int)));
    const int init[4] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};
    HIP_TRY(c, hipMemcpyAsync(box.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_crop_rect, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, d_img, pitch_px, x0, y0, w, h, threshold,
                (int*)box.p);
    HIP_TRY(c, hipGetLastError());
    int out[4];
    HIP_TRY(c, hipMemcpyAsync(out, box.p, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *found = out[0] != 0x7FFFFFFF;
    if (*found) {
        rect[0] = out[0];
        rect[1] = out[1];
        rect[2] = -out[2] - out[0] + 1;
        rect[3] = -out[3] - out[1] + 1;
    } else {
        rect[0] = rect[1] = rect[2] = rect[3] = 0;
    }
    return RTVM_OK;
}

int resize_dev(rtvm_ctx* c, const uint8_t* d_img, int pitch_px, int x0, int y0, int w, int h, uint8_t* out, int nh,
               int nw) {
    std::vector<int> xofs, yofs;
    std::vector<short> ia, ib;
    axis_tables(w, nw, true, xofs, ia);
    axis_tables(h, nh, false, yofs, ib);
    DevBuf dx, da, dy, db, dout;
    const size_t out_bytes = (size_t)nh * nw * 3;
    HIP_TRY(c, hipMalloc(&dx.p, xofs.size() * sizeof(int)));
    HIP_TRY(c, hipMalloc(&da.p, ia.size() * sizeof(short)));
    HIP_TRY(c, hipMalloc(&dy.p, yofs.size() * sizeof(int)));
    HIP_TRY(c, hipMalloc(&db.p, ib.size() * sizeof(short)));
    HIP_TRY(c, hipMalloc(&dout.p, out_bytes));
    HIP_TRY(c, hipMemcpyAsync(dx.p, xofs.data(), xofs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(da.p, ia.data(), ia.size() * sizeof(short), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dy.p, yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(db.p, ib.data(), ib.size() * sizeof(short), hipMemcpyHostToDevice, c->stream));
    RTVM_LAUNCH(c, k_resize_linear, dim3((nw + 63) / 64, (nh + 3) / 4), dim3(256), 0, d_img, pitch_px, x0, y0, w, h,
                (const int*)dx.p, (const short*)da.p, (const int*)dy.p, (const short*)db.p, (uint8_t*)dout.p, nw, nh);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the host tables and device buffers go out of scope
    return RTVM_OK;
}

int upload(rtvm_ctx* c, DevBuf& d, const uint8_t* bgr, int h, int w) {
    HIP_TRY(c, hipMalloc(&d.p, (size_t)h * w * 3));
    HIP_TRY(c, hipMemcpyAsync(d.p, bgr, (size_t)h * w * 3, hipMemcpyHostToDevice, c->stream));
    return RTVM_OK;
}

}  // namespace

extern "C" int rtvm_crop_rect(rtvm_ctx* c, const uint8_t* bgr, int h, int w, int threshold, int rect[4], int* found) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !rect || !found || h < 1 || w < 1) RTVM_FAIL(c, RTVM_E_ARG, "crop_rect: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf img;
    int r = upload(c, img, bgr, h, w);
    if (r) return r;
    return crop_rect_dev(c, (const uint8_t*)img.p, w, 0, 0, w, h, threshold, rect, found);
}

extern "C" int rtvm_canvas_crop_rect(rtvm_ctx* c, int threshold, int rect[4], int* found) {
    if (!c) return RTVM_E_ARG;
    if (!rect || !found) RTVM_FAIL(c, RTVM_E_ARG, "canvas_crop_rect: null argument");
    if (!c->canvas_ready) RTVM_FAIL(c, RTVM_E_STATE, "canvas_crop_rect: no canvas yet");
    HIP_TRY(c, hipSetDevice(c->device));
    return crop_rect_dev(c, c->d_canvas, c->cw, 0, 0, c->cw, c->ch, threshold, rect, found);
}

extern "C" int rtvm_resize_linear(rtvm_ctx* c, const uint8_t* bgr, int h, int w, uint8_t* out, int nh, int nw) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !out || h < 1 || w < 1 || nh < 1 || nw < 1) RTVM_FAIL(c, RTVM_E_ARG, "resize_linear: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf img;
    int r = upload(c, img, bgr, h, w);
    if (r) return r;
    return resize_dev(c, (const uint8_t*)img.p, w, 0, 0, w, h, out, nh, nw);
}

extern "C" int rtvm_canvas_read_scaled(rtvm_ctx* c, int x, int y, int w, int h, uint8_t* out, int nh, int nw) {
    if (!c) return RTVM_E_ARG;
    if (!out || nh < 1 || nw < 1) RTVM_FAIL(c, RTVM_E_ARG, "canvas_read_scaled: bad argument");
    if (!c->canvas_ready) RTVM_FAIL(c, RTVM_E_STATE, "canvas_read_scaled: no canvas yet");
    if (x < 0 || y < 0 || w < 1 || h < 1 || x + w > c->cw || y + h > c->ch)
        RTVM_FAIL(c, RTVM_E_ARG, "canvas_read_scaled: rectangle %d,%d %dx%d outside the %dx%d canvas", x, y, w, h, c->cw,
                  c->ch);
    HIP_TRY(c, hipSetDevice(c->device));
    return resize_dev(c, c->d_canvas, c->cw, x, y, w, h, out, nh, nw);
}
