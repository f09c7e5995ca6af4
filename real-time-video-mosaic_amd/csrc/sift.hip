// cv2.SIFT_create(n).detectAndCompute(gray, None) for gfx950 (reference main.py:33, 111-112,
// 717-718; OpenCV defaults nOctaveLayers = 3, contrastThreshold = 0.04, edgeThreshold = 10,
// sigma = 1.6).  Every kernel carries the frame of a batch in its grid (blockIdx.z, or .y for
// the per-keypoint kernels), so the stage entry point rtvm_sift (one host frame) and the
// device-resident stream path (rtvm_launch_sift: the frames of a batch) share them.
//
//   k_sift_base      gray (15-bit fixed point) as float, doubled with the arithmetic of
//                    cv::resize(INTER_LINEAR) on floats
//   k_sift_blur_rows / k_sift_blur_cols
//                    cv::GaussianBlur(Size(), sigma) on floats: row pass s = k0*x0,
//                    fma(x_i, k_i, s); column pass centre then fma(r[+i] + r[-i], k_i, s); the
//                    differences of Gaussians are not stored (see SDog)
//   k_sift_decimate  next octave = every second pixel of layer 3
//   k_sift_extrema   26-neighbour extrema above the contrast floor -> candidate keys (every
//                    octave and layer in one launch)
//   k_sift_refine    one wave per candidate: adjustLocalExtrema, orientation histogram, peaks;
//                    keypoints appended to a list in whatever order the waves finish
//   k_sift_rank / k_sift_mark / k_sift_compact
//                    KeyPointsFilter::removeDuplicatedSorted and retainBest(nfeatures) without a
//                    sort: every keypoint counts the keypoints that precede it in the sorted
//                    order (a total order: equal records are identical), the duplicates are
//                    marked, every unique keypoint counts the unique ones with a larger
//                    response (kept iff fewer than nfeatures), and the kept ones are compacted
//                    in sorted order
//   k_sift_describe  one wave per keypoint: calcSIFTDescriptor (samples in parallel, sums in order)
// Sums that OpenCV accumulates sequentially (orientation and descriptor histograms) are
// accumulated sequentially by one thread here as well; exp, atan2, cos, sin and pow are
// evaluated in double precision and rounded to float.
#include "common.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <new>
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
};

// Difference-of-Gaussians layer d of an octave = Gaussian layer d + 1 minus layer d.  The
// differences are not stored: whoever needs one subtracts the two Gaussian values (the same float
// subtraction a stored plane would hold), which saves writing and reading five planes per octave.
struct SDog {
    const float* hi;
    const float* lo;
    int h, w;
};
__device__ __forceinline__ SDog dog_layer(const SPyr* __restrict__ P, int octave, int d) {
    const SImg a = P->gp[octave * (SL + 3) + d + 1], b = P->gp[octave * (SL + 3) + d];
    return SDog{a.px, b.px, a.h, a.w};
}
#define DAT(dg, y, x) ((dg).hi[(size_t)(y) * (dg).w + (x)] - (dg).lo[(size_t)(y) * (dg).w + (x)])

// Everything one frame of a batch needs, in device memory; the kernels index an array of these
// with the frame's position in the batch.  All frames of a batch have the same geometry.
struct SiftFrame {
    SPyr pyr;                   // planes of the frame's Gaussian pyramid
    float* scratch;             // row-pass plane (size of the doubled image)
    const uint8_t* bgr;         // frame pixels
    unsigned long long* keys;   // scale-space extrema
    int* counts;                // [0] extrema found, [1] keypoints listed, [2] keypoints kept, [3] overflow bits
    SKp* list;                  // keypoints as the refinement appends them
    SKp* sorted;                // the same in removeDuplicatedSorted order
    int* gt;                    // per sorted keypoint: -1 = duplicate, else unique keypoints with a larger response
    SKp* kept;                  // the filtered keypoints, coordinates of the doubled image
    float* xy;                  // outputs: n x 2, level-0 pixel coordinates
    float* size;
    float* angle;
    float* response;
    int32_t* octave;
    float* desc;                // n x 128
    int* n_out;
};

// plane ids: Gaussian layer g (octave * (SL + 3) + layer), or the row-pass plane of the two-pass blur
constexpr int PLANE_SCRATCH = 1024;
__device__ __forceinline__ float* plane(const SiftFrame& f, int id) {
    return id == PLANE_SCRATCH ? f.scratch : f.pyr.gp[id].px;
}

__device__ __forceinline__ int reflect101s(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Gray (main.py:717) and the 2x INTER_LINEAR upsampling that SIFT starts from.  Output x samples the
// source at (x + 0.5) / 2 - 0.5: column 2i lies between source columns i - 1 and i with weights
// 0.25 / 0.75, column 2i + 1 between i and i + 1 with 0.75 / 0.25; at the image border the weight
// of the one source pixel is 1 (cv::resize replicates the border).  A thread takes source pixel
// (i, j): the gray values of its 3 x 3 neighbourhood (coordinates clamped) give the 2 x 2 outputs;
// every output is g(y0, x0) * a0 + g(y0, x1) * a1 per row, then row0 * b0 + row1 * b1.
__global__ __launch_bounds__(256) void k_sift_base(const SiftFrame* __restrict__ F, int h, int w, int dst_id, GrayCoef gc) {
    const SiftFrame& f = F[blockIdx.z];
    const auto bgr = as_global<const uint8_t>(f.bgr);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) f.counts[threadIdx.x] = 0;  // nothing found yet
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= w || j >= h) return;
    float G[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int yy = min(max(j - 1 + a, 0), h - 1);
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int xx = min(max(i - 1 + b, 0), w - 1);
            // offsets in 32 bits and 24-bit multiplies (full rate): rows, widths and coefficients are below 2^24
            const auto p = bgr + (__umul24((uint32_t)yy, (uint32_t)w) + (uint32_t)xx) * 3u;
            G[a][b] = (float)((__umul24(p[0], (uint32_t)gc.cb) + __umul24(p[1], (uint32_t)gc.cg) + __umul24(p[2], (uint32_t)gc.cr) +
                               (1u << (gc.shift - 1))) >> gc.shift);
        }
    }
    // weights of the left / upper source pixel (a0, b0) and of the other one (a1 = fx, a0 = 1 - fx)
    const float ax1[2] = {i == 0 ? 0.f : 0.75f, i == w - 1 ? 0.f : 0.25f};
    const float by1[2] = {j == 0 ? 0.f : 0.75f, j == h - 1 ? 0.f : 0.25f};
    const auto out = as_global(plane(f, dst_id) + (size_t)(2 * j) * (2 * w) + 2 * i);
#pragma unroll
    for (int py = 0; py < 2; py++) {
        const float b1 = by1[py], b0 = 1.f - b1;
        float o[2];
#pragma unroll
        for (int px = 0; px < 2; px++) {
            const float a1 = ax1[px], a0 = 1.f - a1;
            const float r0 = G[py][px] * a0 + G[py][px + 1] * a1;
            const float r1 = G[py + 1][px] * a0 + G[py + 1][px + 1] * a1;
            o[px] = r0 * b0 + r1 * b1;
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<gptr<f32x2>>(out + (size_t)py * (2 * w)) = f32x2{o[0], o[1]};
    }
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

// The general form of the two passes, one output per thread, for any kernel length (OpenCV's
// GaussianBlur as sepFilter2D: all rows first, then all columns); the lengths SIFT actually uses
// go through k_sift_blur below.
__global__ __launch_bounds__(256) void k_sift_blur_rows(const SiftFrame* __restrict__ F, int src_id, int dst_id, int h, int w,
                                                        Taps t) {
    const SiftFrame& f = F[blockIdx.z];
    const float* __restrict__ src = plane(f, src_id);
    float* __restrict__ dst = plane(f, dst_id);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int half = t.n / 2;
    const float* __restrict__ row = src + (size_t)y * w;
    dst[(size_t)y * w + x] = (x >= half && x + half < w) ? row_taps(t, row + (x - half)) : row_taps_border(t, row, x, w);
}

// Column pass.
__global__ __launch_bounds__(256) void k_sift_blur_cols(const SiftFrame* __restrict__ F, int src_id, int dst_id, int h, int w,
                                                        Taps t) {
    const SiftFrame& f = F[blockIdx.z];
    const float* __restrict__ src = plane(f, src_id);
    float* __restrict__ dst = plane(f, dst_id);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int half = t.n / 2;
    dst[(size_t)y * w + x] =
        (y >= half && y + half < h) ? col_taps(t, src + (size_t)y * w + x, w) : col_taps_border(t, src + x, y, h, w);
}

// 16-byte LDS reads by hand.  lds_offset: the LDS address of a __shared__ object (the low half of
// its generic address); lds_read16: ds_read_b128 at addr + OFF; lds_arrived<LATER>: its result is
// valid once all but the LATER reads issued after it have returned (LDS answers a wave in order;
// the compiler does not track reads it did not issue itself, so the registers are an in/out operand
// of the wait and no use of them can move above it).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_offset(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ f32x4 lds_read16(uint32_t addr, int off) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
    return v;
}
template <int LATER>
__device__ __forceinline__ void lds_arrived(f32x4& u) {
    static_assert(LATER >= 0 && LATER <= 15, "the counter has four bits");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(u) : "n"(LATER) : "memory");
}
// u[0..K) were read in this order and nothing else since
template <int K, int I = 0>
__device__ __forceinline__ void lds_arrived_all(f32x4 (&u)[K]) {
    if constexpr (I < K) {
        lds_arrived<(K - 1 - I < 15 ? K - 1 - I : 15)>(u[I]);
        lds_arrived_all<K, I + 1>(u);
    }
}

// Both passes of one blur for a tile of SBT_W x SBT_H outputs, compiled per kernel length N (11,
// 13, 17, 21, 27 for sigma = 1.6 and three layers per octave): the row-passed values of the tile's
// rows and of the N - 1 rows around them stay in LDS, so the plane between the passes never
// reaches memory and the source is read once (plus the tiles' margins, from the caches).
//   row pass: a wave takes four source rows at a time; their samples (the tile's columns plus half
//     a kernel on either side, mirrored at the image border) are staged in LDS by consecutive lanes
//     loading consecutive floats; 16 lanes per row then compute 8 outputs each from a window of
//     8 + N - 1 samples read as 16-byte words; every output is its own chain of fused
//     multiply-adds in tap order;
//   column pass: a thread takes 16 consecutive rows of one column from a window of 16 + N - 1
//     row-passed values in registers (centre first, then the symmetric pairs outwards) and writes
//     the layer.
constexpr int SBT_W = 128, SBT_H = 32, SB_R = 8, SB_C = 16, SBT_STAGE = SBT_W + 32;
// The row pass reads its windows as 16-byte words at a lane stride of 32 bytes, four rows of 16 lanes
// per wave.  A 16-byte read is served in four groups of 16 lanes that mix two rows (lanes 0-3, 12-15
// and 20-27 first): with rows 160 floats apart the lanes of a group meet on the same banks.  A row
// stride of 164 floats (36 banks on) puts the second row's lanes on the banks the first leaves idle.
constexpr int SBT_STAGE_PAD = SBT_STAGE + 4;
__device__ __forceinline__ int sb_pad(int e) { return e; }
// The row-passed tile is written as 16-byte words at the same lane stride (eight consecutive lanes per
// group, banks counted modulo 32): columns with bit 5 set keep their two 16-byte words in swapped
// order (column c lives at c ^ 4 there), which costs no space.
__device__ __forceinline__ int rp_col(int c) { return c ^ ((c >> 3) & 4); }

template <int N>
__global__ __launch_bounds__(256) void k_sift_blur(const SiftFrame* __restrict__ F, int src_id, int dst_id, int h, int w, Taps t) {
    constexpr int half = N / 2, RB = SBT_H + N - 1, NIN = SBT_W + N - 1, WIN4 = (SB_R + N - 1 + 3) / 4;
    static_assert(N % 2 == 1 && NIN <= SBT_STAGE && (SBT_W - SB_R) + 4 * WIN4 <= SBT_STAGE, "windows fit the staged row");
    static_assert(SBT_W == 16 * SB_R && SBT_H == 2 * SB_C, "thread mapping of the two passes");
    __shared__ __align__(16) float s_in[4][4][SBT_STAGE_PAD];
    __shared__ __align__(16) float s_rp[RB][SBT_W];
    const SiftFrame& f = F[blockIdx.z];
    const auto src = as_global(plane(f, src_id));
    const int tiles_x = (w + SBT_W - 1) / SBT_W;
    const int tile = xcd_tile(blockIdx.x, tiles_x * ((h + SBT_H - 1) / SBT_H));
    if (tile < 0) return;
    const int x0 = (tile % tiles_x) * SBT_W, y0 = (tile / tiles_x) * SBT_H;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool inner = x0 >= half && x0 + SBT_W + half <= w;
    const int q = lane >> 4, seg = lane & 15;
    // the samples of the next four rows are requested before the current four are filtered
    constexpr int NK = (NIN + 63) / 64;
    float pre[4][NK];
    auto request = [&](int rb) {
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const int r = min(rb + qq, RB - 1);
            const auto row = src + (size_t)reflect101s(y0 - half + r, h) * w;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int i = min(lane + 64 * k, NIN - 1);
                pre[qq][k] = row[inner ? x0 - half + i : reflect101s(x0 - half + i, w)];
            }
        }
    };
    request(4 * wv);
    for (int rb = 4 * wv; rb < RB; rb += 16) {
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int i = lane + 64 * k;
                if (i < SBT_STAGE) s_in[wv][qq][sb_pad(i)] = pre[qq][k];
            }
        }
        if (rb + 16 < RB) request(rb + 16);
        // the rows staged by this wave are read by this wave only: LDS keeps a wave's accesses in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int r = rb + q;
        if (r < RB) {
            // the window as WIN4 16-byte reads, spelled out: left to itself the compiler cuts the window
            // into some forty 4- and 8-byte reads (half the LDS rate, banks counted modulo 32)
            f32x4 u[WIN4];
            const uint32_t wa = lds_offset(&s_in[wv][q][sb_pad(seg * SB_R)]);
#pragma unroll
            for (int i = 0; i < WIN4; i++) u[i] = lds_read16(wa, 16 * i);
            lds_arrived_all(u);
            float v[4 * WIN4];
#pragma unroll
            for (int i = 0; i < WIN4; i++) v[4 * i] = u[i].x, v[4 * i + 1] = u[i].y, v[4 * i + 2] = u[i].z, v[4 * i + 3] = u[i].w;
            float a[SB_R];
#pragma unroll
            for (int o = 0; o < SB_R; o++) a[o] = t.k[0] * v[o];
#pragma unroll
            for (int k = 1; k < N; k++) {
#pragma unroll
                for (int o = 0; o < SB_R; o++) a[o] = __builtin_fmaf(v[o + k], t.k[k], a[o]);
            }
            *reinterpret_cast<float4*>(&s_rp[r][rp_col(seg * SB_R)]) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(&s_rp[r][rp_col(seg * SB_R + 4)]) = make_float4(a[4], a[5], a[6], a[7]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int cx = threadIdx.x & (SBT_W - 1), ry = (threadIdx.x / SBT_W) * SB_C;
    const int x = x0 + cx, yo = y0 + ry;
    if (x >= w || yo >= h) return;
    float v[SB_C + N - 1];
#pragma unroll
    for (int i = 0; i < SB_C + N - 1; i++) v[i] = s_rp[ry + i][rp_col(cx)];
    float a[SB_C];
#pragma unroll
    for (int o = 0; o < SB_C; o++) a[o] = t.c[0] * v[o + half];
#pragma unroll
    for (int k = 1; k <= half; k++) {
#pragma unroll
        for (int o = 0; o < SB_C; o++) a[o] = __builtin_fmaf(v[o + half + k] + v[o + half - k], t.c[k], a[o]);
    }
    const auto od = as_global(plane(f, dst_id) + (size_t)yo * w + x);
#pragma unroll
    for (int o = 0; o < SB_C; o++)
        if (yo + o < h) od[(size_t)o * w] = a[o];
}

__global__ __launch_bounds__(256) void k_sift_decimate(const SiftFrame* __restrict__ F, int src_id, int sw, int dst_id, int h,
                                                       int w) {
    const SiftFrame& f = F[blockIdx.z];
    const float* __restrict__ src = plane(f, src_id);
    float* __restrict__ dst = plane(f, dst_id);
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
// into the pyramid, and `cur` replaced by the new layer.
__device__ __forceinline__ void small_layer(const SPyr* P, int o, int i, const Taps& t, float* cur, float* rowp, int h, int w) {
    const int tid = threadIdx.x, npx = h * w, half = t.n / 2;
    float* dst = P->gp[o * (SL + 3) + i].px;
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
        cur[p] = a;  // only this thread reads cur[p] in this pass
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sift_small_octaves(const SiftFrame* F, int o0, int o1, TapsAll T, int lds_px) {
    const SPyr* P = &F[blockIdx.z].pyr;
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

// Tiles of EX_COLS x 4 * EX_ROWS positions of every octave in one launch: tile_start[o] is the
// first linear tile of octave o.
struct ExtremaTiles {
    int n_oct;
    int tile_start[MAX_OCT + 1];
    int tiles_x[MAX_OCT];
};

// positions per wave: 62 columns (lanes 1..62; lanes 0 and 63 only carry the neighbouring columns) x 4 rows per lane
constexpr int EX_COLS = 62, EX_ROWS = 4;

// the value of the lane to the left / right (wave-wide shifts of the vector ALU)
__device__ __forceinline__ float lane_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// A lane takes EX_ROWS consecutive positions of one column, all SL layers.  It loads its own column
// only (EX_ROWS + 2 rows of the SL + 3 Gaussian planes, whose differences are the SL + 2 difference planes); the largest and smallest value of a
// 3 x 3 neighbourhood is the column's maximum over three rows, then the maximum over the lane and
// its two neighbours.  val >= every one of the 27 values  <=>  val >= the largest of them
// (likewise the smallest).
__global__ __launch_bounds__(256) void k_sift_extrema(const SiftFrame* __restrict__ F, ExtremaTiles T, float threshold, int cap) {
    const SiftFrame& f = F[blockIdx.z];
    const SPyr* __restrict__ P = &f.pyr;
    unsigned long long* __restrict__ keys = f.keys;
    int* __restrict__ count = f.counts;
    const int lin = xcd_tile(blockIdx.x, T.tile_start[T.n_oct]);
    if (lin < 0) return;
    int octave = 0;
    while (octave + 1 < T.n_oct && lin >= T.tile_start[octave + 1]) octave++;
    const int tile = lin - T.tile_start[octave];
    const int ty = tile / T.tiles_x[octave], tx = tile - ty * T.tiles_x[octave];
    const int iw = P->gp[octave * (SL + 3)].w, ih = P->gp[octave * (SL + 3)].h;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = SBORDER + tx * EX_COLS + lane - 1;
    const int r0 = SBORDER + (ty * 4 + wv) * EX_ROWS;
    if (r0 >= ih - SBORDER) return;  // wave-uniform
    const int cc = min(c, iw - 1);
    float v[SL + 2][EX_ROWS + 2];
    {
        float g[SL + 3][EX_ROWS + 2];
#pragma unroll
        for (int l = 0; l < SL + 3; l++) {
            const auto pl = as_global<const float>(P->gp[octave * (SL + 3) + l].px);
#pragma unroll
            for (int k = 0; k < EX_ROWS + 2; k++) g[l][k] = pl[(size_t)min(r0 - 1 + k, ih - 1) * iw + cc];
        }
#pragma unroll
        for (int l = 0; l < SL + 2; l++)
#pragma unroll
            for (int k = 0; k < EX_ROWS + 2; k++) v[l][k] = g[l + 1][k] - g[l][k];
    }
    float hi[SL + 2][EX_ROWS], lo[SL + 2][EX_ROWS];
#pragma unroll
    for (int l = 0; l < SL + 2; l++)
#pragma unroll
        for (int k = 0; k < EX_ROWS; k++) {
            const float m = fmaxf(fmaxf(v[l][k], v[l][k + 1]), v[l][k + 2]);
            const float n = fminf(fminf(v[l][k], v[l][k + 1]), v[l][k + 2]);
            hi[l][k] = fmaxf(fmaxf(lane_left(m), m), lane_right(m));
            lo[l][k] = fminf(fminf(lane_left(n), n), lane_right(n));
        }
    const bool col_ok = lane != 0 && lane != 63 && c < iw - SBORDER;
#pragma unroll
    for (int l = 0; l < SL; l++)
#pragma unroll
        for (int k = 0; k < EX_ROWS; k++) {
            const float val = v[l + 1][k + 1];
            bool hit = col_ok && fabsf(val) > threshold && r0 + k < ih - SBORDER;
            if (hit) {
                const bool mx = val > 0 && val >= fmaxf(fmaxf(hi[l][k], hi[l + 1][k]), hi[l + 2][k]);
                const bool mn = val < 0 && val <= fminf(fminf(lo[l][k], lo[l + 1][k]), lo[l + 2][k]);
                hit = mx || mn;
            }
            // one addition to the frame's counter per wave and (layer, row), not one per candidate
            const unsigned long long m = __ballot(hit);
            if (m == 0) continue;  // wave-uniform
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, __popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (hit) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < cap)
                    keys[pos] = ((unsigned long long)octave << 56) | ((unsigned long long)(l + 1) << 48) |
                                ((unsigned long long)(r0 + k) << 24) | (unsigned long long)c;
            }
        }
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
        const SDog img = dog_layer(P, octv, layer);
        const SDog prev = dog_layer(P, octv, layer - 1);
        const SDog next = dog_layer(P, octv, layer + 1);
        const float dD[3] = {(DAT(img, r, c + 1) - DAT(img, r, c - 1)) * deriv_scale,
                             (DAT(img, r + 1, c) - DAT(img, r - 1, c)) * deriv_scale,
                             (DAT(next, r, c) - DAT(prev, r, c)) * deriv_scale};
        const float v2 = DAT(img, r, c) * 2;
        const float dxx = (DAT(img, r, c + 1) + DAT(img, r, c - 1) - v2) * second;
        const float dyy = (DAT(img, r + 1, c) + DAT(img, r - 1, c) - v2) * second;
        const float dss = (DAT(next, r, c) + DAT(prev, r, c) - v2) * second;
        const float dxy = (DAT(img, r + 1, c + 1) - DAT(img, r + 1, c - 1) - DAT(img, r - 1, c + 1) + DAT(img, r - 1, c - 1)) * cross;
        const float dxs = (DAT(next, r, c + 1) - DAT(next, r, c - 1) - DAT(prev, r, c + 1) + DAT(prev, r, c - 1)) * cross;
        const float dys = (DAT(next, r + 1, c) - DAT(next, r - 1, c) - DAT(prev, r + 1, c) + DAT(prev, r - 1, c)) * cross;
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
    const SDog img = dog_layer(P, octv, layer);
    const SDog prev = dog_layer(P, octv, layer - 1);
    const SDog next = dog_layer(P, octv, layer + 1);
    const float dD[3] = {(DAT(img, r, c + 1) - DAT(img, r, c - 1)) * deriv_scale,
                         (DAT(img, r + 1, c) - DAT(img, r - 1, c)) * deriv_scale,
                         (DAT(next, r, c) - DAT(prev, r, c)) * deriv_scale};
    const float t = (dD[0] * xc + dD[1] * xr) + dD[2] * xi;
    const float contr = DAT(img, r, c) * img_scale + t * 0.5f;
    if (fabsf(contr) * SL < SCONTRAST) return false;
    const float v2 = DAT(img, r, c) * 2.f;
    const float dxx = (DAT(img, r, c + 1) + DAT(img, r, c - 1) - v2) * second;
    const float dyy = (DAT(img, r + 1, c) + DAT(img, r - 1, c) - v2) * second;
    const float dxy = (DAT(img, r + 1, c + 1) - DAT(img, r + 1, c - 1) - DAT(img, r - 1, c + 1) + DAT(img, r - 1, c - 1)) * cross;
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_sift_refine(const SiftFrame* __restrict__ F, int cap, int list_cap) {
    const SiftFrame& f = F[blockIdx.y];
    const SPyr* __restrict__ P = &f.pyr;
    const unsigned long long* __restrict__ keys = f.keys;
    const int* __restrict__ count = f.counts;
    SKp* __restrict__ list = f.list;
    int* __restrict__ list_count = f.counts + 1;
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
// Sorted order of KeyPointsFilter::removeDuplicatedSorted (x, y, size descending, angle, response
// descending, octave descending).
__device__ __forceinline__ bool kp_before(const SKp& a, const SKp& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    return a.octave > b.octave;
}

// Position of every listed keypoint in the sorted order = number of keypoints that precede it
// (records that compare equal are equal in every field; the list index breaks those ties, so
// the positions are a permutation).  grid = (list blocks, frames).
__global__ __launch_bounds__(256) void k_sift_rank(const SiftFrame* __restrict__ F, int list_cap) {
    const SiftFrame& f = F[blockIdx.y];
    const int n = min(f.counts[1], list_cap);
    if ((int)blockIdx.x * 64 >= n) return;
    // 64 keypoints per workgroup; wave `sub` counts over a quarter of every tile of 256 list entries
    __shared__ SKp s[256];
    __shared__ __align__(16) unsigned long long s_key[256];
    __shared__ int s_part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, sub = tid >> 6, i = blockIdx.x * 64 + lane;
    const SKp me = f.list[min(i, n - 1)];
    // x and y are positive, so their bit patterns order like the values: one 64-bit compare decides
    // all pairs except those at the same position
    const unsigned long long my_key = ((unsigned long long)__float_as_uint(me.x) << 32) | __float_as_uint(me.y);
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        if (j0 + tid < n) {
            const SKp o = f.list[j0 + tid];
            s[tid] = o;
            s_key[tid] = ((unsigned long long)__float_as_uint(o.x) << 32) | __float_as_uint(o.y);
        } else {
            s_key[tid] = ~0ull;   // sorts behind everything, equals nothing
        }
        __syncthreads();
#pragma unroll 8
        for (int u2 = 0; u2 < 32; u2++) {
            const int u = 64 * sub + 2 * u2;
            const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s_key[u]);
            const unsigned long long ks[2] = {kk.x, kk.y};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (ks[e] == my_key) {
                    // same position: the remaining fields decide, then the place in the list (records that
                    // compare equal are equal in every field)
                    const SKp o = s[u + e];
                    const bool lt = kp_before(o, me), gtb = kp_before(me, o);
                    rank += (lt || (!gtb && j0 + u + e < i)) ? 1 : 0;
                } else {
                    rank += ks[e] < my_key ? 1 : 0;
                }
            }
        }
    }
    s_part[sub][lane] = rank;
    __syncthreads();
    if (sub == 0 && i < n) f.sorted[s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane]] = me;
}

// removeDuplicatedSorted keeps a keypoint iff it differs from its predecessor in x, y, size or
// angle; retainBest(nfeatures) then keeps a keypoint iff its response is at least the
// nfeatures-th largest, i.e. iff fewer than nfeatures unique keypoints have a larger response.
__global__ __launch_bounds__(256) void k_sift_mark(const SiftFrame* __restrict__ F, int list_cap) {
    const SiftFrame& f = F[blockIdx.y];
    const int n = min(f.counts[1], list_cap);
    if ((int)blockIdx.x * 256 >= n) return;
    __shared__ float s_r[256];
    __shared__ int s_u[256];
    const int tid = threadIdx.x, p = blockIdx.x * 256 + tid;
    const SKp* __restrict__ sorted = f.sorted;
    const float my_r = sorted[min(p, n - 1)].response;
    int gt = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        const int q = j0 + tid;
        if (q < n) {
            const SKp a = sorted[q];
            bool uniq = true;
            if (q > 0) {
                const SKp b = sorted[q - 1];
                uniq = a.x != b.x || a.y != b.y || a.size != b.size || a.angle != b.angle;
            }
            s_r[tid] = a.response;
            s_u[tid] = uniq ? 1 : 0;
        }
        __syncthreads();
        const int m = min(256, n - j0);
        for (int u = 0; u < m; u++) gt += (s_u[u] && s_r[u] > my_r) ? 1 : 0;
    }
    if (p < n) {
        bool uniq = true;
        if (p > 0) {
            const SKp a = sorted[p], b = sorted[p - 1];
            uniq = a.x != b.x || a.y != b.y || a.size != b.size || a.angle != b.angle;
        }
        f.gt[p] = uniq ? gt : -1;
    }
}

// The kept keypoints in sorted order, at most out_cap of them: the list the descriptor kernel
// reads and the caller's arrays (coordinates and size back to the input image, octave field
// re-based for firstOctave = -1).
__global__ __launch_bounds__(256) void k_sift_compact(const SiftFrame* __restrict__ F, int list_cap, int nfeatures, int out_cap) {
    const SiftFrame& f = F[blockIdx.y];
    const int n = min(f.counts[1], list_cap);
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            f.counts[2] = 0;
            *f.n_out = 0;
        }
        return;
    }
    if ((int)blockIdx.x * 256 >= n) return;
    __shared__ int s_k[256];
    const int tid = threadIdx.x, p = blockIdx.x * 256 + tid;
    const int* __restrict__ gt = f.gt;
    int before = 0;  // kept keypoints in front of p
    for (int j0 = 0; j0 < (int)blockIdx.x * 256 + 256 && j0 < n; j0 += 256) {
        __syncthreads();
        const int q = j0 + tid;
        s_k[tid] = (q < n && gt[q] >= 0 && gt[q] < nfeatures) ? 1 : 0;
        __syncthreads();
        const int m = min(256, p - j0);
        for (int u = 0; u < m; u++) before += s_k[u];
    }
    if (p >= n) return;
    const bool keep = gt[p] >= 0 && gt[p] < nfeatures;
    if (p == n - 1) {
        const int total = min(before + (keep ? 1 : 0), out_cap);
        f.counts[2] = total;
        *f.n_out = total;
    }
    if (!keep || before >= out_cap) return;
    const SKp kp = f.sorted[p];
    f.kept[before] = kp;
    f.xy[2 * before] = kp.x * 0.5f;
    f.xy[2 * before + 1] = kp.y * 0.5f;
    f.size[before] = kp.size * 0.5f;
    f.angle[before] = kp.angle;
    f.response[before] = kp.response;
    f.octave[before] = (kp.octave & ~255) | ((kp.octave - 1) & 255);
}

// grid = (out_cap, frames): wave q of frame y describes kept keypoint q, if there is one.
__global__ __launch_bounds__(64) void k_sift_describe(const SiftFrame* __restrict__ F) {
    const SiftFrame& f = F[blockIdx.y];
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= f.counts[2]) return;
    const SPyr* __restrict__ P = &f.pyr;
    float* __restrict__ desc = f.desc;
    const SKp kp = f.kept[q];
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
    // About half of the scanned square lies inside the descriptor's rotated window, and a sample inside it costs
    // some 700 instructions (exp and atan2 in double): the scan only tests (a few operations per position) and
    // queues the positions inside, in scan order; whenever 64 are queued all lanes compute one sample each, and
    // the eight shares of every sample are added to the bins in queue order, i.e. in scan order as before.
    __shared__ unsigned s_q[128];
    int nq = 0;   // queued positions (the same for all lanes)
    auto inside = [&](unsigned t, float& c_rot, float& r_rot, float& rbin, float& cbin, int& r, int& c) {
        const int i = (int)(t / (unsigned)side) - radius, j = (int)(t % (unsigned)side) - radius;
        c_rot = j * cos_t - i * sin_t;
        r_rot = j * sin_t + i * cos_t;
        rbin = r_rot + d / 2 - 0.5f;
        cbin = c_rot + d / 2 - 0.5f;
        r = py + i, c = px + j;
        return rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img.h - 1 && c > 0 && c < img.w - 1;
    };
    auto flush = [&](int count) {   // the first count <= 64 queued samples
        int idx = -1;
        if (lane < count) {
            float c_rot, r_rot, rbin, cbin;
            int r, c;
            inside(s_q[lane], c_rot, r_rot, rbin, cbin, r, c);
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
        s_idx[lane] = idx;
        __syncthreads();
        // the eight shares of a sample go to eight different bins: lane e < 8 adds share e of every
        // sample, in sample order, so every bin still receives its terms in scan order
        if (lane < 8) {
            const int ofs = (lane & 1) + ((lane >> 1) & 1) * (nb + 2) + (lane >> 2) * (d + 2) * (nb + 2);
            for (int l0 = 0; l0 < count; l0 += 8) {
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
    };
    for (unsigned base = 0; base < total; base += 64) {
        const unsigned t = base + lane;
        bool in = false;
        if (t < total) {
            float c_rot, r_rot, rbin, cbin;
            int r, c;
            in = inside(t, c_rot, r_rot, rbin, cbin, r, c);
        }
        const unsigned long long hit = __ballot(in);
        if (in) s_q[nq + __popcll(hit & ((1ull << lane) - 1ull))] = t;
        nq += __popcll(hit);
        __syncthreads();
        if (nq >= 64) {
            flush(64);
            const unsigned keep = lane < nq - 64 ? s_q[64 + lane] : 0u;
            __syncthreads();
            s_q[lane] = keep;
            nq -= 64;
            __syncthreads();
        }
    }
    if (nq > 0) flush(nq);
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

dim3 grid3(int w, int h, int nb) { return dim3((w + 63) / 64, (h + 3) / 4, nb); }
// the two passes of the blur: SB_R outputs per thread along the row / down the column
dim3 grid_rows(int w, int h, int nb) { return dim3((w + 64 * SB_R - 1) / (64 * SB_R), (h + 3) / 4, nb); }
dim3 grid_cols(int w, int h, int nb) { return dim3((w + 63) / 64, (h + 4 * SB_C - 1) / (4 * SB_C), nb); }

// What depends on the frame size alone: octaves, plane offsets inside a frame's arena (in
// floats), filter taps, the tiling of the extrema kernel.
struct SiftGeom {
    int h = 0, w = 0, bh, bw, n_oct, o_small, lds_px;
    int oh[MAX_OCT], ow[MAX_OCT];
    size_t off_gp[MAX_OCT * (SL + 3)], scratch_off, total;
    Taps base, taps[SL + 3];
    TapsAll small;
    ExtremaTiles tiles;
    float threshold;
};

int sift_geometry(rtvm_ctx* c, int h, int w, SiftGeom& g) {
    if (h < 16 || w < 16 || h > 8192 || w > 8192) RTVM_FAIL(c, RTVM_E_ARG, "sift: frame %dx%d (16..8192 pixels a side)", h, w);
    g.h = h, g.w = w, g.bh = 2 * h, g.bw = 2 * w;
    g.n_oct = cv_round_d(std::log((double)std::min(g.bw, g.bh)) / std::log(2.) - 2) + 1;
    if (g.n_oct < 1 || g.n_oct > MAX_OCT) RTVM_FAIL(c, RTVM_E_ARG, "sift: %d octaves", g.n_oct);
    double sig[SL + 3];
    sig[0] = SSIGMA;
    const double kf = std::pow(2., 1. / SL);
    for (int i = 1; i < SL + 3; i++) {
        const double sp = std::pow(kf, (double)(i - 1)) * SSIGMA, st = sp * kf;
        sig[i] = std::sqrt(st * st - sp * sp);
    }
    size_t total = 0;
    int oh = g.bh, ow = g.bw;
    for (int o = 0; o < g.n_oct; o++) {
        g.oh[o] = oh, g.ow[o] = ow;
        for (int i = 0; i < SL + 3; i++) {
            g.off_gp[o * (SL + 3) + i] = total;
            total += (size_t)oh * ow;
        }
        oh /= 2, ow /= 2;
        if (oh < 1 || ow < 1) RTVM_FAIL(c, RTVM_E_ARG, "sift: frame too small for %d octaves", g.n_oct);
    }
    g.scratch_off = total;
    total += (size_t)g.bh * g.bw;
    g.total = total;
    const float sig_diff = std::sqrt(std::max(SSIGMA * SSIGMA - 1.f, 0.01f));
    g.base = gauss_taps((double)sig_diff);
    memset(&g.small, 0, sizeof(g.small));
    for (int i = 1; i < SL + 3; i++) {
        g.taps[i] = gauss_taps(sig[i]);
        g.small.t[i] = g.taps[i];
    }
    g.threshold = (float)(int)std::floor(0.5 * SCONTRAST / SL * 255);
    // octaves from o_small on are built by one workgroup in one launch
    g.o_small = g.n_oct;
    for (int o = g.n_oct - 1; o >= 1 && g.oh[o] * g.ow[o] <= SMALL_OCTAVE_PX; o--) g.o_small = o;
    g.lds_px = g.o_small < g.n_oct ? g.oh[g.o_small] * g.ow[g.o_small] : 0;
    memset(&g.tiles, 0, sizeof(g.tiles));
    for (int o = 0; o < g.n_oct; o++) {
        if (!(g.oh[o] > 2 * SBORDER && g.ow[o] > 2 * SBORDER)) break;  // the octaves only get smaller
        const int tx = (g.ow[o] - 2 * SBORDER + EX_COLS - 1) / EX_COLS, ty = (g.oh[o] - 2 * SBORDER + 4 * EX_ROWS - 1) / (4 * EX_ROWS);
        g.tiles.tiles_x[o] = tx;
        g.tiles.tile_start[o + 1] = g.tiles.tile_start[o] + tx * ty;
        g.tiles.n_oct = o + 1;
    }
    return RTVM_OK;
}

// Sizes of the per-frame work areas.
constexpr int SIFT_CAND_CAP = 1 << 18, SIFT_LIST_CAP = 1 << 16;

size_t sift_work_bytes(int out_cap) {
    size_t b = 0;
    b += (size_t)SIFT_CAND_CAP * 8;                // keys
    b += 256;                                      // counts
    b += 3 * ((size_t)SIFT_LIST_CAP * sizeof(SKp) + 256);  // list, sorted, kept
    b += (size_t)SIFT_LIST_CAP * sizeof(int) + 256;        // gt
    (void)out_cap;
    return b + 1024;
}

// The host copy of one frame's descriptor: planes inside `arena` (g.total floats), lists inside
// `work` (sift_work_bytes); frame pixels and outputs are filled in by the caller.
void sift_frame_init(SiftFrame& f, const SiftGeom& g, float* arena, char* work) {
    memset(&f, 0, sizeof(f));
    for (int o = 0; o < g.n_oct; o++) {
        for (int i = 0; i < SL + 3; i++) f.pyr.gp[o * (SL + 3) + i] = SImg{arena + g.off_gp[o * (SL + 3) + i], g.oh[o], g.ow[o]};
    }
    f.scratch = arena + g.scratch_off;
    auto take = [&](size_t bytes) {
        char* p = work;
        work += (bytes + 255) & ~(size_t)255;
        return p;
    };
    f.keys = (unsigned long long*)take((size_t)SIFT_CAND_CAP * 8);
    f.counts = (int*)take(256);
    f.list = (SKp*)take((size_t)SIFT_LIST_CAP * sizeof(SKp));
    f.sorted = (SKp*)take((size_t)SIFT_LIST_CAP * sizeof(SKp));
    f.kept = (SKp*)take((size_t)SIFT_LIST_CAP * sizeof(SKp));
    f.gt = (int*)take((size_t)SIFT_LIST_CAP * sizeof(int));
}

// The two passes of one blur for nb frames: the kernel compiled for the length of `t`, or the
// general one.
int sift_blur(rtvm_ctx* c, const SiftFrame* d_frames, int nb, int src, int dst, int h, int w, const Taps& t) {
#define SIFT_BLUR_N(N)                                                                                                   \
    case N:                                                                                                              \
        RTVM_LAUNCH(c, k_sift_blur<N>, dim3(xcd_grid(((w + SBT_W - 1) / SBT_W) * ((h + SBT_H - 1) / SBT_H)), 1, nb), dim3(256),  \
                    0, d_frames, src, dst, h, w, t);                                                                     \
        break;
    switch (t.n) {
        SIFT_BLUR_N(11)
        SIFT_BLUR_N(13)
        SIFT_BLUR_N(17)
        SIFT_BLUR_N(21)
        SIFT_BLUR_N(27)
        default:
            RTVM_LAUNCH(c, k_sift_blur_rows, grid3(w, h, nb), dim3(256), 0, d_frames, src, PLANE_SCRATCH, h, w, t);
            RTVM_LAUNCH(c, k_sift_blur_cols, grid3(w, h, nb), dim3(256), 0, d_frames, PLANE_SCRATCH, dst, h, w, t);
    }
#undef SIFT_BLUR_N
    return RTVM_OK;
}

// Detection and description of nb frames (descriptors d_frames[0 .. nb-1], all of geometry g) on
// the context's current stream.
int sift_launch(rtvm_ctx* c, const SiftGeom& g, const SiftFrame* d_frames, int nb, int nfeatures, int out_cap) {
    const int bh = g.bh, bw = g.bw;
    // base image: doubled, blurred by sqrt(sigma^2 - 1) (into layer 0 via layer 1 of octave 0 as a temporary)
    {
        const int dbl = 1;  // free until layer 1 is made from layer 0
        RTVM_LAUNCH(c, k_sift_base, grid3(g.w, g.h, nb), dim3(256), 0, d_frames, g.h, g.w, dbl, rtvm_gray_coef(c));
        sift_blur(c, d_frames, nb, dbl, 0, bh, bw, g.base);
    }
    for (int o = 0; o < g.o_small; o++) {
        const int oh = g.oh[o], ow = g.ow[o];
        for (int i = 0; i < SL + 3; i++) {
            const int dst = o * (SL + 3) + i;
            if (i == 0) {
                if (o == 0) continue;
                RTVM_LAUNCH(c, k_sift_decimate, grid3(ow, oh, nb), dim3(256), 0, d_frames, (o - 1) * (SL + 3) + SL, g.ow[o - 1],
                            dst, oh, ow);
            } else {
                sift_blur(c, d_frames, nb, dst - 1, dst, oh, ow, g.taps[i]);
            }
        }
    }
    if (g.o_small < g.n_oct) {
        const size_t lds_bytes = (size_t)2 * g.lds_px * sizeof(float);
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sift_small_octaves),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        RTVM_LAUNCH(c, k_sift_small_octaves, dim3(1, 1, nb), dim3(1024), lds_bytes, d_frames, g.o_small, g.n_oct, g.small,
                    g.lds_px);
    }
    if (g.tiles.n_oct > 0)
        RTVM_LAUNCH(c, k_sift_extrema, dim3(xcd_grid(g.tiles.tile_start[g.tiles.n_oct]), 1, nb), dim3(256), 0, d_frames, g.tiles,
                    g.threshold, SIFT_CAND_CAP);
    // refinement of every candidate, queued behind the extrema without a host round trip; then
    // removeDuplicatedSorted + retainBest on the device, then the descriptors of what is kept
    RTVM_LAUNCH(c, k_sift_refine, dim3(nb > 1 ? 2048 : 4096, nb), dim3(64), 0, d_frames, SIFT_CAND_CAP, SIFT_LIST_CAP);
    const dim3 gl(SIFT_LIST_CAP / 256, nb);
    RTVM_LAUNCH(c, k_sift_rank, dim3(SIFT_LIST_CAP / 64, nb), dim3(256), 0, d_frames, SIFT_LIST_CAP);
    RTVM_LAUNCH(c, k_sift_mark, gl, dim3(256), 0, d_frames, SIFT_LIST_CAP);
    RTVM_LAUNCH(c, k_sift_compact, gl, dim3(256), 0, d_frames, SIFT_LIST_CAP, nfeatures, out_cap);
    RTVM_LAUNCH(c, k_sift_describe, dim3(out_cap, nb), dim3(64), 0, d_frames);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
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
#define SIFT_BUF(c, slot, ptr, bytes)                                 \
    do {                                                              \
        int r_ = sift_buffer((c), (slot), (size_t)(bytes), &(ptr));   \
        if (r_ != RTVM_OK) return r_;                                 \
    } while (0)

// Frame pixels and output arrays of the frames of a batch: slot sb + j feeds work area j.
__global__ void k_sift_bind(SiftFrame* __restrict__ F, const FrameSlot* __restrict__ slots, int sb, int nb) {
    const int j = threadIdx.x;
    if (j >= nb) return;
    const FrameSlot& S = slots[sb + j];
    F[j].bgr = S.bgr;
    F[j].xy = S.feat.xy;
    F[j].size = S.feat.size;
    F[j].angle = S.feat.angle;
    F[j].response = S.feat.response;
    F[j].octave = S.feat.octave;
    F[j].desc = reinterpret_cast<float*>(S.feat.desc);
    F[j].n_out = S.feat.count;
}

// SIFT detection for the stream path: one work area (pyramids, lists) per frame of a batch, in two sets.  The
// batches in the even slot sets are detected on one stream with the first set of areas, those in the odd slot
// sets on the other stream with the second: the areas of a set serve their batches in turn, in stream order.
// (Contexts with one detection stream use the first set only.)
struct SiftPipe {
    SiftGeom g;
    int sets = 1;                   // sets of work areas in use
    int n = 0;                      // work areas allocated per set
    SiftFrame* d_frames = nullptr;  // [2][RTVM_MAX_BATCH]
    void* arena[2][RTVM_MAX_BATCH] = {};
    void* work[2][RTVM_MAX_BATCH] = {};
};

}  // namespace

int rtvm_sift_pipe_create(rtvm_ctx* c) {
    SiftPipe* p = new (std::nothrow) SiftPipe;
    if (!p) RTVM_FAIL(c, RTVM_E_NOMEM, "host allocation failed");
    c->sift_pipe = p;
    int r = sift_geometry(c, c->fh, c->fw, p->g);
    if (r) return r;
    p->sets = c->stream_front2 != c->stream_front ? 2 : 1;
    hipError_t e = hipMalloc((void**)&p->d_frames, sizeof(SiftFrame) * RTVM_MAX_BATCH * 2);
    if (e != hipSuccess) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: hipMalloc failed: %s", hipGetErrorString(e));
    return RTVM_OK;
}

int rtvm_sift_pipe_reserve(rtvm_ctx* c, int n) {
    SiftPipe* p = (SiftPipe*)c->sift_pipe;
    if (!p) RTVM_FAIL(c, RTVM_E_STATE, "sift: the context was not created with the SIFT detector");
    for (int j = p->n; j < n && j < RTVM_MAX_BATCH; j++) {
        for (int k = 0; k < p->sets; k++) {
            const size_t ab = p->g.total * sizeof(float), wb = sift_work_bytes(c->orb.max_kp);
            hipError_t e = hipMalloc(&p->arena[k][j], ab);
            if (e == hipSuccess) e = hipMalloc(&p->work[k][j], wb);
            if (e != hipSuccess)
                RTVM_FAIL(c, RTVM_E_NOMEM, "sift: hipMalloc(%zu + %zu bytes) for the pyramids of batch frame %d failed: %s", ab, wb, j,
                          hipGetErrorString(e));
            SiftFrame f;
            sift_frame_init(f, p->g, (float*)p->arena[k][j], (char*)p->work[k][j]);
            HIP_TRY(c, hipMemcpy(p->d_frames + k * RTVM_MAX_BATCH + j, &f, sizeof(SiftFrame), hipMemcpyHostToDevice));
        }
        p->n = j + 1;
    }
    return RTVM_OK;
}

void rtvm_sift_pipe_free(rtvm_ctx* c) {
    SiftPipe* p = (SiftPipe*)c->sift_pipe;
    if (!p) return;
    for (int k = 0; k < 2; k++)
        for (int j = 0; j < RTVM_MAX_BATCH; j++) {
            if (p->arena[k][j]) (void)hipFree(p->arena[k][j]);
            if (p->work[k][j]) (void)hipFree(p->work[k][j]);
        }
    if (p->d_frames) (void)hipFree(p->d_frames);
    delete p;
    c->sift_pipe = nullptr;
}

// cv2.SIFT_create(nfeatures).detectAndCompute for the frames bound to slots sb .. sb+nb-1: features into slot.feat
int rtvm_launch_sift(rtvm_ctx* c, int sb, int nb) {
    SiftPipe* p = (SiftPipe*)c->sift_pipe;
    if (!p || nb < 1 || nb > p->n) RTVM_FAIL(c, RTVM_E_STATE, "sift: batch of %d frames, work areas for %d", nb, p ? p->n : 0);
    // the work areas that belong to the stream this is queued on
    const int k = c->stream == c->stream_front2 && c->stream_front2 != c->stream_front ? 1 : 0;
    if (k && !p->arena[1][nb - 1]) RTVM_FAIL(c, RTVM_E_STATE, "sift: the second set of work areas holds fewer than %d frames", nb);
    SiftFrame* frames = p->d_frames + k * RTVM_MAX_BATCH;
    RTVM_LAUNCH(c, k_sift_bind, dim3(1), dim3(64), 0, frames, (const FrameSlot*)c->d_slots, sb, nb);
    return sift_launch(c, p->g, frames, nb, c->nfeatures, c->orb.max_kp);
}

extern "C" int rtvm_sift_max_keypoints(int nfeatures) { return nfeatures + 256; }

extern "C" int rtvm_sift(rtvm_ctx* c, const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy, float* size,
                         float* angle_deg_out, float* response, int32_t* octave, float* desc) {
    if (!c) return RTVM_E_ARG;
    if (!bgr || !n_out || !xy || !size || !angle_deg_out || !response || !octave || !desc)
        RTVM_FAIL(c, RTVM_E_ARG, "sift: null argument");
    if (nfeatures < 1 || nfeatures > SIFT_LIST_CAP - 256) RTVM_FAIL(c, RTVM_E_ARG, "sift: %d features (1..%d)", nfeatures, SIFT_LIST_CAP - 256);
    HIP_TRY(c, hipSetDevice(c->device));
    SiftGeom g;
    int r = sift_geometry(c, h, w, g);
    if (r) return r;
    const int out_cap = rtvm_sift_max_keypoints(nfeatures);
    // buffers: pyramids, frame pixels, lists, frame descriptor, outputs
    void *arena, *frame, *work, *fdesc, *outs;
    SIFT_BUF(c, 0, arena, g.total * sizeof(float));
    SIFT_BUF(c, 1, frame, (size_t)h * w * 3);
    SIFT_BUF(c, 2, work, sift_work_bytes(out_cap));
    SIFT_BUF(c, 4, fdesc, sizeof(SiftFrame));
    const size_t row = (2 + 1 + 1 + 1 + 1 + 128) * sizeof(float);
    SIFT_BUF(c, 7, outs, (size_t)out_cap * row + 8 * 256);
    SiftFrame f;
    sift_frame_init(f, g, (float*)arena, (char*)work);
    f.bgr = (const uint8_t*)frame;
    {
        char* o = (char*)outs;
        auto take = [&](size_t bytes) {
            char* p = o;
            o += (bytes + 255) & ~(size_t)255;
            return p;
        };
        f.xy = (float*)take((size_t)out_cap * 8);
        f.size = (float*)take((size_t)out_cap * 4);
        f.angle = (float*)take((size_t)out_cap * 4);
        f.response = (float*)take((size_t)out_cap * 4);
        f.octave = (int32_t*)take((size_t)out_cap * 4);
        f.desc = (float*)take((size_t)out_cap * 512);
        f.n_out = (int*)take(4);
    }
    // the descriptor goes through pinned memory: a pageable source would make the copy synchronous
    SiftFrame* hf = (SiftFrame*)((char*)c->h_pinned + 8192);
    static_assert(sizeof(SiftFrame) <= 8192, "frame descriptor staging");
    *hf = f;
    HIP_TRY(c, hipMemcpyAsync(frame, bgr, (size_t)h * w * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(fdesc, hf, sizeof(SiftFrame), hipMemcpyHostToDevice, c->stream));
    if ((r = sift_launch(c, g, (const SiftFrame*)fdesc, 1, nfeatures, out_cap))) return r;
    int* counts = (int*)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(counts, f.counts, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (counts[0] > SIFT_CAND_CAP) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: %d scale-space extrema, room for %d", counts[0], SIFT_CAND_CAP);
    if (counts[1] > SIFT_LIST_CAP) RTVM_FAIL(c, RTVM_E_NOMEM, "sift: %d keypoints before filtering, room for %d", counts[1], SIFT_LIST_CAP);
    const int n = counts[2];
    if (n < 0 || n > out_cap) RTVM_FAIL(c, RTVM_E_HIP, "sift: device returned keypoint count %d", n);
    *n_out = n;
    if (n == 0) return RTVM_OK;
    HIP_TRY(c, hipMemcpyAsync(xy, f.xy, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(size, f.size, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(angle_deg_out, f.angle, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(response, f.response, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(octave, f.octave, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(desc, f.desc, (size_t)n * 512, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTVM_OK;
}
