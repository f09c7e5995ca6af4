// cv2.cvtColor(BGR2GRAY) + cv2.ORB_create(n).detectAndCompute for gfx950
// (reference main.py:36,111-112,717-718).
//
// Pipeline (every stage keeps its data in HBM planes of one pyramid buffer, one
// u8 plane per level, pitch a multiple of 64):
//   k_gray        BGR -> level 0
//   k_resize_chain  levels 1-3 and 4-7, bit-exact fixed-point bilinear, level by level through LDS
//                 (two launches)
//   k_fast        FAST-9/16 on 64x32 tiles in LDS, all levels in one launch: quick test, eight-pair
//                 test and arc scores on compacted lists, 3x3 non-maximum suppression, 31-pixel
//                 border, aggregated append of candidates + per-level score histogram
//   k_orb_select  one workgroup per level: retainBest(2n) from the histogram, raster
//                 sort (bitonic, LDS), Harris response, retainBest(n) by rank counting
//   k_orb_describe one wave per keypoint: intensity-centroid angle, steered BRIEF-256 on the 7x7
//                 Gaussian (integer, exact) of the level, evaluated at the points the tests read
// All integer stages are exact; the float stages (Harris response, angle, pattern
// rotation) follow OpenCV's operation order without fused multiply-add.
#include "common.h"

#include <cfloat>
#include <cmath>
#include <vector>

namespace {

constexpr int EDGE = 31;
constexpr int FAST_T = 20;
constexpr int SEL_CAP = 8192;  // FAST survivors per level the selection workgroup can hold

__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ------------------------------------------------------------------ gray / pyramid
// Also clears the slot's candidate counters and score histograms for the FAST stage.
__global__ __launch_bounds__(256) void k_gray(const FrameSlot* __restrict__ slots, int sb, unsigned off0, int w, int h,
                                              int pitch, GrayCoef gc) {
    // 8-bit pixels times 15-bit coefficients: 24-bit multiplies (full rate; a 32-bit one takes four times as long)
    const uint32_t cb = gc.cb, cg = gc.cg, cr = gc.cr, half = 1u << (gc.shift - 1), sh = gc.shift;
    const FrameSlot& S = slots[sb + blockIdx.z];
    const uint8_t* __restrict__ bgr = S.bgr;
    uint8_t* __restrict__ out = S.pyr + off0;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < RTVM_NLEVELS + RTVM_NLEVELS * 256; i += 256) S.cand_count[i] = 0;
    // four pixels per lane: twelve bytes in (three unaligned-safe word loads), one word out
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* p = bgr + ((size_t)y * w + x) * 3;
    if (x + 4 <= w) {
        uint32_t v[3];
        __builtin_memcpy(v, p, 12);
        const uint32_t b0 = v[0] & 255, g0 = (v[0] >> 8) & 255, r0 = (v[0] >> 16) & 255, b1 = v[0] >> 24;
        const uint32_t g1 = v[1] & 255, r1 = (v[1] >> 8) & 255, b2 = (v[1] >> 16) & 255, g2 = v[1] >> 24;
        const uint32_t r2 = v[2] & 255, b3 = (v[2] >> 8) & 255, g3 = (v[2] >> 16) & 255, r3 = v[2] >> 24;
        const uint32_t y0 = (__umul24(b0, cb) + __umul24(g0, cg) + __umul24(r0, cr) + half) >> sh;
        const uint32_t y1 = (__umul24(b1, cb) + __umul24(g1, cg) + __umul24(r1, cr) + half) >> sh;
        const uint32_t y2 = (__umul24(b2, cb) + __umul24(g2, cg) + __umul24(r2, cr) + half) >> sh;
        const uint32_t y3 = (__umul24(b3, cb) + __umul24(g3, cg) + __umul24(r3, cr) + half) >> sh;
        *reinterpret_cast<uint32_t*>(out + (size_t)y * pitch + x) = y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
    } else {
        for (int i = 0; x + i < w; i++)
            out[(size_t)y * pitch + x + i] = (uint8_t)((p[3 * i] * cb + p[3 * i + 1] * cg + p[3 * i + 2] * cr + half) >> sh);
    }
}

// INTER_LINEAR_EXACT: 8.8 coefficients (xw/yw = weight of the second tap, 0xFFFF = clamped
// single tap, evaluated as weight 0 on a repeated tap: (256-0)*a + 0*a = a << 8):
//   h0 = (256-cx)*a00 + cx*a01, h1 = (256-cx)*a10 + cx*a11, v = ((256-cy)*h0 + cy*h1 + 32768) >> 16.

// ---- several levels per launch
// The rectangle of a level a chain workgroup computes (x0, y0, w, h; x0 and w multiples of 4) and the part
// of it the workgroup writes to memory (columns ox0..ox1, rows oy0..oy1, exclusive ends).
struct RcRegion {
    int x0, y0, w, h, ox0, oy0, ox1, oy1;
};
static_assert(sizeof(RcRegion) == 32, "the context reserves 32 bytes per rectangle");

constexpr int RC_TW = 64, RC_TH = 16;  // tile of the chain's last level
constexpr int RC_MAXN = 4;             // levels per chain
// bytes of the two staging buffers (the levels alternate between them; the chain's source level, the largest
// rectangle, is in the first) of the chain of three levels and of the chain of four: sized for the pyramid factor
// 1.2 with a margin, so that LDS does not limit the workgroups per CU; the host checks every tile against them
constexpr int RC3_B0 = 6144, RC3_B1 = 4096, RC4_B0 = 9216, RC4_B1 = 6144;

// Rectangles of levels a .. a+n (R[0] .. R[n]) for tile (tx0, ty0) of level a+n.  Level j+1 reads level j
// through the tables xofs/yofs (+ tab_off[j+1]): a workgroup computes of every level what its next level
// reads, and owns (writes) the pixels between the first sources of its own tile and of the next tile:
// those ranges partition every level, so that each pixel is written once, by the workgroup that has it
// anyway.  Column boundaries are rounded down to 4 (words are stored); the last tile of a row or column
// takes the level's remainder.  Host and device run the same code on the same tables.
__host__ __device__ inline void rc_regions(const OrbDev& P, int a, int n, int tx0, int ty0, const uint16_t* xofs,
                                           const uint16_t* yofs, RcRegion* R) {
    const int b = a + n;
    const int w4b = (P.w[b] + 3) & ~3;
    R[n].ox0 = tx0;
    R[n].oy0 = ty0;
    R[n].ox1 = tx0 + RC_TW < P.w[b] ? tx0 + RC_TW : w4b;
    R[n].oy1 = ty0 + RC_TH < P.h[b] ? ty0 + RC_TH : P.h[b];
    R[n].x0 = R[n].ox0;
    R[n].y0 = R[n].oy0;
    R[n].w = R[n].ox1 - R[n].ox0;
    R[n].h = R[n].oy1 - R[n].oy0;
    for (int k = n - 1; k >= 0; k--) {
        const int j = a + k;  // this level; level j+1 is R[k+1]
        const uint16_t* xo = xofs + P.xtab_off[j + 1];
        const uint16_t* yo = yofs + P.ytab_off[j + 1];
        const RcRegion& U = R[k + 1];
        const int uw = P.w[j + 1], uh = P.h[j + 1];
        const int lastx = U.x0 + U.w - 1 < uw - 1 ? U.x0 + U.w - 1 : uw - 1;
        const int lasty = U.y0 + U.h - 1 < uh - 1 ? U.y0 + U.h - 1 : uh - 1;
        // taps of the upper rectangle: first source .. last source + 1 (inside the level)
        int nx0 = xo[U.x0 < uw - 1 ? U.x0 : uw - 1], nx1 = (int)xo[lastx] + 2;   // exclusive
        int ny0 = yo[U.y0 < uh - 1 ? U.y0 : uh - 1], ny1 = (int)yo[lasty] + 2;
        const int w4 = (P.w[j] + 3) & ~3;
        if (nx1 > w4) nx1 = w4;
        if (ny1 > P.h[j]) ny1 = P.h[j];
        RcRegion& T = R[k];
        T.ox0 = U.ox0 == 0 ? 0 : (int)xo[U.ox0] & ~3;
        T.oy0 = U.oy0 == 0 ? 0 : (int)yo[U.oy0];
        T.ox1 = U.ox1 >= uw ? w4 : (int)xo[U.ox1] & ~3;
        T.oy1 = U.oy1 >= uh ? P.h[j] : (int)yo[U.oy1];
        const int x0 = (nx0 & ~3) < T.ox0 ? (nx0 & ~3) : T.ox0, x1 = ((nx1 + 3) & ~3) > T.ox1 ? ((nx1 + 3) & ~3) : T.ox1;
        const int y0 = ny0 < T.oy0 ? ny0 : T.oy0, y1 = ny1 > T.oy1 ? ny1 : T.oy1;
        T.x0 = x0;
        T.y0 = y0;
        T.w = x1 - x0;
        T.h = y1 - y0;
    }
}

// Levels a+1 .. a+N of the pyramid in one launch: one workgroup per 64 x 16 tile of level a+N stages the
// rectangle of level a under it in LDS and resizes level by level from LDS to LDS (the arithmetic above),
// four pixels of a row per thread, storing the pixels it owns as words.  What neighbouring workgroups both
// need of a level both compute (about a third more pixels than the levels have); no level is read back
// from memory and no launch waits for another.  The host has checked every rectangle against its staging buffer.
template <int N>
__global__ __launch_bounds__(256) void k_resize_chain(const OrbDev P, const FrameSlot* __restrict__ slots, int sb, int a,
                                                      const uint16_t* __restrict__ xofs, const uint32_t* __restrict__ xsel,
                                                      const uint32_t* __restrict__ xwp,
                                                      const uint16_t* __restrict__ yofs, const uint16_t* __restrict__ yw,
                                                      const RcRegion* __restrict__ regions) {
    constexpr int B0 = N == 3 ? RC3_B0 : RC4_B0, B1 = N == 3 ? RC3_B1 : RC4_B1;
    static_assert(N == 3 || N == 4, "staging buffers exist for the chains of three and of four levels");
    __shared__ __align__(16) uint8_t s_buf0[B0], s_buf1[B1];
    __shared__ RcRegion s_reg[N + 1];
    const FrameSlot& S = slots[sb + blockIdx.z];
    const int b = a + N;
    const int tiles_x = (P.w[b] + RC_TW - 1) / RC_TW;
    const int tile = xcd_tile(blockIdx.x, tiles_x * ((P.h[b] + RC_TH - 1) / RC_TH));
    if (tile < 0) return;
    const int tid = threadIdx.x;
    // the tile's rectangles, computed by the host at set-up with rc_regions (a chain of dependent table look-ups)
    if (tid <= N) s_reg[tid] = regions[tile * (N + 1) + tid];
    __syncthreads();
    {   // level a: the rectangle from memory, aligned words, clamped to the plane
        const RcRegion R = s_reg[0];
        const uint8_t* __restrict__ src = S.pyr + P.off[a];
        const int sp = P.pitch[a], sh = P.h[a];
        const int qw = R.w >> 2, nq = qw * R.h;
        const float inv = __builtin_amdgcn_rcpf((float)qw);
        for (int i = tid; i < nq; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv), q = i - r * qw;   // i / qw: exact for these sizes
            const int y = min(R.y0 + r, sh - 1), x = min(R.x0 + 4 * q, sp - 4);
            *reinterpret_cast<uint32_t*>(&s_buf0[r * R.w + 4 * q]) = *reinterpret_cast<const uint32_t*>(src + (unsigned)(y * sp + x));
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k <= N; k++) {
        const int j = a + k;
        const RcRegion R = s_reg[k], U = s_reg[k - 1];   // this level, the level it reads
        const uint8_t* __restrict__ sbuf = ((k - 1) & 1) ? s_buf1 : s_buf0;
        uint8_t* __restrict__ dbuf = (k & 1) ? s_buf1 : s_buf0;
        uint8_t* __restrict__ dst = S.pyr + P.off[j];
        const uint16_t* __restrict__ xo_t = xofs + P.xtab_off[j];
        const uint32_t* __restrict__ xs_t = xsel + P.xtab_off[j];
        const uint32_t* __restrict__ xp_t = xwp + P.xtab_off[j];
        const uint16_t* __restrict__ yo_t = yofs + P.ytab_off[j];
        const uint16_t* __restrict__ yw_t = yw + P.ytab_off[j];
        const int dp = P.pitch[j], dh = P.h[j];
        const int qw = R.w >> 2, nq = qw * R.h;
        const float inv = __builtin_amdgcn_rcpf((float)qw);
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        for (int i = tid; i < nq; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv), q = i - r * qw;
            const int x = R.x0 + 4 * q, y = R.y0 + r;   // x < w4: the tables are padded to there
            const int yc = min(y, dh - 1);
            const int yo = yo_t[yc];
            const unsigned cyr = yw_t[yc];
            const unsigned cy = cyr == 0xFFFFu ? 0u : cyr;
            // The four outputs read at most six consecutive bytes of a row, from the quad's first source on: the
            // eight bytes from there are cut out of three aligned words, an output's two taps out of those by a
            // byte selector from the table (as 16-bit halves), and the horizontal sum is one v_dot2_u32_u16 with
            // the output's weight pair {256 - w, w} (a clamped single tap: the same byte twice, weights {256, 0}).
            const int b0 = (yo - U.y0) * U.w - U.x0 + (int)xo_t[x], b1 = cyr == 0xFFFFu ? b0 : b0 + U.w;
            const uint4 sel4 = *reinterpret_cast<const uint4*>(xs_t + x), wp4 = *reinterpret_cast<const uint4*>(xp_t + x);
            const uint32_t* w0p = reinterpret_cast<const uint32_t*>(sbuf + (b0 & ~3));
            const uint32_t* w1p = reinterpret_cast<const uint32_t*>(sbuf + (b1 & ~3));
            const uint32_t sh = (uint32_t)(b0 & 3);   // the rows of a rectangle are a multiple of 4 bytes apart
            const uint32_t lo0 = __builtin_amdgcn_alignbyte(w0p[1], w0p[0], sh), hi0 = __builtin_amdgcn_alignbyte(w0p[2], w0p[1], sh);
            const uint32_t lo1 = __builtin_amdgcn_alignbyte(w1p[1], w1p[0], sh), hi1 = __builtin_amdgcn_alignbyte(w1p[2], w1p[1], sh);
            const uint32_t sels[4] = {sel4.x, sel4.y, sel4.z, sel4.w}, wps[4] = {wp4.x, wp4.y, wp4.z, wp4.w};
            uint32_t word = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const us2 wp = __builtin_bit_cast(us2, wps[e]);
                const unsigned h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi0, lo0, sels[e])), wp, 0u, false);
                const unsigned h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi1, lo1, sels[e])), wp, 0u, false);
                // 9-bit weights, 16-bit sums: 24-bit multiplies
                unsigned v = __umul24(256u - cy, h0) + __umul24(cy, h1);
                v = (v + 32768u) >> 16;
                word |= (v > 255u ? 255u : v) << (8 * e);
            }
            if (k < N) *reinterpret_cast<uint32_t*>(&dbuf[r * R.w + 4 * q]) = word;
            if (x >= R.ox0 && x < R.ox1 && y >= R.oy0 && y < R.oy1)
                *reinterpret_cast<uint32_t*>(dst + (unsigned)(y * dp + x)) = word;
        }
        if (k < N) __syncthreads();
    }
}

// ------------------------------------------------------------------ FAST
__device__ __forceinline__ int tile_level(const OrbDev& P, int tile, int* local) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < RTVM_NLEVELS; k++)
        if (tile >= P.tile_off[k]) l = k;
    *local = tile - P.tile_off[l];
    return l;
}

// cv::FAST (type 9_16, threshold 20) on a pixel tile in LDS, in two steps so that the
// expensive one only runs on the few pixels that can be corners.
// ring_diffs: v - p for the 16 ring pixels of the pixel at LDS position (row, col).
template <int LP>
__device__ __forceinline__ void ring_diffs(const uint8_t (*s)[LP], int row, int col, int* d) {
    const int v = s[row][col];
    d[0] = v - s[row + 3][col];
    d[1] = v - s[row + 3][col + 1];
    d[2] = v - s[row + 2][col + 2];
    d[3] = v - s[row + 1][col + 3];
    d[4] = v - s[row][col + 3];
    d[5] = v - s[row - 1][col + 3];
    d[6] = v - s[row - 2][col + 2];
    d[7] = v - s[row - 3][col + 1];
    d[8] = v - s[row - 3][col];
    d[9] = v - s[row - 3][col - 1];
    d[10] = v - s[row - 2][col - 2];
    d[11] = v - s[row - 1][col - 3];
    d[12] = v - s[row][col - 3];
    d[13] = v - s[row + 1][col - 3];
    d[14] = v - s[row + 2][col - 2];
    d[15] = v - s[row + 3][col - 1];
}

// Necessary condition for a corner: an arc of 9 contains one pixel of every opposite pair,
// so every pair must have a member brighter (or every pair a member darker) than the
// threshold allows.
template <int LP>
__device__ __forceinline__ bool fast_candidate(const uint8_t (*s)[LP], int row, int col) {
    const int v = s[row][col];
    const int d0 = v - s[row + 3][col], d8 = v - s[row - 3][col];
    if (!(max(d0, d8) > FAST_T || min(d0, d8) < -FAST_T)) return false;
    const int d4 = v - s[row][col + 3], d12 = v - s[row][col - 3];
    if (!(max(d4, d12) > FAST_T || min(d4, d12) < -FAST_T)) return false;
    int d[16];
    ring_diffs<LP>(s, row, col, d);
    bool bright = true, dark = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bright &= max(d[k], d[k + 8]) > FAST_T;
        dark &= min(d[k], d[k + 8]) < -FAST_T;
    }
    return bright || dark;
}

// Corner score: max over the 16 arcs of 9 contiguous ring pixels of min(v - p) (bright
// centre) or min(p - v) (dark centre), minus 1; 0 when no arc clears the threshold.
template <int LP>
__device__ __forceinline__ int fast_score_at(const uint8_t (*s)[LP], int row, int col) {
    int d[16];
    ring_diffs<LP>(s, row, col, d);
    int best = 0;
    // min over arcs of 9 by doubling: m2, m4, m8, then one more element
    int m2[16], m4[16], M2[16], M4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        m2[k] = min(d[k], d[(k + 1) & 15]);
        M2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        m4[k] = min(m2[k], m2[(k + 2) & 15]);
        M4[k] = max(M2[k], M2[(k + 2) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int m9 = min(min(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int M9 = max(max(M4[k], M4[(k + 4) & 15]), d[(k + 8) & 15]);
        best = max(best, max(m9, -M9));
    }
    return best > FAST_T ? best - 1 : 0;
}

// The two axis pairs of the FAST test for two positions at once (16-bit halves): bit 15 of a half is
// set when both the vertical pair (t, b) and the horizontal pair (l, r) hold a pixel brighter than
// v + FAST_T or both hold one darker than v - FAST_T ... in the sense of fast_candidate's first two lines.
__device__ __forceinline__ uint32_t quick_pair(uint32_t v, uint32_t t, uint32_t b, uint32_t l, uint32_t r) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 V = __builtin_bit_cast(us2, v), T = __builtin_bit_cast(us2, t), B = __builtin_bit_cast(us2, b);
    const us2 L = __builtin_bit_cast(us2, l), R = __builtin_bit_cast(us2, r);
    const us2 th = {FAST_T, FAST_T};
    const us2 vt = V + th;
    const us2 a = ((__builtin_elementwise_min(T, B) + th) - V) | (vt - __builtin_elementwise_max(T, B));
    const us2 c = ((__builtin_elementwise_min(L, R) + th) - V) | (vt - __builtin_elementwise_max(L, R));
    return __builtin_bit_cast(uint32_t, a & c) & 0x80008000u;
}

// FAST + 3x3 non-maximum suppression for one tile of 64 x 32 pixels of the region that
// survives ORB's 31-pixel border: scores of the tile and a one-pixel rim in LDS, then a
// pixel is kept iff its score beats its 8 neighbours strictly.  Keeps are appended as
// (score << 24 | y << 12 | x) to the level's candidate list and counted in the level's
// score histogram (aggregated per workgroup in LDS: one global atomic per workgroup
// and per occupied bin).
__global__ __launch_bounds__(256) void k_fast(OrbDev P, const FrameSlot* __restrict__ slots, int sb) {
    const FrameSlot& SL = slots[sb + blockIdx.y];
    uint32_t* __restrict__ cand = SL.cand;
    int* __restrict__ cand_count = SL.cand_count;
    int* __restrict__ hist = SL.cand_count + RTVM_NLEVELS;
    int local;
    const int lin = xcd_tile_runs(blockIdx.x, P.tile_off[RTVM_NLEVELS]);
    if (lin < 0) return;
    const int l = tile_level(P, lin, &local);
    const int W = P.w[l], H = P.h[l], pitch = P.pitch[l];
    const int tiles_x = (W - 2 * EDGE + 63) / 64;
    const int tx0 = EDGE + (local % tiles_x) * 64, ty0 = EDGE + (local / tiles_x) * 32;
    const uint8_t* __restrict__ img = SL.pyr + P.off[l];
    // pixels: rows ty0-4 .. ty0+35, columns tx0-7 .. tx0+72 (tx0-7 is a multiple of 4)
    __shared__ __align__(16) uint8_t s[40][80];
    __shared__ __align__(4) uint8_t sc[34][68];
    __shared__ int s_hist[256];
    __shared__ uint32_t s_list[512];
    __shared__ uint16_t s_surv[34 * 66], s_quick[34 * 66];
    __shared__ int s_cnt, s_base, s_nsurv, s_nquick;
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    if (tid == 0) s_cnt = 0, s_nsurv = 0, s_nquick = 0;
    for (int i = tid; i < 40 * 20; i += 256) {
        const int r = i / 20, q = i % 20;
        const int y = min(ty0 - 4 + r, H - 1), x = min(tx0 - 7 + 4 * q, pitch - 4);
        *reinterpret_cast<uint32_t*>(&s[r][4 * q]) = *reinterpret_cast<const uint32_t*>(img + (size_t)y * pitch + x);
    }
    __syncthreads();
    // score-tile position (r, c) is pixel (tx0 - 1 + c, ty0 - 1 + r) = LDS (r + 3, c + 6).
    // First the two axis pairs on every position (five pixels), the few positions that pass into a
    // list; then the eight-pair necessary test on that list, its survivors into a second list ...
    // A thread takes four neighbouring positions of a row at a time: their centres, the pixels three
    // rows above and below and three columns to the left and right are five words cut out of aligned
    // LDS words, and the test runs on 16-bit halves, two positions per operation:
    //   max(v - p, v - q) >  t  <=>  (min(p, q) + t) - v < 0
    //   min(v - p, v - q) < -t  <=>  (v + t) - max(p, q) < 0        (differences of at most 275: no wrap)
    for (int g0 = 0; g0 < 34 * 17; g0 += 256) {
        const int gi = g0 + tid;
        uint32_t flags = 0;
        int r = 0, c = 0;
        if (gi < 34 * 17) {
            r = gi / 17, c = 4 * (gi % 17);
            *reinterpret_cast<uint32_t*>(&sc[r][c]) = 0;
            const uint32_t* mid = reinterpret_cast<const uint32_t*>(&s[r + 3][c]);  // columns c .. c + 15
            const uint32_t* top = reinterpret_cast<const uint32_t*>(&s[r][c + 4]);  // columns c + 4 .. c + 11
            const uint32_t* bot = reinterpret_cast<const uint32_t*>(&s[r + 6][c + 4]);
            const uint32_t m0 = mid[0], m1 = mid[1], m2 = mid[2], m3 = mid[3];
            const uint32_t t0 = top[0], t1 = top[1], b0 = bot[0], b1 = bot[1];
            // byte selects: the pixels of positions c, c + 2 (even) and c + 1, c + 3 (odd) as 16-bit halves.
            // perm(hi, lo, sel) picks bytes 0..3 of lo, 4..7 of hi, 0x0c = zero.
            // centres: columns c + 6 .. c + 9 of (m2:m1) = bytes 2 .. 5
            const uint32_t Ve = __builtin_amdgcn_perm(m2, m1, 0x0c040c02u), Vo = __builtin_amdgcn_perm(m2, m1, 0x0c050c03u);
            // left: columns c + 3 .. c + 6 of (m1:m0) = bytes 3 .. 6; right: c + 9 .. c + 12 of (m3:m2) = bytes 1 .. 4
            const uint32_t Le = __builtin_amdgcn_perm(m1, m0, 0x0c050c03u), Lo = __builtin_amdgcn_perm(m1, m0, 0x0c060c04u);
            const uint32_t Re = __builtin_amdgcn_perm(m3, m2, 0x0c030c01u), Ro = __builtin_amdgcn_perm(m3, m2, 0x0c040c02u);
            // above and below: columns c + 6 .. c + 9 of (t1:t0) = bytes 2 .. 5
            const uint32_t Te = __builtin_amdgcn_perm(t1, t0, 0x0c040c02u), To = __builtin_amdgcn_perm(t1, t0, 0x0c050c03u);
            const uint32_t Be = __builtin_amdgcn_perm(b1, b0, 0x0c040c02u), Bo = __builtin_amdgcn_perm(b1, b0, 0x0c050c03u);
            const uint32_t fe = quick_pair(Ve, Te, Be, Le, Re);
            const uint32_t fo = quick_pair(Vo, To, Bo, Lo, Ro);
            // positions c, c + 2 in the halves of fe, c + 1, c + 3 in those of fo
            flags = ((fe >> 15) & 1u) | ((fo >> 14) & 2u) | ((fe >> 29) & 4u) | ((fo >> 28) & 8u);
            if (c == 64) flags &= 3u;  // the score tile is 66 positions wide
        }
        // place in the list (its order does not matter): per position of the group, the lanes before
        // this one that pass; waves in the order they arrive
        const unsigned long long q0 = __ballot(flags & 1u), q1 = __ballot(flags & 2u), q2 = __ballot(flags & 4u),
                                 q3 = __ballot(flags & 8u);
        const int c0 = __popcll(q0), c1 = __popcll(q1), c2 = __popcll(q2), total = c0 + c1 + c2 + __popcll(q3);
        if (total) {
            int base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&s_nquick, total);
            base = __builtin_amdgcn_readfirstlane(base);
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
            const uint16_t pos = (uint16_t)(r * 68 + c);
            if (flags & 1u) s_quick[base + __popcll(q0 & below)] = pos;
            if (flags & 2u) s_quick[base + c0 + __popcll(q1 & below)] = (uint16_t)(pos + 1);
            if (flags & 4u) s_quick[base + c0 + c1 + __popcll(q2 & below)] = (uint16_t)(pos + 2);
            if (flags & 8u) s_quick[base + c0 + c1 + c2 + __popcll(q3 & below)] = (uint16_t)(pos + 3);
        }
    }
    __syncthreads();
    const int nquick = s_nquick;
    for (int i0 = 0; i0 < nquick; i0 += 256) {
        const int i = i0 + tid;
        bool cand = false;
        int pos = 0;
        if (i < nquick) {
            pos = s_quick[i];
            cand = fast_candidate<80>(s, pos / 68 + 3, pos % 68 + 6);
        }
        const unsigned long long m = __ballot(cand);
        if (m) {
            int base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&s_nsurv, __popcll(m));
            base = __shfl(base, 0);
            if (cand) s_surv[base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)pos;
        }
    }
    __syncthreads();
    // ... then the arc scores of the survivors, one per lane
    const int nsurv = s_nsurv;
    for (int i = tid; i < nsurv; i += 256) {
        const int pos = s_surv[i], r = pos / 68, c = pos % 68;
        sc[r][c] = (uint8_t)fast_score_at<80>(s, r + 3, c + 6);
    }
    __syncthreads();
    // only a survivor can be a maximum: the 3x3 test runs on the list (positions of the tile proper,
    // inside the level's border), all eight neighbours read at once
    for (int i = tid; i < nsurv; i += 256) {
        const int pos = s_surv[i], r = pos / 68, c = pos % 68;
        const int x = tx0 + c - 1, y = ty0 + r - 1;
        if (r < 1 || r > 32 || c < 1 || c > 64 || x >= W - EDGE || y >= H - EDGE) continue;
        const int v = sc[r][c];
        const int n0 = sc[r - 1][c - 1], n1 = sc[r - 1][c], n2 = sc[r - 1][c + 1], n3 = sc[r][c - 1];
        const int n4 = sc[r][c + 1], n5 = sc[r + 1][c - 1], n6 = sc[r + 1][c], n7 = sc[r + 1][c + 1];
        if (v && v > max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)))) {
            const int at = atomicAdd(&s_cnt, 1);
            s_list[at] = ((uint32_t)v << 24) | ((uint32_t)y << 12) | (uint32_t)x;
            atomicAdd(&s_hist[v], 1);
        }
    }
    __syncthreads();
    const int n = s_cnt;  // a strict 3x3 maximum: at most one per 2x2 block, 512 per tile
    if (n == 0) return;
    if (tid == 0) s_base = atomicAdd(&cand_count[l], n);
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < n; i += 256)
        if (base + i < P.cand_cap[l]) cand[P.cand_off[l] + base + i] = s_list[i];
    const int hcount = s_hist[tid];
    if (hcount) atomicAdd(&hist[l * 256 + tid], hcount);
}

// ------------------------------------------------------------------ selection
// The 9 x 9 pixels around the point as nine 12-byte loads (a byte load per sample would touch a cache line
// per lane and sample: 189 of them per point), the sums from registers with compile-time byte positions.
__device__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x0, int y0) {
    uint32_t w[9][3];   // byte k of row r = pixel (x0 - 4 + k, y0 - 4 + r)
    const uint8_t* base = img + (size_t)(y0 - 4) * pitch + (x0 - 4);
#pragma unroll
    for (int r = 0; r < 9; r++) __builtin_memcpy(w[r], base + (size_t)r * pitch, 12);
    auto px = [&](int r, int k) { return (int)((w[r][k >> 2] >> (8 * (k & 3))) & 255u); };
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int i = -3; i <= 3; i++) {
        const int r1 = i + 4, r0 = r1 - 1, r2 = r1 + 1;
#pragma unroll
        for (int j = -3; j <= 3; j++) {
            const int k = j + 4;
            const int Ix = (px(r1, k + 1) - px(r1, k - 1)) * 2 + (px(r0, k + 1) - px(r0, k - 1)) + (px(r2, k + 1) - px(r2, k - 1));
            const int Iy = (px(r2, k) - px(r0, k)) * 2 + (px(r2, k - 1) - px(r0, k - 1)) + (px(r2, k + 1) - px(r0, k + 1));
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

// One workgroup per level.  KeyPointsFilter::retainBest keeps everything whose
// response is >= the n-th largest; a point is kept iff fewer than n points have a
// strictly larger response.  Survivors keep row-major order.
__global__ __launch_bounds__(1024) void k_orb_select(OrbDev P, const FrameSlot* __restrict__ slots, int sb) {
    const FrameSlot& SL = slots[sb + blockIdx.y];
    const uint8_t* __restrict__ pyr = SL.pyr;
    const uint32_t* __restrict__ cand = SL.cand;
    const int* __restrict__ cand_count = SL.cand_count;
    const int* __restrict__ hist = SL.cand_count + RTVM_NLEVELS;
    int* __restrict__ sel_count = SL.sel_count;
    int32_t* __restrict__ sel_lxy = SL.sel_lxy;
    float* __restrict__ sel_resp = SL.sel_resp;
    const int l = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t s_key[SEL_CAP];
    __shared__ float s_resp[SEL_CAP];
    __shared__ int s_n, s_thr, s_base;
    __shared__ int s_wave[16];
    const int quota = P.quota[l];
    const int n_cand = min(cand_count[l], P.cand_cap[l]);
    // The score histogram of the level and its sums from the top: S[t] = candidates with a score of at least t.
    // (One thread walking down the histogram in memory until the sum reaches the quota was a chain of some two
    // hundred dependent loads: most of the kernel's time.)
    __shared__ int s_h[256], s_suf[256], s_wsum[4], s_wthr[4];
    const int want = 2 * quota;
    if (tid < 256) {
        const int hv = hist[l * 256 + tid];
        s_h[tid] = hv;
        int v = hv;   // inclusive sum over the lanes at and after this one
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_down(v, d);
            if (lane + d < 64) v += o;
        }
        s_suf[tid] = v;
        if (lane == 0) s_wsum[wave] = v;
    }
    __syncthreads();
    if (tid < 256) {
        int v = s_suf[tid];
        for (int w = wave + 1; w < 4; w++) v += s_wsum[w];
        s_suf[tid] = v;
        // the largest score t of this wave's 64 with S[t] >= want
        const unsigned long long b = __ballot(v >= want);
        if (lane == 0) s_wthr[wave] = b ? 64 * wave + 63 - __clzll((long long)b) : -1;
    }
    __syncthreads();
    if (tid == 0) {
        s_n = 0;
        s_base = 0;
        int thr = 0;
        if (want == 0) thr = 256;
        else if (n_cand > want) {
            for (int w = 3; w >= 0; w--)
                if (s_wthr[w] >= 0) {
                    thr = s_wthr[w];
                    break;
                }
            int cum = s_suf[thr];   // all candidates when no score reaches the quota (thr = 0)
            // more ties at the cut than the workgroup can hold: raise the cut
            while (cum > SEL_CAP && thr < 256) {
                cum -= s_h[thr];
                thr++;
            }
        } else if (n_cand > SEL_CAP) {
            int cum = n_cand;
            while (cum > SEL_CAP && thr < 256) {
                cum -= s_h[thr];
                thr++;
            }
        }
        s_thr = thr;
    }
    __syncthreads();
    const int thr = s_thr;
    for (int i = tid; i < n_cand; i += 1024) {
        const uint32_t e = cand[P.cand_off[l] + i];
        if ((int)(e >> 24) >= thr) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < SEL_CAP) s_key[pos] = e & 0x00FFFFFFu;
        }
    }
    __syncthreads();
    const int n = min(s_n, SEL_CAP);
    // raster order: ascending (y << 12 | x)
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += 1024) s_key[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = s_key[i], b = s_key[ixj];
                    if ((a > b) == ((i & k) == 0)) {
                        s_key[i] = b;
                        s_key[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    const uint8_t* img = pyr + P.off[l];
    for (int i = tid; i < n; i += 1024) {
        const uint32_t e = s_key[i];
        s_resp[i] = harris_response(img, P.pitch[l], (int)(e & 0xFFFu), (int)(e >> 12));
    }
    __syncthreads();
    const int out_cap = quota + 64;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        if (i < n) {
            keep = true;
            if (n > quota) {
                const float r = s_resp[i];
                int greater = 0;
                for (int j = 0; j < n; j++) greater += s_resp[j] > r;
                keep = greater < quota;
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wave[w];
        const int base = s_base;
        if (keep) {
            const int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < out_cap) {
                const uint32_t e = s_key[i];
                sel_lxy[2 * (P.sel_off[l] + pos)] = (int)(e & 0xFFFu);
                sel_lxy[2 * (P.sel_off[l] + pos) + 1] = (int)(e >> 12);
                sel_resp[P.sel_off[l] + pos] = s_resp[i];
            }
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wave[w];
            s_base = base + tot;
        }
        __syncthreads();
    }
    if (tid == 0) sel_count[l] = min(s_base, out_cap);
}

// ------------------------------------------------------------------ describe
__device__ float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
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

// One wave per output keypoint: orientation by intensity centroid on the level
// image (31 lanes sum one row of the disc each), then the 256 steered BRIEF tests
// on the blurred level (4 per lane, packed through ballot-free shuffles).
__global__ __launch_bounds__(64) void k_orb_describe(OrbDev P, const FrameSlot* __restrict__ slots, int sb) {
    const FrameSlot& SL = slots[sb + blockIdx.y];
    const uint8_t* __restrict__ pyr = SL.pyr;
    const int* __restrict__ sel_count = SL.sel_count;
    const int32_t* __restrict__ sel_lxy = SL.sel_lxy;
    const float* __restrict__ sel_resp = SL.sel_resp;
    const FeatSet F = SL.feat;
    const int slot = blockIdx.x, lane = threadIdx.x;
    int l = -1, idx = 0, total = 0;
#pragma unroll
    for (int k = 0; k < RTVM_NLEVELS; k++) {
        const int c = sel_count[k];
        if (l < 0 && slot < total + c) {
            l = k;
            idx = slot - total;
        }
        total += c;
    }
    if (slot == 0 && lane == 0) *F.count = min(total, P.max_kp);
    if (l < 0 || slot >= P.max_kp) return;
    const int x0 = sel_lxy[2 * (P.sel_off[l] + idx)], y0 = sel_lxy[2 * (P.sel_off[l] + idx) + 1];
    const int pitch = P.pitch[l];
    const uint8_t* img = pyr + P.off[l];
    // The patch of the level the keypoint reads, as 16-byte loads into LDS instead of a byte gather per sample (a
    // gather of 64 bytes from 64 rows touches 64 cache lines: the kernel was bound by that): DESC_R + 3 rows and
    // DESC_R + 5 columns on every side of (x0, y0) and more to the right (rows of 64 bytes).  It serves the moments
    // (radius 15) and the tests: the pattern's points lie within 18.4 pixels of the centre, and the 7 x 7 Gaussian
    // of cv::ORB's blurred image is evaluated at the 512 points the tests read, not on the whole pyramid.
    // Keypoints keep 31 pixels from the border, so neither the patch nor a 7 x 7 window meets a mirrored pixel
    // (bytes past a row's 45th column are staged but never used; the plane has a spare row behind its last one).
    constexpr int DESC_R = 19, PATCH_Y = DESC_R + 3, PATCH_X = DESC_R + 5, PATCH_ROWS = 2 * PATCH_Y + 1, PATCH_PITCH = 64;
    __shared__ __align__(16) uint8_t s_raw[PATCH_ROWS][PATCH_PITCH];
    {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
        for (int i = lane; i < 4 * PATCH_ROWS; i += 64) {
            const int r = i >> 2, sg = i & 3;
            const uint8_t* g = img + (size_t)(y0 - PATCH_Y + r) * pitch + (x0 - PATCH_X + 16 * sg);
            *reinterpret_cast<u32x4*>(&s_raw[r][16 * sg]) = *reinterpret_cast<const u32x4u*>(g);
        }
    }
    wave_lds_phase();
    // moments over the disc of radius 15: lane v+15 handles row v
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        const int v = lane - 15;
        const int d = P.umax[v < 0 ? -v : v];
        int sum = 0;
#pragma unroll
        for (int u = -15; u <= 15; u++) {
            const int raw = s_raw[PATCH_Y - 15 + lane][PATCH_X + u];   // read unconditionally: no branch per sample
            const int val = (u < 0 ? -u : u) <= d ? raw : 0;
            m10 += u * val;
            sum += val;
        }
        m01 = v * sum;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        m10 += __shfl_xor(m10, m);
        m01 += __shfl_xor(m01, m);
    }
    const float angle = fast_atan2((float)m01, (float)m10);
    const float sf = P.scale[l];
    if (lane == 0) {
        F.xy[2 * slot] = (float)x0 * sf;
        F.xy[2 * slot + 1] = (float)y0 * sf;
        F.size[slot] = 31 * sf;
        F.angle[slot] = angle;
        F.response[slot] = sel_resp[P.sel_off[l] + idx];
        F.octave[slot] = l;
        F.lxy[2 * slot] = x0;
        F.lxy[2 * slot + 1] = y0;
    }
    float ang = angle;
    ang *= (float)(M_PI / 180.f);
    const float a = (float)cos((double)ang), b = (float)sin((double)ang);
    // cv::GaussianBlur(7 x 7, sigma 2) of the level at patch position (ix, iy) relative to the keypoint: seven
    // row sums of seven bytes (two v_dot4_u32_u8 on the bytes cut from three aligned words), their weighted sum,
    // divided by 65536 with halves rounded to even and saturated: integer taps cvRound(k * 256), exact sums.
    const uint32_t k0 = P.gk7[0], k1 = P.gk7[1], k2 = P.gk7[2], k3 = P.gk7[3], k4 = P.gk7[4], k5 = P.gk7[5], k6 = P.gk7[6];
    const uint32_t tlo = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), thi = k4 | (k5 << 8) | (k6 << 16);
    auto blurred = [&](int ix, int iy) -> int {
        const int c = ix + PATCH_X - 3, r = iy + PATCH_Y - 3;   // first byte and first row of the window
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&s_raw[r][c & ~3]);
        const uint32_t sh = (uint32_t)(c & 3);
        const uint32_t kt[7] = {k0, k1, k2, k3, k4, k5, k6};
        uint32_t acc = 0;
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const uint32_t w0 = w[t * (PATCH_PITCH / 4)], w1 = w[t * (PATCH_PITCH / 4) + 1], w2 = w[t * (PATCH_PITCH / 4) + 2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi4 = __builtin_amdgcn_alignbyte(w2, w1, sh);
            const uint32_t rs = __builtin_amdgcn_udot4(hi4, thi, __builtin_amdgcn_udot4(lo, tlo, 0u, false), false);
            acc += __umul24(kt[t], rs);   // an 8-bit tap and a 16-bit row sum: the full-rate multiply
        }
        return (int)min((acc + 0x7FFFu + ((acc >> 16) & 1u)) >> 16, 255u);
    };
    unsigned nib = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int8_t* p = &c_pattern[(lane * 4 + t) * 4];
        const float px1 = (float)p[0], py1 = (float)p[1], px2 = (float)p[2], py2 = (float)p[3];
        const float x1 = px1 * a - py1 * b, y1 = px1 * b + py1 * a;
        const float x2 = px2 * a - py2 * b, y2 = px2 * b + py2 * a;
        // (the clamps never bind: they keep an index inside the patch whatever the arithmetic does)
        const int ix1 = min(max(__float2int_rn(x1), -DESC_R), DESC_R), iy1 = min(max(__float2int_rn(y1), -DESC_R), DESC_R);
        const int ix2 = min(max(__float2int_rn(x2), -DESC_R), DESC_R), iy2 = min(max(__float2int_rn(y2), -DESC_R), DESC_R);
        const int t0 = blurred(ix1, iy1);
        const int t1 = blurred(ix2, iy2);
        nib |= (unsigned)(t0 < t1) << t;
    }
    const unsigned hi = __shfl_down(nib, 1);
    if ((lane & 1) == 0) F.desc[(size_t)slot * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

}  // namespace

// ---------------------------------------------------------------------- host
static inline int cv_round_d(double v) { return (int)std::nearbyint(v); }

static void lin_coeffs(int src, int dst, uint16_t* ofs, uint16_t* c1) {
    const double inv_scale = (double)dst / src;
    const double scale = 1.0 / inv_scale;
    for (int v = 0; v < dst; v++) {
        const double f = scale * ((double)v + 0.5) - 0.5;
        const int i = (int)std::floor(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                ofs[v] = (uint16_t)i;
                c1[v] = (uint16_t)cv_round_d((f - (double)i) * 256.0);
            } else {
                ofs[v] = (uint16_t)(src - 1);
                c1[v] = 0xFFFF;
            }
        } else {
            ofs[v] = 0;
            c1[v] = 0xFFFF;
        }
    }
}

// Level geometry, quotas and tables of cv::ORB (nlevels 8, scaleFactor 1.2f).
int rtvm_orb_plan(rtvm_ctx* c) {
    OrbDev& P = c->orb;
    memset(&P, 0, sizeof(P));
    const int W = c->fw, H = c->fh, n = c->nfeatures;
    if (W > 4095 || H > 4095) RTVM_FAIL(c, RTVM_E_ARG, "ORB: frame side above 4095 is not supported");
    size_t off = 0;
    int cand = 0, sel = 0, xt = 0, yt = 0, tiles = 0, btiles = 0;
    for (int l = 0; l < RTVM_NLEVELS; l++) {
        const float scale = (float)std::pow((double)1.2f, (double)l);
        const float inv = 1.0f / scale;
        P.scale[l] = scale;
        P.w[l] = cv_round_d(W * inv);
        P.h[l] = cv_round_d(H * inv);
        P.pitch[l] = (P.w[l] + 63) & ~63;
        P.off[l] = (unsigned)off;
        off += (size_t)P.pitch[l] * (P.h[l] + 1);
        off = (off + 255) & ~(size_t)255;
        P.cand_off[l] = cand;
        P.cand_cap[l] = (P.w[l] * P.h[l]) / 4 + 1024;
        cand += P.cand_cap[l];
        P.xtab_off[l] = xt;
        P.ytab_off[l] = yt;
        xt += ((P.w[l] + 3) & ~3) + 4;   // entries up to the next multiple of 4 exist (the last one repeated)
        yt += P.h[l];
        P.tile_off[l] = tiles;
        if (P.w[l] > 2 * EDGE && P.h[l] > 2 * EDGE)
            tiles += ((P.w[l] - 2 * EDGE + 63) / 64) * ((P.h[l] - 2 * EDGE + 31) / 32);
        P.btile_off[l] = btiles;
        btiles += ((P.w[l] + 63) / 64) * ((P.h[l] + 31) / 32);
    }
    P.tile_off[RTVM_NLEVELS] = tiles;
    P.btile_off[RTVM_NLEVELS] = btiles;
    c->orb_pyr_bytes = off;
    c->orb_rc_total = ((P.w[3] + RC_TW - 1) / RC_TW) * ((P.h[3] + RC_TH - 1) / RC_TH) * 4 +
                      ((P.w[7] + RC_TW - 1) / RC_TW) * ((P.h[7] + RC_TH - 1) / RC_TH) * 5;
    c->orb_cand_total = cand;
    c->orb_xtab_total = xt;
    c->orb_ytab_total = yt;
    {  // nfeaturesPerLevel
        const float factor = (float)(1.0 / (double)1.2f);
        float nd = n * (1 - factor) / (1 - (float)std::pow((double)factor, (double)RTVM_NLEVELS));
        int sum = 0;
        for (int l = 0; l < RTVM_NLEVELS - 1; l++) {
            P.quota[l] = (int)std::nearbyintf(nd);
            sum += P.quota[l];
            nd *= factor;
        }
        P.quota[RTVM_NLEVELS - 1] = n - sum > 0 ? n - sum : 0;
    }
    for (int l = 0; l < RTVM_NLEVELS; l++) {
        P.sel_off[l] = sel;
        sel += P.quota[l] + 64;
    }
    c->orb_sel_total = sel;
    P.max_kp = n + 64 * RTVM_NLEVELS;
    {  // umax of the circular patch (half size 15)
        const int hp = 15;
        int v, v0, vmax = (int)std::floor(hp * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(hp * std::sqrt(2.f) / 2);
        for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round_d(std::sqrt((double)hp * hp - v * v));
        for (v = hp, v0 = 0; v >= vmin; --v) {
            while (P.umax[v0] == P.umax[v0 + 1]) ++v0;
            P.umax[v] = v0;
            ++v0;
        }
    }
    {  // 7-tap Gaussian, sigma 2, as 8-bit fixed point
        double t[7], sum = 0;
        const double sigma = 2.0, scale2 = -0.5 / (sigma * sigma);
        for (int i = 0; i < 7; i++) {
            const double x = i - 3.0;
            t[i] = std::exp(scale2 * x * x);
            sum += t[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) P.gk7[i] = (int)std::nearbyintf((float)(t[i] * sum) * 256.f);
    }
    return RTVM_OK;
}

// Resize tables of levels 1..7 (level l is resized from level l-1).
int rtvm_orb_setup(rtvm_ctx* c) {
    const OrbDev& P = c->orb;
    std::vector<uint16_t> xo(c->orb_xtab_total + 1), xw(c->orb_xtab_total + 1), yo(c->orb_ytab_total + 1),
        yw(c->orb_ytab_total + 1);
    for (int l = 1; l < RTVM_NLEVELS; l++) {
        lin_coeffs(P.w[l - 1], P.w[l], &xo[P.xtab_off[l]], &xw[P.xtab_off[l]]);
        for (int v = P.w[l]; v < ((P.w[l] + 3) & ~3) + 4; v++) {
            xo[P.xtab_off[l] + v] = xo[P.xtab_off[l] + P.w[l] - 1];
            xw[P.xtab_off[l] + v] = xw[P.xtab_off[l] + P.w[l] - 1];
        }
        lin_coeffs(P.h[l - 1], P.h[l], &yo[P.ytab_off[l]], &yw[P.ytab_off[l]]);
    }
    HIP_TRY(c, hipMemcpy(c->d_xofs, xo.data(), xo.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_xw, xw.data(), xw.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_yofs, yo.data(), yo.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_yw, yw.data(), yw.size() * 2, hipMemcpyHostToDevice));
    // The chained resize takes an output's taps out of the eight bytes that start at its quad's first source
    bool sel_ok = true;
    std::vector<uint32_t> xsel(c->orb_xtab_total + 4, 0u), xwp(c->orb_xtab_total + 4, 0u);
    for (int l = 1; l < RTVM_NLEVELS; l++)
        for (int v = 0; v < ((P.w[l] + 3) & ~3) + 4; v++) {
            const int t = P.xtab_off[l] + v;
            const int off = (int)xo[t] - (int)xo[P.xtab_off[l] + (v & ~3)];
            const bool single = xw[t] == 0xFFFFu;
            const int off1 = single ? off : off + 1;
            if (off < 0 || off1 > 7) sel_ok = false;
            xsel[t] = 0x0c000c00u | ((uint32_t)(off1 & 7) << 16) | (uint32_t)(off & 7);
            const uint32_t cx = single ? 0u : xw[t];
            xwp[t] = (256u - cx) | (cx << 16);
        }
    HIP_TRY(c, hipMemcpy(c->d_xsel, xsel.data(), xsel.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_xwp, xwp.data(), xwp.size() * 4, hipMemcpyHostToDevice));
    // Levels 1-3 and 4-7 as two chained launches: every rectangle of every tile must fit the staging buffers
    // (checked here for every tile; true for any frame size the pyramid factor 1.2 leaves above a few pixels).
    // Frames of fewer than 29 pixels a side have levels too small for the chains, and no level of theirs is
    // wide enough for a keypoint (63 pixels): their pyramid is not built at all.
    c->orb_chain = sel_ok ? 1 : 0;
    const int chain_a[2] = {0, 3}, chain_n[2] = {3, 4};
    std::vector<RcRegion> regs;
    regs.reserve(c->orb_rc_total + 1);
    for (int g = 0; g < 2; g++) {
        const int a = chain_a[g], n = chain_n[g], b = a + n;
        if (P.w[b] < 8 || P.h[b] < 8) c->orb_chain = 0;
        for (int ty = 0; ty < P.h[b]; ty += RC_TH)      // tiles in the kernel's order
            for (int tx = 0; tx < P.w[b]; tx += RC_TW) {
                RcRegion R[RC_MAXN + 1];
                rc_regions(P, a, n, tx, ty, xo.data(), yo.data(), R);
                for (int k = 0; k <= n; k++) {
                    const int room = (k & 1) ? (n == 3 ? RC3_B1 : RC4_B1) : (n == 3 ? RC3_B0 : RC4_B0);
                    if (R[k].w <= 0 || R[k].h <= 0 || R[k].w * R[k].h + 16 > room || R[k].x0 < 0 || R[k].y0 < 0 ||
                        R[k].x0 + R[k].w > P.pitch[a + k] || R[k].w * R[k].h / 4 >= 4096)
                        c->orb_chain = 0;
                    regs.push_back(R[k]);
                }
            }
    }
    if ((int)regs.size() != c->orb_rc_total) RTVM_FAIL(c, RTVM_E_STATE, "resize chain: %d rectangles, %d planned", (int)regs.size(), c->orb_rc_total);
    HIP_TRY(c, hipMemcpy(c->d_rc_regions, regs.data(), regs.size() * sizeof(RcRegion), hipMemcpyHostToDevice));
    if (!c->orb_chain && P.tile_off[RTVM_NLEVELS] > 0)
        RTVM_FAIL(c, RTVM_E_ARG, "ORB pyramid: a %d x %d frame does not fit the resize kernels' staging buffers", P.w[0], P.h[0]);
    return RTVM_OK;
}

// Queues detectAndCompute for the frames of slots sb .. sb+nb-1 (slot.bgr -> slot.feat).
int rtvm_launch_orb(rtvm_ctx* c, int sb, int nb) {
    const OrbDev& P = c->orb;
    const FrameSlot* sl = c->d_slots;
    RTVM_LAUNCH(c, k_gray, dim3((P.w[0] + 1023) / 1024, P.h[0], nb), dim3(256), 0, sl, sb, P.off[0], P.w[0], P.h[0],
                P.pitch[0], rtvm_gray_coef(c));
    if (c->orb_chain) {
        static_assert(RTVM_NLEVELS == 8, "the chains cover levels 1-3 and 4-7");
        RTVM_LAUNCH(c, k_resize_chain<3>, dim3(xcd_grid(((P.w[3] + RC_TW - 1) / RC_TW) * ((P.h[3] + RC_TH - 1) / RC_TH)), 1, nb),
                    dim3(256), 0, P, sl, sb, 0, c->d_xofs, c->d_xsel, c->d_xwp, c->d_yofs, c->d_yw, (const RcRegion*)c->d_rc_regions);
        RTVM_LAUNCH(c, k_resize_chain<4>, dim3(xcd_grid(((P.w[7] + RC_TW - 1) / RC_TW) * ((P.h[7] + RC_TH - 1) / RC_TH)), 1, nb),
                    dim3(256), 0, P, sl, sb, 3, c->d_xofs, c->d_xsel, c->d_xwp, c->d_yofs, c->d_yw,
                    (const RcRegion*)c->d_rc_regions + ((P.w[3] + RC_TW - 1) / RC_TW) * ((P.h[3] + RC_TH - 1) / RC_TH) * 4);
    }
    if (P.tile_off[RTVM_NLEVELS] > 0) {
        RTVM_LAUNCH(c, k_fast, dim3(xcd_grid_runs(P.tile_off[RTVM_NLEVELS]), nb), dim3(256), 0, P, sl, sb);
    }
    RTVM_LAUNCH(c, k_orb_select, dim3(RTVM_NLEVELS, nb), dim3(1024), 0, P, sl, sb);
    RTVM_LAUNCH(c, k_orb_describe, dim3(P.max_kp, nb), dim3(64), 0, P, sl, sb);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}
