// 3x3 chamfer distance transform (cv2.distanceTransform(mask, DIST_L2, 3),
// reference main.py:888-889) for gfx950.
//
// OpenCV's algorithm is two sequential raster passes; its result equals the
// closed form  d(p) = min over zero pixels z of  a*(max(|dx|,|dy|) - min(|dx|,|dy|))
// + b*min(|dx|,|dy|)  in Q16 (a = 62587, b = 89738).  Every shortest chamfer
// path can be ordered "diagonal steps first, then axial steps", so
//     E(q) = b * (steps from q to the nearest zero along the two diagonals through q)
//     d(p) = min( min over q in p's row    of E(q) + a*|dx| ,
//                 min over q in p's column of E(q) + a*|dy| )
// which is three families of independent 1-D min-plus scans: along diagonals,
// rows and columns.  All arithmetic is integer, so the result is bit-exact.
//
// The image is cut into chunks of 32 rows.  A diagonal line crosses a chunk in 32
// pixels, whose zero flags form one 32-bit word; a column of a chunk is 32 values
// held in the registers of one lane.
//   k_dt_bits    mask bytes -> one zero bit per pixel (32 pixels per lane)
//   k_dt_diag_a  per (line, chunk): the 32 zero bits of the line inside the chunk, by
//                32x32 bit-matrix transposes of windows of the bit rows; per (line,
//                16 chunks): which chunks hold a zero
//   k_dt_diag_b  per tile of 32 rows x 192 columns: steps to the nearest zero along
//                both diagonals (bit scans inside the chunk, the chunk summaries
//                beyond it), minimum of the two through LDS, one u16 plane out
//   k_dt_scan    row scans of the output window's rows (wave prefix minima), and
//                per (column, chunk) the values leaving the chunk downwards / upwards
//   k_dt_cols    carries from the other chunks' summaries, column scan in registers,
//                min with the row scan, distance out
#include "common.h"

namespace {

constexpr uint32_t DT_A = 62587u;              // cvRound(0.955  * 65536)
constexpr uint32_t DT_B = 89738u;              // cvRound(1.3693 * 65536)
constexpr uint32_t DT_MAX = 0xFFFFFFFFu - DT_B;  // OpenCV's DIST_MAX
constexpr uint32_t INF32 = 0xFFFFFFFFu;
constexpr int CH = 32;                         // rows per chunk
constexpr int GR = 16;                         // chunks per summary group
constexpr int TW = 192;                        // columns per tile of k_dt_diag_b: TW + 31 lines fit 256 threads
constexpr int DB_THREADS = 256;

__device__ __forceinline__ uint32_t sat_add(uint32_t v, uint32_t k) {
    return v > INF32 - k ? INF32 : v + k;
}

// E(q) in Q16 from the step plane (0xFFFF = no zero on either diagonal).
__device__ __forceinline__ uint32_t e_of(uint32_t s) {
    return s == 0xFFFFu ? INF32 : s * DT_B;
}

// minimum of the two 16-bit halves of a and b
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 r = __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(uint32_t, r);
}

// A launch serves jobs[0 .. n_first) and then jobs[second ...]: the new-mask and the
// old-mask jobs of a batch live in two arrays.
__device__ __forceinline__ int job_index(int i, int n_first, int second) {
    return i < n_first ? i : second + (i - n_first);
}

// 4 mask bytes -> 4 bits (bit i set iff byte i is zero)
__device__ __forceinline__ uint32_t zero_nibble(uint32_t v) {
    const uint32_t t = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);  // 0x80 in every zero byte
    return (((t >> 7) * 0x01020408u) >> 24) & 0xFu;
}

// One lane per 32 pixels of a row of the rectangle: bit i of word w of row y is set iff
// pixel (32 w + i, y) lies inside the rectangle and is zero.
__global__ __launch_bounds__(256) void k_dt_bits(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z, n_first, second)]);
    const int w0 = jb.x0 >> 5;
    const int w = w0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = jb.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= jb.y1 || w * 32 >= jb.x1) return;
    const int xa = w * 32;
    const auto row = jb.mask + (size_t)y * jb.pitch;
    uint32_t bits = 0;
    if (xa + 32 <= jb.pitch && (((uintptr_t)(row + xa) & 3) == 0)) {
        const auto p = reinterpret_cast<gptr<const uint32_t>>(row + xa);
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = p[i];
#pragma unroll
        for (int i = 0; i < 8; i++) bits |= zero_nibble(v[i]) << (4 * i);
    } else {
        for (int i = 0; i < 32; i++)
            if (xa + i < jb.pitch && row[xa + i] == 0) bits |= 1u << i;
    }
    // pixels outside the rectangle are walls, not zeros
    if (xa < jb.x0) bits &= 0xFFFFFFFFu << (jb.x0 - xa);
    if (xa + 32 > jb.x1) bits &= 0xFFFFFFFFu >> (xa + 32 - jb.x1);
    jb.B[(size_t)y * jb.wpitch + w] = bits;
}

// Lines of orientation 0 run along x - y = const (x grows with y), those of orientation 1
// along x + y = const.  Line `line` passes through column xs at row y0 of the rectangle.
struct LineGeom {
    int W, H, nlines, nch;
};
__device__ __forceinline__ LineGeom line_geom(const DtJobG& jb) {
    LineGeom g;
    g.W = jb.x1 - jb.x0;
    g.H = jb.y1 - jb.y0;
    g.nlines = g.W + g.H - 1;
    g.nch = (g.H + CH - 1) / CH;
    return g;
}

// blockIdx.z = 2 * job + orientation; a wave takes 64 lines of one chunk at a time, one workgroup
// (four waves, four chunks each) the 16 chunks of a group.  Bit r of line j's word is bit
// (x_top(j) +- r) of bit row r, and x_top grows by one per line: with w_r = the 32 bits of row r
// starting at x_top(first line) +- r, the words of 32 consecutive lines are the transpose of the
// 32x32 bit matrix (w_r).  Each half-wave does one such block: lane r loads its window (two words),
// 32 ballots transpose it.  (Small workgroups on purpose: a workgroup of 16 waves waits for half a
// compute unit to be free at once, which rarely happens while the other streams keep the GPU busy.)
__global__ __launch_bounds__(256) void k_dt_diag_a(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z >> 1, n_first, second)]);
    const int orient = blockIdx.z & 1;
    const LineGeom g = line_geom(jb);
    if (g.W <= 0 || g.H <= 0 || (int)blockIdx.x * 64 >= g.nlines || (int)blockIdx.y * GR >= g.nch) return;
    __shared__ uint32_t s_any[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int line = blockIdx.x * 64 + lane;
    const int grp = lane >> 5, r = lane & 31;
    const int dx = orient ? -1 : 1;
    const int line0 = blockIdx.x * 64 + 32 * grp;
    const int xs0 = orient ? jb.x0 + line0 : jb.x0 - (g.H - 1) + line0;
    const int nw = jb.wpitch - 2;
    // all windows first, so that the loads of the four chunks are in flight together
    uint32_t w[GR / 4];
#pragma unroll
    for (int k = 0; k < GR / 4; k++) {
        const int c = blockIdx.y * GR + wave * (GR / 4) + k;
        const int ya = jb.y0 + c * CH;
        const int xstart = xs0 + dx * (ya - jb.y0) + dx * r;  // first column of row r's window
        const int y = ya + r;
        w[k] = 0;
        if (c < g.nch && y < jb.y1) {
            const int wi = xstart >> 5, sh = xstart & 31;
            const auto rowp = jb.B + (size_t)y * jb.wpitch;
            const uint32_t lo = (wi >= 0 && wi < nw) ? rowp[wi] : 0u;
            const uint32_t hi = (wi + 1 >= 0 && wi + 1 < nw) ? rowp[wi + 1] : 0u;
            uint32_t v = (uint32_t)(((((uint64_t)hi) << 32) | lo) >> sh);
            // columns outside the rectangle are walls (their bits may be stale)
            const int j0 = jb.x0 - xstart, j1 = jb.x1 - xstart;  // window bits [j0, j1) are inside
            if (j0 > 0) v &= j0 >= 32 ? 0u : 0xFFFFFFFFu << j0;
            if (j1 < 32) v &= j1 <= 0 ? 0u : 0xFFFFFFFFu >> (32 - j1);
            w[k] = v;
        }
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < GR / 4; k++) {
        const int c = blockIdx.y * GR + wave * (GR / 4) + k;
        if (c >= g.nch) break;  // wave-uniform
        uint32_t z = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const unsigned long long b = __ballot((w[k] >> j) & 1u);
            const uint32_t half = grp ? (uint32_t)(b >> 32) : (uint32_t)b;
            if (r == j) z = half;
        }
        if (line < g.nlines) jb.Z[((size_t)orient * jb.znch + c) * jb.zpitch + line] = z;
        any |= (z != 0 ? 1u : 0u) << (wave * (GR / 4) + k);
    }
    s_any[wave][lane] = any;
    __syncthreads();
    if (wave == 0 && line < g.nlines)
        jb.Z16[((size_t)orient * jb.zg + blockIdx.y) * jb.zpitch + line] = s_any[0][lane] | s_any[1][lane] | s_any[2][lane] | s_any[3][lane];
}

// Steps to the nearest zero along one line for the 32 rows of chunk c, written into the LDS tile
// of the line's orientation.
__device__ __forceinline__ void diag_steps(const DtJobG& jb, const LineGeom& g, int orient, int c, int line, int col0,
                                           int dcol, uint16_t (*T)[TW]) {
    const auto Z = jb.Z + (size_t)orient * jb.znch * jb.zpitch + line;
    const auto Z16 = jb.Z16 + (size_t)orient * jb.zg * jb.zpitch + line;
    const uint32_t z = Z[(size_t)c * jb.zpitch];
    // nearest chunk with a zero above / below this one
    const int gi = c / GR, ci = c % GR, ng = (g.nch + GR - 1) / GR;
    int ka = -1, kb = -1;
    {
        const uint32_t own = Z16[(size_t)gi * jb.zpitch];
        uint32_t m = own & ((1u << ci) - 1u);
        if (m)
            ka = gi * GR + (31 - __clz((int)m));
        else
            for (int k = gi - 1; k >= 0; k--) {
                m = Z16[(size_t)k * jb.zpitch];
                if (m) {
                    ka = k * GR + (31 - __clz((int)m));
                    break;
                }
            }
        m = own & ~((2u << ci) - 1u);
        if (m)
            kb = gi * GR + (__ffs((int)m) - 1);
        else
            for (int k = gi + 1; k < ng; k++) {
                m = Z16[(size_t)k * jb.zpitch];
                if (m) {
                    kb = k * GR + (__ffs((int)m) - 1);
                    break;
                }
            }
    }
    // distance in rows from the chunk's first row up to the zero above, and from its last
    // row down to the zero below (large = none)
    int up = 1 << 20, down = 1 << 20;
    if (ka >= 0) up = (c - ka) * CH - (31 - __clz((int)Z[(size_t)ka * jb.zpitch]));
    if (kb >= 0) down = (kb - c) * CH + (__ffs((int)Z[(size_t)kb * jb.zpitch]) - 1) - (CH - 1);
    // steps to the nearest zero below (e) and above (run) as running counts that a zero resets:
    // count = (count + 1) & (zero here ? 0 : ~0); both stay below 2^21
    const uint32_t nz = ~z;
    int e[CH];
    int run = down - 1;
#pragma unroll
    for (int r = CH - 1; r >= 0; r--) {
        run = (run + 1) & __builtin_amdgcn_sbfe(nz, r, 1);
        e[r] = run;
    }
    run = up - 1;
#pragma unroll
    for (int r = 0; r < CH; r++) {
        run = (run + 1) & __builtin_amdgcn_sbfe(nz, r, 1);
        const int col = col0 + dcol * r;
        if (col < 0 || col >= TW) continue;
        T[r][col] = (uint16_t)min(min(run, e[r]), 0xFFFF);
    }
}

// One workgroup per tile of 32 rows x TW columns (columns on a grid of TW from 0, so the
// rows of a tile are 16-byte aligned when the pitch is a multiple of 8).  Four waves and 24 KB of
// LDS: a workgroup that needs more than the kernels of the other streams do waits for them.  Tiles that touch
// neither the rows nor the columns of the output window are never read and are skipped.
__global__ __launch_bounds__(DB_THREADS) void k_dt_diag_b(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z, n_first, second)]);
    const LineGeom g = line_geom(jb);
    const int c = blockIdx.y;
    const int X = ((jb.x0 / TW) + (int)blockIdx.x) * TW;
    if (g.W <= 0 || g.H <= 0 || c >= g.nch || X >= jb.x1) return;
    const int ya = jb.y0 + c * CH;
    const bool rows_needed = ya < jb.oy1 && ya + CH > jb.oy0;
    const bool cols_needed = X < jb.ox1 && X + TW > jb.ox0;
    if (!rows_needed && !cols_needed) return;
    // one tile per orientation; a pixel of the rectangle lies on exactly one line of each family,
    // pixels outside it keep "no zero" and are never stored
    __shared__ __align__(16) uint16_t T[2][CH][TW];
    const int i = threadIdx.x;
    if (jb.x0 > X || jb.x1 < X + TW) {
        for (int e = i; e < 2 * CH * TW / 8; e += DB_THREADS) reinterpret_cast<uint4*>(&T[0][0][0])[e] = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
    }
    if (i < TW + CH - 1) {
        // orientation 0: the line through column X - 31 + i at the chunk's first row
        const int line0 = (X - 31 + i) - jb.x0 + (g.H - 1) - c * CH;
        if (line0 >= 0 && line0 < g.nlines) diag_steps(jb, g, 0, c, line0, i - 31, 1, T[0]);
        // orientation 1: the line through column X + i at the chunk's first row
        const int line1 = (X + i) - jb.x0 + c * CH;
        if (line1 >= 0 && line1 < g.nlines) diag_steps(jb, g, 1, c, line1, i, -1, T[1]);
    }
    __syncthreads();
    const int nrows = min(CH, jb.y1 - ya);
    if ((jb.pitch & 7) == 0 && X + TW <= jb.pitch) {
        for (int e = i; e < nrows * (TW / 8); e += DB_THREADS) {
            const int r = e / (TW / 8), q = e % (TW / 8);
            const int x = X + q * 8;
            if (x + 8 <= jb.x0 || x >= jb.x1) continue;
            const uint4 a = *reinterpret_cast<const uint4*>(&T[0][r][q * 8]), b = *reinterpret_cast<const uint4*>(&T[1][r][q * 8]);
            store_u4(jb.S + (size_t)(ya + r) * jb.pitch + x, make_uint4(pk_min_u16(a.x, b.x), pk_min_u16(a.y, b.y),
                                                                      pk_min_u16(a.z, b.z), pk_min_u16(a.w, b.w)));
        }
    } else {
        for (int e = i; e < nrows * TW; e += DB_THREADS) {
            const int r = e / TW, x = X + e % TW;
            if (x < jb.x0 || x >= jb.x1) continue;
            jb.S[(size_t)(ya + r) * jb.pitch + x] = min(T[0][r][e % TW], T[1][r][e % TW]);
        }
    }
}

// ---- row scans ---------------------------------------------------------------------
// R(x) = min_k E(k) + a*|x-k| = min(prefix minimum of E(k) + a*(XR-k) minus a*(XR-x),
//                                   suffix minimum of E(k) + a*k minus a*x)
// with k, x relative to an 8-aligned origin.  A thread owns 8 consecutive pixels, a
// workgroup 2048; wider rows take several passes with the running minimum carried over.

__device__ __forceinline__ uint32_t wg_prefix_min_excl(uint32_t v, uint32_t* s_w, uint32_t carry_in, uint32_t* total) {
    // exclusive prefix minimum over the 256 threads (thread order), seeded with carry_in
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t p = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(p, o);
        if (lane >= o) p = min(p, u);
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = p;
    __syncthreads();
    uint32_t ex = __shfl_up(p, 1);
    if (lane == 0) ex = INF32;
    uint32_t c = carry_in;
    for (int w = 0; w < 4; w++)
        if (w < wave) c = min(c, s_w[w]);
    *total = min(min(min(s_w[0], s_w[1]), min(s_w[2], s_w[3])), carry_in);
    return min(ex, c);
}

__device__ __forceinline__ uint32_t wg_suffix_min_excl(uint32_t v, uint32_t* s_w, uint32_t carry_in, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t p = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_down(p, o);
        if (lane + o < 64) p = min(p, u);
    }
    __syncthreads();
    if (lane == 0) s_w[wave] = p;
    __syncthreads();
    uint32_t ex = __shfl_down(p, 1);
    if (lane == 63) ex = INF32;
    uint32_t c = carry_in;
    for (int w = 0; w < 4; w++)
        if (w > wave) c = min(c, s_w[w]);
    *total = min(min(min(s_w[0], s_w[1]), min(s_w[2], s_w[3])), carry_in);
    return min(ex, c);
}

// E of the 8 pixels starting at x (8-aligned relative to the plane when the pitch allows)
__device__ __forceinline__ void load_e8(const DtJobG& jb, size_t row, int x, uint32_t* E) {
    uint16_t s[8];
    if ((jb.pitch & 7) == 0 && x >= 0 && x + 8 <= jb.pitch) {
        const uint4 v = load_u4(jb.S + row + x);
        s[0] = v.x & 0xFFFF, s[1] = v.x >> 16, s[2] = v.y & 0xFFFF, s[3] = v.y >> 16;
        s[4] = v.z & 0xFFFF, s[5] = v.z >> 16, s[6] = v.w & 0xFFFF, s[7] = v.w >> 16;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = (x + i >= jb.x0 && x + i < jb.x1) ? jb.S[row + x + i] : (uint16_t)0xFFFF;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) E[i] = (x + i >= jb.x0 && x + i < jb.x1) ? e_of(s[i]) : INF32;
}

__device__ void dt_row(const DtJobG& jb, int y, uint32_t* s_w) {
    const int xo = jb.x0 & ~7;                          // origin of the relative coordinate
    const int span = jb.x1 - xo;
    const int P = (span + 2047) / 2048;                 // passes
    const uint32_t XR = (uint32_t)(P * 2048 - 1);
    const size_t row = (size_t)y * jb.pitch;
    const int t = threadIdx.x;
    if (P == 1) {
        const int xr = t * 8, x = xo + xr;
        uint32_t E[8], F[8];
        if (xr < span)
            load_e8(jb, row, x, E);
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) E[i] = INF32;
        }
        uint32_t mf = INF32, mb = INF32;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (E[i] != INF32) {
                mf = min(mf, E[i] + DT_A * (XR - (uint32_t)(xr + i)));
                mb = min(mb, E[i] + DT_A * (uint32_t)(xr + i));
            }
        }
        uint32_t tot;
        uint32_t run = wg_prefix_min_excl(mf, s_w, INF32, &tot);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (E[i] != INF32) run = min(run, E[i] + DT_A * (XR - (uint32_t)(xr + i)));
            F[i] = run == INF32 ? INF32 : run - DT_A * (XR - (uint32_t)(xr + i));
        }
        run = wg_suffix_min_excl(mb, s_w, INF32, &tot);
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if (E[i] != INF32) run = min(run, E[i] + DT_A * (uint32_t)(xr + i));
            const uint32_t b = run == INF32 ? INF32 : run - DT_A * (uint32_t)(xr + i);
            const uint32_t v = min(F[i], b);
            F[i] = v > DT_MAX ? DT_MAX : v;
        }
        if (xr < span && x + 8 > jb.ox0 && x < jb.ox1) {
            if ((jb.pitch & 3) == 0 && x >= jb.ox0 && x + 8 <= jb.ox1) {
                const auto o = jb.R + row + x;
                store_u4(o, make_uint4(F[0], F[1], F[2], F[3]));
                store_u4(o + 4, make_uint4(F[4], F[5], F[6], F[7]));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (x + i >= jb.ox0 && x + i < jb.ox1) jb.R[row + x + i] = F[i];
            }
        }
        return;
    }
    // wider rows are scanned by k_dt_rows_wide
}

// The 8 step values starting at x as four words (0xFFFF outside the rectangle).
__device__ __forceinline__ uint4 load_s8(const DtJobG& jb, size_t row, int x) {
    if ((jb.pitch & 7) == 0 && x >= 0 && x + 8 <= jb.pitch) return load_u4(jb.S + row + x);
    uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int i = 0; i < 8; i++)
        if (x + i >= jb.x0 && x + i < jb.x1) {
            const uint32_t v = jb.S[row + x + i];
            w[i >> 1] = (i & 1) ? (w[i >> 1] & 0x0000FFFFu) | (v << 16) : (w[i >> 1] & 0xFFFF0000u) | v;
        }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint32_t e_of_s8(const DtJobG& jb, const uint4& sv, int x, int i) {
    const uint32_t w = i < 2 ? sv.x : i < 4 ? sv.y : i < 6 ? sv.z : sv.w;
    const uint32_t sx = (i & 1) ? w >> 16 : w & 0xFFFFu;
    return (x + i >= jb.x0 && x + i < jb.x1) ? e_of(sx) : INF32;
}

// Rows wider than one workgroup pass of dt_row (2048 pixels): thread t owns the 8 pixels at
// 8*(256 p + t) of every pass p.  The wave scans of all passes run side by side and one
// exchange through LDS gives every (pass, wave) the minima of everything before / after it;
// the passes that hold pixels of the output window are then loaded again and finished.
// One workgroup per row of the output window; blockIdx.y = job.
__global__ __launch_bounds__(256) void k_dt_rows_wide(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.y, n_first, second)]);
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    const int y = jb.oy0 + blockIdx.x;
    if (W <= 0 || H <= 0 || y >= jb.oy1) return;
    const int xo = jb.x0 & ~7;
    const int span = jb.x1 - xo;
    const int P = (span + 2047) / 2048;
    if (P == 1) return;  // done by k_dt_scan
    constexpr int PM = 9;  // 16384 pixels + alignment slack
    const uint32_t XR = (uint32_t)(P * 2048 - 1);
    const size_t row = (size_t)y * jb.pitch;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ uint32_t s_f[PM * 4], s_b[PM * 4];  // totals of (pass, wave): forward / backward keys
    uint32_t pf[PM], pb[PM];
#pragma unroll
    for (int p = 0; p < PM; p++) {
        uint32_t mf = INF32, mb = INF32;
        const int xr = (p * 256 + t) * 8, x = xo + xr;
        if (p < P && xr < span) {
            const uint4 sv = load_s8(jb, row, x);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t e = e_of_s8(jb, sv, x, i);
                if (e != INF32) {
                    mf = min(mf, e + DT_A * (XR - (uint32_t)(xr + i)));
                    mb = min(mb, e + DT_A * (uint32_t)(xr + i));
                }
            }
        }
        pf[p] = mf;
        pb[p] = mb;
    }
    // inclusive wave scans of all passes at once
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int p = 0; p < PM; p++) {
            const uint32_t uf = __shfl_up(pf[p], o), ub = __shfl_down(pb[p], o);
            if (lane >= o) pf[p] = min(pf[p], uf);
            if (lane + o < 64) pb[p] = min(pb[p], ub);
        }
    }
#pragma unroll
    for (int p = 0; p < PM; p++) {
        if (lane == 63) s_f[p * 4 + wave] = pf[p];
        if (lane == 0) s_b[p * 4 + wave] = pb[p];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PM; p++) {
        const int xr = (p * 256 + t) * 8, x = xo + xr;
        // minima of everything before (forward) / after (backward) this thread's pixels in its wave
        uint32_t runf = __shfl_up(pf[p], 1), runb = __shfl_down(pb[p], 1);
        if (p < P && xr < span && x + 8 > jb.ox0 && x < jb.ox1) {  // some of the window is here
            if (lane == 0) runf = INF32;
            if (lane == 63) runb = INF32;
            for (int q = 0; q < PM * 4; q++) {
                if (q < p * 4 + wave) runf = min(runf, s_f[q]);
                if (q > p * 4 + wave) runb = min(runb, s_b[q]);
            }
            const uint4 sv = load_s8(jb, row, x);
            uint32_t F[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t e = e_of_s8(jb, sv, x, i);
                if (e != INF32) runf = min(runf, e + DT_A * (XR - (uint32_t)(xr + i)));
                F[i] = runf == INF32 ? INF32 : runf - DT_A * (XR - (uint32_t)(xr + i));
            }
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                const uint32_t e = e_of_s8(jb, sv, x, i);
                if (e != INF32) runb = min(runb, e + DT_A * (uint32_t)(xr + i));
                const uint32_t bk = runb == INF32 ? INF32 : runb - DT_A * (uint32_t)(xr + i);
                const uint32_t v = min(F[i], bk);
                F[i] = v > DT_MAX ? DT_MAX : v;
            }
            if ((jb.pitch & 3) == 0 && x >= jb.ox0 && x + 8 <= jb.ox1) {
                const auto o = jb.R + row + x;
                store_u4(o, make_uint4(F[0], F[1], F[2], F[3]));
                store_u4(o + 4, make_uint4(F[4], F[5], F[6], F[7]));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (x + i >= jb.ox0 && x + i < jb.ox1) jb.R[row + x + i] = F[i];
            }
        }
    }
}

// E of the 32 rows of a chunk of column x (INF below the rectangle)
__device__ __forceinline__ void load_chunk_e(const DtJobG& jb, int x, int ya, uint32_t* E) {
    uint16_t s[CH];
#pragma unroll
    for (int r = 0; r < CH; r++) {
        const int y = min(ya + r, jb.y1 - 1);
        s[r] = jb.S[(size_t)y * jb.pitch + x];
    }
#pragma unroll
    for (int r = 0; r < CH; r++) E[r] = ya + r < jb.y1 ? e_of(s[r]) : INF32;
}

// Per (column, chunk): fe = value at the chunk's last row from the chunk's own pixels
// (going down), be = value at its first row (going up).  Only columns of the output window.
__device__ void dt_col_summary(const DtJobG& jb, int x, int c) {
    uint32_t E[CH];
    load_chunk_e(jb, x, jb.y0 + c * CH, E);
    uint32_t fe = INF32, be = INF32;
#pragma unroll
    for (int r = 0; r < CH; r++) {
        fe = min(sat_add(fe, DT_A), E[r]);
        be = min(be, sat_add(E[r], DT_A * (uint32_t)r));
    }
    jb.CS[((size_t)0 * jb.znch + c) * jb.cpitch + x] = fe;
    jb.CS[((size_t)1 * jb.znch + c) * jb.cpitch + x] = be;
}

// blockIdx.y = job.  The first `nrow_blocks` workgroups scan one row of the output window
// each; the others take 64 columns x 4 chunks of the column summaries.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_dt_scan(const DtJob* __restrict__ jobs, int n_first, int second, int nrow_blocks,
                                                 int ncol_x) {
    const DtJobG jb(jobs[job_index(blockIdx.y, n_first, second)]);
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (W <= 0 || H <= 0) return;
    __shared__ uint32_t s_w[4];
    if ((int)blockIdx.x < nrow_blocks) {
        const int y = jb.oy0 + blockIdx.x;
        if (y < jb.oy1) dt_row(jb, y, s_w);
        return;
    }
    const int b = blockIdx.x - nrow_blocks;
    const int nch = (H + CH - 1) / CH;
    const int x = (jb.ox0 & ~63) + (b % ncol_x) * 64 + (threadIdx.x & 63);  // column groups on the plane's 64-column grid
    const int c = (b / ncol_x) * 4 + (threadIdx.x >> 6);
    if (c >= nch || x < jb.ox0 || x >= jb.ox1) return;
    dt_col_summary(jb, x, c);
}

// Scan over the chunks of a column, in place over the chunk summaries: plane 0 of CS becomes
// what reaches each chunk's first row from the chunks above it, plane 1 what reaches its last
// row from the chunks below (chunk k's summary travels 32*|c-k| - 31 rows to chunk c).
// blockIdx.x = 64 columns of the output window, blockIdx.y = job, blockIdx.z = plane.
__global__ __launch_bounds__(64) void k_dt_chunks(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.y, n_first, second)]);
    const int H = jb.y1 - jb.y0, nch = (H + CH - 1) / CH;
    const int x = (jb.ox0 & ~63) + blockIdx.x * 64 + threadIdx.x;
    if (H <= 0 || x < jb.ox0 || x >= jb.ox1) return;
    const bool down = blockIdx.z == 0;
    const auto cs = jb.CS + (down ? 0 : (size_t)jb.znch * jb.cpitch) + x;
    uint32_t run = INF32;
    for (int i0 = 0; i0 < nch; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u, k = down ? i : nch - 1 - i;
            v[u] = i < nch ? cs[(size_t)k * jb.cpitch] : INF32;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u, k = down ? i : nch - 1 - i;
            if (i < nch) {
                cs[(size_t)k * jb.cpitch] = run;
                run = min(sat_add(run, DT_A * (uint32_t)CH), sat_add(v[u], DT_A));
            }
        }
    }
}

// Final column scan of the chunks that intersect the output window, in registers;
// d = min(column scan, row scan), clamped to DIST_MAX, as float / 65536.
__global__ __launch_bounds__(64) void k_dt_cols(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z, n_first, second)]);
    const int H = jb.y1 - jb.y0, nch = (H + CH - 1) / CH;
    const int x = (jb.ox0 & ~63) + blockIdx.x * 64 + threadIdx.x;  // a wave's row accesses start on a cache line
    const int c = (jb.oy0 - jb.y0) / CH + blockIdx.y;
    if (H <= 0 || c >= nch || x < jb.ox0 || x >= jb.ox1) return;
    const int ya = jb.y0 + c * CH;
    if (ya >= jb.oy1) return;
    uint32_t E[CH], F[CH], Rv[CH];
    // values arriving from the other chunks of the column: at the first row from above,
    // at the last row from below
    const uint32_t cf = jb.CS[(size_t)c * jb.cpitch + x];
    const uint32_t cb = jb.CS[((size_t)jb.znch + c) * jb.cpitch + x];
    load_chunk_e(jb, x, ya, E);
#pragma unroll
    for (int r = 0; r < CH; r++) {
        const int y = ya + r;
        Rv[r] = (y >= jb.oy0 && y < jb.oy1) ? jb.R[(size_t)y * jb.pitch + x] : INF32;
    }
    uint32_t run = cf;
#pragma unroll
    for (int r = 0; r < CH; r++) {
        run = min(run, E[r]);
        F[r] = run;
        run = sat_add(run, DT_A);
    }
    run = cb;  // only chunks that are not the last have a chunk below, and those are full
#pragma unroll
    for (int r = CH - 1; r >= 0; r--) {
        const int y = ya + r;
        if (y >= jb.y1) continue;
        run = min(run, E[r]);
        if (y >= jb.oy0 && y < jb.oy1) {
            uint32_t d = min(min(run, F[r]), Rv[r]);
            d = d > DT_MAX ? DT_MAX : d;
            jb.dist[(size_t)y * jb.pitch + x] = (float)d * (1.f / 65536.f);
        }
        run = sat_add(run, DT_A);
    }
}

}  // namespace

// Queues the scan kernels for njobs jobs described in device memory; max_w/max_h
// bound every job's rectangle (grid sizing).
int rtvm_launch_dt_bits(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h) {
    if (max_w <= 0 || max_h <= 0) return RTVM_OK;
    RTVM_LAUNCH(c, k_dt_bits, dim3((max_w / 32 + 2 + 63) / 64, (max_h + 3) / 4, njobs), dim3(256), 0, d_jobs, njobs, 0);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// Jobs d_jobs[0 .. n_first) and d_jobs[second .. second + n_second).
int rtvm_launch_dt2(rtvm_ctx* c, const DtJob* d_jobs, int n_first, int second, int n_second, int max_w, int max_h,
                    int with_bits) {
    if (max_w <= 0 || max_h <= 0) return RTVM_OK;
    if (max_w > 16384 || max_h > 16384) RTVM_FAIL(c, RTVM_E_ARG, "distance transform: image side above 16384");
    const int njobs = n_first + n_second;
    const int nlines = max_w + max_h - 1;
    const int nch = (max_h + CH - 1) / CH;
    const int ncol_x = (max_w + 63) / 64 + 1;  // column groups on the 64-column grid: one more than the width needs
    if (with_bits)
        RTVM_LAUNCH(c, k_dt_bits, dim3((max_w / 32 + 2 + 63) / 64, (max_h + 3) / 4, njobs), dim3(256), 0, d_jobs, n_first,
                    second);
    RTVM_LAUNCH(c, k_dt_diag_a, dim3((nlines + 63) / 64, (nch + GR - 1) / GR, 2 * njobs), dim3(256), 0, d_jobs, n_first,
                second);
    RTVM_LAUNCH(c, k_dt_diag_b, dim3(max_w / TW + 2, nch, njobs), dim3(DB_THREADS), 0, d_jobs, n_first, second);
    RTVM_LAUNCH(c, k_dt_scan, dim3(max_h + ncol_x * ((nch + 3) / 4), njobs), dim3(256), 0, d_jobs, n_first, second, max_h,
                ncol_x);
    if (max_w + 7 > 2048)  // rows that take more than one pass of the row scan
        RTVM_LAUNCH(c, k_dt_rows_wide, dim3(max_h, njobs), dim3(256), 0, d_jobs, n_first, second);
    RTVM_LAUNCH(c, k_dt_chunks, dim3(ncol_x, njobs, 2), dim3(64), 0, d_jobs, n_first, second);
    RTVM_LAUNCH(c, k_dt_cols, dim3(ncol_x, nch + 1, njobs), dim3(64), 0, d_jobs, n_first, second);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

int rtvm_launch_dt(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h, int with_bits) {
    return rtvm_launch_dt2(c, d_jobs, njobs, 0, 0, max_w, max_h, with_bits);
}
