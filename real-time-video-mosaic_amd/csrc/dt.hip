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
// Diagonal and column scans are cut into chunks of 32 rows, one lane per
// (line, chunk): lanes of a wave sit on consecutive lines, so every step touches
// consecutive pixels of one image row (coalesced), a chunk's 32 values live in
// registers, and the chunks of a line are stitched through small summary arrays:
//   k_dt_diag_a  32 mask bytes -> one 32-bit zero bitmap per (line, chunk)
//   k_dt_diag_b  steps to the nearest zero along the line from the bitmaps
//   k_dt_rows    row scans (one workgroup per row), only rows of the output window
//   k_dt_cols_a  per (column, chunk): value leaving the chunk downwards / upwards
//   k_dt_cols_b  carries from the other chunks' summaries, final column scan in
//                registers, min with the row scan, distance out
#include "common.h"

namespace {

constexpr uint32_t DT_A = 62587u;              // cvRound(0.955  * 65536)
constexpr uint32_t DT_B = 89738u;              // cvRound(1.3693 * 65536)
constexpr uint32_t DT_MAX = 0xFFFFFFFFu - DT_B;  // OpenCV's DIST_MAX
constexpr uint32_t INF32 = 0xFFFFFFFFu;
constexpr int CH = 32;                         // rows per chunk

__device__ __forceinline__ uint32_t sat_add(uint32_t v, uint32_t k) {
    return v > INF32 - k ? INF32 : v + k;
}

// E(q) in Q16 from the two diagonal step planes (0xFFFF = no zero on that diagonal).
__device__ __forceinline__ uint32_t e_of(uint16_t a, uint16_t b) {
    const uint32_t s = a < b ? a : b;
    return s == 0xFFFFu ? INF32 : s * DT_B;
}

// blockIdx.z = 2 * job + orientation; orientation 0: lines x - y = const, 1: x + y = const.
// x of line `line` at row y0 is xs; it moves by dx per row.
struct DiagGeom {
    int W, H, nlines, nch, line, dx, xs, ya;
    bool ok;
};
__device__ __forceinline__ DiagGeom diag_geom(const DtJob& jb, int orient) {
    DiagGeom g;
    g.W = jb.x1 - jb.x0;
    g.H = jb.y1 - jb.y0;
    g.nlines = g.W + g.H - 1;
    g.nch = (g.H + CH - 1) / CH;
    g.line = blockIdx.x * 64 + threadIdx.x;
    g.dx = orient ? -1 : 1;
    g.xs = orient ? jb.x0 + g.line : jb.x0 - (g.H - 1) + g.line;
    g.ya = jb.y0 + (int)blockIdx.y * CH;
    g.ok = g.W > 0 && g.H > 0 && (int)blockIdx.y < g.nch && (int)blockIdx.x * 64 < g.nlines;
    return g;
}

__global__ __launch_bounds__(64) void k_dt_diag_a(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.z >> 1];
    const int orient = blockIdx.z & 1;
    const DiagGeom g = diag_geom(jb, orient);
    if (!g.ok) return;
    uint32_t z = 0;
    if (g.line < g.nlines) {
        uint8_t v[CH];
#pragma unroll
        for (int r = 0; r < CH; r++) {
            const int y = g.ya + r;
            const int x = g.xs + g.dx * (y - jb.y0);
            const bool in = y < jb.y1 && x >= jb.x0 && x < jb.x1;
            v[r] = in ? jb.mask[(size_t)y * jb.pitch + x] : (uint8_t)1;
        }
#pragma unroll
        for (int r = 0; r < CH; r++) z |= (uint32_t)(v[r] == 0) << r;
    }
    if (g.line < g.nlines) jb.Z[((size_t)orient * jb.znch + blockIdx.y) * jb.zpitch + g.line] = z;
}

__global__ __launch_bounds__(64) void k_dt_diag_b(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.z >> 1];
    const int orient = blockIdx.z & 1;
    const DiagGeom g = diag_geom(jb, orient);
    if (!g.ok || g.line >= g.nlines) return;
    const uint32_t* __restrict__ Z = jb.Z + (size_t)orient * jb.znch * jb.zpitch + g.line;
    const int c = blockIdx.y;
    const uint32_t z = Z[(size_t)c * jb.zpitch];
    // nearest zero of the line above / below this chunk, as a row index
    long long above = -(1ll << 40), below = 1ll << 40;
    for (int k = c - 1; k >= 0; k--) {
        const uint32_t zz = Z[(size_t)k * jb.zpitch];
        if (zz) {
            above = jb.y0 + k * CH + (31 - __clz((int)zz));
            break;
        }
    }
    for (int k = c + 1; k < g.nch; k++) {
        const uint32_t zz = Z[(size_t)k * jb.zpitch];
        if (zz) {
            below = jb.y0 + k * CH + (__ffs((int)zz) - 1);
            break;
        }
    }
    uint16_t* __restrict__ S = orient ? jb.sB : jb.sA;
#pragma unroll
    for (int r = 0; r < CH; r++) {
        const int y = g.ya + r;
        const int x = g.xs + g.dx * (y - jb.y0);
        if (y >= jb.y1 || x < jb.x0 || x >= jb.x1) continue;
        const uint32_t lower = z & (0xFFFFFFFFu >> (31 - r));
        const uint32_t upper = z >> r;
        const long long du = lower ? (long long)(r - (31 - __clz((int)lower))) : (long long)y - above;
        const long long dd = upper ? (long long)(__ffs((int)upper) - 1) : below - (long long)y;
        const long long d = du < dd ? du : dd;
        S[(size_t)y * jb.pitch + x] = (uint16_t)(d > 0xFFFE ? 0xFFFF : d);
    }
}

// Row scans: one workgroup of 256 threads per row of the output window; thread t
// owns a contiguous segment.  R(x) = min_k E(k) + a*|x-k| as prefix/suffix minima of
// E(k) -/+ a*k.
__global__ __launch_bounds__(256) void k_dt_rows(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.y];
    const int W = jb.x1 - jb.x0;
    const int y = jb.oy0 + blockIdx.x;
    if (W <= 0 || y >= jb.oy1) return;
    const size_t row = (size_t)y * jb.pitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = (W + 255) / 256;
    const int xa = jb.x0 + tid * S, xb = min(xa + S, jb.x1);
    constexpr long long BIG = 1ll << 60;
    const long long a = DT_A;
    long long mf = BIG, mb = BIG;
    for (int x = xa; x < xb; x++) {
        const uint32_t e = e_of(jb.sA[row + x], jb.sB[row + x]);
        if (e != INF32) {
            mf = min(mf, (long long)e - a * x);
            mb = min(mb, (long long)e + a * x);
        }
    }
    __shared__ long long s_f[4], s_b[4];
    long long pf = mf, pb = mb;
    for (int o = 1; o < 64; o <<= 1) {
        const long long uf = __shfl_up(pf, o), ub = __shfl_down(pb, o);
        if (lane >= o) pf = min(pf, uf);
        if (lane + o < 64) pb = min(pb, ub);
    }
    if (lane == 63) s_f[wave] = pf;
    if (lane == 0) s_b[wave] = pb;
    __syncthreads();
    long long cf = __shfl_up(pf, 1), cb = __shfl_down(pb, 1);
    if (lane == 0) cf = BIG;
    if (lane == 63) cb = BIG;
    for (int w = 0; w < 4; w++) {
        if (w < wave) cf = min(cf, s_f[w]);
        if (w > wave) cb = min(cb, s_b[w]);
    }
    for (int x = xa; x < xb; x++) {
        const uint32_t e = e_of(jb.sA[row + x], jb.sB[row + x]);
        if (e != INF32) cf = min(cf, (long long)e - a * x);
        const long long v = cf >= BIG / 2 ? (long long)INF32 : cf + a * x;
        jb.R[row + x] = v > (long long)DT_MAX ? DT_MAX : (uint32_t)v;
    }
    for (int x = xb - 1; x >= xa; x--) {
        const uint32_t e = e_of(jb.sA[row + x], jb.sB[row + x]);
        if (e != INF32) cb = min(cb, (long long)e + a * x);
        const long long v = cb >= BIG / 2 ? (long long)INF32 : cb - a * x;
        const uint32_t r = v > (long long)DT_MAX ? DT_MAX : (uint32_t)v;
        if (r < jb.R[row + x]) jb.R[row + x] = r;
    }
}

// E of the 32 rows of chunk c of column x (INF below the rectangle)
__device__ __forceinline__ void load_chunk_e(const DtJob& jb, int x, int ya, uint32_t* E) {
    uint16_t a[CH], b[CH];
#pragma unroll
    for (int r = 0; r < CH; r++) {
        const int y = min(ya + r, jb.y1 - 1);
        const size_t p = (size_t)y * jb.pitch + x;
        a[r] = jb.sA[p];
        b[r] = jb.sB[p];
    }
#pragma unroll
    for (int r = 0; r < CH; r++) E[r] = ya + r < jb.y1 ? e_of(a[r], b[r]) : INF32;
}

// Per (column, chunk): fe = value at the chunk's last row from the chunk's own pixels
// (going down), be = value at its first row (going up).  Only columns of the output window.
__global__ __launch_bounds__(64) void k_dt_cols_a(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.z];
    const int H = jb.y1 - jb.y0, nch = (H + CH - 1) / CH;
    const int x = jb.ox0 + blockIdx.x * 64 + threadIdx.x;
    const int c = blockIdx.y;
    if (H <= 0 || c >= nch || x >= jb.ox1) return;
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

// Final column scan of the chunks that intersect the output window, in registers;
// d = min(column scan, row scan), clamped to DIST_MAX, as float / 65536.
__global__ __launch_bounds__(64) void k_dt_cols_b(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.z];
    const int H = jb.y1 - jb.y0, nch = (H + CH - 1) / CH;
    const int x = jb.ox0 + blockIdx.x * 64 + threadIdx.x;
    const int c = (jb.oy0 - jb.y0) / CH + blockIdx.y;
    if (H <= 0 || c >= nch || x >= jb.ox1) return;
    const int ya = jb.y0 + c * CH;
    if (ya >= jb.oy1) return;
    const size_t plane = (size_t)jb.znch * jb.cpitch;
    uint32_t E[CH], F[CH];
    load_chunk_e(jb, x, ya, E);
    // values arriving from the other chunks of the column: at the first row from above,
    // at the last row from below (chunk k's summary travels 32*|c-k| - 31 rows)
    uint32_t cf = INF32, cb = INF32;
    for (int k = 0; k < nch; k++) {
        if (k < c) {
            const uint32_t fe = jb.CS[0 * plane + (size_t)k * jb.cpitch + x];
            cf = min(cf, sat_add(fe, DT_A * (uint32_t)(CH * (c - k) - (CH - 1))));
        } else if (k > c) {
            const uint32_t be = jb.CS[1 * plane + (size_t)k * jb.cpitch + x];
            cb = min(cb, sat_add(be, DT_A * (uint32_t)(CH * (k - c) - (CH - 1))));
        }
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
            const size_t p = (size_t)y * jb.pitch + x;
            uint32_t d = min(min(run, F[r]), jb.R[p]);
            d = d > DT_MAX ? DT_MAX : d;
            jb.dist[p] = (float)d * (1.f / 65536.f);
        }
        run = sat_add(run, DT_A);
    }
}

}  // namespace

// Queues the scan kernels for njobs jobs described in device memory; max_w/max_h
// bound every job's rectangle (grid sizing).
int rtvm_launch_dt(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h) {
    if (max_w <= 0 || max_h <= 0) return RTVM_OK;
    if (max_w > 16384 || max_h > 16384) RTVM_FAIL(c, RTVM_E_ARG, "distance transform: image side above 16384");
    const int nlines = max_w + max_h - 1;
    const int nch = (max_h + CH - 1) / CH;
    RTVM_LAUNCH(c, k_dt_diag_a, dim3((nlines + 63) / 64, nch, 2 * njobs), dim3(64), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_diag_b, dim3((nlines + 63) / 64, nch, 2 * njobs), dim3(64), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_rows, dim3(max_h, njobs), dim3(256), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_cols_a, dim3((max_w + 63) / 64, nch, njobs), dim3(64), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_cols_b, dim3((max_w + 63) / 64, nch + 1, njobs), dim3(64), 0, d_jobs);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}
