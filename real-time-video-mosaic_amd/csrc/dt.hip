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
// which is three families of independent 1-D min-plus scans: along diagonals
// (k_dt_diag), along rows (k_dt_rows) and along columns (k_dt_cols).  The
// diagonal and column scans walk down the rows with one lane per line, so that
// a wave touches consecutive pixels of one row at every step (coalesced); each
// line is cut into NCH chunks that are walked by different waves and stitched
// through LDS.  All arithmetic is integer, so the result is bit-exact.
#include "common.h"

namespace {

constexpr uint32_t DT_A = 62587u;              // cvRound(0.955  * 65536)
constexpr uint32_t DT_B = 89738u;              // cvRound(1.3693 * 65536)
constexpr uint32_t DT_MAX = 0xFFFFFFFFu - DT_B;  // OpenCV's DIST_MAX
constexpr uint32_t INF32 = 0xFFFFFFFFu;
constexpr int NCH = 16;                        // chunks per line

__device__ __forceinline__ uint32_t sat_add(uint32_t v, uint32_t k) {
    return v > INF32 - k ? INF32 : v + k;
}
__device__ __forceinline__ uint32_t sat_mul(uint32_t v, uint32_t k) {
    const unsigned long long r = (unsigned long long)v * k;
    return r > 0xFFFFFFFFull ? INF32 : (uint32_t)r;
}

// E(q) in Q16 from the two diagonal step planes (0xFFFF = no zero on that diagonal).
__device__ __forceinline__ uint32_t e_of(uint16_t a, uint16_t b) {
    const uint32_t s = a < b ? a : b;
    return s == 0xFFFFu ? INF32 : s * DT_B;
}

// Diagonal scans.  blockIdx.y: 0 = lines x - y = const, 1 = lines x + y = const.
// Lane = line, threadIdx.y = chunk of rows.  Pass 1 finds the first and last
// zero of the chunk; the carries (last zero above, first zero below) come from
// the other chunks through LDS; pass 2 walks down writing the steps to the zero
// above, pass 3 walks up and merges the steps to the zero below.
__global__ __launch_bounds__(64 * NCH) void k_dt_diag(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.z];
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (W <= 0 || H <= 0) return;
    const int nlines = W + H - 1;
    if ((int)blockIdx.x * 64 >= nlines) return;
    const int line = blockIdx.x * 64 + threadIdx.x;
    const int orient = blockIdx.y;
    const int dx = orient ? -1 : 1;
    const int xs = orient ? jb.x0 + line : jb.x0 - (H - 1) + line;  // x of the line at row y0
    const int L = (H + NCH - 1) / NCH;
    const int c = threadIdx.y;
    const int ya = jb.y0 + c * L;
    const int yb = min(ya + L, jb.y1);
    uint16_t* __restrict__ S = orient ? jb.sB : jb.sA;
    __shared__ int s_first[NCH][64];
    __shared__ int s_last[NCH][64];
    int first = 0x7FFFFFFF, last = -0x40000000;
    if (line < nlines) {
        for (int y = ya; y < yb; y++) {
            const int x = xs + dx * (y - jb.y0);
            if (x >= jb.x0 && x < jb.x1 && jb.mask[(size_t)y * jb.pitch + x] == 0) {
                first = min(first, y);
                last = y;
            }
        }
    }
    s_first[c][threadIdx.x] = first;
    s_last[c][threadIdx.x] = last;
    __syncthreads();
    if (line >= nlines) return;
    int above = -0x40000000, below = 0x7FFFFFFF;
    for (int k = 0; k < NCH; k++) {
        if (k < c) above = max(above, s_last[k][threadIdx.x]);
        if (k > c) below = min(below, s_first[k][threadIdx.x]);
    }
    for (int y = ya; y < yb; y++) {  // steps to the nearest zero at or above
        const int x = xs + dx * (y - jb.y0);
        if (x < jb.x0 || x >= jb.x1) continue;
        const size_t p = (size_t)y * jb.pitch + x;
        if (jb.mask[p] == 0) above = y;
        const int d = y - above;
        S[p] = (uint16_t)(d > 0xFFFE ? 0xFFFF : d);
    }
    for (int y = yb - 1; y >= ya; y--) {  // merge the steps to the nearest zero at or below
        const int x = xs + dx * (y - jb.y0);
        if (x < jb.x0 || x >= jb.x1) continue;
        const size_t p = (size_t)y * jb.pitch + x;
        if (jb.mask[p] == 0) below = y;
        const long long d = (long long)below - y;
        const uint16_t v = (uint16_t)(d > 0xFFFE ? 0xFFFF : d);
        if (v < S[p]) S[p] = v;
    }
}

// Row scans: one workgroup of 256 threads per row; thread t owns a contiguous
// segment.  R(x) = min_k E(k) + a*|x-k| as prefix/suffix minima of E(k) -/+ a*k.
__global__ __launch_bounds__(256) void k_dt_rows(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.y];
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (W <= 0 || (int)blockIdx.x >= H) return;
    const int y = jb.y0 + blockIdx.x;
    const size_t row = (size_t)y * jb.pitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = (W + 255) / 256;
    const int xa = jb.x0 + tid * S, xb = min(xa + S, jb.x1);
    constexpr long long BIG = 1ll << 60;
    const long long a = DT_A;
    // segment minima of E(k) - a*k (forward) and E(k) + a*k (backward)
    long long mf = BIG, mb = BIG;
    for (int x = xa; x < xb; x++) {
        const uint32_t e = e_of(jb.sA[row + x], jb.sB[row + x]);
        if (e != INF32) {
            mf = min(mf, (long long)e - a * x);
            mb = min(mb, (long long)e + a * x);
        }
    }
    // exclusive prefix-min of mf over lower threads, exclusive suffix-min of mb over higher threads
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
    // forward values into R, then backward values merged
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

// Column scans and final result.  Lane = column, threadIdx.y = chunk of rows.
__global__ __launch_bounds__(64 * NCH) void k_dt_cols(const DtJob* __restrict__ jobs) {
    const DtJob jb = jobs[blockIdx.y];
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (W <= 0 || H <= 0 || (int)blockIdx.x * 64 >= W) return;
    const int x = jb.x0 + blockIdx.x * 64 + threadIdx.x;
    const bool act = x < jb.x1;
    const int L = (H + NCH - 1) / NCH;
    const int c = threadIdx.y;
    const int ya = jb.y0 + c * L;
    const int yb = min(ya + L, jb.y1);
    __shared__ uint32_t s_fe[NCH][64];  // value carried out of the chunk's last row, going down
    __shared__ uint32_t s_be[NCH][64];  // value carried out of the chunk's first row, going up
    uint32_t fe = INF32, be = INF32;
    if (act) {
        for (int y = ya; y < yb; y++) {
            const size_t p = (size_t)y * jb.pitch + x;
            const uint32_t e = e_of(jb.sA[p], jb.sB[p]);
            fe = min(sat_add(fe, DT_A), e);
            be = min(be, sat_add(e, DT_A * (uint32_t)(y - ya)));
        }
    }
    s_fe[c][threadIdx.x] = fe;
    s_be[c][threadIdx.x] = be;
    __syncthreads();
    if (!act || ya >= yb) return;
    // carry arriving at the first row from above / at the last row from below
    uint32_t cf = INF32, cb = INF32;
    for (int k = 0; k < NCH; k++) {
        const int ka = jb.y0 + k * L, kb = min(ka + L, jb.y1);
        if (ka >= kb) continue;
        if (k < c) cf = min(cf, sat_add(s_fe[k][threadIdx.x], sat_mul(DT_A, (uint32_t)(ya - (kb - 1)))));
        if (k > c) cb = min(cb, sat_add(s_be[k][threadIdx.x], sat_mul(DT_A, (uint32_t)(ka - (yb - 1)))));
    }
    uint32_t run = cf;  // value at row ya-1 plus a, i.e. what arrives at ya
    for (int y = ya; y < yb; y++) {
        const size_t p = (size_t)y * jb.pitch + x;
        const uint32_t e = e_of(jb.sA[p], jb.sB[p]);
        run = min(run, e);
        jb.F[p] = run;
        run = sat_add(run, DT_A);
    }
    run = cb;
    for (int y = yb - 1; y >= ya; y--) {
        const size_t p = (size_t)y * jb.pitch + x;
        const uint32_t e = e_of(jb.sA[p], jb.sB[p]);
        run = min(run, e);
        uint32_t d = min(min(run, jb.F[p]), jb.R[p]);
        d = d > DT_MAX ? DT_MAX : d;
        jb.dist[p] = (float)d * (1.f / 65536.f);
        run = sat_add(run, DT_A);
    }
}

}  // namespace

// Queues the three scan kernels for njobs jobs described in device memory;
// max_w/max_h bound every job's rectangle (grid sizing).
int rtvm_launch_dt(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h) {
    if (max_w <= 0 || max_h <= 0) return RTVM_OK;
    if (max_w > 16384 || max_h > 16384) RTVM_FAIL(c, RTVM_E_ARG, "distance transform: image side above 16384");
    const int nlines = max_w + max_h - 1;
    RTVM_LAUNCH(c, k_dt_diag, dim3((nlines + 63) / 64, 2, njobs), dim3(64, NCH), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_rows, dim3(max_h, njobs), dim3(256), 0, d_jobs);
    RTVM_LAUNCH(c, k_dt_cols, dim3((max_w + 63) / 64, njobs), dim3(64, NCH), 0, d_jobs);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}
