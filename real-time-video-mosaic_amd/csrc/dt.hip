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
//   k_dt_sum     per tile of 32 rows x 64 columns: what leaves the tile along its columns
//                (downwards / upwards) and along its rows (to the right / left)
//   k_dt_chunks  scans over those summaries: what arrives at every tile from the other tiles
//                of its columns and of its rows
//   k_dt_final   per tile: column scan in registers, row scan over the lanes, the arrivals,
//                minimum of the two, distance out
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
constexpr int TPAD = 8, TP = TW + 2 * TPAD;    // a tile row in LDS: TPAD spare columns on either side (rows stay 16-byte aligned)

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
// Transpose of the 32 x 32 bit matrix held by each half of the wave (lane r of a half holds row r): afterwards bit i
// of lane r's word is bit r of what lane i held.  Five exchange steps with the lane 16, 8, 4, 2, 1 away (a swizzle
// inside each half): a lane keeps the bits of the block it shares with its partner's row and takes the rest from the
// partner's word, rotated by the distance; tm = the bits kept, tsh = the rotation (both per lane and step).
// Three instructions per step where a ballot per bit and a select took four per bit and two wait states.
__device__ __forceinline__ uint32_t transpose32(uint32_t a, const uint32_t (&tm)[5], const uint32_t (&tsh)[5]) {
#define RTVM_T32_STEP(st, J)                                                                               \
    {                                                                                                      \
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, ((J) << 10) | 0x1F);               \
        a = (a & tm[st]) | (__builtin_amdgcn_alignbit(b, b, tsh[st]) & ~tm[st]);                           \
    }
    RTVM_T32_STEP(0, 16)
    RTVM_T32_STEP(1, 8)
    RTVM_T32_STEP(2, 4)
    RTVM_T32_STEP(3, 2)
    RTVM_T32_STEP(4, 1)
#undef RTVM_T32_STEP
    return a;
}

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
    // the lane's constants of the five exchange steps of the bit-matrix transpose (distance 16 >> s)
    uint32_t tm[5], tsh[5];
#pragma unroll
    for (int st = 0; st < 5; st++) {
        const int j = 16 >> st;
        const uint32_t m = st == 0 ? 0x0000FFFFu : st == 1 ? 0x00FF00FFu : st == 2 ? 0x0F0F0F0Fu : st == 3 ? 0x33333333u : 0x55555555u;
        tm[st] = (r & j) ? ~m : m;
        tsh[st] = (r & j) ? j : 32 - j;
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < GR / 4; k++) {
        const int c = blockIdx.y * GR + wave * (GR / 4) + k;
        if (c >= g.nch) break;  // wave-uniform
        const uint32_t z = transpose32(w[k], tm, tsh);
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
                                           int dcol, uint16_t (*T)[TP]) {
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
        // a column outside the tile goes to one of the two pad columns beside it: a store per row, no branch
        const int col = min(max(col0 + dcol * r, -1), TW);
        T[r][TPAD + col] = (uint16_t)min(min(run, e[r]), 0xFFFF);
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
    __shared__ __align__(16) uint16_t T[2][CH][TP];
    const int i = threadIdx.x;
    if (jb.x0 > X || jb.x1 < X + TW) {
        for (int e = i; e < 2 * CH * TP / 8; e += DB_THREADS) reinterpret_cast<uint4*>(&T[0][0][0])[e] = make_uint4(~0u, ~0u, ~0u, ~0u);
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
            const uint4 a = *reinterpret_cast<const uint4*>(&T[0][r][TPAD + q * 8]), b = *reinterpret_cast<const uint4*>(&T[1][r][TPAD + q * 8]);
            store_u4(jb.S + (size_t)(ya + r) * jb.pitch + x, make_uint4(pk_min_u16(a.x, b.x), pk_min_u16(a.y, b.y),
                                                                      pk_min_u16(a.z, b.z), pk_min_u16(a.w, b.w)));
        }
    } else {
        for (int e = i; e < nrows * TW; e += DB_THREADS) {
            const int r = e / TW, x = X + e % TW;
            if (x < jb.x0 || x >= jb.x1) continue;
            jb.S[(size_t)(ya + r) * jb.pitch + x] = min(T[0][r][TPAD + e % TW], T[1][r][TPAD + e % TW]);
        }
    }
}

// ---- row and column scans --------------------------------------------------------------
// The plane is cut into tiles of 32 rows (a chunk) x 64 columns (a group, on the plane's
// 64-column grid): one wave per tile, lane = column, the 32 rows of the column in registers.
//   column direction  sequential in the lane; what arrives from the other chunks of the column
//                     comes from the chunk summaries (k_dt_sum -> k_dt_chunks)
//   row direction     R(i) = min_j E(j) + a*|i-j| inside the tile is a prefix and a suffix minimum
//                     over the lanes of E(j) + a*(63-j) and E(j) + a*j (data-parallel-primitive row
//                     shifts, no shared memory); what arrives from the other groups of the row comes
//                     from the row summaries (k_dt_sum -> k_dt_chunks)
// Inside the tile kernels "no zero reachable" is BIG rather than all ones, so that the few axial
// steps added inside a tile cannot wrap; everything from BIG_MIN on is infinite (a real distance is
// at most 16384 * (a + b) < 2^31.3).  The summary planes hold all ones for infinite.
constexpr uint32_t BIG = 0xF0000000u;
constexpr uint32_t BIG_MIN = 0xE0000000u;

__device__ __forceinline__ uint32_t e_soft(uint32_t s) { return s == 0xFFFFu ? BIG : s * DT_B; }
__device__ __forceinline__ uint32_t inf_out(uint32_t v) { return v >= BIG_MIN ? INF32 : v; }

// x of lane (lane op n) where the operation names a valid lane of the row of 16, `ident` elsewhere
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_or(uint32_t ident, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)x, CTRL, ROW_MASK, 0xF, false);
}

// inclusive minimum over lanes 0..lane (lane 63 ends up with the minimum of the wave); all 64 lanes active
__device__ __forceinline__ uint32_t wave_prefix_min(uint32_t x) {
    x = min(x, dpp_or<0x111>(INF32, x));        // row_shr:1
    x = min(x, dpp_or<0x112>(INF32, x));        // row_shr:2
    x = min(x, dpp_or<0x114>(INF32, x));        // row_shr:4
    x = min(x, dpp_or<0x118>(INF32, x));        // row_shr:8
    x = min(x, dpp_or<0x142, 0xA>(INF32, x));   // row_bcast:15 into rows 1 and 3
    x = min(x, dpp_or<0x143, 0xC>(INF32, x));   // row_bcast:31 into rows 2 and 3
    return x;
}

// inclusive minimum over lanes lane..63; all 64 lanes active
__device__ __forceinline__ uint32_t wave_suffix_min(uint32_t x, int lane) {
    x = min(x, dpp_or<0x101>(INF32, x));        // row_shl:1
    x = min(x, dpp_or<0x102>(INF32, x));
    x = min(x, dpp_or<0x104>(INF32, x));
    x = min(x, dpp_or<0x108>(INF32, x));
    // the minima of the rows of 16 are in lanes 0, 16, 32, 48: every row takes those of the rows after it
    const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    const uint32_t t2 = min((uint32_t)__builtin_amdgcn_readlane((int)x, 32), t3);
    const uint32_t t1 = min((uint32_t)__builtin_amdgcn_readlane((int)x, 16), t2);
    const uint32_t after = lane < 16 ? t1 : lane < 32 ? t2 : lane < 48 ? t3 : INF32;
    return min(x, after);
}

// `old` with lane `lane` (a constant) replaced by the wave-uniform value v
__device__ __forceinline__ int write_lane(int v, int lane, int old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(lane));
    return old;
}

// E of the 32 rows of the chunk starting at row ya for column x; BIG outside the rectangle
__device__ __forceinline__ void load_tile_e(const DtJobG& jb, int x, int ya, uint32_t* E) {
    const bool col_in = x >= jb.x0 && x < jb.x1;
    const int xc = min(max(x, jb.x0), jb.x1 - 1);
    uint16_t s[CH];
#pragma unroll
    for (int r = 0; r < CH; r++) {
        const int y = min(ya + r, jb.y1 - 1);
        s[r] = jb.S[(size_t)y * jb.pitch + xc];
    }
    // rows of the lane that lie inside the rectangle, as a per-lane number: tests against it are selects.  (A test
    // on the wave-uniform row limit becomes a branch per row, and every such branch copies all of E.)
    const int rows_in = col_in ? jb.y1 - ya : 0;
#pragma unroll
    for (int r = 0; r < CH; r++) E[r] = r < rows_in ? e_soft(s[r]) : BIG;
}

// Summaries of the tiles: per (column, chunk) what leaves the chunk downwards (value at its last
// row from the chunk's own pixels) and upwards (at its first row), for the columns of the output
// window; per (row, group) what leaves the group to the right (value at its last column from the
// group's own pixels) and to the left, for the rows of the output window.
// blockIdx.x = group of the rectangle, blockIdx.y = four chunks (one per wave), blockIdx.z = job.
__global__ __launch_bounds__(256) void k_dt_sum(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z, n_first, second)]);
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (W <= 0 || H <= 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = (jb.x0 >> 6) + blockIdx.x;
    const int c = blockIdx.y * 4 + wave, nch = (H + CH - 1) / CH;
    if (G * 64 >= jb.x1 || c >= nch) return;  // the same for the whole wave, like every exit below
    const int ya = jb.y0 + c * CH;
    const bool rows_needed = ya < jb.oy1 && ya + CH > jb.oy0;
    const bool cols_needed = G * 64 < jb.ox1 && G * 64 + 64 > jb.ox0;
    if (!rows_needed && !cols_needed) return;
    const int x = G * 64 + lane;
    uint32_t E[CH];
    load_tile_e(jb, x, ya, E);
    if (cols_needed) {
        uint32_t fe = BIG, be = BIG;
#pragma unroll
        for (int r = 0; r < CH; r++) {
            fe = min(fe + DT_A, E[r]);
            be = min(be, E[r] + DT_A * (uint32_t)r);
        }
        if (x >= jb.ox0 && x < jb.ox1) {
            jb.CS[(size_t)c * jb.cpitch + x] = inf_out(fe);
            jb.CS[((size_t)jb.znch + c) * jb.cpitch + x] = inf_out(be);
        }
    }
    if (rows_needed) {
        const uint32_t kf = DT_A * (uint32_t)(63 - lane), kb = DT_A * (uint32_t)lane;
        int fo = 0, bo = 0;  // lane r keeps the pair of row r
#pragma unroll
        for (int r = 0; r < CH; r++) {
            const int y = ya + r;
            if (y < jb.oy0 || y >= jb.oy1) continue;
            const uint32_t tf = wave_prefix_min(E[r] + kf), tb = wave_prefix_min(E[r] + kb);
            fo = write_lane(__builtin_amdgcn_readlane((int)tf, 63), r, fo);
            bo = write_lane(__builtin_amdgcn_readlane((int)tb, 63), r, bo);
        }
        const int y = ya + lane;
        if (lane < CH && y >= jb.oy0 && y < jb.oy1) {
            jb.RS[(size_t)G * jb.rpitch + y] = inf_out((uint32_t)fo);
            jb.RS[((size_t)jb.ng + G) * jb.rpitch + y] = inf_out((uint32_t)bo);
        }
    }
}

// Scans over the summaries, in place.  Workgroups below ncol_x take 64 columns of the output
// window each and scan the chunks of a column: plane 0 of CS becomes what reaches each chunk's
// first row from the chunks above it, plane 1 what reaches its last row from the chunks below
// (chunk k's summary travels 32*|c-k| - 31 rows to chunk c).  The other workgroups take 64 rows
// of the output window each and scan the groups of a row the same way: plane 0 of RS becomes what
// reaches each group's first column from the groups left of it, plane 1 what reaches its last
// column from the groups right of it.  blockIdx.y = job, blockIdx.z = plane.
__global__ __launch_bounds__(64) void k_dt_chunks(const DtJob* __restrict__ jobs, int n_first, int second, int ncol_x) {
    const DtJobG jb(jobs[job_index(blockIdx.y, n_first, second)]);
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0;
    if (H <= 0 || W <= 0) return;
    const bool down = blockIdx.z == 0;
    gptr<uint32_t> cs;
    size_t stride;
    int n;
    uint32_t step;
    if ((int)blockIdx.x < ncol_x) {
        const int x = (jb.ox0 & ~63) + blockIdx.x * 64 + threadIdx.x;
        if (x < jb.ox0 || x >= jb.ox1) return;
        cs = jb.CS + (down ? 0 : (size_t)jb.znch * jb.cpitch) + x;
        stride = jb.cpitch;
        n = (H + CH - 1) / CH;
        step = DT_A * (uint32_t)CH;
    } else {
        const int y = jb.oy0 + ((int)blockIdx.x - ncol_x) * 64 + threadIdx.x;
        if (y >= jb.oy1) return;
        const int g0 = jb.x0 >> 6;
        cs = jb.RS + ((down ? 0 : (size_t)jb.ng) + g0) * jb.rpitch + y;
        stride = jb.rpitch;
        n = ((jb.x1 - 1) >> 6) - g0 + 1;
        step = DT_A * 64u;
    }
    uint32_t run = INF32;
    for (int i0 = 0; i0 < n; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u, k = down ? i : n - 1 - i;
            v[u] = i < n ? cs[(size_t)k * stride] : INF32;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u, k = down ? i : n - 1 - i;
            if (i < n) {
                cs[(size_t)k * stride] = run;
                run = min(sat_add(run, step), sat_add(v[u], DT_A));
            }
        }
    }
}

// The distances of one tile of the output window: column scan in registers with the arrivals from
// the other chunks, row scan over the lanes with the arrivals from the other groups,
// d = min of the two, clamped to DIST_MAX, as float / 65536.
// blockIdx.x = group of the output window, blockIdx.y = chunk, blockIdx.z = job; one wave.
__global__ __launch_bounds__(64) void k_dt_final(const DtJob* __restrict__ jobs, int n_first, int second) {
    const DtJobG jb(jobs[job_index(blockIdx.z, n_first, second)]);
    const int W = jb.x1 - jb.x0, H = jb.y1 - jb.y0, nch = (H + CH - 1) / CH;
    const int lane = threadIdx.x;
    const int G = (jb.ox0 >> 6) + blockIdx.x;
    const int c = (jb.oy0 - jb.y0) / CH + blockIdx.y;
    if (H <= 0 || W <= 0 || jb.ox1 <= jb.ox0 || c >= nch || G * 64 >= jb.ox1) return;  // whole wave
    const int ya = jb.y0 + c * CH;
    if (ya >= jb.oy1) return;
    const int x = G * 64 + lane;
    const bool col_out = x >= jb.ox0 && x < jb.ox1;
    const int xw = min(max(x, jb.ox0), jb.ox1 - 1);
    // arrivals from the other chunks of the column: at the first row from above, at the last row from below
    const uint32_t cf = min(jb.CS[(size_t)c * jb.cpitch + xw], BIG);
    const uint32_t cb = min(jb.CS[((size_t)jb.znch + c) * jb.cpitch + xw], BIG);
    // arrivals from the other groups of the rows: lane r holds what reaches row r's first column from
    // the left, lane 32 + r what reaches its last column from the right
    const int yr = min(max(ya + (lane & 31), jb.oy0), jb.oy1 - 1);
    const uint32_t io = min(jb.RS[((lane >> 5) * (size_t)jb.ng + G) * jb.rpitch + yr], BIG);
    uint32_t E[CH], F[CH];
    load_tile_e(jb, x, ya, E);
    uint32_t run = cf;
#pragma unroll
    for (int r = 0; r < CH; r++) {
        run = min(run, E[r]);
        F[r] = run;
        run += DT_A;
    }
    const uint32_t kf = DT_A * (uint32_t)(63 - lane), kb = DT_A * (uint32_t)lane;
    run = cb;  // only chunks that are not the last have a chunk below, and those are full
#pragma unroll
    for (int r = CH - 1; r >= 0; r--) {
        const int y = ya + r;
        if (y >= jb.y1) continue;
        run = min(run, E[r]);
        if (y >= jb.oy0 && y < jb.oy1) {
            const uint32_t lin = (uint32_t)__builtin_amdgcn_readlane((int)io, r) + 63u * DT_A;
            const uint32_t rin = (uint32_t)__builtin_amdgcn_readlane((int)io, 32 + r) + 63u * DT_A;
            const uint32_t rf = min(wave_prefix_min(E[r] + kf), lin) - kf;
            const uint32_t rb = min(wave_suffix_min(E[r] + kb, lane), rin) - kb;
            uint32_t d = min(min(run, F[r]), min(rf, rb));
            d = d >= BIG_MIN ? DT_MAX : d;
            if (col_out) jb.dist[(size_t)y * jb.pitch + x] = (float)d * (1.f / 65536.f);
        }
        run += DT_A;
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
    RTVM_LAUNCH(c, k_dt_sum, dim3(ncol_x, (nch + 3) / 4, njobs), dim3(256), 0, d_jobs, n_first, second);
    RTVM_LAUNCH(c, k_dt_chunks, dim3(ncol_x + (max_h + 63) / 64, njobs, 2), dim3(64), 0, d_jobs, n_first, second, ncol_x);
    RTVM_LAUNCH(c, k_dt_final, dim3(ncol_x, nch + 1, njobs), dim3(64), 0, d_jobs, n_first, second);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

int rtvm_launch_dt(rtvm_ctx* c, const DtJob* d_jobs, int njobs, int max_w, int max_h, int with_bits) {
    return rtvm_launch_dt2(c, d_jobs, njobs, 0, 0, max_w, max_h, with_bits);
}
