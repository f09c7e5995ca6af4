// cv2.findHomography(src, dst, cv2.RANSAC, thresh) for gfx950 (reference
// main.py:856-857).
//
// OpenCV's loop is sequential (draw a 4-point subset, solve, score, maybe shrink
// the iteration budget), but three facts make it data-parallel without changing
// its result:
//   * the subset sequence depends only on the points and on the fixed-seed RNG
//     stream, never on the scores, so the subsets of many iterations can be
//     generated up front (k_rs_subsets; the raw RNG outputs are a constant table);
//   * solving and scoring a hypothesis is independent of every other hypothesis
//     (k_rs_hyp: one wave per hypothesis, 9x9 Jacobi eigen-solve in LDS, inlier
//     count by ballot + popcount);
//   * the keep/stop rule only needs the inlier counts in iteration order
//     (k_rs_replay, one lane).
// Iterations are evaluated in three stages, 0..15, 16..63 and 64..max_iters-1; with a
// usable inlier ratio the budget has dropped below 16 (or 64) by the time a stage has
// been replayed and the later stages exit at once.
// k_rs_refine then recomputes the winner's inlier mask, re-solves on all inliers
// and runs the 10-iteration Levenberg-Marquardt polish, one wave per problem.
//
// Sums over all inlier points use the order "lane l adds points l, l+64, ... in
// turn, then a butterfly over lane masks 32,16,8,4,2,1", which the CPU
// restatement mirrors; everything else follows OpenCV's operation order in
// IEEE double/float without contraction.
#include "common.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int RS_BATCH0 = 64;  // iterations evaluated in the first two stages (a quarter of them in the first)

#ifdef RTVM_RS_TRACE
__device__ long long g_tr[16];
__device__ __forceinline__ long long tnow() { return (long long)__builtin_readcyclecounter(); }
#define TR(i) do { if ((threadIdx.x & 63) == 0) g_tr[i] = tnow(); } while (0)
#define TRADD(i, v) do { if ((threadIdx.x & 63) == 0) g_tr[i] += (v); } while (0)
#else
#define TR(i)
#define TRADD(i, v)
#endif

struct RsState {
    int niters, max_good, best, done, found, fail_at, next_pos, next_iter;
};

// ------------------------------------------------------------------ helpers
// sqrt(x) for 1 <= x <= 2: the refinement sqrt() itself runs (reciprocal square root, then
// Goldschmidt steps and two corrections on fused multiply-adds), without its scaling of very small
// arguments and its handling of zero, infinity and NaN, none of which 1 + q*q with q <= 1 can be.
__device__ __forceinline__ double sqrt_1to2(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, e, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// cv::hypot as Jacobi uses it (the larger magnitude times sqrt(1 + q*q), q the ratio of the smaller to
// the larger; 0 when both are 0) for a != 0, as every pivot is: the larger magnitude is then not 0 and the
// zero case cannot occur.  As selects: one copy of the division and the root, no branches.
__device__ __forceinline__ double cv_hypot_nz(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    const bool ab = a > b;
    const double hi = ab ? a : b, lo = ab ? b : a;
    const double q = lo / hi;
    return hi * sqrt_1to2(1 + q * q);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

// Lane j of a row of 16 lanes receives the value of lane j - N of the same row (row_shr:N on the
// data-parallel-primitive path of the vector ALU: no trip through LDS as a shuffle would make);
// lanes without a source keep `fill`.
template <int N>
__device__ __forceinline__ int row_shr_i32(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ double row_shr_f64(double v, double fill) {
    const int lo = row_shr_i32<N>(__double2loint(v), __double2loint(fill));
    const int hi = row_shr_i32<N>(__double2hiint(v), __double2hiint(fill));
    return __hiloint2double(hi, lo);
}
// v_max_f64 as it is (fmax() spends two more of them on quieting signalling NaNs first)
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the same shift with +0.0 for lanes without a source (no registers to preset)
template <int N>
__device__ __forceinline__ double row_shr_f64_zero(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// largest value of lanes 0..15, the same in every lane of the wave (values >= 0 or -1, at least one >= 0)
__device__ __forceinline__ double row0_max(double v) {
    v = max_f64(v, row_shr_f64_zero<1>(v));
    v = max_f64(v, row_shr_f64_zero<2>(v));
    v = max_f64(v, row_shr_f64_zero<4>(v));
    v = max_f64(v, row_shr_f64_zero<8>(v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 15), hi = __builtin_amdgcn_readlane(__double2hiint(v), 15);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane: the same in every lane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// every lane of a row of 16 lanes receives the largest value of the row: four exchanges (lane ^ 1,
// lane ^ 2, then the mirror images inside 8 and 16 lanes, which meet lanes ^ 4 and ^ 8 of equal halves)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    // (every lane has a source under these patterns; "zero without one" only spares presetting the target)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_all_max(double v) {
    v = max_f64(v, dpp_f64<0xB1>(v));   // quad_perm:[1,0,3,2]
    v = max_f64(v, dpp_f64<0x4E>(v));   // quad_perm:[2,3,0,1]
    v = max_f64(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = max_f64(v, dpp_f64<0x140>(v));  // row_mirror
    return v;
}
// Every lane receives the value its image in the lower half of the wave (lane & 31) holds (lo) and the value
// its image in the upper half (lane | 32) holds (hi): one exchange of the halves of two copies.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void halves_i32(int v, int& lo, int& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    lo = (int)r.x;
    hi = (int)r.y;
}
__device__ __forceinline__ void halves_f64(double v, double& lo, double& hi) {
    int l0, h0, l1, h1;
    halves_i32(__double2loint(v), l0, h0);
    halves_i32(__double2hiint(v), l1, h1);
    lo = __hiloint2double(l1, l0);
    hi = __hiloint2double(h1, h0);
}

// cv::JacobiImpl_<double> executed by one full wave (A: n x n in LDS, only the upper triangle is used and it
// is destroyed; V: rows are the eigenvectors; W: eigenvalues, descending; n <= 9).  The pivot choice, the
// rotation formulas and the bookkeeping of the per-row / per-column maxima follow the sequential algorithm
// exactly; lanes only share out independent element updates.
//
// A lone wave is bound by the chain of dependent instructions of one rotation, so nothing on that chain goes
// through LDS.  The sequential code keeps, per row r, the column indR[r] of the largest |a| right of the
// diagonal and, per column c, the row indC[c] of the largest |a| above it, finds the pivot among the elements
// those (possibly stale) indices point at, and after a rotation scans rows and columns k and l again.  Here
// lane i16 of the groups of 16 lanes 0 and 2 holds (indR[i16], |A[i16][indR[i16]]|), of groups 1 and 3
// (indC[i16], |A[indC[i16]][i16]|), and W[i16], in registers:
//   * the pivot is a maximum over registers (rows in group 0, columns in group 1);
//   * every group computes the rotated pairs of A (lane i16 < n: the pair that row / column i16 shares with k
//     and l), so a lane refreshes the value its index points at from its own pair, and the four scans (row k,
//     column k, row l, column l: groups 0..3) are maxima over the pairs' registers;
//   * the halves of the wave exchange the scans' results (lanes k and l take them);
//   * the pairs of V belong to lanes 9..15 of the four groups.
// LDS holds A and V for the pair loads of the next rotation (requested as soon as k and l are known, consumed
// after the coefficients' divisions and square roots) and the sign of the pivot.
__device__ __forceinline__ void jacobi_wave(double* A, double* W, double* V, int* perm, int n) {
    const int lane = threadIdx.x & 63;
    const double eps = DBL_EPSILON;
    for (int e = lane; e < n * n; e += 64) V[e] = (e / n == e % n) ? 1. : 0.;
    const int i16 = lane & 15, grp = lane >> 4;
    const bool colk = (grp & 1) != 0;   // this lane keeps a column's maximum (else a row's)
    const int of_l = grp >> 1;          // its group scans row / column l after a rotation (else k)
    // rows 0..n-2 and columns 1..n-1 have candidates
    const bool cand = colk ? (i16 >= 1 && i16 < n) : (i16 < n - 1);
    int sind = 0;
    double sabs = -1.;
    double Wi = i16 < n ? A[(n + 1) * i16] : 0.;
    if (cand) {
        // first largest |a| of row i16 right of the diagonal (elements n*i16 + i16+1+t) / of column i16 above
        // it (elements n*t + i16)
        const int cnt = colk ? i16 : n - 1 - i16;
        const int e0 = colk ? i16 : n * i16 + i16 + 1, es = colk ? n : 1, m0 = colk ? 0 : i16 + 1;
        int m = m0;
        double mv = fabs(A[e0]);
        for (int t = 1; t < cnt; t++) {
            const double val = fabs(A[e0 + es * t]);
            if (mv < val) mv = val, m = m0 + t;
        }
        sind = m;
        sabs = mv;
    }
    __syncthreads();
    const int maxIters = n * n * 30;
    int jac_rot = 0;
    const int voff = (int)(V - A);
    // The pairs: lanes i16 < n take the pair of A of index i16: elements n * min(i16, k) + max(i16, k) and the same
    // with l (none for i16 = k, l); lanes i16 >= 9 take the pair of V of column vi: elements voff + n * k + vi and
    // the same with l, which is the same formula for an index rr = vi + 64 beyond every k and a base 64 lower.
    const bool isA = i16 < n;
    const int vi = grp * 7 + i16 - 9;
    const bool live = isA || (i16 >= 9 && vi < n);
    const int rr = isA ? i16 : vi + 64, rbase = isA ? 0 : voff - 64;
    const bool writes = !isA || grp == 0;   // the pairs of A are the same in the four groups: group 0 stores them
    if (n > 1)
        for (int iters = 0; iters < maxIters; iters++) {
            // pivot: first maximum over rows 0..n-2 (group 0), then over columns 1..n-1 (group 1) if larger
            const double v = cand ? sabs : -1.;
            const double gm = row_all_max(v);
            const unsigned long long hit = __ballot(v == gm);
            const double m1 = readlane_f64(gm, 0), m2 = readlane_f64(gm, 16);
            const bool from_col = m2 > m1;
            const double pabs = from_col ? m2 : m1;   // |p|
            if (pabs <= eps) break;
            const unsigned h16 = (unsigned)(from_col ? hit >> 16 : hit) & 0xFFFFu;
            const int win = __builtin_ctz(h16);   // some lane holds the maximum
            const int other = __builtin_amdgcn_readlane(sind, win + (from_col ? 16 : 0));
            const int k = from_col ? other : win, l = from_col ? win : other;
            const double wk = readlane_f64(Wi, k), wl = readlane_f64(Wi, l);
            // The pairs and the pivot (for its sign) are requested now and arrive while the divisions and
            // square roots of the rotation's coefficients run.
            const bool act = live && rr != k && rr != l;
            const int e0 = rbase + n * min(rr, k) + max(rr, k);
            const int e1 = rbase + n * min(rr, l) + max(rr, l);
            double a0 = 0, b0 = 0;
            if (act) a0 = A[e0], b0 = A[e1];
            const double p = A[n * k + l];
#ifdef FAKE_MATH
            double c = 0.8 + 1e-9 * (wl - wk), s = 0.6 * (p > 0 ? 1. : -1.), t = pabs * 0.3;
#else
            const double y = (wl - wk) * 0.5;
            double t = fabs(y) + cv_hypot_nz(pabs, y);  // |p| > eps
            // hypot(p, t): t >= hypot(p, y) >= |p|, so t is the larger magnitude (for t = |p| both orders give
            // the same), and its ratio q = |p| / t is, but for the sign of p, the quotient p / t of the next
            // line of the sequential code: (p / t) * p = q * |p| bit for bit.  One division less.
            const double q = pabs / t;
            double s = t * sqrt_1to2(1 + q * q);
            const double c = t / s;
            s = p / s;
            t = q * pabs;
            if (y < 0) s = -s, t = -t;
#endif
            // W[k] -= t, W[l] += t on the values the pivot search already holds
            double wk_new = wk - t, wl_new = wl + t, a1 = a0 * c - b0 * s, b1 = a0 * s + b0 * c;
            // All of that in every lane and in one piece: left to itself the compiler moves the division by t
            // under "lane 0" and the rest under "lanes with a pair", one chain of dependent operations after
            // the other, where side by side each fills the other's waits.
            asm volatile("" : "+v"(wk_new), "+v"(wl_new), "+v"(a1), "+v"(b1));
            wave_lds_phase();  // every lane has read its pair and the pivot
            if (lane == 0) A[n * k + l] = 0;
            if (act && writes) {
                A[e0] = a1;
                A[e1] = b1;
            }
            if (i16 == k) Wi = wk_new;
            if (i16 == l) Wi = wl_new;
            // the element a kept index points at changed when the index is k or l: row i16 then holds A[i16][k]
            // = a1 (i16 < k) or A[i16][l] = b1 (i16 < l), column i16 A[k][i16] = a1 (i16 > k) or A[l][i16] = b1
            // (i16 > l): the pair's halves either way.  (Lanes k and l take the scans' results below.)
            const double na = fabs(a1), nb = fabs(b1);
            sabs = sind == k ? na : sind == l ? nb : sabs;
            // the maxima of row k, column k, row l, column l again: the sequential scans "first largest |a| of
            // row idx right of the diagonal" / "... of column idx above it", one per group: row k is a1 of lanes
            // > k (0 at l), column k a1 of lanes < k, row l b1 of lanes > l, column l b1 of lanes < l (0 at k)
            {
                const int idx = of_l ? l : k, zl = of_l ? k : l;
                const bool has = colk ? i16 < idx : (i16 > idx && i16 < n);
                const double sv = has ? (i16 == zl ? 0. : (of_l ? nb : na)) : -1.;
                const double sg = row_all_max(sv);
                const unsigned long long eq = __ballot(sv == sg);
                const int first = __ffs((unsigned)(eq >> (lane & 48)) & 0xFFFFu) - 1;
                // groups 0 and 1 hold the results for k, groups 2 and 3 those for l, rows in the even groups
                // and columns in the odd ones, as the lanes that keep them
                int fk, fl;
                double gk, gl;
                halves_i32(first, fk, fl);
                halves_f64(sg, gk, gl);
                if (i16 == k) sabs = gk, sind = fk;
                if (i16 == l) sabs = gl, sind = fl;
            }
            wave_lds_phase();
            jac_rot++;
#ifdef MAXROT
            if (jac_rot >= MAXROT) break;
#endif
        }
    if (lane == 0) perm[15] = jac_rot;
    if (lane < n) W[lane] = Wi;
    __syncthreads();
    // selection sort (descending) with row swaps: the permutation is found with compile-time indices
    // only (values and positions stay in registers; arrays indexed at run time would live in scratch
    // memory, a chain of dependent loads and stores), by every lane alike; every lane then moves its
    // share of V through registers
    {
        constexpr int NM = 9;
        double w[NM];
        int pr[NM];
#pragma unroll
        for (int i = 0; i < NM; i++) w[i] = i < n ? W[i] : 0., pr[i] = i;
#pragma unroll
        for (int k = 0; k < NM - 1; k++) {
            if (k >= n - 1) break;
            // first largest of w[k..n-1]
            double wm = w[k];
            int m = k, pm = pr[k];
#pragma unroll
            for (int i = k + 1; i < NM; i++)
                if (i < n && wm < w[i]) wm = w[i], m = i, pm = pr[i];
            // swap positions k and m
#pragma unroll
            for (int i = k + 1; i < NM; i++)
                if (i == m) w[i] = w[k], pr[i] = pr[k];
            w[k] = wm;
            pr[k] = pm;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NM; i++)
                if (i < n) W[i] = w[i], perm[i] = pr[i];
        }
    }
    __syncthreads();
    double v0 = 0, v1 = 0;
    const int e0 = lane, e1 = lane + 64;
    if (e0 < n * n) v0 = V[n * perm[e0 / n] + e0 % n];
    if (e1 < n * n) v1 = V[n * perm[e1 / n] + e1 % n];
    __syncthreads();
    if (e0 < n * n) V[e0] = v0;
    if (e1 < n * n) V[e1] = v1;
    __syncthreads();
}

__device__ __forceinline__ void mat3mul(const double* a, const double* b, double* d) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

struct JacobiLds {
    double A[81], V[81], W[9];
    int perm[16]; int indR[1], indC[1];
};

// HomographyEstimatorCallback::runKernel for `count` point pairs (M -> m), one wave.
// Point i belongs to lane i % 64.  Returns 0 for a degenerate configuration,
// else 1 with H (9 doubles, identical in every lane).
__device__ int run_kernel_wave(const float* __restrict__ M, const float* __restrict__ m, int count, JacobiLds* L,
                               double* H) {
    const int lane = threadIdx.x & 63;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = lane; i < count; i += 64) {
        a0 += m[2 * i];
        a1 += m[2 * i + 1];
        a2 += M[2 * i];
        a3 += M[2 * i + 1];
    }
    const double cmx = wave_sum(a0) / count, cmy = wave_sum(a1) / count;
    const double cMx = wave_sum(a2) / count, cMy = wave_sum(a3) / count;
    a0 = a1 = a2 = a3 = 0;
    for (int i = lane; i < count; i += 64) {
        a0 += fabs(m[2 * i] - cmx);
        a1 += fabs(m[2 * i + 1] - cmy);
        a2 += fabs(M[2 * i] - cMx);
        a3 += fabs(M[2 * i + 1] - cMy);
    }
    double smx = wave_sum(a0), smy = wave_sum(a1), sMx = wave_sum(a2), sMy = wave_sum(a3);
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double acc[45];
#pragma unroll
    for (int e = 0; e < 45; e++) acc[e] = 0;
    for (int i = lane; i < count; i += 64) {
        const double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        const double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        int e = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) acc[e++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    __syncthreads();
    {
        int e = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) {
                const double v = wave_sum(acc[e++]);
                if (lane == 0) {
                    L->A[j * 9 + k] = v;
                    L->A[k * 9 + j] = v;
                }
            }
    }
    __syncthreads();
    TR(2);
    jacobi_wave(L->A, L->W, L->V, L->perm, 9);
    TR(3);
    double H0v[9], Htemp[9], H0[9];
    for (int i = 0; i < 9; i++) H0v[i] = L->V[72 + i];
    mat3mul(invHnorm, H0v, Htemp);
    mat3mul(Htemp, Hnorm2, H0);
    const double sc = 1. / H0[8];
    for (int i = 0; i < 9; i++) H[i] = H0[i] * sc;
    __syncthreads();
    return 1;
}

// HomographyEstimatorCallback::computeError in float32; inlier iff err <= thr
__device__ __forceinline__ bool is_inlier(const float* Hf, float Mx, float My, float mx, float my, float thr) {
    const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
    const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx;
    const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my;
    const float err = dx * dx + dy * dy;
    return err <= thr;
}

// ------------------------------------------------------------------ subsets
__device__ bool have_collinear4(const float* p) {
    const int i = 3;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}
__device__ __forceinline__ double det3pts(const float* p, int a, int b, int c) {
    const double m00 = p[2 * a], m01 = p[2 * a + 1], m10 = p[2 * b], m11 = p[2 * b + 1], m20 = p[2 * c], m21 = p[2 * c + 1];
    return m00 * (m11 * 1. - 1. * m21) - m01 * (m10 * 1. - 1. * m20) + 1. * (m10 * m21 - m11 * m20);
}
__device__ bool check_subset(const float* ms1, const float* ms2) {
    if (have_collinear4(ms1) || have_collinear4(ms2)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++)
        negative += det3pts(ms1, tt[i][0], tt[i][1], tt[i][2]) * det3pts(ms2, tt[i][0], tt[i][1], tt[i][2]) < 0;
    return negative == 0 || negative == 4;
}

// One RANSAC iteration's getSubset starting at draw `start` of the raw RNG stream:
// returns the number of draws it consumes (or -1: no valid subset within the table
// or within 10000 attempts) and the four indices.
__device__ int sim_subset(const float* __restrict__ src, const float* __restrict__ dst, int count,
                          const uint32_t* __restrict__ raw, int raw_n, int start, int* idx) {
    int pos = start;
    float ms1[8], ms2[8];
    for (int attempt = 0; attempt < 10000; attempt++) {
        for (int i = 0; i < 4; i++) {
            int v;
            for (;;) {
                if (pos >= raw_n) return -1;
                v = (int)(raw[pos++] % (uint32_t)count);
                bool dup = false;
                for (int k = 0; k < i; k++) dup |= idx[k] == v;
                if (!dup) break;
            }
            idx[i] = v;
            ms1[2 * i] = src[2 * v];
            ms1[2 * i + 1] = src[2 * v + 1];
            ms2[2 * i] = dst[2 * v];
            ms2[2 * i + 1] = dst[2 * v + 1];
        }
        if (check_subset(ms1, ms2)) return pos - start;
    }
    return -1;
}

// Subsets of iterations [h0, h1) of one problem, 64 at a time: lane L guesses the
// stream position of iteration h0+L, simulates it, the guesses are corrected by a
// prefix sum of the consumed draws and the round repeats until nothing moves.
__global__ __launch_bounds__(64) void k_rs_subsets(const FrameSlot* __restrict__ slots, int sb,
                                                   const uint32_t* __restrict__ raw, int raw_n, int h0, int h1,
                                                   double confidence, int max_iters, int first) {
    const FrameSlot& S = slots[sb + blockIdx.x];
    const float* __restrict__ src = S.rs_src;
    const float* __restrict__ dst = S.rs_dst;
    const int* __restrict__ count_p = S.rs_count;
    RsState* __restrict__ st = (RsState*)S.rs_state;
    int* __restrict__ idx_out = S.rs_idx;
    const int lane = threadIdx.x;
    const int count = *count_p;
    if (first && lane == 0) {
        st->niters = max_iters > 1 ? max_iters : 1;
        st->max_good = 0;
        st->best = -1;
        st->done = count <= 4 ? 1 : 0;
        st->found = 0;
        st->fail_at = -1;
        st->next_pos = 0;
        st->next_iter = 0;
    }
    __syncthreads();
    if (count <= 4) return;
    if (!first && (st->done || st->fail_at >= 0)) return;
    int base = first ? 0 : st->next_pos;
    if (h1 > max_iters) h1 = max_iters;
    if (!first && h1 > st->niters) h1 = st->niters;  // the budget only ever shrinks
    for (int c0 = h0; c0 < h1; c0 += 64) {
        const int h = c0 + lane;
        int pos = base + 4 * lane, cons = 0, idx[4] = {0, 0, 0, 0};
        bool fresh = true;
        for (int round = 0; round < 65; round++) {
            if (fresh) cons = sim_subset(src, dst, count, raw, raw_n, pos, idx);
            // exclusive prefix sum of the draws consumed (a failed lane counts as 0)
            int inc = cons > 0 ? cons : 0;
            int scan = inc;
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(scan, o);
                if (lane >= o) scan += u;
            }
            const int npos = base + scan - inc;
            fresh = npos != pos;
            pos = npos;
            if (!__any(fresh)) break;
        }
        // first failing lane ends the stream
        const unsigned long long bad = __ballot(cons < 0 && h < h1);
        const int first_bad = bad ? __ffsll((long long)bad) - 1 : 64;
        if (h < h1 && lane < first_bad) {
            idx_out[4 * h] = idx[0];
            idx_out[4 * h + 1] = idx[1];
            idx_out[4 * h + 2] = idx[2];
            idx_out[4 * h + 3] = idx[3];
        }
        if (first_bad < 64) {
            if (lane == 0) st->fail_at = c0 + first_bad;
            return;
        }
        base = __shfl(pos + cons, min(63, h1 - 1 - c0));  // where the last iteration of this stage ends
    }
    if (lane == 0) {
        st->next_pos = base;
        st->next_iter = h1;
    }
}

// One wave per hypothesis: solve the 4-point homography, count the inliers.  Workgroup x of a problem
// takes iterations h0 + x, h0 + x + gridDim.x, ... below h1 and below the problem's budget: a late stage,
// whose budget has almost always dropped below h0 already, is a small grid that exits at once instead of
// one workgroup for every iteration it might have had to run.
__global__ __launch_bounds__(64) void k_rs_hyp(const FrameSlot* __restrict__ slots, int sb, int h0, int h1,
                                               float thr) {
    const FrameSlot& S = slots[sb + blockIdx.y];
    const float* __restrict__ src = S.rs_src;
    const float* __restrict__ dst = S.rs_dst;
    const int* __restrict__ count_p = S.rs_count;
    const RsState* __restrict__ st = (const RsState*)S.rs_state;
    const int* __restrict__ idx = S.rs_idx;
    int* __restrict__ good = S.rs_good;
    double* __restrict__ hypH = S.rs_hypH;
    const int count = *count_p;
    if (count <= 4 || st->done) return;
    int hend = min(h1, st->niters);
    if (st->fail_at >= 0) hend = min(hend, st->fail_at);
    __shared__ JacobiLds L;
    __shared__ float ms1[8], ms2[8];
    const int lane = threadIdx.x;
    for (int h = h0 + blockIdx.x; h < hend; h += gridDim.x) {
        __syncthreads();  // the points of the iteration before are no longer read
        if (lane < 4) {
            const int v = idx[4 * h + lane];
            ms1[2 * lane] = src[2 * v];
            ms1[2 * lane + 1] = src[2 * v + 1];
            ms2[2 * lane] = dst[2 * v];
            ms2[2 * lane + 1] = dst[2 * v + 1];
        }
        __syncthreads();
        double H[9];
        if (!run_kernel_wave(ms1, ms2, 4, &L, H)) {
            if (lane == 0) good[h] = -1;
            continue;
        }
        float Hf[8];
        for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
        int g = 0;
        for (int i0 = 0; i0 < count; i0 += 64) {
            const int i = i0 + lane;
            bool in = false;
            if (i < count) in = is_inlier(Hf, src[2 * i], src[2 * i + 1], dst[2 * i], dst[2 * i + 1], thr);
            g += __popcll(__ballot(in));
        }
        if (lane == 0) good[h] = g;
        if (lane < 9) hypH[9 * h + lane] = H[lane];
    }
}

__device__ int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : __double2int_rn(num / denom);
}

// The sequential keep/stop rule of RANSACPointSetRegistrator::run over the inlier
// counts of iterations [h0, h1).
__global__ __launch_bounds__(64) void k_rs_replay(const FrameSlot* __restrict__ slots, int sb, int h0, int h1, double confidence,
                                                  int last) {
    const FrameSlot& S = slots[sb + blockIdx.x];
    const int* __restrict__ count_p = S.rs_count;
    RsState* __restrict__ st = (RsState*)S.rs_state;
    const int* __restrict__ good = S.rs_good;
    const int count = *count_p;
    if (count <= 4 || st->done) return;   // the same for the wave
    int niters = st->niters, max_good = st->max_good, best = st->best;
    const int fail_at = st->fail_at;
    bool done = false;
    // the hypotheses' inlier counts come 64 at a time, one load per lane; lane 0 then walks through them in order
    // (a load per hypothesis behind the test on the one before it made the walk a chain of memory round trips)
    __shared__ int s_good[64];
    const int lane = threadIdx.x;
    for (int hb = h0; hb < h1 && !done; hb += 64) {
        s_good[lane] = hb + lane < h1 ? good[hb + lane] : -1;
        wave_lds_phase();
        if (lane == 0) {
            const int he = min(hb + 64, h1);
            for (int h = hb; h < he; h++) {
                if (h >= niters || fail_at == h) {
                    done = true;
                    break;
                }
                const int g = s_good[h - hb];
                if (g < 0) continue;
                if (g > (max_good > 3 ? max_good : 3)) {
                    best = h;
                    max_good = g;
                    niters = update_num_iters(confidence, (double)(count - g) / count, 4, niters);
                }
            }
        }
        done = __shfl((int)done, 0) != 0;
        wave_lds_phase();
    }
    if (lane != 0) return;
    if (h1 >= niters || last) done = true;
    st->niters = niters;
    st->max_good = max_good;
    st->best = best;
    st->done = done ? 1 : 0;
}

// ------------------------------------------------------------------ refine
struct RefineLds {
    JacobiLds J;
    double A[64], Ap[64], v[8], d[8], D[8], x[8], xd[8], diag[8];
};

// position of JtJ[a][b], a <= b, in the row-major list of the upper triangle of an 8x8
__device__ constexpr int acc_index(int a, int b) { return a * 8 - a * (a - 1) / 2 + (b - a); }

// residuals at parameters h over the compacted inliers: returns |r|^2; when
// with_J, also JtJ -> S->A (8x8), Jt r -> S->v, and max |r| -> *rinf
__device__ double lm_compute_wave(const float* __restrict__ M, const float* __restrict__ m, int count,
                                  const double* h, bool with_J, RefineLds* S, double* rinf) {
    const int lane = threadIdx.x & 63;
    double acc[44];
#pragma unroll
    for (int e = 0; e < 44; e++) acc[e] = 0;
    double ss = 0, rmax = 0;
    for (int i = lane; i < count; i += 64) {
        const double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double ex = xi - m[2 * i], ey = yi - m[2 * i + 1];
        ss += ex * ex + ey * ey;
        rmax = fmax(rmax, fmax(fabs(ex), fabs(ey)));
        if (with_J) {
            // Jx = {Mx ww, My ww, ww, 0, 0, 0, -Mx ww xi, -My ww xi}, Jy = {0, 0, 0, Mx ww, My ww, ww, -Mx ww yi,
            // -My ww yi}; JtJ[a][b] += Jx[a] Jx[b] + Jy[a] Jy[b], Jt r[a] += Jx[a] ex + Jy[a] ey.  A product with
            // one of the zeros is +-0 (the entries are finite: ww is guarded above), and adding +-0 changes
            // neither the other product nor a sum that started at +0: the products of the zeros are left out,
            // and the nine entries between parameters 0..2 and 3..5 stay +0.
            const double j0 = Mx * ww, j1 = My * ww, j2 = ww;
            const double x6 = -Mx * ww * xi, x7 = -My * ww * xi, y6 = -Mx * ww * yi, y7 = -My * ww * yi;
            const double J3[3] = {j0, j1, j2};
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = a; b < 3; b++) {
                    acc[acc_index(a, b)] += J3[a] * J3[b];          // Jx Jx
                    acc[acc_index(a + 3, b + 3)] += J3[a] * J3[b];  // Jy Jy: the same value, its own sum
                }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                acc[acc_index(a, 6)] += J3[a] * x6;
                acc[acc_index(a, 7)] += J3[a] * x7;
                acc[acc_index(a + 3, 6)] += J3[a] * y6;
                acc[acc_index(a + 3, 7)] += J3[a] * y7;
                acc[36 + a] += J3[a] * ex;
                acc[36 + a + 3] += J3[a] * ey;
            }
            acc[acc_index(6, 6)] += x6 * x6 + y6 * y6;
            acc[acc_index(6, 7)] += x6 * x7 + y6 * y7;
            acc[acc_index(7, 7)] += x7 * x7 + y7 * y7;
            acc[36 + 6] += x6 * ex + y6 * ey;
            acc[36 + 7] += x7 * ex + y7 * ey;
        }
    }
    if (with_J) {
        __syncthreads();
        int e = 0;
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int b = a; b < 8; b++) {
                const bool zero = a < 3 && b >= 3 && b < 6;  // never touched: a sum of +0 terms
                const double v = zero ? 0. : wave_sum(acc[e]);
                e++;
                if (lane == 0) {
                    S->A[a * 8 + b] = v;
                    S->A[b * 8 + a] = v;
                }
            }
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const double v = wave_sum(acc[36 + a]);
            if (lane == 0) S->v[a] = v;
        }
        *rinf = wave_max(rmax);
        __syncthreads();
    }
    return wave_sum(ss);
}

// Jacobi of the 8x8 in S->J.A (a copy), then either x = solve (to S->d) or diag(inverse) (to S->diag)
__device__ void eig8_apply(RefineLds* S, const double* Asrc, bool solve) {
    const int lane = threadIdx.x & 63;
    __syncthreads();
    S->J.A[lane] = Asrc[lane];
    __syncthreads();
    jacobi_wave(S->J.A, S->J.W, S->J.V, S->J.perm, 8);
    // back-substitution through the eigenvectors, as cv::solve(DECOMP_EIG) does it: for every
    // eigenpair above the threshold, s = (v_i . b) / w_i and x += s v_i, the pairs taken in order.
    // Lane i < 8 forms s of pair i (its own sequential dot product); lane j < 8 then adds the
    // eight terms of x[j] in that order.
    double threshold = 0;
    for (int i = 0; i < 8; i++) threshold += S->J.W[i];
    threshold *= DBL_EPSILON * 2;
    const int r = lane & 7;
    const double* vi = &S->J.V[r * 8];
    double wi = S->J.W[r];
    const bool keep = fabs(wi) > threshold;
    wi = 1 / wi;
    double si = 0;
    if (solve) {
        for (int j = 0; j < 8; j++) si += vi[j] * S->v[j];
        si *= wi;
    }
    double out = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double s_i = readlane_f64(si, i), w_i = readlane_f64(wi, i);
        const bool keep_i = __builtin_amdgcn_readlane((int)keep, i) != 0;
        const double vij = S->J.V[i * 8 + r];
        if (keep_i) out = solve ? out + s_i * vij : out + vij * (vij * w_i);
    }
    if (lane < 8) (solve ? S->d : S->diag)[lane] = out;
    __syncthreads();
}

__device__ __forceinline__ double dot8(const double* a, const double* b) {
    double r = 0;
    r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    r += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
    return r;
}

// One wave per problem: the winner's inlier mask, runKernel on all inliers, LM polish.
__global__ __launch_bounds__(64) void k_rs_refine(const FrameSlot* __restrict__ slots, int sb, float thr) {
    const FrameSlot& SL = slots[sb + blockIdx.x];
    const float* __restrict__ src = SL.rs_src;
    const float* __restrict__ dst = SL.rs_dst;
    const int* __restrict__ count_p = SL.rs_count;
    RsState* __restrict__ st = (RsState*)SL.rs_state;
    const double* __restrict__ hypH = SL.rs_hypH;
    float* __restrict__ csrc = SL.rs_csrc;
    float* __restrict__ cdst = SL.rs_cdst;
    double* __restrict__ outH = SL.rs_H;
    uint8_t* __restrict__ mask = SL.rs_mask;
    int* __restrict__ found = SL.rs_found;
    __shared__ RefineLds S;
    const int lane = threadIdx.x;
#ifdef RTVM_RS_TRACE
    if (lane == 0) for (int i = 0; i < 16; i++) g_tr[i] = 0;
    TR(0);
#endif
    const int count = *count_p;
    double H[9];
    if (count < 4) {
        if (lane == 0) *found = 0;
        if (lane < 9) outH[lane] = 0;
        return;
    }
    if (count == 4) {
        const int ok = run_kernel_wave(src, dst, 4, &S.J, H);
        if (lane == 0) *found = ok;
        if (lane < 9) outH[lane] = ok ? H[lane] : 0;
        if (lane < 4) mask[lane] = 1;
        return;
    }
    const int best = st->best;
    if (best < 0 || st->max_good <= 0) {
        if (lane == 0) *found = 0;
        if (lane < 9) outH[lane] = 0;
        for (int i = lane; i < count; i += 64) mask[i] = 0;
        return;
    }
    for (int i = 0; i < 9; i++) H[i] = hypH[9 * best + i];
    float Hf[8];
    for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
    // inlier mask + order-preserving compaction
    int n = 0;
    for (int i0 = 0; i0 < count; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        float sx = 0, sy = 0, dx = 0, dy = 0;
        if (i < count) {
            sx = src[2 * i];
            sy = src[2 * i + 1];
            dx = dst[2 * i];
            dy = dst[2 * i + 1];
            in = is_inlier(Hf, sx, sy, dx, dy, thr);
            mask[i] = in ? 1 : 0;
        }
        const unsigned long long bal = __ballot(in);
        if (in) {
            const int p = n + __popcll(bal & ((1ull << lane) - 1ull));
            csrc[2 * p] = sx;
            csrc[2 * p + 1] = sy;
            cdst[2 * p] = dx;
            cdst[2 * p + 1] = dy;
        }
        n += __popcll(bal);
    }
    __syncthreads();
    TR(1);
    if (n > 0) {
        double H2[9];
        if (run_kernel_wave(csrc, cdst, n, &S.J, H2))
            for (int i = 0; i < 9; i++) H[i] = H2[i];
        // LMSolverImpl::run, maxIters = 10, eps = FLT_EPSILON, 8 parameters
        const int maxIters = 10;
        const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
        if (lane < 8) S.x[lane] = H[lane];
        __syncthreads();
        double rinf = 0;
        TR(4);
        double Sres = lm_compute_wave(csrc, cdst, n, S.x, true, &S, &rinf);
        TR(5);
        if (lane < 8) S.D[lane] = S.A[lane * 8 + lane];
        const double Rlo = 0.25, Rhi = 0.75;
        double lambda = 1, lc = 0.75;
        int iter = 0;
        for (;;) {
            __syncthreads();
            S.Ap[lane] = ((lane >> 3) == (lane & 7)) ? S.A[lane] + lambda * S.D[lane & 7] : S.A[lane];
#ifdef RTVM_RS_TRACE
            const long long te0 = tnow();
#endif
            eig8_apply(&S, S.Ap, true);
#ifdef RTVM_RS_TRACE
            TRADD(11, tnow() - te0); TRADD(12, 1);
            const long long tl0 = tnow();
#endif
            if (lane < 8) S.xd[lane] = S.x[lane] - S.d[lane];
            __syncthreads();
            const double Sd = lm_compute_wave(csrc, cdst, n, S.xd, false, &S, nullptr);
#ifdef RTVM_RS_TRACE
            TRADD(13, tnow() - tl0);
#endif
            double temp_d[8], d[8], v[8];
            for (int i = 0; i < 8; i++) d[i] = S.d[i], v[i] = S.v[i];
            for (int i = 0; i < 8; i++) {
                double s = 0;
                for (int k = 0; k < 8; k++) s += S.A[i * 8 + k] * d[k];
                temp_d[i] = s * -1. + v[i] * 2.;
            }
            const double dS = dot8(d, temp_d);
            const double R = (Sres - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
            if (R > Rhi) {
                lambda *= 0.5;
                if (lambda < lc) lambda = 0;
            } else if (R < Rlo) {
                const double t = dot8(d, v);
                double nu = (Sd - Sres) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
                nu = fmin(fmax(nu, 2.), 10.);
                if (lambda == 0) {
                    TRADD(14, 1);
                    eig8_apply(&S, S.A, false);
                    double maxval = DBL_EPSILON;
                    for (int i = 0; i < 8; i++) maxval = fmax(maxval, fabs(S.diag[i]));
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            if (Sd < Sres) {
                __syncthreads();
                if (lane < 8) S.x[lane] = S.xd[lane];
                __syncthreads();
#ifdef RTVM_RS_TRACE
                const long long tj0 = tnow();
#endif
                Sres = lm_compute_wave(csrc, cdst, n, S.x, true, &S, &rinf);
#ifdef RTVM_RS_TRACE
                TRADD(15, tnow() - tj0); TRADD(9, 1);
#endif
            }
            iter++;
            double dinf = 0;
            for (int i = 0; i < 8; i++) dinf = fmax(dinf, fabs(d[i]));
            const bool proceed = iter < maxIters && dinf >= epsx && rinf >= epsf;
            if (!proceed) break;
        }
        __syncthreads();
        for (int i = 0; i < 8; i++) H[i] = S.x[i];
        const double sc = fabs(H[8]) > FLT_EPSILON ? 1. / H[8] : 1.;
        for (int i = 0; i < 9; i++) H[i] *= sc;
    }
    if (lane == 0) *found = 1;
    if (lane < 9) outH[lane] = H[lane];
#ifdef RTVM_RS_TRACE
    TR(6);
    __syncthreads();
    if (lane == 0) {
        // cycles: compaction, DLT sums, DLT Jacobi, DLT tail, first J, whole; counts
        double* o = outH;
        o[0] = (double)(g_tr[1] - g_tr[0]); o[1] = (double)(g_tr[2] - g_tr[1]); o[2] = (double)(g_tr[3] - g_tr[2]);
        o[3] = (double)(g_tr[4] - g_tr[3]); o[4] = (double)(g_tr[5] - g_tr[4]); o[5] = (double)(g_tr[6] - g_tr[0]);
        o[6] = (double)g_tr[10] + 1e-3 * (double)g_tr[12] + 1e-6 * (double)g_tr[14] + 1e-9 * (double)g_tr[9];
        o[7] = (double)g_tr[11]; o[8] = (double)g_tr[13] + 1e-9 * (double)g_tr[15];
    }
#endif
}

}  // namespace

size_t rtvm_ransac_state_bytes() { return sizeof(RsState); }

// raw outputs of cv::RNG((uint64)-1): state = (uint32)state * 4164903690 + (state >> 32)
void rtvm_ransac_raw_table(uint32_t* raw, int n) {
    uint64_t state = 0xFFFFFFFFFFFFFFFFull;
    for (int i = 0; i < n; i++) {
        state = (uint64_t)(uint32_t)state * 4164903690ull + (uint32_t)(state >> 32);
        raw[i] = (uint32_t)state;
    }
}

// findHomography(RANSAC) for slots sb .. sb+nb-1 in two parts: the search for the best
// 4-point hypothesis, then the polish (inlier mask, least squares on the inliers,
// Levenberg-Marquardt).  Point pairs in slot.rs_src / rs_dst (count in slot.rs_count);
// results in slot.rs_H (9 doubles), slot.rs_mask (count bytes), slot.rs_found.
int rtvm_launch_ransac_search(rtvm_ctx* c, int sb, int nb, double thresh, int max_iters, double confidence) {
    if (max_iters > RTVM_RANSAC_MAX_ITERS)
        RTVM_FAIL(c, RTVM_E_ARG, "findHomography: maxIters above %d is not supported", RTVM_RANSAC_MAX_ITERS);
    if (max_iters < 1) max_iters = 1;
    if (!(confidence > 0 && confidence < 1)) RTVM_FAIL(c, RTVM_E_ARG, "findHomography: confidence must be in (0,1)");
    const float thr = (float)(thresh * thresh);
    const FrameSlot* sl = c->d_slots;
    // Iterations in stages: with a usable inlier ratio the budget (RsState::niters) has dropped below the
    // end of the first or second stage by the time it has been replayed, and the later stages exit at once.
    // One or two problems leave the GPU idle either way: 64 hypotheses take as long as 16 (one wave each), so
    // the first stage is skipped and a round of launches saved; a batch pays for every hypothesis it evaluates.
    // The same goes for a batch with nothing else in flight (the first of a video, a short burst): the stage
    // it saves is a fifth of a millisecond of latency nothing else would fill.
    const int ends[3] = {nb <= 2 || c->rs_alone ? RS_BATCH0 : RS_BATCH0 / 4, RS_BATCH0, max_iters};
    int h0 = 0;
    for (int s = 0; s < 3 && h0 < max_iters; s++) {
        const int h1 = ends[s] < max_iters ? ends[s] : max_iters;
        if (h1 <= h0) continue;
        RTVM_LAUNCH(c, k_rs_subsets, dim3(nb), dim3(64), 0, sl, sb, c->d_rs_raw, c->rs_raw_n, h0, h1, confidence,
                    max_iters, h0 == 0 ? 1 : 0);
        RTVM_LAUNCH(c, k_rs_hyp, dim3(h1 - h0 < 256 ? h1 - h0 : 256, nb), dim3(64), 0, sl, sb, h0, h1, thr);
        RTVM_LAUNCH(c, k_rs_replay, dim3(nb), dim3(64), 0, sl, sb, h0, h1, confidence, h1 >= max_iters ? 1 : 0);
        h0 = h1;
    }
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

int rtvm_launch_ransac_refine(rtvm_ctx* c, int sb, int nb, double thresh) {
    const float thr = (float)(thresh * thresh);
    RTVM_LAUNCH(c, k_rs_refine, dim3(nb), dim3(64), 0, c->d_slots, sb, thr);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

int rtvm_launch_ransac(rtvm_ctx* c, int sb, int nb, double thresh, int max_iters, double confidence) {
    int r = rtvm_launch_ransac_search(c, sb, nb, thresh, max_iters, confidence);
    if (r != RTVM_OK) return r;
    return rtvm_launch_ransac_refine(c, sb, nb, thresh);
}
