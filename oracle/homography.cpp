// CPU restatement of cv2.findHomography(src, dst, cv2.RANSAC, thresh)
// (reference main.py:856-857): OpenCV 4.x's RANSACPointSetRegistrator::run,
// HomographyEstimatorCallback (runKernel / computeError / checkSubset),
// RANSACUpdateNumIters, and the Levenberg-Marquardt polish (LMSolverImpl::run
// with HomographyRefineCallback), restated from their published algorithm.
// Test infrastructure only (see oracle.h).  Parity unpinned.
//
// One deliberate difference from OpenCV: sums that run over all inlier points
// (centroids, scales, LtL, JtJ, Jt r, |r|^2) are accumulated in a fixed 64-way
// interleaved order (sum64 below) instead of strictly sequentially, so that a
// 64-lane reduction reproduces them bit for bit.  The results differ from the
// sequential order by rounding only (relative 1e-16).
#include "oracle.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline int cv_round(double v) { return (int)std::nearbyint(v); }

// partial[l] = sum of terms i with i % 64 == l in increasing i, then the
// butterfly p[l] += p[l ^ 32], ^16, ^8, ^4, ^2, ^1.
struct Sum64 {
    double p[64];
    Sum64() {
        for (int l = 0; l < 64; l++) p[l] = 0.;
    }
    inline void add(int i, double v) { p[i & 63] += v; }
    double total() const {
        double a[64], b[64];
        for (int l = 0; l < 64; l++) a[l] = p[l];
        for (int m = 32; m >= 1; m >>= 1) {
            for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ m];
            for (int l = 0; l < 64; l++) a[l] = b[l];
        }
        return a[0];
    }
};

inline double cv_hypot(double a, double b) {
    a = std::fabs(a);
    b = std::fabs(b);
    if (a > b) {
        b /= a;
        return a * std::sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * std::sqrt(1 + a * a);
    }
    return 0;
}

// cv::JacobiImpl_<double>: eigenvalues W (descending) and eigenvectors (rows of V)
// of the symmetric n x n matrix A (only its upper triangle is used; destroyed).
void jacobi(double* A, int astep, double* W, double* V, int vstep, int n) {
    const double eps = DBL_EPSILON;
    int i, j, k, m;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * vstep + j] = 0;
        V[i * vstep + i] = 1;
    }
    int iters, maxIters = n * n * 30;
    int indR[16], indC[16];
    double mv = 0;
    for (k = 0; k < n; k++) {
        W[k] = A[(astep + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = std::fabs(A[astep * k + m]), i = k + 2; i < n; i++) {
                double val = std::fabs(A[astep * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = std::fabs(A[k]), i = 1; i < k; i++) {
                double val = std::fabs(A[astep * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1)
        for (iters = 0; iters < maxIters; iters++) {
            for (k = 0, mv = std::fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                double val = std::fabs(A[astep * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                double val = std::fabs(A[astep * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            double p = A[astep * k + l];
            if (std::fabs(p) <= eps) break;
            double y = (W[l] - W[k]) * 0.5;
            double t = std::fabs(y) + cv_hypot(p, y);
            double s = cv_hypot(p, t);
            double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[astep * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define ROTATE(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
            for (i = 0; i < k; i++) ROTATE(A[astep * i + k], A[astep * i + l]);
            for (i = k + 1; i < l; i++) ROTATE(A[astep * k + i], A[astep * i + l]);
            for (i = l + 1; i < n; i++) ROTATE(A[astep * k + i], A[astep * l + i]);
            for (i = 0; i < n; i++) ROTATE(V[vstep * k + i], V[vstep * l + i]);
#undef ROTATE
            for (j = 0; j < 2; j++) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = std::fabs(A[astep * idx + m]), i = idx + 2; i < n; i++) {
                        double val = std::fabs(A[astep * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = std::fabs(A[idx]), i = 1; i < idx; i++) {
                        double val = std::fabs(A[astep * i + idx]);
                        if (mv < val) mv = val, m = i;
                    }
                    indC[idx] = m;
                }
            }
        }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            std::swap(W[m], W[k]);
            for (i = 0; i < n; i++) std::swap(V[vstep * m + i], V[vstep * k + i]);
        }
    }
}

inline void mat3mul(const double* a, const double* b, double* d) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// HomographyEstimatorCallback::runKernel on points M (source) -> m (destination).
// Returns 0 when the configuration is degenerate.
int run_kernel(const float* M, const float* m, int count, double* H) {
    Sum64 scmx, scmy, scMx, scMy;
    for (int i = 0; i < count; i++) {
        scmx.add(i, m[2 * i]);
        scmy.add(i, m[2 * i + 1]);
        scMx.add(i, M[2 * i]);
        scMy.add(i, M[2 * i + 1]);
    }
    double cmx = scmx.total() / count, cmy = scmy.total() / count;
    double cMx = scMx.total() / count, cMy = scMy.total() / count;
    Sum64 ssmx, ssmy, ssMx, ssMy;
    for (int i = 0; i < count; i++) {
        ssmx.add(i, std::fabs(m[2 * i] - cmx));
        ssmy.add(i, std::fabs(m[2 * i + 1] - cmy));
        ssMx.add(i, std::fabs(M[2 * i] - cMx));
        ssMy.add(i, std::fabs(M[2 * i + 1] - cMy));
    }
    double smx = ssmx.total(), smy = ssmy.total(), sMx = ssMx.total(), sMy = ssMy.total();
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON ||
        std::fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx;
    smy = count / smy;
    sMx = count / sMx;
    sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    std::vector<Sum64> acc(45);
    for (int i = 0; i < count; i++) {
        const double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        const double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        int e = 0;
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) acc[e++].add(i, Lx[j] * Lx[k] + Ly[j] * Ly[k]);
    }
    double LtL[81], W[9], V[81];
    int e = 0;
    for (int j = 0; j < 9; j++)
        for (int k = j; k < 9; k++) {
            const double v = acc[e++].total();
            LtL[j * 9 + k] = v;
            LtL[k * 9 + j] = v;
        }
    jacobi(LtL, 9, W, V, 9, 9);
    double Htemp[9], H0[9];
    mat3mul(invHnorm, V + 72, Htemp);
    mat3mul(Htemp, Hnorm2, H0);
    const double sc = 1. / H0[8];
    for (int i = 0; i < 9; i++) H[i] = H0[i] * sc;
    return 1;
}

// HomographyEstimatorCallback::computeError + findInliers (float32 arithmetic)
int find_inliers(const float* M, const float* m, int count, const double* H, float thresh_sq, uint8_t* mask) {
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    int good = 0;
    for (int i = 0; i < count; i++) {
        const float Mx = M[2 * i], My = M[2 * i + 1];
        const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
        const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - m[2 * i];
        const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - m[2 * i + 1];
        const float err = dx * dx + dy * dy;
        const int f = err <= thresh_sq;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

bool have_collinear(const float* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

inline double det3pts(const float* p, int a, int b, int c) {
    const double m00 = p[2 * a], m01 = p[2 * a + 1], m10 = p[2 * b], m11 = p[2 * b + 1], m20 = p[2 * c], m21 = p[2 * c + 1];
    // cv::determinant of [[x,y,1]...]: m00*(m11*m22 - m12*m21) - m01*(m10*m22 - m12*m20) + m02*(m10*m21 - m11*m20)
    return m00 * (m11 * 1. - 1. * m21) - m01 * (m10 * 1. - 1. * m20) + 1. * (m10 * m21 - m11 * m20);
}

bool check_subset(const float* ms1, const float* ms2) {
    if (have_collinear(ms1, 4) || have_collinear(ms2, 4)) return false;
    static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; i++) {
        const int* t = tt[i];
        negative += det3pts(ms1, t[0], t[1], t[2]) * det3pts(ms2, t[0], t[1], t[2]) < 0;
    }
    return negative == 0 || negative == 4;
}

struct Rng {
    uint64_t state;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool get_subset(const float* m1, const float* m2, int count, Rng& rng, float* ms1, float* ms2, int maxAttempts) {
    int idx[4];
    for (int iters = 0; iters < maxAttempts; ++iters) {
        for (int i = 0; i < 4; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng.uniform(0, count);
                bool dup = false;
                for (int k = 0; k < i; k++) dup |= idx[k] == idx_i;
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[2 * i] = m1[2 * idx_i];
            ms1[2 * i + 1] = m1[2 * idx_i + 1];
            ms2[2 * i] = m2[2 * idx_i];
            ms2[2 * i + 1] = m2[2 * idx_i + 1];
        }
        if (!check_subset(ms1, ms2)) continue;
        return true;
    }
    return false;
}

int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::fmax(p, 0.);
    p = std::fmin(p, 1.);
    ep = std::fmax(ep, 0.);
    ep = std::fmin(ep, 1.);
    double num = std::fmax(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round(num / denom);
}

// HomographyRefineCallback::compute: residuals r (2 per point) and, when JtJ != 0,
// the normal matrix JtJ (8x8), Jt r (8) accumulated over the points; returns |r|^2.
double lm_compute(const float* M, const float* m, int count, const double* h, double* JtJ, double* Jtr, double* rinf) {
    std::vector<Sum64> acc(JtJ ? 44 : 0);
    Sum64 ss;
    double rmax = 0;
    for (int i = 0; i < count; i++) {
        const double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double ex = xi - m[2 * i], ey = yi - m[2 * i + 1];
        ss.add(i, ex * ex + ey * ey);
        rmax = std::fmax(rmax, std::fmax(std::fabs(ex), std::fabs(ey)));
        if (JtJ) {
            const double Jx[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            const double Jy[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            int e = 0;
            for (int a = 0; a < 8; a++)
                for (int b = a; b < 8; b++) acc[e++].add(i, Jx[a] * Jx[b] + Jy[a] * Jy[b]);
            for (int a = 0; a < 8; a++) acc[36 + a].add(i, Jx[a] * ex + Jy[a] * ey);
        }
    }
    if (JtJ) {
        int e = 0;
        for (int a = 0; a < 8; a++)
            for (int b = a; b < 8; b++) {
                const double v = acc[e++].total();
                JtJ[a * 8 + b] = v;
                JtJ[b * 8 + a] = v;
            }
        for (int a = 0; a < 8; a++) Jtr[a] = acc[36 + a].total();
    }
    if (rinf) *rinf = rmax;
    return ss.total();
}

// cv::solve(A, b, x, DECOMP_EIG) for a symmetric 8x8: Jacobi + SVBkSb
void solve_eig8(const double* Ain, const double* b, double* x) {
    double A[64], W[8], V[64];
    memcpy(A, Ain, sizeof(A));
    jacobi(A, 8, W, V, 8, 8);
    double threshold = 0;
    for (int i = 0; i < 8; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < 8; j++) x[j] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < 8; j++) s += V[i * 8 + j] * b[j];
        s *= wi;
        for (int j = 0; j < 8; j++) x[j] = x[j] + s * V[i * 8 + j];
    }
}

// diagonal of cv::invert(A, DECOMP_EIG) for a symmetric 8x8
void inv_eig8_diag(const double* Ain, double* diag) {
    double A[64], W[8], V[64];
    memcpy(A, Ain, sizeof(A));
    jacobi(A, 8, W, V, 8, 8);
    double threshold = 0;
    for (int i = 0; i < 8; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int j = 0; j < 8; j++) diag[j] = 0;
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (std::fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int j = 0; j < 8; j++) diag[j] = diag[j] + V[i * 8 + j] * (V[i * 8 + j] * wi);
    }
}

inline double dot8(const double* a, const double* b) {
    double r = 0;
    r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    r += a[4] * b[4] + a[5] * b[5] + a[6] * b[6] + a[7] * b[7];
    return r;
}

// LMSolverImpl::run with maxIters = 10, eps = FLT_EPSILON, on the 8 free parameters
void lm_refine(const float* M, const float* m, int count, double* H) {
    const int lx = 8, maxIters = 10;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], D[8], temp_d[8];
    for (int i = 0; i < 8; i++) x[i] = H[i];
    double rinf;
    double S = lm_compute(M, m, count, x, A, v, &rinf);
    for (int i = 0; i < lx; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
        solve_eig8(Ap, v, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        const double Sd = lm_compute(M, m, count, xd, nullptr, nullptr, nullptr);
        for (int i = 0; i < lx; i++) {  // temp_d = -A*d + 2*v
            double s = 0;
            for (int k = 0; k < lx; k++) s += A[i * 8 + k] * d[k];
            temp_d[i] = s * -1. + v[i] * 2.;
        }
        const double dS = dot8(d, temp_d);
        const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            const double t = dot8(d, v);
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::fmin(std::fmax(nu, 2.), 10.);
            if (lambda == 0) {
                double diag[8];
                inv_eig8_diag(A, diag);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = std::fmax(maxval, std::fabs(diag[i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            for (int i = 0; i < lx; i++) x[i] = xd[i];
            S = lm_compute(M, m, count, x, A, v, &rinf);
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < lx; i++) dinf = std::fmax(dinf, std::fabs(d[i]));
        const bool proceed = iter < maxIters && dinf >= epsx && rinf >= epsf;
        if (!proceed) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
}

}  // namespace

extern "C" void rtvm_oracle_chain3x3(const double A[9], const double B[9], double C[9]) {
    double out[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            out[3 * r + c] = __builtin_fma(A[3 * r + 2], B[6 + c], __builtin_fma(A[3 * r + 1], B[3 + c], A[3 * r] * B[c]));
    for (int i = 0; i < 9; i++) C[i] = out[i];
}

extern "C" int rtvm_oracle_find_homography_ransac(const float* src_xy, const float* dst_xy, int count, double thresh,
                                                  int max_iters, double confidence, double H[9], uint8_t* inlier_mask,
                                                  int* found) {
    if (!found || count < 0) return -1;
    *found = 0;
    for (int i = 0; i < 9; i++) H[i] = 0;
    std::vector<uint8_t> mask(count > 0 ? count : 1, 0), best(count > 0 ? count : 1, 0);
    if (count < 4) return 0;  // cv::findHomography returns an empty matrix
    double model[9], bestModel[9];
    bool result = false;
    if (count == 4) {
        result = run_kernel(src_xy, dst_xy, 4, bestModel) > 0;
        for (int i = 0; i < count; i++) best[i] = 1;
    } else {
        const float thr = (float)(thresh * thresh);
        int niters = max_iters > 1 ? max_iters : 1, maxGood = 0;
        Rng rng{0xFFFFFFFFFFFFFFFFull};
        float ms1[8], ms2[8];
        for (int iter = 0; iter < niters; iter++) {
            if (!get_subset(src_xy, dst_xy, count, rng, ms1, ms2, 10000)) {
                if (iter == 0) return 0;
                break;
            }
            if (run_kernel(ms1, ms2, 4, model) <= 0) continue;
            const int good = find_inliers(src_xy, dst_xy, count, model, thr, mask.data());
            if (good > (maxGood > 3 ? maxGood : 3)) {
                std::swap(mask, best);
                memcpy(bestModel, model, sizeof(model));
                maxGood = good;
                niters = update_num_iters(confidence, (double)(count - good) / count, 4, niters);
            }
        }
        result = maxGood > 0;
    }
    if (!result) return 0;
    if (count > 4) {
        std::vector<float> s, d;
        for (int i = 0; i < count; i++)
            if (best[i]) {
                s.push_back(src_xy[2 * i]);
                s.push_back(src_xy[2 * i + 1]);
                d.push_back(dst_xy[2 * i]);
                d.push_back(dst_xy[2 * i + 1]);
            }
        const int n = (int)s.size() / 2;
        if (n > 0) {
            run_kernel(s.data(), d.data(), n, bestModel);
            lm_refine(s.data(), d.data(), n, bestModel);
            // H.convertTo(H, type, scaleFor(H(2,2)))
            const double sc = std::fabs(bestModel[8]) > FLT_EPSILON ? 1. / bestModel[8] : 1.;
            for (int i = 0; i < 9; i++) bestModel[i] *= sc;
        }
    }
    memcpy(H, bestModel, 9 * sizeof(double));
    if (inlier_mask) memcpy(inlier_mask, best.data(), count);
    *found = 1;
    return 0;
}
