// CPU restatement of the two brute-force matchers on the reference's hot path.
// Test infrastructure only (see oracle.h).  Parity unpinned: restates OpenCV 4.x
// BFMatcher semantics from its published algorithm; no reference fixture exists.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

int rtvm_oracle_nthreads = 1;
extern "C" int rtvm_oracle_set_threads(int n) {
    const int old = rtvm_oracle_nthreads;
    rtvm_oracle_nthreads = n < 1 ? 1 : n;
    return old;
}

namespace {

inline int hamming32(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
}

// nearest row of B for every row of A; first minimum wins (strict '<' scan in
// ascending index), as cv::batchDistance does for K=1
void nearest(const uint8_t* A, int nA, const uint8_t* B, int nB, std::vector<int>& idx,
             std::vector<int>& dst) {
    idx.assign(nA, -1);
    dst.assign(nA, 0x7fffffff);
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int i = 0; i < nA; i++)
        for (int j = 0; j < nB; j++) {
            int d = hamming32(A + 32 * i, B + 32 * j);
            if (d < dst[i]) {
                dst[i] = d;
                idx[i] = j;
            }
        }
}

}  // namespace

// main.py:37,695: BFMatcher(NORM_HAMMING, crossCheck=True).match(des_cur, des_prev).
// A pair (i, j) is emitted iff j is i's nearest train row and i is j's nearest
// query row; output in ascending i.  sorted != 0 applies main.py:698
// (sorted(matches, key=distance), a stable sort).
extern "C" int rtvm_oracle_match_hamming_xcheck(const uint8_t* q, int nq, const uint8_t* t, int nt,
                                                int32_t* qidx, int32_t* tidx, int32_t* dist,
                                                int* m_out, int sorted) {
    if (nq < 0 || nt < 0 || !m_out) return -1;
    std::vector<int> sidx, sdst, ridx, rdst;
    nearest(q, nq, t, nt, sidx, sdst);
    nearest(t, nt, q, nq, ridx, rdst);
    struct M {
        int q, t, d;
    };
    std::vector<M> ms;
    for (int i = 0; i < nq; i++) {
        int j = sidx[i];
        if (j >= 0 && ridx[j] == i) ms.push_back({i, j, sdst[i]});
    }
    if (sorted)
        std::stable_sort(ms.begin(), ms.end(), [](const M& a, const M& b) { return a.d < b.d; });
    for (size_t k = 0; k < ms.size(); k++) {
        qidx[k] = ms[k].q;
        tidx[k] = ms[k].t;
        dist[k] = ms[k].d;
    }
    *m_out = (int)ms.size();
    return 0;
}

// The same call under BFMatcher's cross-check before OpenCV 4.2 / 3.4.9 (the match is taken from
// the train -> query direction): query i gets the nearest of the train rows whose nearest query is
// i (lowest train index on ties); queries that are no train row's nearest emit nothing.
extern "C" int rtvm_oracle_match_hamming_legacy(const uint8_t* q, int nq, const uint8_t* t, int nt,
                                                int32_t* qidx, int32_t* tidx, int32_t* dist, int* m_out, int sorted) {
    if (nq < 0 || nt < 0 || !m_out) return -1;
    std::vector<int> ridx, rdst;
    nearest(t, nt, q, nq, ridx, rdst);
    struct M {
        int q, t, d;
    };
    std::vector<M> best(nq, M{-1, -1, 0x7fffffff});
    for (int j = 0; j < nt; j++) {
        const int i = ridx[j];
        if (i >= 0 && rdst[j] < best[i].d) best[i] = M{i, j, rdst[j]};
    }
    std::vector<M> ms;
    for (int i = 0; i < nq; i++)
        if (best[i].t >= 0) ms.push_back(best[i]);
    if (sorted)
        std::stable_sort(ms.begin(), ms.end(), [](const M& a, const M& b) { return a.d < b.d; });
    for (size_t k = 0; k < ms.size(); k++) {
        qidx[k] = ms[k].q;
        tidx[k] = ms[k].t;
        dist[k] = ms[k].d;
    }
    *m_out = (int)ms.size();
    return 0;
}

// main.py:34,688: BFMatcher() (NORM_L2, no cross-check).knnMatch(q, t, k=2).
// Distance is sqrt of the float32 sum of squared differences; the two smallest
// per query in ascending order, earlier train index first on ties (insertion
// with strict comparisons, as cv::batchDistance does for K=2).
extern "C" int rtvm_oracle_match_l2_knn2(const float* q, int nq, const float* t, int nt, int dim,
                                         int32_t* idx, float* dist) {
    if (nq < 0 || nt < 2 || dim <= 0) return -1;
#pragma omp parallel for schedule(static) num_threads(rtvm_oracle_nthreads)
    for (int i = 0; i < nq; i++) {
        float d0 = INFINITY, d1 = INFINITY;
        int i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            float s = 0.f;
            for (int k = 0; k < dim; k++) {
                float e = q[(size_t)i * dim + k] - t[(size_t)j * dim + k];
                s += e * e;
            }
            float d = std::sqrt(s);
            if (d < d1) {
                if (d0 > d) {
                    d1 = d0;
                    i1 = i0;
                    d0 = d;
                    i0 = j;
                } else {
                    d1 = d;
                    i1 = j;
                }
            }
        }
        idx[2 * i] = i0;
        idx[2 * i + 1] = i1;
        dist[2 * i] = d0;
        dist[2 * i + 1] = d1;
    }
    return 0;
}
