// cv2.BFMatcher().knnMatch(q, t, k=2) with NORM_L2 (reference main.py:34,688) for
// gfx950: for every query the two nearest train descriptors by float32 Euclidean
// distance, ascending, earlier train index first on ties.
//
// SIFT descriptors are float32 holding the integers 0..255.  For such data every
// partial sum of squares is an integer below 2^24, so sum((q-t)^2) =
// |q|^2 + |t|^2 - 2 q.t exactly in float32 in any order, and bf16 holds the values
// exactly: the dot products run on the matrix cores (v_mfma_f32_32x32x16_bf16,
// k_l2_knn2_mfma).  Anything else (non-integer values, dim not a multiple of 16)
// takes k_l2_knn2_direct, which evaluates sum((q-t)^2) in index order with separate
// multiply and add, exactly as the sequential float32 loop does.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Top2 {
    float d0, d1;
    int i0, i1;
};
// insertion of (d, j) under the order "smaller d first, then smaller j"
__device__ __forceinline__ void top2_insert(Top2& t, float d, int j) {
    if (d < t.d1 || (d == t.d1 && j < t.i1)) {
        if (d < t.d0 || (d == t.d0 && j < t.i0)) {
            t.d1 = t.d0;
            t.i1 = t.i0;
            t.d0 = d;
            t.i0 = j;
        } else {
            t.d1 = d;
            t.i1 = j;
        }
    }
}

// flag[0] = 1 iff every value is an integer in [0, 255]
__global__ __launch_bounds__(256) void k_l2_check(const float* __restrict__ a, size_t na, const float* __restrict__ b,
                                                  size_t nb, int* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < na) {
        const float v = a[i];
        bad |= !(v >= 0.f && v <= 255.f && v == floorf(v));
    }
    if (i < nb) {
        const float v = b[i];
        bad |= !(v >= 0.f && v <= 255.f && v == floorf(v));
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

__global__ __launch_bounds__(256) void k_l2_norms(const float* __restrict__ a, int n, int dim, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < dim; k++) s += a[(size_t)i * dim + k] * a[(size_t)i * dim + k];
    out[i] = s;
}

__device__ __forceinline__ bf16x8 load_bf16x8(const float* __restrict__ p) {
    const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 r;
    r[0] = (short)(__float_as_uint(lo.x) >> 16);
    r[1] = (short)(__float_as_uint(lo.y) >> 16);
    r[2] = (short)(__float_as_uint(lo.z) >> 16);
    r[3] = (short)(__float_as_uint(lo.w) >> 16);
    r[4] = (short)(__float_as_uint(hi.x) >> 16);
    r[5] = (short)(__float_as_uint(hi.y) >> 16);
    r[6] = (short)(__float_as_uint(hi.z) >> 16);
    r[7] = (short)(__float_as_uint(hi.w) >> 16);
    return r;
}

// One wave per 32 queries.  C = T_tile (32 train x K) * Q^T (K x 32 queries): the
// accumulator column of a lane is its query, its 16 registers are 16 train rows, so
// the running top-2 of a query lives in two lanes (l, l+32) and is merged at the end.
// DIMS = dim / 16 k-steps; the query fragments stay in registers.
template <int DIMS>
__global__ __launch_bounds__(64) void k_l2_knn2_mfma(const float* __restrict__ q, int nq, const float* __restrict__ t,
                                                     int nt, const float* __restrict__ qn,
                                                     const float* __restrict__ tn, const int* __restrict__ flag,
                                                     int32_t* __restrict__ idx, float* __restrict__ dist) {
    if (!*flag) return;  // data is not 8-bit integer valued: the direct kernel handles it
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int dim = DIMS * 16;
    const int qi = min(blockIdx.x * 32 + r, nq - 1);
    bf16x8 bq[DIMS];
#pragma unroll
    for (int s = 0; s < DIMS; s++) bq[s] = load_bf16x8(q + (size_t)qi * dim + s * 16 + 8 * h);
    const float nqv = qn[qi];
    Top2 best = {INFINITY, INFINITY, -1, -1};
    for (int t0 = 0; t0 < nt; t0 += 32) {
        const int tj = min(t0 + r, nt - 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < DIMS; s++) {
            const bf16x8 at = load_bf16x8(t + (size_t)tj * dim + s * 16 + 8 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bq[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int j = t0 + row;
            if (j < nt) {
                const float d2 = (nqv + tn[j]) - 2.f * acc[e];
                top2_insert(best, sqrtf(d2), j);
            }
        }
    }
    // merge the two half-waves
    Top2 o;
    o.d0 = __shfl_xor(best.d0, 32);
    o.d1 = __shfl_xor(best.d1, 32);
    o.i0 = __shfl_xor(best.i0, 32);
    o.i1 = __shfl_xor(best.i1, 32);
    if (o.i0 >= 0) top2_insert(best, o.d0, o.i0);
    if (o.i1 >= 0) top2_insert(best, o.d1, o.i1);
    const int qo = blockIdx.x * 32 + r;
    if (h == 0 && qo < nq) {
        idx[2 * qo] = best.i0;
        idx[2 * qo + 1] = best.i1;
        dist[2 * qo] = best.d0;
        dist[2 * qo + 1] = best.d1;
    }
}

// Generic float data: 64 queries x 4 train sub-lanes per workgroup, train tiles of 64
// staged in LDS, sum((q-t)^2) in index order.
__global__ __launch_bounds__(256) void k_l2_knn2_direct(const float* __restrict__ q, int nq,
                                                        const float* __restrict__ t, int nt, int dim,
                                                        const int* __restrict__ flag, int use_flag,
                                                        int32_t* __restrict__ idx, float* __restrict__ dist) {
    if (use_flag && *flag) return;  // handled by the matrix-core kernel
    extern __shared__ float lds[];
    const int pitch = dim + 1;
    float* sq = lds;                 // 64 x pitch
    float* st = lds + 64 * pitch;    // 64 x pitch
    const int tid = threadIdx.x;
    const int qi = tid & 63, sub = tid >> 6;
    const int q0 = blockIdx.x * 64;
    for (int e = tid; e < 64 * dim; e += 256) {
        const int rr = e / dim, k = e % dim;
        sq[rr * pitch + k] = q[(size_t)min(q0 + rr, nq - 1) * dim + k];
    }
    Top2 best = {INFINITY, INFINITY, -1, -1};
    for (int t0 = 0; t0 < nt; t0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * dim; e += 256) {
            const int rr = e / dim, k = e % dim;
            st[rr * pitch + k] = t[(size_t)min(t0 + rr, nt - 1) * dim + k];
        }
        __syncthreads();
        for (int jj = 0; jj < 16; jj++) {
            const int jl = sub + 4 * jj, j = t0 + jl;
            if (j >= nt) break;
            float s = 0.f;
            for (int k = 0; k < dim; k++) {
                const float e = sq[qi * pitch + k] - st[jl * pitch + k];
                s += e * e;
            }
            top2_insert(best, sqrtf(s), j);
        }
    }
    // merge the 4 sub-lanes of a query through LDS
    __syncthreads();
    Top2* sb = reinterpret_cast<Top2*>(lds);
    sb[tid] = best;
    __syncthreads();
    if (sub == 0) {
        for (int s2 = 1; s2 < 4; s2++) {
            const Top2 o = sb[qi + 64 * s2];
            if (o.i0 >= 0) top2_insert(best, o.d0, o.i0);
            if (o.i1 >= 0) top2_insert(best, o.d1, o.i1);
        }
        if (q0 + qi < nq) {
            idx[2 * (q0 + qi)] = best.i0;
            idx[2 * (q0 + qi) + 1] = best.i1;
            dist[2 * (q0 + qi)] = best.d0;
            dist[2 * (q0 + qi) + 1] = best.d1;
        }
    }
}

}  // namespace

// dq: nq x dim, dt: nt x dim (device); d_idx / d_dist: nq x 2; d_norms: nq + nt floats; d_flag: 1 int
int rtvm_launch_l2_knn2(rtvm_ctx* c, const float* dq, int nq, const float* dt, int nt, int dim, int32_t* d_idx,
                        float* d_dist, float* d_norms, int* d_flag) {
    if (nq <= 0) return RTVM_OK;
    // two 64-row tiles; the same memory then holds the 256 partial results of the final merge
    size_t lds_bytes = (size_t)2 * 64 * (dim + 1) * sizeof(float);
    if (lds_bytes < 256 * sizeof(Top2)) lds_bytes = 256 * sizeof(Top2);
    if (lds_bytes > 160 * 1024) RTVM_FAIL(c, RTVM_E_ARG, "match_l2_knn2: descriptor dimension %d is too large", dim);
    const bool mfma = dim == 128 || dim == 64;
    if (mfma) {
        HIP_TRY(c, hipMemsetAsync(d_flag, 1, sizeof(int), c->stream));  // non-zero = "all values are 8-bit integers"

        const size_t na = (size_t)nq * dim, nb = (size_t)nt * dim;
        const size_t nmax = na > nb ? na : nb;
        RTVM_LAUNCH(c, k_l2_check, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, dq, na, dt, nb, d_flag);
        RTVM_LAUNCH(c, k_l2_norms, dim3((nq + 255) / 256), dim3(256), 0, dq, nq, dim, d_norms);
        RTVM_LAUNCH(c, k_l2_norms, dim3((nt + 255) / 256), dim3(256), 0, dt, nt, dim, d_norms + nq);
        if (dim == 128)
            RTVM_LAUNCH(c, k_l2_knn2_mfma<8>, dim3((nq + 31) / 32), dim3(64), 0, dq, nq, dt, nt, d_norms, d_norms + nq,
                        d_flag, d_idx, d_dist);
        else
            RTVM_LAUNCH(c, k_l2_knn2_mfma<4>, dim3((nq + 31) / 32), dim3(64), 0, dq, nq, dt, nt, d_norms, d_norms + nq,
                        d_flag, d_idx, d_dist);
    }
    (void)hipFuncSetAttribute((const void*)k_l2_knn2_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    RTVM_LAUNCH(c, k_l2_knn2_direct, dim3((nq + 63) / 64), dim3(256), lds_bytes, dq, nq, dt, nt, dim, d_flag,
                mfma ? 1 : 0, d_idx, d_dist);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}
