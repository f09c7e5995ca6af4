// cv2.BFMatcher().knnMatch(q, t, k=2) with NORM_L2 (reference main.py:34,688) for
// gfx950: for every query the two nearest train descriptors by float32 Euclidean
// distance, ascending, earlier train index first on ties.
//
// SIFT descriptors are float32 holding the integers 0..255.  For such data every
// partial sum of squares is an integer below 2^24, so sum((q-t)^2) =
// |q|^2 + |t|^2 - 2 q.t exactly in float32 in any order, and bf16 holds the values
// exactly: the dot products run on the matrix cores (v_mfma_f32_32x32x16_bf16: k_l2_prep,
// k_l2_tile, k_l2_merge; rows that hold other values or whose second distance reaches 2^22
// are recomputed by k_l2_far).  Descriptors whose length is not a multiple of 16 take
// k_l2_knn2_direct.  Both evaluate sum((q-t)^2) in index order with separate multiply and
// add, exactly as the sequential float32 loop does.
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

// ---- matrix-core path -------------------------------------------------------------------------
//
// k_l2_prep   converts both descriptor sets to bf16 once, in the order the matrix-core operands
//             are consumed: for every block of 32 rows and every k-step of 16 columns, the 64
//             operand registers of a wave (lane = 32 * half + row, 8 consecutive columns each)
//             are 1 KiB of consecutive memory.  Also the squared norms and the "all values are
//             integers 0..255" flag.
// k_l2_tile   one workgroup per 128 queries x 128 train rows: the train tile goes through LDS
//             (coalesced 16-byte copies in, conflict-free 16-byte operand reads out), every wave
//             keeps its 32 queries' operands in registers and runs 4 x KS MFMAs.  C = T * Q^T,
//             so the accumulator column of a lane is its query and its registers are train
//             rows: the running top-2 of a query is two packed keys per lane,
//             (d2 << 7 | row in tile), updated with three integer min/max per element.
// k_l2_merge  top-2 of every query over the column tiles, sqrt, output.
//
// Order of the results: OpenCV compares the float32 distances sqrt(d2), earlier train row first
// on ties.  Two different integers d2 can round to the same float sqrt only from 2^22 up (below,
// neighbouring integers' roots are at least 2 ulp apart), so keys in d2 give the same order
// whenever the second-best d2 of a query is below 2^22: always for real SIFT descriptors
// (norm 512, d2 <= 2^20).  Queries beyond that go to k_l2_far, which orders by the rounded roots.

constexpr int L2_TILE = 128;
constexpr uint32_t L2_NOKEY = 0xFFFFFFFFu;
constexpr uint32_t L2_FAR = 1u << 22;

// grid.y = 0: queries, 1: train rows.  One thread per (row, group of 8 columns); KS = dim / 16.
template <int KS>
__device__ __forceinline__ void l2_prep_rows(const float* __restrict__ src, int n, int n_pad, uint4* __restrict__ dstb,
                                             float* __restrict__ dstn, int* __restrict__ flag) {
    constexpr int G = 2 * KS;  // groups of 8 columns per row
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int row = id / G, g = id % G;
    if (row >= n_pad) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < n) {
        const float4 lo = *reinterpret_cast<const float4*>(src + (size_t)row * (16 * KS) + 8 * g);
        const float4 hi = *reinterpret_cast<const float4*>(src + (size_t)row * (16 * KS) + 8 * g + 4);
        v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
    bool bad = false;
    float ss = 0.f;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        bad |= !(v[i] >= 0.f && v[i] <= 255.f && v[i] == floorf(v[i]));
        ss += v[i] * v[i];   // exact for 8-bit integers in any order (below 2^24)
        const uint32_t h = __float_as_uint(v[i]) >> 16;
        if (i & 1)
            w[i >> 1] |= h << 16;
        else
            w[i >> 1] = h;
    }
    if (flag && __any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
    // the G threads of a row are consecutive lanes of one wave (G divides 64)
#pragma unroll
    for (int d = 1; d < G; d <<= 1) ss += __shfl_xor(ss, d);
    if (g == 0) dstn[row] = ss;
    const int s = g >> 1, h = g & 1;
    dstb[((size_t)(row >> 5) * KS + s) * 64 + h * 32 + (row & 31)] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int KS>
__global__ __launch_bounds__(256) void k_l2_prep(const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
                                                 uint4* __restrict__ qb, uint4* __restrict__ tb,
                                                 float* __restrict__ qn, float* __restrict__ tn, int* __restrict__ flag) {
    const bool train = blockIdx.y != 0;
    const int n = train ? nt : nq;
    l2_prep_rows<KS>(train ? t : q, n, (n + L2_TILE - 1) / L2_TILE * L2_TILE, train ? tb : qb, train ? tn : qn, flag);
}

// The same for the frames of a batch: blockIdx.z = frame, descriptors and counts from the slots
// (SIFT descriptors are 8-bit integers by construction: no check).
__global__ __launch_bounds__(256) void k_l2_prep_slots(const FrameSlot* __restrict__ slots, int sb, int cap_pad) {
    const FrameSlot& S = slots[sb + blockIdx.z];
    const bool train = blockIdx.y != 0;
    const FeatSet& F = train ? S.prev : S.feat;
    l2_prep_rows<8>(reinterpret_cast<const float*>(F.desc), min(*F.count, cap_pad), cap_pad, train ? S.l2_tb : S.l2_qb,
                    train ? S.l2_tn : S.l2_qn, nullptr);
}

__device__ __forceinline__ void key_top2(uint32_t& k0, uint32_t& k1, uint32_t key) {
    const uint32_t lo = min(k0, key), hi = max(k0, key);
    k0 = lo;
    k1 = min(k1, hi);
}

// blockIdx.x = query tile, blockIdx.y = train tile.  part: [train tiles][padded queries][2] keys.
template <int KS>
__device__ __forceinline__ void l2_tile_body(const uint4* __restrict__ qb, const uint4* __restrict__ tb,
                                             const float* __restrict__ qn, const float* __restrict__ tn, int nt, int nq_pad,
                                             uint32_t* __restrict__ part) {
    __shared__ uint4 s_t[4 * KS * 64];
    __shared__ __align__(16) float s_tn[L2_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int t0 = blockIdx.y * L2_TILE;
    {
        const uint4* __restrict__ src = tb + (size_t)blockIdx.y * (4 * KS * 64);
#pragma unroll
        for (int i = 0; i < KS; i++) s_t[tid + 256 * i] = src[tid + 256 * i];
        if (tid < L2_TILE) s_tn[tid] = tn[t0 + tid];
    }
    // the wave's 32 queries: operand registers for all k-steps
    const int qblock = blockIdx.x * 4 + wave;
    bf16x8 bq[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const uint4 u = qb[((size_t)qblock * KS + s) * 64 + lane];
        bq[s] = __builtin_bit_cast(bf16x8, u);
    }
    const float nqv = qn[qblock * 32 + (lane & 31)];
    __syncthreads();
    uint32_t k0 = L2_NOKEY, k1 = L2_NOKEY;
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const bf16x8 at = __builtin_bit_cast(bf16x8, s_t[(rb * KS + s) * 64 + lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bq[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            // accumulator registers 4*g4 .. 4*g4+3 are train rows 8*g4 + 4*h + (0..3) of the row block
            const int r0 = 32 * rb + 8 * g4 + 4 * h;
            const float4 n4 = *reinterpret_cast<const float4*>(&s_tn[r0]);
            const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float d2 = (nqv + nn[i]) - 2.f * acc[4 * g4 + i];   // exact: integers below 2^24
                const uint32_t key = ((uint32_t)d2 << 7) | (uint32_t)(r0 + i);
                key_top2(k0, k1, t0 + r0 + i < nt ? key : L2_NOKEY);
            }
        }
    }
    // the two lanes of a query
    const uint32_t o0 = __shfl_xor(k0, 32), o1 = __shfl_xor(k1, 32);
    key_top2(k0, k1, o0);
    key_top2(k0, k1, o1);
    if (h == 0) {
        const size_t o = ((size_t)blockIdx.y * nq_pad + qblock * 32 + lane) * 2;
        part[o] = k0;
        part[o + 1] = k1;
    }
}

template <int KS>
__global__ __launch_bounds__(256) void k_l2_tile(const uint4* __restrict__ qb, const uint4* __restrict__ tb,
                                                 const float* __restrict__ qn, const float* __restrict__ tn, int nt,
                                                 int nq_pad, const int* __restrict__ flag, uint32_t* __restrict__ part) {
    if (!*flag) return;  // not 8-bit integer data: k_l2_knn2_direct computes the result
    l2_tile_body<KS>(qb, tb, qn, tn, nt, nq_pad, part);
}

// blockIdx.z = frame of the batch; tiles beyond the frame's descriptor counts have nothing to do.
__global__ __launch_bounds__(256) void k_l2_tile_slots(const FrameSlot* __restrict__ slots, int sb, int cap_pad) {
    const FrameSlot& S = slots[sb + blockIdx.z];
    const int nq = min(*S.feat.count, cap_pad), nt = min(*S.prev.count, cap_pad);
    if ((int)blockIdx.x * L2_TILE >= nq || (int)blockIdx.y * L2_TILE >= nt) return;
    l2_tile_body<8>(S.l2_qb, S.l2_tb, S.l2_qn, S.l2_tn, nt, cap_pad, S.l2_part);
}

// main.py:688-698 behind the tiles, one workgroup per frame: the two nearest train rows of every
// query (merge over the column tiles, roots), Lowe's ratio test `d0 < 0.7 * d1` in double
// precision as Python evaluates it, compaction in query order, stable sort by distance (keys:
// bit pattern of the float distance, then the position before sorting).  SIFT descriptors have
// norm 512 at most, so d2 stays below 2^22 and the order of d2 is the order of the roots.
__global__ __launch_bounds__(1024) void k_l2_ratio_sort(const FrameSlot* __restrict__ slots, int sb, int cap_pad) {
    const FrameSlot& S = slots[sb + blockIdx.x];
    __shared__ unsigned long long s_key[RTVM_SIFT_STREAM_MAX_KP];
    __shared__ uint16_t s_q[RTVM_SIFT_STREAM_MAX_KP], s_t[RTVM_SIFT_STREAM_MAX_KP];
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = min(*S.feat.count, cap_pad), nt = min(*S.prev.count, cap_pad);
    const int n_tiles = (nt + L2_TILE - 1) / L2_TILE;
    const uint32_t* __restrict__ part = S.l2_part;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        float d0 = 0.f;
        int j0 = 0;
        if (i < nq && nt >= 2) {  // with fewer than two train rows the reference has no pair to test
            unsigned long long b0 = ~0ull, b1 = ~0ull;
            for (int tI = 0; tI < n_tiles; tI++) {
                const uint2 k = *reinterpret_cast<const uint2*>(part + ((size_t)tI * cap_pad + i) * 2);
                const uint32_t ks[2] = {k.x, k.y};
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (ks[u] == L2_NOKEY) continue;
                    const unsigned long long key =
                        ((unsigned long long)(ks[u] >> 7) << 32) | (uint32_t)(tI * L2_TILE + (ks[u] & 127));
                    const unsigned long long lo = key < b0 ? key : b0, hi = key < b0 ? b0 : key;
                    b0 = lo;
                    b1 = hi < b1 ? hi : b1;
                }
            }
            d0 = sqrtf((float)(uint32_t)(b0 >> 32));
            const float d1 = sqrtf((float)(uint32_t)(b1 >> 32));
            j0 = (int)(uint32_t)b0;
            ok = (double)d0 < 0.7 * (double)d1;
        }
        const unsigned long long bal = __ballot(ok);
        const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wave[w];
        const int base = s_base;
        if (ok) {
            const int pos = base + woff + wrank;
            s_key[pos] = ((unsigned long long)__float_as_uint(d0) << 32) | (uint32_t)pos;
            s_q[pos] = (uint16_t)i;
            s_t[pos] = (uint16_t)j0;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wave[w];
            s_base = base + tot;
        }
        __syncthreads();
    }
    const int m = s_base;
    if (m > 1) {
        int n2 = 1;
        while (n2 < m) n2 <<= 1;
        for (int i = m + tid; i < n2; i += 1024) s_key[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = s_key[i], b = s_key[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) {
                            s_key[i] = b;
                            s_key[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    for (int p = tid; p < m; p += 1024) {
        const unsigned long long key = s_key[p];
        const int src = (int)(uint32_t)key;  // position before sorting
        S.m_qidx[p] = s_q[src];
        S.m_tidx[p] = s_t[src];
        S.m_dist[p] = (int)(uint32_t)(key >> 32);  // bit pattern of the float distance
    }
    if (tid == 0) *S.m_count = m;
}

// One thread per query: top-2 over the train tiles (tiles ascend with the row index, keys inside
// a tile order by d2 then row), sqrt, output; queries whose second-best d2 reaches 2^22 are listed
// for k_l2_far.
__global__ __launch_bounds__(256) void k_l2_merge(const uint32_t* __restrict__ part, int n_tiles, int nq, int nq_pad,
                                                  const int* __restrict__ flag, int32_t* __restrict__ idx,
                                                  float* __restrict__ dist, int* __restrict__ far_count,
                                                  int* __restrict__ far_list) {
    if (!*flag) return;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    unsigned long long b0 = ~0ull, b1 = ~0ull;
    for (int tI = 0; tI < n_tiles; tI++) {
        const uint2 k = *reinterpret_cast<const uint2*>(part + ((size_t)tI * nq_pad + qi) * 2);
        const uint32_t ks[2] = {k.x, k.y};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (ks[u] == L2_NOKEY) continue;
            const unsigned long long key = ((unsigned long long)(ks[u] >> 7) << 32) | (uint32_t)(tI * L2_TILE + (ks[u] & 127));
            const unsigned long long lo = key < b0 ? key : b0, hi = key < b0 ? b0 : key;
            b0 = lo;
            b1 = hi < b1 ? hi : b1;
        }
    }
    const uint32_t d2a = (uint32_t)(b0 >> 32), d2b = (uint32_t)(b1 >> 32);
    if (d2b >= L2_FAR) {  // the order of the rounded roots may differ from the order of d2
        far_list[atomicAdd(far_count, 1)] = qi;
        return;
    }
    idx[2 * qi] = (int)(uint32_t)b0;
    idx[2 * qi + 1] = (int)(uint32_t)b1;
    dist[2 * qi] = sqrtf((float)d2a);
    dist[2 * qi + 1] = sqrtf((float)d2b);
}

// The listed queries, one wave each (grid-stride over the list): every train row's sqrt(d2) in
// float32 as the sequential loop computes it, top-2 under "smaller distance, then earlier row".
template <int KS>
__global__ __launch_bounds__(64) void k_l2_far(const float* __restrict__ q, const float* __restrict__ t, int nt,
                                               const int* __restrict__ flag, const int* __restrict__ far_count,
                                               const int* __restrict__ far_list, int32_t* __restrict__ idx,
                                               float* __restrict__ dist) {
    if (!*flag) return;
    const int lane = threadIdx.x, dim = 16 * KS;
    const int n_far = *far_count;
    for (int e = blockIdx.x; e < n_far; e += gridDim.x) {
        const int qi = far_list[e];
        Top2 best = {INFINITY, INFINITY, -1, -1};
        for (int j = lane; j < nt; j += 64) {
            float s = 0.f;
            for (int k = 0; k < dim; k++) {
                const float d = q[(size_t)qi * dim + k] - t[(size_t)j * dim + k];
                s += d * d;
            }
            top2_insert(best, sqrtf(s), j);
        }
        for (int m = 1; m < 64; m <<= 1) {
            Top2 o;
            o.d0 = __shfl_xor(best.d0, m);
            o.d1 = __shfl_xor(best.d1, m);
            o.i0 = __shfl_xor(best.i0, m);
            o.i1 = __shfl_xor(best.i1, m);
            if (o.i0 >= 0) top2_insert(best, o.d0, o.i0);
            if (o.i1 >= 0) top2_insert(best, o.d1, o.i1);
        }
        if (lane == 0) {
            idx[2 * qi] = best.i0;
            idx[2 * qi + 1] = best.i1;
            dist[2 * qi] = best.d0;
            dist[2 * qi + 1] = best.d1;
        }
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

size_t rtvm_l2_work_bytes(int nq, int nt, int dim) {
    const size_t nq_pad = (size_t)(nq + L2_TILE - 1) / L2_TILE * L2_TILE, nt_pad = (size_t)(nt + L2_TILE - 1) / L2_TILE * L2_TILE;
    size_t b = 256;                                   // flag, far_count
    b += (nq_pad + nt_pad) * sizeof(float) + 512;     // norms
    b += (nq_pad + nt_pad) * dim * 2 + 512;           // bf16 operands
    b += (nt_pad / L2_TILE) * nq_pad * 2 * sizeof(uint32_t) + 256;  // per-tile top-2 keys
    b += nq_pad * sizeof(int) + 256;                  // far list
    return b;
}

// dq: nq x dim, dt: nt x dim (device); d_idx / d_dist: nq x 2; work: rtvm_l2_work_bytes(nq, nt, dim) bytes.
int rtvm_launch_l2_knn2(rtvm_ctx* c, const float* dq, int nq, const float* dt, int nt, int dim, int32_t* d_idx,
                        float* d_dist, void* work) {
    if (nq <= 0) return RTVM_OK;
    // two 64-row tiles; the same memory then holds the 256 partial results of the final merge
    size_t lds_bytes = (size_t)2 * 64 * (dim + 1) * sizeof(float);
    if (lds_bytes < 256 * sizeof(Top2)) lds_bytes = 256 * sizeof(Top2);
    if (lds_bytes > 160 * 1024) RTVM_FAIL(c, RTVM_E_ARG, "match_l2_knn2: descriptor dimension %d is too large", dim);
    const bool mfma = dim == 128 || dim == 64;
    int* d_flag = (int*)work;
    if (mfma) {
        const int nq_pad = (nq + L2_TILE - 1) / L2_TILE * L2_TILE, nt_pad = (nt + L2_TILE - 1) / L2_TILE * L2_TILE;
        const int n_tiles = nt_pad / L2_TILE;
        char* w = (char*)work + 256;
        int* d_far_count = d_flag + 1;
        float* qn = (float*)w;
        float* tn = qn + nq_pad;
        w += ((size_t)(nq_pad + nt_pad) * sizeof(float) + 255) / 256 * 256;
        uint4* qb = (uint4*)w;
        uint4* tb = qb + (size_t)nq_pad * dim / 8;
        w += ((size_t)(nq_pad + nt_pad) * dim * 2 + 255) / 256 * 256;
        uint32_t* part = (uint32_t*)w;
        w += ((size_t)n_tiles * nq_pad * 2 * sizeof(uint32_t) + 255) / 256 * 256;
        int* far_list = (int*)w;
        // flag = 1 ("all values are 8-bit integers" until a value says otherwise), far_count = 0
        HIP_TRY(c, hipMemsetAsync(d_flag, 0, 2 * sizeof(int), c->stream));
        HIP_TRY(c, hipMemsetAsync(d_flag, 1, 1, c->stream));
        const int n_max = nq_pad > nt_pad ? nq_pad : nt_pad;
        const dim3 gp((unsigned)(((size_t)n_max * (dim / 8) + 255) / 256), 2), gt(nq_pad / L2_TILE, n_tiles);
        if (dim == 128) {
            RTVM_LAUNCH(c, k_l2_prep<8>, gp, dim3(256), 0, dq, nq, dt, nt, qb, tb, qn, tn, d_flag);
            RTVM_LAUNCH(c, k_l2_tile<8>, gt, dim3(256), 0, (const uint4*)qb, (const uint4*)tb, (const float*)qn,
                        (const float*)tn, nt, nq_pad, (const int*)d_flag, part);
        } else {
            RTVM_LAUNCH(c, k_l2_prep<4>, gp, dim3(256), 0, dq, nq, dt, nt, qb, tb, qn, tn, d_flag);
            RTVM_LAUNCH(c, k_l2_tile<4>, gt, dim3(256), 0, (const uint4*)qb, (const uint4*)tb, (const float*)qn,
                        (const float*)tn, nt, nq_pad, (const int*)d_flag, part);
        }
        RTVM_LAUNCH(c, k_l2_merge, dim3((nq + 255) / 256), dim3(256), 0, (const uint32_t*)part, n_tiles, nq, nq_pad,
                    (const int*)d_flag, d_idx, d_dist, d_far_count, far_list);
        const int far_grid = nq < 1024 ? nq : 1024;
        if (dim == 128)
            RTVM_LAUNCH(c, k_l2_far<8>, dim3(far_grid), dim3(64), 0, dq, dt, nt, (const int*)d_flag,
                        (const int*)d_far_count, (const int*)far_list, d_idx, d_dist);
        else
            RTVM_LAUNCH(c, k_l2_far<4>, dim3(far_grid), dim3(64), 0, dq, dt, nt, (const int*)d_flag,
                        (const int*)d_far_count, (const int*)far_list, d_idx, d_dist);
    }
    (void)hipFuncSetAttribute((const void*)k_l2_knn2_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    RTVM_LAUNCH(c, k_l2_knn2_direct, dim3((nq + 63) / 64), dim3(256), lds_bytes, dq, nq, dt, nt, dim, (const int*)d_flag,
                mfma ? 1 : 0, d_idx, d_dist);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// knn-2 + ratio + sort for slots sb .. sb+nb-1 (contexts with the SIFT detector): matches into
// slot.m_qidx / m_tidx / m_dist (distances as float bit patterns) / m_count.
int rtvm_launch_l2_slots(rtvm_ctx* c, int sb, int nb, int cap) {
    if (cap > RTVM_SIFT_STREAM_MAX_KP) RTVM_FAIL(c, RTVM_E_ARG, "L2 matcher: at most %d descriptors per frame", RTVM_SIFT_STREAM_MAX_KP);
    const int cap_pad = (cap + L2_TILE - 1) / L2_TILE * L2_TILE;
    const int tiles = cap_pad / L2_TILE;
    RTVM_LAUNCH(c, k_l2_prep_slots, dim3((cap_pad * 16 + 255) / 256, 2, nb), dim3(256), 0, (const FrameSlot*)c->d_slots, sb,
                cap_pad);
    RTVM_LAUNCH(c, k_l2_tile_slots, dim3(tiles, tiles, nb), dim3(256), 0, (const FrameSlot*)c->d_slots, sb, cap_pad);
    RTVM_LAUNCH(c, k_l2_ratio_sort, dim3(nb), dim3(1024), 0, (const FrameSlot*)c->d_slots, sb, cap_pad);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}
