// Brute-force descriptor matchers for gfx950.
//
// Hamming (ORB): all-pairs distance between 256-bit descriptors, nearest
// neighbour in both directions, mutual-nearest-neighbour filter, optional
// stable sort by distance.  Replaces cv2.BFMatcher(NORM_HAMMING,
// crossCheck=True).match and the sorted() that follows it (reference
// main.py:37,695,698).
#include "common.h"

#include <mutex>

namespace {

// One wave per (64 rows of A) x (64 rows of B).  Lane l keeps row l of the A
// tile in registers and row l of the B chunk; the B rows are then broadcast one
// at a time through v_readlane (scalar operands), so the inner loop is 8 xor +
// 8 popcount-accumulate per pair and touches no memory.  The per-row minimum
// is merged across B chunks with one atomicMin on a packed (dist << 16 | index)
// key, whose integer order is exactly "smaller distance first, then lower
// index", i.e. OpenCV's first-minimum rule.
// blockIdx.z = 2 * frame + direction; direction 0: A = query, B = train; 1: A = train, B = query.
__global__ __launch_bounds__(64) void k_hamming_argmin(const FrameSlot* __restrict__ slots, int sb) {
    const FrameSlot& S = slots[sb + (blockIdx.z >> 1)];
    const uint8_t* __restrict__ q = S.feat.desc;
    const uint8_t* __restrict__ t = S.prev.desc;
    const int* __restrict__ nq_p = S.feat.count;
    const int* __restrict__ nt_p = S.prev.count;
    uint32_t* __restrict__ qkey = S.qkey;
    uint32_t* __restrict__ tkey = S.tkey;
    const int dir = blockIdx.z & 1;
    const uint8_t* A = dir ? t : q;
    const uint8_t* B = dir ? q : t;
    const int nA = dir ? *nt_p : *nq_p;
    const int nB = dir ? *nq_p : *nt_p;
    uint32_t* key = dir ? tkey : qkey;
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    if (a0 >= nA || b0 >= nB) return;
    const int lane = threadIdx.x;
    const int ia = min(a0 + lane, nA - 1), ib = min(b0 + lane, nB - 1);
    const uint4* pa = reinterpret_cast<const uint4*>(A + (size_t)ia * 32);
    const uint4* pb = reinterpret_cast<const uint4*>(B + (size_t)ib * 32);
    const uint4 a_lo = pa[0], a_hi = pa[1];
    const uint4 b_lo = pb[0], b_hi = pb[1];
    const int nb = min(64, nB - b0);
    // the tile's 64 rows of B go through LDS: every lane reads row k as two 16-byte broadcast reads (eight
    // v_readlane per row took three times as long as the instructions they feed)
    __shared__ __align__(16) uint4 s_b[64][2];
    s_b[lane][0] = b_lo;
    s_b[lane][1] = b_hi;
    wave_lds_phase();
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll 8
    for (int k = 0; k < 64; k++) {
        const uint4 r_lo = s_b[k][0], r_hi = s_b[k][1];
        uint32_t d;
        d = __popc(a_lo.x ^ r_lo.x);
        d += __popc(a_lo.y ^ r_lo.y);
        d += __popc(a_lo.z ^ r_lo.z);
        d += __popc(a_lo.w ^ r_lo.w);
        d += __popc(a_hi.x ^ r_hi.x);
        d += __popc(a_hi.y ^ r_hi.y);
        d += __popc(a_hi.z ^ r_hi.z);
        d += __popc(a_hi.w ^ r_hi.w);
        // rows past the end of B were loaded clamped; keep them out of the minimum
        best = min(best, k < nb ? ((d << 16) | (uint32_t)(b0 + k)) : 0xFFFFFFFFu);
    }
    if (a0 + lane < nA) atomicMin(&key[a0 + lane], best);
}

// Nearest-neighbour keys start at "no neighbour" for both directions.
__global__ __launch_bounds__(256) void k_match_init(const FrameSlot* __restrict__ slots, int sb, int cap_q, int cap_t) {
    const FrameSlot& S = slots[sb + blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < cap_q) S.qkey[i] = 0xFFFFFFFFu;
    if (i < cap_t) S.tkey[i] = 0xFFFFFFFFu;
}

// Single workgroup: mutual-NN filter, order-preserving compaction (ascending
// query index) and, when sorted != 0, a bitonic sort of (dist << 16 | position)
// keys, which is the stable sort by distance of main.py:698.
// legacy != 0: the cross-check of BFMatcher before OpenCV 4.2: query i is matched with the
// nearest (lowest index on ties) of the train rows whose nearest query is i; a query's own
// nearest train row plays no part.  The per-query keys are rebuilt from the per-train keys
// (the distance of train row j to its nearest query i is d(i, j)).
__global__ __launch_bounds__(1024) void k_xcheck_compact(const FrameSlot* __restrict__ slots, int sb, int sorted, int legacy) {
    const FrameSlot& S = slots[sb + blockIdx.x];
    if (legacy) {
        const int nq = *S.feat.count, nt = *S.prev.count;
        for (int i = threadIdx.x; i < nq; i += 1024) S.qkey[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (int j = threadIdx.x; j < nt; j += 1024) {
            const uint32_t kt = S.tkey[j];
            if (kt != 0xFFFFFFFFu) atomicMin(&S.qkey[kt & 0xFFFFu], (kt & 0xFFFF0000u) | (uint32_t)j);
        }
        __syncthreads();
    }
    const uint32_t* __restrict__ qkey = S.qkey;
    const uint32_t* __restrict__ tkey = S.tkey;
    const int* __restrict__ nq_p = S.feat.count;
    const int* __restrict__ nt_p = S.prev.count;
    int32_t* __restrict__ m_qidx = S.m_qidx;
    int32_t* __restrict__ m_tidx = S.m_tidx;
    int32_t* __restrict__ m_dist = S.m_dist;
    int* __restrict__ m_count = S.m_count;
    __shared__ uint32_t s_key[RTVM_MAX_MATCH];  // (dist << 16 | position)
    __shared__ uint16_t s_q[RTVM_MAX_MATCH];
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = *nq_p, nt = *nt_p;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        uint32_t kq = 0;
        if (i < nq && nt > 0) {
            kq = qkey[i];
            const uint32_t j = kq & 0xFFFFu;
            ok = legacy ? kq != 0xFFFFFFFFu : (tkey[j] & 0xFFFFu) == (uint32_t)i;
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
            s_key[pos] = (kq & 0xFFFF0000u) | (uint32_t)pos;
            s_q[pos] = (uint16_t)i;
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
    if (sorted && m > 1) {
        int n2 = 1;
        while (n2 < m) n2 <<= 1;
        for (int i = m + tid; i < n2; i += 1024) s_key[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const uint32_t a = s_key[i], b = s_key[ixj];
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
        const uint32_t key = s_key[p];
        const int src = (int)(key & 0xFFFFu);  // position before sorting
        const int qi = s_q[src];
        m_qidx[p] = qi;
        m_tidx[p] = (int)(qkey[qi] & 0xFFFFu);
        m_dist[p] = (int)(key >> 16);
    }
    if (tid == 0) *m_count = m;
}

}  // namespace

// Matches slot.feat (query) against slot.prev (train) for slots sb .. sb+nb-1; cap_q /
// cap_t bound the descriptor counts (grid sizing).  Results land in slot.m_*.
int rtvm_launch_hamming(rtvm_ctx* c, int sb, int nb, int cap_q, int cap_t, int sorted) {
    if (cap_q > RTVM_MAX_MATCH || cap_t > RTVM_MAX_MATCH)
        RTVM_FAIL(c, RTVM_E_ARG, "hamming matcher: at most %d descriptors per side", RTVM_MAX_MATCH);
    const int nfill = cap_q > cap_t ? cap_q : cap_t;
    if (nfill > 0) {
        RTVM_LAUNCH(c, k_match_init, dim3((nfill + 255) / 256, nb), dim3(256), 0, c->d_slots, sb, cap_q, cap_t);
        const int tiles = (nfill + 63) / 64;
        RTVM_LAUNCH(c, k_hamming_argmin, dim3(tiles, tiles, 2 * nb), dim3(64), 0, c->d_slots, sb);
    }
    RTVM_LAUNCH(c, k_xcheck_compact, dim3(nb), dim3(1024), 0, c->d_slots, sb, sorted,
                (c->flags & RTVM_FLAG_MATCH_LEGACY) ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    return RTVM_OK;
}

// ---- the matcher for callers that cannot allocate ------------------------------------------
//
// Same kernels as rtvm_match_hamming_xcheck[_sorted]; what that entry point keeps in buffers of
// its context (the slot descriptor the kernels read, the two descriptor counts, the nearest-
// neighbour keys of both directions) lives in one file-scope device array instead, so the call
// needs no context and allocates nothing.  Every GPU holds its own copy of the array at its own
// address: the address is looked up once per device (hipGetSymbolAddress, with that device
// current) and kept; host code uses only that pointer.

namespace {

struct NoAllocScratch {
    FrameSlot slot;                  // what the kernels read as slots[0]
    int counts[4];                   // nq, nt
    uint32_t qkey[RTVM_NOALLOC_MAX_DESCRIPTORS];
    uint32_t tkey[RTVM_NOALLOC_MAX_DESCRIPTORS];
};

static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

// k_xcheck_compact sizes its LDS arrays by RTVM_MAX_MATCH and the packed keys hold 16-bit indices: the public
// limit of the entry point may not outgrow either.
static_assert(RTVM_NOALLOC_MAX_DESCRIPTORS <= RTVM_MAX_MATCH, "the compaction kernel's LDS arrays hold RTVM_MAX_MATCH entries");
static_assert(RTVM_NOALLOC_MAX_DESCRIPTORS <= 65535, "the nearest-neighbour keys carry 16-bit indices");

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on device d, null until looked up
std::mutex g_scratch_mutex;
thread_local char g_noalloc_err[256];

// Publishes the slot descriptor and the descriptor counts of this call.
__global__ void k_noalloc_bind(NoAllocScratch* __restrict__ sc, FrameSlot slot, int nq, int nt) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    sc->slot = slot;
    sc->counts[0] = nq;
    sc->counts[1] = nt;
}

int noalloc_fail(int code, const char* msg) {
    snprintf(g_noalloc_err, sizeof(g_noalloc_err), "%s", msg);
    return code;
}

}  // namespace

extern "C" const char* rtvm_noalloc_last_error(void) { return g_noalloc_err; }

extern "C" int rtvm_match_hamming_xcheck_noalloc(const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt,
                                                 int32_t* qidx_dev, int32_t* tidx_dev, int32_t* dist_dev,
                                                 int32_t* m_out_dev, int sorted, void* hip_stream) {
    g_noalloc_err[0] = 0;
    if (nq < 0 || nt < 0 || !m_out_dev || (nq > 0 && (!q_dev || !qidx_dev || !tidx_dev || !dist_dev)) || (nt > 0 && !t_dev))
        return noalloc_fail(RTVM_E_ARG, "match_hamming_xcheck_noalloc: bad argument");
    if (((uintptr_t)q_dev | (uintptr_t)t_dev) & 15)   // the kernels read descriptors as 16-byte words
        return noalloc_fail(RTVM_E_ARG, "match_hamming_xcheck_noalloc: q_dev and t_dev must be 16-byte aligned");
    if (nq > RTVM_NOALLOC_MAX_DESCRIPTORS || nt > RTVM_NOALLOC_MAX_DESCRIPTORS) {
        snprintf(g_noalloc_err, sizeof(g_noalloc_err),
                 "match_hamming_xcheck_noalloc: %d x %d descriptors, the fixed scratch holds at most %d per side", nq, nt,
                 RTVM_NOALLOC_MAX_DESCRIPTORS);
        return RTVM_E_ARG;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return noalloc_fail(RTVM_E_HIP, "match_hamming_xcheck_noalloc: no current HIP device");
    NoAllocScratch* sc;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                snprintf(g_noalloc_err, sizeof(g_noalloc_err), "hipGetSymbolAddress(scratch) on device %d failed: %s", dev,
                         hipGetErrorString(e));
                return RTVM_E_HIP;
            }
            g_scratch_ptr[dev] = p;
        }
        sc = static_cast<NoAllocScratch*>(g_scratch_ptr[dev]);
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    FrameSlot slot;
    memset(&slot, 0, sizeof(slot));
    slot.feat.desc = const_cast<uint8_t*>(q_dev);
    slot.prev.desc = const_cast<uint8_t*>(t_dev);
    slot.feat.count = &sc->counts[0];
    slot.prev.count = &sc->counts[1];
    slot.qkey = sc->qkey;
    slot.tkey = sc->tkey;
    slot.m_qidx = qidx_dev;
    slot.m_tidx = tidx_dev;
    slot.m_dist = dist_dev;
    slot.m_count = m_out_dev;
    const FrameSlot* slots = &sc->slot;
    hipLaunchKernelGGL(k_noalloc_bind, dim3(1), dim3(64), 0, stream, sc, slot, nq, nt);
    const int nfill = nq > nt ? nq : nt;
    if (nfill > 0) {
        // the keys keep the previous call's values: both directions start at "no neighbour" again
        hipLaunchKernelGGL(k_match_init, dim3((nfill + 255) / 256, 1), dim3(256), 0, stream, slots, 0, nq, nt);
        const int tiles = (nfill + 63) / 64;
        hipLaunchKernelGGL(k_hamming_argmin, dim3(tiles, tiles, 2), dim3(64), 0, stream, slots, 0);
    }
    hipLaunchKernelGGL(k_xcheck_compact, dim3(1), dim3(1024), 0, stream, slots, 0, sorted ? 1 : 0, 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_noalloc_err, sizeof(g_noalloc_err), "match_hamming_xcheck_noalloc: launch failed: %s", hipGetErrorString(e));
        return RTVM_E_HIP;
    }
    return RTVM_OK;
}
