// Issue cost per instruction for a lone wave: eight independent chains interleaved (so no instruction waits for the
// one before it), against the dependent-chain figures of lat.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
#define BENCH8(name, decl, op)                                                   \
    __global__ void k_##name(double* out, long long* cyc, double seed) {         \
        decl;                                                                    \
        long long t0 = __builtin_readcyclecounter();                             \
        _Pragma("unroll 8") for (int i = 0; i < N; i++) {                        \
            op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)                      \
        }                                                                        \
        long long t1 = __builtin_readcyclecounter();                             \
        double s = 0;                                                            \
        for (int j = 0; j < 8; j++) s += (double)x[j];                           \
        out[threadIdx.x] = s;                                                    \
        if (threadIdx.x == 0) *cyc = t1 - t0;                                    \
    }
#define DECL_D double x[8]; for (int j = 0; j < 8; j++) x[j] = seed + j + threadIdx.x * 1e-9; double y = seed * 0.37
#define DECL_F float x[8]; for (int j = 0; j < 8; j++) x[j] = (float)seed + j + threadIdx.x * 1e-6f; float y = (float)seed * 0.37f
#define DECL_I int x[8]; for (int j = 0; j < 8; j++) x[j] = (int)seed + j + threadIdx.x
#define OP_FMA64(j) x[j] = __builtin_fma(x[j], y, 0.25);
#define OP_FMA32(j) x[j] = __builtin_fmaf(x[j], y, 0.25f);
#define OP_ADDI(j) x[j] = x[j] * 3 + 1;
#define OP_DPP(j) x[j] = __builtin_amdgcn_update_dpp(0, x[j], 0xB1, 0xF, 0xF, true);
#define OP_RDL(j) x[j] = __builtin_amdgcn_readlane(x[j], 3) + x[(j + 1) & 7];
#define OP_CND(j) x[j] = x[j] > 5 ? x[(j + 1) & 7] : x[(j + 2) & 7];
#define OP_RCP(j) x[j] = __builtin_amdgcn_rcp(x[j]);
#define OP_MAX64(j) x[j] = fmax(x[j], y + j);
#define OP_SWAP(j) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)x[j], (unsigned)x[j], false, false); x[j] = (int)(r[0] + r[1]); }
BENCH8(fma64, DECL_D, OP_FMA64)
BENCH8(fma32, DECL_F, OP_FMA32)
BENCH8(mad_i32, DECL_I, OP_ADDI)
BENCH8(dpp_mov, DECL_I, OP_DPP)
BENCH8(readlane_add, DECL_I, OP_RDL)
BENCH8(cmp_cndmask, DECL_I, OP_CND)
BENCH8(rcp64, DECL_D, OP_RCP)
BENCH8(max64, DECL_D, OP_MAX64)
BENCH8(permlane32_swap, DECL_I, OP_SWAP)
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMallocManaged(&cyc, 8);
#define RUN(name, per) for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, out, cyc, 1.37); hipDeviceSynchronize(); } \
    printf("%-16s %6.2f cycles per op (%d instructions each)\n", #name, (double)*cyc / (N * 8), per);
    RUN(fma64, 1) RUN(fma32, 1) RUN(mad_i32, 1) RUN(dpp_mov, 1) RUN(readlane_add, 2) RUN(cmp_cndmask, 2) RUN(rcp64, 1) RUN(max64, 1) RUN(permlane32_swap, 3)
    return 0;
}
