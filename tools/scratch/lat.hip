// Dependent-issue latency of the instructions on the Jacobi rotation's chain, one wave alone on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
#define BENCH(name, init, body)                                             \
    __global__ void k_##name(double* out, long long* cyc, double seed) {    \
        double x = seed + threadIdx.x * 1e-9, y = seed * 0.37;              \
        init;                                                               \
        long long t0 = __builtin_readcyclecounter();                        \
        _Pragma("unroll 16") for (int i = 0; i < N; i++) { body; }          \
        long long t1 = __builtin_readcyclecounter();                        \
        out[threadIdx.x] = x;                                               \
        if (threadIdx.x == 0) *cyc = t1 - t0;                               \
    }
BENCH(fma, , x = __builtin_fma(x, y, 0.25))
BENCH(mul, , x = x * y)
BENCH(add, , x = x + y)
BENCH(rcp, , x = __builtin_amdgcn_rcp(x))
BENCH(rsq, , x = __builtin_amdgcn_rsq(x))
BENCH(div, , x = y / x)
BENCH(sqrt, , x = sqrt(x) + 1.0)
BENCH(fixup, , x = __builtin_amdgcn_div_fixup(x, y, 1.5))
BENCH(fmas, , x = __builtin_amdgcn_div_fmas(x, y, 0.25, false))
BENCH(scale, , x = __builtin_amdgcn_div_scale(x, y, true, nullptr) + 0.0)
BENCH(f32fma, float f = (float)x; float g = (float)y, f = __builtin_fmaf(f, g, 0.25f); x = f)
BENCH(readlane, int v = (int)x, v = __builtin_amdgcn_readlane(v, 3) + threadIdx.x; x = v)
BENCH(dpp, int v = (int)x, v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true) + 1; x = v)
__global__ void k_lds(double* out, long long* cyc, double seed) {
    __shared__ double s[64];
    double x = seed + threadIdx.x;
    s[threadIdx.x] = x;
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        s[idx] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = s[(idx + 1) & 63] + 1.0;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMallocManaged(&cyc, 8);
#define RUN(name) for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, out, cyc, 1.37); hipDeviceSynchronize(); } \
    printf("%-10s %6.1f cycles per dependent op\n", #name, (double)*cyc / N);
    RUN(fma) RUN(mul) RUN(add) RUN(rcp) RUN(rsq) RUN(div) RUN(sqrt) RUN(fixup) RUN(fmas) RUN(scale) RUN(f32fma) RUN(readlane) RUN(dpp) RUN(lds)
    return 0;
}
