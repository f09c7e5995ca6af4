// Cycles per Jacobi rotation of jacobi_wave on a lone wave (the RANSAC kernels' bound).  Includes the product source.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I real-time-video-mosaic_amd/csrc \
//   '-DJAC_SRC="../../real-time-video-mosaic_amd/csrc/ransac.hip"' '-DJAC_CALL(N)=jacobi_wave(L.A,L.W,L.V,L.indR,L.indC,L.perm,N)' \
//   -o tools/scratch/jac_old tools/scratch/jac_bench.hip
#include JAC_SRC
#include <cstdio>
#include <cstdlib>
namespace {
__global__ __launch_bounds__(64) void k_jac(const double* Ain, int n, double* Wout, long long* cyc, int reps) {
    __shared__ JacobiLds L;
    const int lane = threadIdx.x;
    long long total = 0;
    for (int r = 0; r < reps; r++) {
        __syncthreads();
        for (int e = lane; e < n * n; e += 64) L.A[e] = Ain[e];
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        if (n == 8) JAC_CALL(8); else JAC_CALL(9);
        total += __builtin_readcyclecounter() - t0;
    }
    if (lane < n) Wout[lane] = L.W[lane];
    if (lane == 0) { cyc[0] = total; cyc[1] = 0; }
}
}
int main() {
    for (int n = 8; n <= 9; n++) {
        double h[81];
        srand(7 + n);
        double B[81];
        for (int i = 0; i < n * n; i++) B[i] = (rand() % 2001 - 1000) / 100.0;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += B[i * n + k] * B[j * n + k]; h[i * n + j] = s; }
        double *dA, *dW; long long* cyc;
        hipMalloc(&dA, 81 * 8); hipMalloc(&dW, 9 * 8); hipMallocManaged(&cyc, 16);
        hipMemcpy(dA, h, 81 * 8, hipMemcpyHostToDevice);
        for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k_jac, dim3(1), dim3(64), 0, 0, dA, n, dW, cyc, 20); hipDeviceSynchronize(); }
        double W[9]; hipMemcpy(W, dW, 72, hipMemcpyDeviceToHost);
        printf("n=%d: %.0f cycles per decomposition, %lld rotations, %.0f cycles each; W0 %.17g W%d %.17g\n", n, (double)*cyc / 20, cyc[1], (double)*cyc / 20 / (double)(cyc[1] ? cyc[1] : 1), W[0], n - 1, W[n - 1]);
    }
    return 0;
}
int rtvm_prof_begin(rtvm_ctx*, const char*) { return -1; }
void rtvm_prof_end(rtvm_ctx*, int) {}
