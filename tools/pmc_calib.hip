// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for narrow per-lane accesses: each kernel streams a
// known number of bytes once, with 2, 4, 8 or 16 bytes per lane (consecutive lanes, consecutive addresses).
// Build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/bin/pmc_calib
// Run under `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (separate runs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename T>
__global__ __launch_bounds__(256) void k_calib_read(const T* __restrict__ src, size_t n, uint32_t* __restrict__ sink, uint32_t never) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T v = src[i];
    uint32_t acc = 0;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    for (unsigned k = 0; k < sizeof(T); k++) acc += b[k];
    if (acc == never) sink[0] = acc;  // `never` is above any byte sum, which the compiler cannot know: keeps the load alive
}

template <typename T>
__global__ __launch_bounds__(256) void k_calib_write(T* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T v;
    unsigned char* b = reinterpret_cast<unsigned char*>(&v);
    for (unsigned k = 0; k < sizeof(T); k++) b[k] = (unsigned char)(i + k);
    dst[i] = v;
}


int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB per kernel: far beyond the caches
    void *src, *dst;
    uint32_t* sink;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    (void)hipMemset(src, 1, bytes);
    (void)hipMemset(dst, 0, bytes);
    (void)hipDeviceSynchronize();
#define RUN(T)                                                                                            \
    {                                                                                                     \
        const size_t n = bytes / sizeof(T);                                                               \
        hipLaunchKernelGGL(k_calib_read<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const T*)src, n, sink, 0xFFFFFFFFu); \
        hipLaunchKernelGGL(k_calib_write<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (T*)dst, n); \
        (void)hipDeviceSynchronize();                                                                         \
    }
    RUN(uint16_t)
    RUN(uint32_t)
    RUN(uint2)
    RUN(uint4)
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
