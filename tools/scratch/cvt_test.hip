// What v_cvt_pk_u8_f32 does with halves, negatives, values above 255 and NaN, against rint + clamp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main() {
    const int n = 1 << 20;
    float* h = new float[n];
    for (int i = 0; i < n; i++) h[i] = -2.f + (float)i * (260.f / n);   // fine steps over [-2, 258)
    for (int i = 0; i < 600; i++) h[i] = -1.f + 0.5f * i;                // all halves
    h[700] = NAN; h[701] = INFINITY; h[702] = -INFINITY; h[703] = 254.5f; h[704] = 255.5f; h[705] = 0.5f; h[706] = 1.5f; h[707] = 2.5f;
    float* d; unsigned* o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, o, n);
    unsigned* r = new unsigned[n];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        float v = h[i];
        int e = std::isnan(v) ? 0 : (int)std::fmin(std::fmax(std::nearbyint(v), 0.f), 255.f);
        if ((int)r[i] != e && bad++ < 12) printf("x=%g got %u want %d\n", v, r[i], e);
    }
    printf("mismatches vs rint+clamp: %d; 0.5->%u 1.5->%u 2.5->%u 254.5->%u 255.5->%u nan->%u\n", bad, r[705], r[706], r[707], r[703], r[704], r[700]);
    return 0;
}
