// Which streams share a hardware queue?  Kernel A on stream X spins until a flag is set or 200 us have passed;
// kernel B on stream Y sets the flag.  If B cannot start before A has ended, X and Y share a queue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(volatile int* flag, int* seen, long long limit) {
    const long long t0 = wall_clock64();
    int s = 0;
    while (wall_clock64() - t0 < limit) {
        if (*flag) {
            s = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    *seen = s;
}
__global__ void k_set(int* flag) { *flag = 1; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    int *flag, *seen;
    hipMalloc(&flag, 4);
    hipMalloc(&seen, 4);
    std::vector<hipStream_t> st(n + 1);
    st[0] = nullptr;   // the null stream
    for (int i = 1; i <= n; i++) hipStreamCreate(&st[i]);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    const long long limit = (long long)rate * 300 / 1000;               // 300 us
    printf("wall clock %d kHz\n", rate);
    for (int rep = 0; rep < 2; rep++) {
        for (int x = 0; x <= n; x++) {
            for (int y = 0; y <= n; y++) {
                if (x == y) {
                    printf(" .");
                    continue;
                }
                hipMemset(flag, 0, 4);
                hipDeviceSynchronize();
                k_spin<<<1, 1, 0, st[x]>>>(flag, seen, limit);
                k_set<<<1, 1, 0, st[y]>>>(flag);
                hipDeviceSynchronize();
                int h = 0;
                hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost);
                printf(" %c", h ? '-' : 'S');
            }
            printf("\n");
        }
        printf("\n");
    }
    return 0;
}
