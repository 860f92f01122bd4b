// Rate of __builtin_readcyclecounter() (s_memtime) against HIP events and wall_clock64(): hipcc --offload-arch=gfx950 -O2 clock_calib.hip -o clock_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    double x = threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) x = __builtin_fma(x, 1.0000001, 1e-12);
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = c1 - c0; out[blockIdx.x * 3 + 1] = w1 - w0; out[blockIdx.x * 3 + 2] = (unsigned long long)(x * 1e9); }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 3 * 8);
    unsigned long long h[3];
    for (int blocks : {1, 256, 4096}) for (int rep = 0; rep < 2; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); spin<<<blocks, 64>>>(d, 4000000); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %5d: %.3f ms (events) | cycle counter %llu -> %.3f GHz | wall_clock64 %llu -> %.1f MHz\n", blocks, ms, h[0], h[0] / (ms * 1e6), h[1], h[1] / (ms * 1e3));
    }
    return 0;
}
