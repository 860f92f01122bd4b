// Do f64 MFMA and f64 VALU work of two wavefronts on the SAME SIMD overlap?  A workgroup of 8 wavefronts (two per SIMD): wavefronts
// 0-3 run a chain of v_mfma_f64_16x16x4_f64, wavefronts 4-7 a chain of independent v_fma_f64; each role is timed alone and together.
// hipcc --offload-arch=gfx950 -O3 -o tests/probes/mfma_valu_overlap tests/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(int mode, int iters, double* out, unsigned long long* cyc) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = w < 4;
    unsigned long long t0 = __builtin_readcyclecounter();
    double r = 0.0;
    if (mf && (mode & 1)) {
        v4d a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}; double x = lane * 1e-3, y = 1.0 + lane * 1e-4;
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    }
    if (!mf && (mode & 2)) {
        double a[8]; for (int q = 0; q < 8; q++) a[q] = lane + q;
        const double b = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = __builtin_fma(a[q], b, c);
        }
        for (int q = 0; q < 8; q++) r += a[q];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc; const int NB = 256, iters = 20000;
    hipMalloc(&out, NB * 512 * 8); hipMalloc(&cyc, NB * 8 * 8);
    unsigned long long h[NB * 8];
    for (int mode = 1; mode <= 3; mode++) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, mode, iters, out, cyc); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0; for (int b = 0; b < NB; b++) { for (int w = 0; w < 4; w++) m += h[b * 8 + w]; for (int w = 4; w < 8; w++) v += h[b * 8 + w]; }
        m /= NB * 4; v /= NB * 4;
        printf("mode %d (%s): MFMA wave %.1f cycles per MFMA (2 per iteration), VALU wave %.2f cycles per v_fma_f64 (32 per iteration)\n", mode,
               mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both, one of each per SIMD", (mode & 1) ? m / (2.0 * iters) : 0.0, (mode & 2) ? v / (32.0 * iters) : 0.0);
    }
    return 0;
}
