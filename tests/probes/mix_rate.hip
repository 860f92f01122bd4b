// Issue model of one SIMD: cycles per loop iteration for bodies of 8 VALU | 8 VALU + 4 SALU | 8 VALU + 4 SALU + 2 LDS reads (+ wait) |
// + 1 branch-like s_cbranch, at 1 and 4 wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tests/probes/mix_rate tests/probes/mix_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, double* out, unsigned long long* cyc) {
    __shared__ double lds[2048];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1.0;
    __syncthreads();
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7, l0 = 0, l1 = 0;
    const double b = 1.0000001, c = 1e-9;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const unsigned addr = threadIdx.x * 8;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (MODE >= 1) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        if (MODE >= 2) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(addr) : "memory");
        if (MODE >= 3) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" :: "s"(s0) : "scc");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + l0 + l1 + s0 + s1 + s2 + s3;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(int wps, double* out, unsigned long long* cyc) {
    const int NB = 256, iters = 20000; unsigned long long h[NB * 16];
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(NB), dim3(256 * wps), 0, 0, iters, out, cyc); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int b = 0; b < NB; b++) for (int w = 0; w < 4 * wps; w++) m += h[b * 16 + w];
    printf("  %d wavefront(s) per SIMD: %.1f cycles per iteration and wavefront -> %.1f SIMD cycles per iteration of one wavefront's work\n", wps, m / (NB * 4 * wps) / iters, m / (NB * 4 * wps) / iters / wps);
}
int main() {
    double* out; unsigned long long* cyc; (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&cyc, 256 * 16 * 8);
    printf("8 VALU:\n"); run<0>(1, out, cyc); run<0>(4, out, cyc);
    printf("8 VALU + 4 SALU:\n"); run<1>(1, out, cyc); run<1>(4, out, cyc);
    printf("8 VALU + 4 SALU + 2 ds_read_b64 + wait:\n"); run<2>(1, out, cyc); run<2>(4, out, cyc);
    printf("8 VALU + 4 SALU + 2 ds_read_b64 + wait + compare/branch/nop:\n"); run<3>(1, out, cyc); run<3>(4, out, cyc);
    return 0;
}
