// Is the SIMD bound by VALU instructions only, or by all instructions?  Four wavefronts per SIMD, each looping over 8 v_fma_f64 plus
// 0 / 8 / 16 SALU instructions (or s_nop, or ds_swizzle): cycles per loop iteration per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define V8 "v_fma_f64 %0, %0, %11, %10\n\tv_fma_f64 %1, %1, %11, %10\n\tv_fma_f64 %2, %2, %11, %10\n\tv_fma_f64 %3, %3, %11, %10\n\tv_fma_f64 %4, %4, %11, %10\n\tv_fma_f64 %5, %5, %11, %10\n\tv_fma_f64 %6, %6, %11, %10\n\tv_fma_f64 %7, %7, %11, %10\n\t"
#define S8 "s_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\ts_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\ts_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\ts_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\t"
#define N8 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#define I8 "v_fma_f64 %0, %0, %11, %10\n\ts_add_u32 %8, %8, 1\n\tv_fma_f64 %1, %1, %11, %10\n\ts_add_u32 %9, %9, 1\n\tv_fma_f64 %2, %2, %11, %10\n\ts_add_u32 %8, %8, 1\n\tv_fma_f64 %3, %3, %11, %10\n\ts_add_u32 %9, %9, 1\n\tv_fma_f64 %4, %4, %11, %10\n\ts_add_u32 %8, %8, 1\n\tv_fma_f64 %5, %5, %11, %10\n\ts_add_u32 %9, %9, 1\n\tv_fma_f64 %6, %6, %11, %10\n\ts_add_u32 %8, %8, 1\n\tv_fma_f64 %7, %7, %11, %10\n\ts_add_u32 %9, %9, 1\n\t"
template <int KIND>
__global__ void k(double* out, long long* cyc, int n) {
    double a[8], s = 1.0 + threadIdx.x * 1e-9, m = 1.0000001; unsigned x = 1, y = 2;
    for (int j = 0; j < 8; j++) a[j] = 1.0 + j + threadIdx.x * 1e-6;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#define OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(x), "+s"(y)
        if (KIND == 0) asm volatile(V8 : OPS : "v"(s), "v"(m) : "scc");
        if (KIND == 1) asm volatile(V8 S8 : OPS : "v"(s), "v"(m) : "scc");
        if (KIND == 2) asm volatile(V8 S8 S8 : OPS : "v"(s), "v"(m) : "scc");
        if (KIND == 3) asm volatile(V8 N8 : OPS : "v"(s), "v"(m) : "scc");
        if (KIND == 4) asm volatile(I8 : OPS : "v"(s), "v"(m) : "scc");
        if (KIND == 5) asm volatile(S8 S8 : OPS : "v"(s), "v"(m) : "scc");
    }
    long long t1 = __builtin_readcyclecounter();
    double r = 0; for (int j = 0; j < 8; j++) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + x + y;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}
template <int KIND> void run(const char* name, double* out, long long* cyc) {
    const int n = 2000; long long h[64];
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double one = (double)h[0] / n;
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(1024), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);
    double mx = 0; for (int w = 0; w < 16; w++) mx = h[w] > mx ? h[w] : mx;
    printf("%-40s %7.1f cycles/iteration alone, %7.1f with 4 waves/SIMD\n", name, one, mx / n);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 64);
    run<0>("8 VALU", out, cyc); run<1>("8 VALU + 8 SALU (blocked)", out, cyc); run<2>("8 VALU + 16 SALU", out, cyc);
    run<3>("8 VALU + 8 s_nop", out, cyc); run<4>("8 VALU + 8 SALU interleaved", out, cyc); run<5>("16 SALU", out, cyc);
    return 0;
}
