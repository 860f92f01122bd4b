// Issue cost of the f64 / DPP instructions the quad-team kernels are built from (gfx950): cycles per instruction for one wavefront
// alone on its SIMD and for four wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O2 -o tests/probes/valu_rate tests/probes/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ void k(double* out, long long* cyc, int n) {
    double a[16], s = 1.0 + threadIdx.x * 1e-9, m = 1.0000001;
    int b[16], si = threadIdx.x, mi = 3;
    for (int j = 0; j < 16; j++) b[j] = j + threadIdx.x;
    for (int j = 0; j < 16; j++) a[j] = 1.0 + j + threadIdx.x * 1e-6;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#define OPSI "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15])
#define OPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
        if (KIND == 0) {
#define X(j) "v_fma_f64 %" #j ", %" #j ", %17, %16\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 1) {
#define X(j) "v_fmac_f64_dpp %" #j ", %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 2) {
#define X(j) "v_mov_b64_dpp %" #j ", %16 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 3) {
#define X(j) "v_mov_b32_dpp %" #j ", %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R16(X) : OPSI : "v"(si), "v"(mi));
#undef X
        } else if (KIND == 4) {
#define X(j) "v_add_f64 %" #j ", %" #j ", %16\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 5) {
#define X(j) "v_fmac_f64 %" #j ", %16, %17\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 6) {
#define X(j) "v_mul_f64 %" #j ", %" #j ", %17\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 7) {
#define X(j) "v_cndmask_b32 %" #j ", %16, %17, vcc\n\t"
            asm volatile(R16(X) : OPSI : "v"(si), "v"(mi) : "vcc");
#undef X
        } else if (KIND == 8) {      // dependent chain of fmac_dpp on one accumulator
#define X(j) "v_fmac_f64_dpp %0, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        } else if (KIND == 9) {      // dependent chain of fma
#define X(j) "v_fma_f64 %0, %0, %17, %16\n\t"
            asm volatile(R16(X) : OPS : "v"(s), "v"(m));
#undef X
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double r = 0; for (int j = 0; j < 16; j++) r += a[j] + b[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND> void run(const char* name, double* out, long long* cyc) {
    const int n = 1000; long long h[64];
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double one = (double)h[0] / (16.0 * n);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(1024), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);   // 16 waves on one CU = 4 per SIMD
    double mx = 0; for (int w = 0; w < 16; w++) mx = h[w] > mx ? h[w] : mx;
    printf("%-44s %6.2f cycles/instr alone, %6.2f per wave with 4 waves/SIMD (= %5.2f per SIMD)\n", name, one, mx / (16.0 * n), mx / (16.0 * n) / 4);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 64);
    run<0>("v_fma_f64 (16 independent)", out, cyc);
    run<5>("v_fmac_f64 (16 independent)", out, cyc);
    run<1>("v_fmac_f64_dpp row_newbcast (16 independent)", out, cyc);
    run<2>("v_mov_b64_dpp row_newbcast", out, cyc);
    run<3>("v_mov_b32_dpp row_ror:8", out, cyc);
    run<4>("v_add_f64", out, cyc);
    run<6>("v_mul_f64", out, cyc);
    run<7>("v_cndmask_b32", out, cyc);
    run<8>("v_fmac_f64_dpp dependent chain", out, cyc);
    run<9>("v_fma_f64 dependent chain", out, cyc);
    return 0;
}
