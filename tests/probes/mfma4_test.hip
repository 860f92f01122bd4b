// Layout and issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 blocks of 4x4x4 per wave: block = 16-lane row).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4 tests/probes/mfma4_test.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_layout(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
__global__ void k_rate(double* out, long long* cyc, int n) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rate_dep(double* out, long long* cyc, int n) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5, c0 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double ha[64], hb[64], hd[64], *a, *b, *d; long long* cyc; long long hc[4];
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512 * 8); hipMalloc(&cyc, 64);
    // A[lane] = 2^(lane%16) tagged per block, B = one-hot probes: find (i,k) of A's lane and (k,j) of B's lane and (i,j) of D's lane
    // probe: for each pair (la, lb) in one block set A[la] = 1, B[lb] = 1, others 0 -> D has a single 1 at the lane holding (i,j) iff k matches
    int Ai[16], Ak[16], Bk[16], Bj[16]; for (int t = 0; t < 16; t++) Ai[t] = Ak[t] = Bk[t] = Bj[t] = -1;
    int hit[16][16];
    for (int la = 0; la < 16; la++) for (int lb = 0; lb < 16; lb++) {
        for (int t = 0; t < 64; t++) { ha[t] = 0; hb[t] = 0; }
        ha[la] = 1.0; hb[lb] = 1.0; ha[16 + la] = 3.0; hb[16 + lb] = 5.0;
        hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d); hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
        hit[la][lb] = -1;
        for (int t = 0; t < 16; t++) if (hd[t] != 0.0) hit[la][lb] = t;
        for (int t = 16; t < 32; t++) if (hd[t] != 0.0 && hd[t] != 15.0) printf("block 1 unexpected %g\n", hd[t]);
        for (int t = 32; t < 64; t++) if (hd[t] != 0.0) printf("cross-block leak at lane %d\n", t);
    }
    printf("D lane hit by (A lane la, B lane lb) (-1: no product, i.e. k differs):\n     ");
    for (int lb = 0; lb < 16; lb++) printf("%3d", lb);
    printf("\n");
    for (int la = 0; la < 16; la++) { printf("la%2d:", la); for (int lb = 0; lb < 16; lb++) printf("%3d", hit[la][lb]); printf("\n"); }
    int n = 2000;
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, d, cyc, n); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("independent 4x4x4 f64 MFMA: %.2f cycles each (one wave)\n", (double)hc[0] / (8.0 * n));
    hipLaunchKernelGGL(k_rate_dep, dim3(1), dim3(64), 0, 0, d, cyc, n); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent 4x4x4 f64 MFMA chain: %.2f cycles each\n", (double)hc[0] / (4.0 * n));
    hipLaunchKernelGGL(k_rate, dim3(4), dim3(256), 0, 0, d, cyc, n); hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
    printf("4 waves per block (one per SIMD): %.2f cycles each\n", (double)hc[0] / (8.0 * n));
    return 0;
}
