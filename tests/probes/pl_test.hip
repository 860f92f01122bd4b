#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* p) {
    double x = p[threadIdx.x];
    int lo = __double2loint(x), hi = __double2hiint(x);
    auto r = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool low = threadIdx.x < 32;
    int plo = low ? r[1] : r[0], phi = low ? s[1] : s[0];
    p[threadIdx.x] = __hiloint2double(phi, plo);
}
int main() { double h[64], *d; for (int i = 0; i < 64; i++) h[i] = i + 0.5; hipMalloc(&d, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int ok = 1; for (int i = 0; i < 64; i++) ok &= (h[i] == (i ^ 32) + 0.5); printf("partner exchange %s: h[0]=%g h[33]=%g\n", ok ? "OK" : "WRONG", h[0], h[33]); return 0; }
