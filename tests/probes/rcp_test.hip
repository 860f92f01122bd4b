// accuracy of v_rcp_f64 and of one / two Newton refinements on gfx950: hipcc --offload-arch=gfx950 -O3 tests/probes/rcp_test.hip -o /tmp/rcp_test && /tmp/rcp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double v = x[i], r = __builtin_amdgcn_rcp(v);
    r0[i] = r; r = fma(fma(-v, r, 1.0), r, r); r1[i] = r; r = fma(fma(-v, r, 1.0), r, r); r2[i] = r;
}
int main() {
    const int n = 1 << 22; std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = ldexp(1.0 + u, (int)(s % 200) - 100) * ((s >> 3) & 1 ? 1 : -1); }
    double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) { long double t = 1.0L / (long double)x[i]; e0 = fmax(e0, fabs((double)((a[i] - t) / t))); e1 = fmax(e1, fabs((double)((b[i] - t) / t))); e2 = fmax(e2, fabs((double)((c[i] - t) / t))); }
    printf("max rel err: rcp %.3e  +1 Newton %.3e  +2 Newton %.3e  (eps = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
    return 0;
}
