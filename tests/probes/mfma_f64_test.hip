#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {   // A 16x4 row-major, B 4x16 row-major, D 16x16
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; i++) { hA[i] = 1.0 + 0.37 * i + (i % 5) * 0.011; hB[i] = 2.0 - 0.21 * i + (i % 7) * 0.013; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s = fma(hA[i * 4 + k], hB[k * 16 + j], s); ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(hD[i] - ref[i]));
    printf("mfma f64 16x16x4 max err %g (ref[17]=%g got %g)\n", e, ref[17], hD[17]);
    return e < 1e-9 ? 0 : 1;
}
