// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for this project's access widths (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  Streams NB bytes (>> Infinity Cache) with 8 B/lane and
// 16 B/lane loads, and writes NB bytes with 8 B/lane and 16 B/lane stores.  Build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ void rd8(const double* p, size_t n, double* out) { double a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i]; if (a == 123.456) out[0] = a; }
__global__ void rd16(const double2_t* p, size_t n, double* out) { double a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2_t v = p[i]; a += v.x + v.y; } if (a == 123.456) out[0] = a; }
__global__ void wr8(double* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i; }
__global__ void wr16(double2_t* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2_t v; v.x = (double)i; v.y = 1.0; p[i] = v; } }
int main() {
    const size_t NB = (size_t)2 << 30;           // 2 GiB
    double* d; double* o; hipMalloc(&d, NB); hipMalloc(&o, 8); hipMemset(d, 0, NB);
    const size_t n8 = NB / 8, n16 = NB / 16;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(rd8, dim3(4096), dim3(64), 0, 0, d, n8, o);
        hipLaunchKernelGGL(rd16, dim3(4096), dim3(64), 0, 0, (const double2_t*)d, n16, o);
        hipLaunchKernelGGL(wr8, dim3(4096), dim3(64), 0, 0, d, n8);
        hipLaunchKernelGGL(wr16, dim3(4096), dim3(64), 0, 0, (double2_t*)d, n16);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %zu\n", NB);
    return 0;
}
