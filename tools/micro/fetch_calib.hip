// FETCH_SIZE / WRITE_SIZE calibration on gfx950 (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`, then again with WRITE_SIZE):
//   k_wide   : coalesced 16-byte loads of a 1 GiB buffer read once                    -> requested = 1 GiB
//   k_dense8 : coalesced 8-byte loads of the same buffer                              -> requested = 1 GiB
//   k_gather8: one 8-byte load per 128-byte line (the access pattern of lanes that each own a stage record)   -> requested = 64 MiB, lines touched = 1 GiB
//   k_reread : a 64 MiB buffer read 8 times by one kernel: after the first pass it sits in the 256 MiB Infinity Cache (not in the 8 x 4 MiB L2)
//              -> 512 MiB requested from L2's point of view, 64 MiB from HBM's.  Tells whether FETCH_SIZE counts Infinity-Cache hits.
//   k_write8 : 8-byte stores, one per 128-byte line, 64 MiB written / 1 GiB of lines touched
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_wide(const double2 *x, size_t n2, double *out) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { double2 v = x[i]; s += v.x + v.y; } if (s == 1.2345) out[0] = s; }
__global__ void k_dense8(const double *x, size_t n, double *out) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i]; if (s == 1.2345) out[0] = s; }
__global__ void k_gather8(const double *x, size_t n, double *out) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i * 16 < n; i += (size_t)gridDim.x * blockDim.x) s += x[i * 16]; if (s == 1.2345) out[0] = s; }
__global__ void k_reread(const double2 *x, size_t n2, double *out) { double s = 0; for (int rep = 0; rep < 8; rep++) for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { double2 v = x[i]; s += v.x + v.y * rep; } if (s == 1.2345) out[0] = s; }
__global__ void k_write8(double *x, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i * 16 < n; i += (size_t)gridDim.x * blockDim.x) x[i * 16] = (double)i; }
int main() {
    const size_t n = (size_t)1 << 27;            // doubles: 1 GiB
    double *x, *out; if (hipMalloc(&x, n * 8) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(x, 0, n * 8); (void)hipDeviceSynchronize();
    const int G = 256 * 16, T = 256;
    hipLaunchKernelGGL(k_wide, dim3(G), dim3(T), 0, 0, (const double2 *)x, n / 2, out); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_dense8, dim3(G), dim3(T), 0, 0, (const double *)x, n, out); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_gather8, dim3(G), dim3(T), 0, 0, (const double *)x, n, out); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_reread, dim3(G), dim3(T), 0, 0, (const double2 *)x, (n / 16) / 2, out); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_write8, dim3(G), dim3(T), 0, 0, x, n); (void)hipDeviceSynchronize();
    printf("done\n"); return 0;
}
