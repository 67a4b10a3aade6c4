// What can one-wavefront workgroups stream, and does the 256 MiB Infinity Cache hold a cyclically re-used working set?
//   G workgroups of 64 threads (forced to `per_cu` per CU by their dynamic LDS), each sweeps ITS OWN region of S doubles `passes` times
//   (read-modify-write, or read only), U loads of W doubles per lane in flight.  Prints GB/s against the total footprint G * S * 8.
// The IPM kernels have exactly this shape: per-instance state streamed once per factorisation pass by one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int U, int W, int RW>
__global__ __launch_bounds__(64) void k_sweep(double *base, size_t S, int passes, double *out) {
    extern __shared__ double lds[];
    double *x = base + (size_t)blockIdx.x * S;
    const int lane = threadIdx.x;
    double acc = 0;
    for (int p = 0; p < passes; p++) {
        for (size_t i = (size_t)lane * W; i < S; i += (size_t)64 * W * U) {
            double v[U][W];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (W == 2) { double2 t = *(const double2 *)(x + i + (size_t)u * 64 * W); v[u][0] = t.x; v[u][W - 1] = t.y; }
                else v[u][0] = x[i + (size_t)u * 64 * W];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int w = 0; w < W; w++) { v[u][w] = v[u][w] * 1.0000001 + 1e-9; acc += v[u][w]; }
                if (RW) {
                    if (W == 2) { double2 t; t.x = v[u][0]; t.y = v[u][W - 1]; *(double2 *)(x + i + (size_t)u * 64 * W) = t; }
                    else x[i + (size_t)u * 64 * W] = v[u][0];
                }
            }
        }
    }
    if (acc == 1.2345) out[0] = acc + lds[lane];
}
template <int U, int W, int RW>
static void run(double *buf, double *out, int G, size_t S, int per_cu, int passes) {
    const size_t lds = (size_t)(160 * 1024 / per_cu) - 256;
    (void)hipFuncSetAttribute((const void *)k_sweep<U, W, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_sweep<U, W, RW>), dim3(G), dim3(64), lds, 0, buf, S, 2, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_sweep<U, W, RW>), dim3(G), dim3(64), lds, 0, buf, S, passes, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)G * S * 8 * passes * (RW ? 2 : 1);
    printf("G %5d per_cu %d S %6zu KB footprint %6.1f MB  U %2d W %d %s  %8.3f ms  %7.1f GB/s  (%.0f clk/pass @2.4GHz)\n", G, per_cu, S * 8 / 1024, G * S * 8 / 1048576.0, U, W,
           RW ? "rw" : "ro", ms, bytes / ms * 1e-6, ms * 2.4e6 / passes);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}
int main() {
    const size_t total = (size_t)2048 * 49152;            // doubles: 768 MiB
    double *buf, *out; if (hipMalloc(&buf, total * 8) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, total * 8); (void)hipDeviceSynchronize();
    const int P = 24;
    // footprint sweep at the kernels' shape: 1024 workgroups, 4 per CU, 8 x 8-byte loads in flight per lane
    for (size_t S : {8192, 16384, 24576, 28672, 32768, 36864, 40960, 49152}) run<8, 1, 1>(buf, out, 1024, S, 4, P);
    for (size_t S : {16384, 24576, 32768, 40960}) run<8, 1, 0>(buf, out, 1024, S, 4, P);
    // memory-level parallelism at a footprint beyond the cache (320 MB) and inside it (128 MB)
    for (size_t S : {16384, 40960}) {
        run<2, 1, 1>(buf, out, 1024, S, 4, P); run<4, 1, 1>(buf, out, 1024, S, 4, P); run<16, 1, 1>(buf, out, 1024, S, 4, P);
        run<4, 2, 1>(buf, out, 1024, S, 4, P); run<8, 2, 1>(buf, out, 1024, S, 4, P); run<16, 2, 1>(buf, out, 1024, S, 4, P);
    }
    // eight workgroups per CU (two wavefronts per SIMD): same total footprint as 1024 x 2S
    for (size_t S : {8192, 12288, 16384, 20480}) run<8, 1, 1>(buf, out, 2048, S, 8, P);
    for (size_t S : {8192, 20480}) run<8, 2, 1>(buf, out, 2048, S, 8, P);
    // sixteen per CU
    for (size_t S : {4096, 10240}) run<8, 1, 1>(buf, out, 4096, S, 16, P);
    return 0;
}
