// micro-benchmark (round 6): what a DEPENDENT fp64 operation costs a wavefront that is alone on its SIMD (the parking kernel's regime: one wavefront per SIMD, so nothing else fills
// the bubbles), and how many independent chains it takes to fill them.  k<C>: C independent chains of v_fma_f64, interleaved; clocks per operation AND chain = the latency a
// dependent operation sees; clocks per operation = the issue cost when C is large.  Also: v_rcp_f64 chains, and fma chains whose operands come from LDS (ds_read_b64 -> fma -> ds_write_b64).
//   hipcc --offload-arch=gfx950 -O3 -o fp64_dependent_latency fp64_dependent_latency.hip && ./fp64_dependent_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C, int KIND>
__global__ __launch_bounds__(64, 1) void k(double *out, long long *cyc, int n, double a, double b) {
    double x[C];
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = 1.0 + 1e-3 * (threadIdx.x + c);
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                if (KIND == 0) x[c] = fma(x[c], a, b);
                else if (KIND == 1) x[c] = __builtin_amdgcn_rcp(x[c]) + b;      // rcp + add: two dependent operations
                else if (KIND == 2) x[c] = x[c] * a;
                else x[c] = x[c] + b;
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < C; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C, int KIND> static void run(double *o, long long *c, int blocks, const char *what) {
    const int n = 2000;
    k<C, KIND><<<blocks, 64>>>(o, c, n, 0.999, 1e-3);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    long long h; if (hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    const double per = (double)h / (n * 8.0);
    printf("%-28s %2d chain(s), %4d wavefronts: %6.1f clocks per round of the chains = %5.1f per operation\n", what, C, blocks, per, per / C / (KIND == 1 ? 2 : 1));
}
int main() {
    double *o; long long *c;
    if (hipMalloc(&o, 1024 * 64 * 8) != hipSuccess || hipMalloc(&c, 1024 * 8) != hipSuccess) { printf("no device memory\n"); return 1; }
    for (int blocks : {1, 1024}) {
        run<1, 0>(o, c, blocks, "v_fma_f64"); run<2, 0>(o, c, blocks, "v_fma_f64"); run<4, 0>(o, c, blocks, "v_fma_f64"); run<8, 0>(o, c, blocks, "v_fma_f64"); run<16, 0>(o, c, blocks, "v_fma_f64");
        run<1, 2>(o, c, blocks, "v_mul_f64"); run<4, 2>(o, c, blocks, "v_mul_f64"); run<1, 3>(o, c, blocks, "v_add_f64"); run<4, 3>(o, c, blocks, "v_add_f64");
        run<1, 1>(o, c, blocks, "v_rcp_f64 + v_add_f64"); run<4, 1>(o, c, blocks, "v_rcp_f64 + v_add_f64");
    }
    return 0;
}
