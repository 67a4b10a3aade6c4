// micro-benchmark: cost of one "LDS write -> workgroup barrier (LDS-only fence) -> dependent LDS read -> short fp64 chain" phase for a
// 128-thread workgroup alone on its CU (the building block of the Riccati sweep).  hipcc --offload-arch=gfx950 -O3 -o lds_lat lds_barrier_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define BAR() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
__global__ void k(double *out, long long *cyc, int iters, int nfma, int dodiv) {
    __shared__ double sh[256];
    const int l = threadIdx.x;
    double v = 1.0 + l * 1e-3;
    sh[l] = v; sh[l + 128] = v;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        double a = sh[(l + 1) & 127] , b = sh[128 + ((l + 7) & 127)];
        for (int j = 0; j < nfma; j++) a = fma(a, 0.999, b);
        if (dodiv) a = 1.0 / a + 1.0 / (a + b);
        sh[l] = a;
        BAR();
    }
    long long t1 = clock64();
    out[blockIdx.x * 128 + l] = sh[l];
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *o; long long *c; hipMalloc(&o, 256 * 128 * 8); hipMalloc(&c, 256 * 8);
    for (int dodiv = 0; dodiv < 2; dodiv++) for (int nf : {0, 6, 12}) for (int blocks : {1, 256, 1024}) {
        k<<<blocks, 128>>>(o, c, 10000, nf, dodiv); hipDeviceSynchronize();
        long long h[1024 > 256 ? 256 : 256]; hipMemcpy(h, c, 256 * 8, hipMemcpyDeviceToHost);
        printf("div %d fma %2d blocks %4d : %.1f cycles per phase\n", dodiv, nf, blocks, h[0] / 10000.0);
    }
    return 0;
}
