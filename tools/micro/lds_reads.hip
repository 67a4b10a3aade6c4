// micro-benchmark 2: phase = nread LDS reads (lane-dependent addresses, issued together) -> tree sum -> LDS write -> LDS-only barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define BAR() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
template <int NR, int NBAR>
__global__ void k(double *out, long long *cyc, int iters) {
    __shared__ double sh[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 128) sh[i] = 1.0 + i * 1e-4;
    __syncthreads();
    const int a = l / 14, c = l % 14;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        double v[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) v[r] = (r & 1) ? sh[256 + (r / 2) * 14 + c] : sh[a * 6 + r / 2];
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int r = 0; r + 1 < NR; r += 4) { s0 = fma(v[r], v[r + 1], s0); if (r + 3 < NR) s1 = fma(v[r + 2], v[r + 3], s1); }
        sh[512 + l] = s0 + s1;
        if (NBAR) BAR(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        sh[l & 63] = sh[512 + ((l + 1) & 127)] * 0.5;      // dependent read of another lane's result
        if (NBAR) BAR(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    }
    long long t1 = clock64();
    out[l] = sh[l];
    if (l == 0) cyc[0] = t1 - t0;
}
template <int NR, int NBAR> void run(double *o, long long *c, int thr) {
    k<NR, NBAR><<<1, thr>>>(o, c, 10000); hipDeviceSynchronize(); long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("reads %2d  %s  threads %3d : %.1f ticks per double-phase\n", NR, NBAR ? "s_barrier" : "wave fence", thr, h / 10000.0);
}
int main() {
    double *o; long long *c; hipMalloc(&o, 1024 * 8); hipMalloc(&c, 8);
    run<2, 1>(o, c, 128); run<4, 1>(o, c, 128); run<12, 1>(o, c, 128); run<24, 1>(o, c, 128);
    run<2, 0>(o, c, 64); run<12, 0>(o, c, 64); run<24, 0>(o, c, 64);
    run<12, 1>(o, c, 64);
    return 0;
}
