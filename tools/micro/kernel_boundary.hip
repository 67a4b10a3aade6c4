// kernel_boundary.hip -- does kernel B see what kernel A (same stream, launched before it) wrote, when B's workgroups read through OTHER XCDs than A's wrote through -- alone, and
// when several processes share the GPU?
//
// Round 5: the solver's results change when another process runs on the GPU at the same time; a wavefront's registers / LDS / scratch and its own HBM writes survive the sharing
// (cwsr_state, own_writes).  What is left is the hand-over BETWEEN kernels of one stream: a device-to-device copy resets the iterates, DualMultWS writes multipliers into them, the
// interior-point kernel reads them -- each kernel's workgroups on whatever XCD the dispatcher picks, so every hand-over crosses XCD L2s and rests on the write-back / invalidate at
// the kernel boundary.  The probe: generation g = 1, 2, ...: hipMemcpyAsync(dev -> dev) of a zero buffer over the data (the solver's reset), kernel A stamps generation g into the
// buffer (block b writes region b), kernel B reads it with a rotated block -> region mapping (block b reads region (b + 3) mod n: another XCD as a rule) and counts words that do not
// carry generation g (stale: an older generation or the zeros of the reset).
//   hipcc --offload-arch=gfx950 -O2 -o kernel_boundary kernel_boundary.hip && ./kernel_boundary [generations]      (run several copies at once)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REGION 8192      // doubles per block (64 KB)

__global__ __launch_bounds__(256) void stamp(double *buf, int g) {
    double *r = buf + (size_t)blockIdx.x * REGION;
    for (int i = threadIdx.x; i < REGION; i += 256) r[i] = (double)g * 65536.0 + i % 65536;
}
__global__ __launch_bounds__(64, 1) void check(const double *buf, int g, int nb, unsigned long long *out /* [0] stale words, [1] of them zeros (the reset), [2] older generations */) {
    extern __shared__ double pad[];      // 40 KB of LDS: four one-wavefront workgroups per CU, as the solver's kernel
    const int reg = (blockIdx.x + 3) % nb;
    const double *r = buf + (size_t)reg * REGION;
    unsigned long long stale = 0, zeros = 0, older = 0;
    for (int i = threadIdx.x; i < REGION; i += 64) {
        const double v = r[i], want = (double)g * 65536.0 + i % 65536;
        if (v != want) { stale++; if (v == 0.0) zeros++; else older++; }
    }
    if (stale) { atomicAdd(&out[0], stale); atomicAdd(&out[1], zeros); atomicAdd(&out[2], older); }
    if (threadIdx.x == 9999) pad[0] = 1;
}

int main(int argc, char **argv) {
    const int gens = argc > 1 ? atoi(argv[1]) : 2000, NB = 1024;
    double *buf, *zero; unsigned long long *dout, h[3];
    CHK(hipMalloc(&buf, (size_t)NB * REGION * 8)); CHK(hipMalloc(&zero, (size_t)NB * REGION * 8)); CHK(hipMalloc(&dout, 3 * 8));
    CHK(hipMemset(zero, 0, (size_t)NB * REGION * 8)); CHK(hipMemset(buf, 0, (size_t)NB * REGION * 8)); CHK(hipMemset(dout, 0, 3 * 8));
    CHK(hipFuncSetAttribute((const void *)check, hipFuncAttributeMaxDynamicSharedMemorySize, 40960));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int g = 1; g <= gens; g++) {
        CHK(hipMemcpyAsync(buf, zero, (size_t)NB * REGION * 8, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(stamp, dim3(NB), dim3(256), 0, s, buf, g);
        hipLaunchKernelGGL(check, dim3(NB), dim3(64), 40960, s, (const double *)buf, g, NB, dout);
    }
    CHK(hipStreamSynchronize(s));
    CHK(hipMemcpy(h, dout, 3 * 8, hipMemcpyDeviceToHost));
    printf("kernel_boundary: %d generations of (device-to-device reset, stamp kernel, check kernel through other XCDs) on one stream, %d x 64 KB: stale words %llu (zeros of the reset %llu, older generations %llu)\n",
           gens, NB, h[0], h[1], h[2]);
    return h[0] ? 1 : 0;
}
