// lds_atomic_order.hip -- does ONE ds_add_f64 whose lanes hit the same LDS address add them in a fixed order?
//
// Round 4 summed the condensed obstacle contributions of a stage with ds_add_f64 (up to 16 lanes of one instruction on one address) and assumed lane order.  fp64 addition is
// not associative, so the assumption decides whether the solve is bit-reproducible.  This probe makes the order visible: lane i of a group of G lanes adds v[i] to one cell;
// the values are chosen so that every order of summation rounds differently (large cancelling terms + small ones).  Each workgroup (one wavefront, as the solver's) repeats
// the add R times and compares the cell with (a) the lane-order sum and (b) the cell of its first repetition; the kernel is launched alone and under load (several streams,
// a second kernel hammering LDS with conflicting traffic on the same CUs).
//
//   hipcc --offload-arch=gfx950 -O2 -o lds_atomic_order lds_atomic_order.hip && ./lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void probe(int G, int R, const double *vals, unsigned long long *out /* per block: [0] mismatches vs lane order, [1] mismatches vs first repetition, [2] bits of first */) {
    __shared__ double cell[64];
    __shared__ double pad[1024];
    const int lane = threadIdx.x, grp = lane / G, pos = lane % G;
    const double v = vals[(blockIdx.x * 64 + lane) % 4096];
    // lane-order reference of the group's sum (computed by the group's first lane from the same values)
    double ref = 0.0;
    for (int i = 0; i < G; i++) ref += vals[(blockIdx.x * 64 + grp * G + i) % 4096];
    unsigned long long bad_order = 0, bad_rep = 0, first = 0;
    for (int r = 0; r < R; r++) {
        if (pos == 0) cell[grp] = 0.0;
        pad[(lane * 17 + r) & 1023] = v;                       // unrelated LDS traffic of the same wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __hip_atomic_fetch_add(&cell[grp], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double got = cell[grp];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(got);
        if (r == 0) first = bits;
        if (pos == 0 && grp * G + G <= 64) {
            if (bits != (unsigned long long)__double_as_longlong(ref)) bad_order++;
            if (bits != first) bad_rep++;
        }
    }
    if (pos == 0 && grp * G + G <= 64) { atomicAdd(&out[3 * blockIdx.x], bad_order); atomicAdd(&out[3 * blockIdx.x + 1], bad_rep); }
    if (lane == 0) out[3 * blockIdx.x + 2] = first;
}

// load generator: wavefronts that keep the LDS pipe of their CU busy with conflicting atomics and strided traffic
__global__ __launch_bounds__(256) void hammer(int R, double *sink) {
    __shared__ double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = i;
    __syncthreads();
    double acc = 0;
    for (int r = 0; r < R; r++) {
        __hip_atomic_fetch_add(&buf[(threadIdx.x * 32 + r) & 4095], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc += buf[(threadIdx.x * 64 + r * 7) & 4095];
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    std::vector<double> h(4096);
    srand(7);
    for (int i = 0; i < 4096; i++) {
        const int e = rand() % 40 - 20;                        // magnitudes over 12 decades, both signs: every order of a 3..16-term sum rounds differently as a rule
        h[i] = ((rand() / (double)RAND_MAX) - 0.5) * pow(2.0, e);
    }
    double *dv, *sink; unsigned long long *dout;
    const int NB = 4096;
    CHK(hipMalloc(&dv, 4096 * 8)); CHK(hipMalloc(&sink, 8)); CHK(hipMalloc(&dout, NB * 3 * 8));
    CHK(hipMemcpy(dv, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipStream_t s[4]; for (auto &x : s) CHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    std::vector<unsigned long long> o(NB * 3), first_alone;
    for (int G : {2, 3, 5, 8, 10, 16}) {
        for (int load = 0; load < 2; load++) {
            CHK(hipMemset(dout, 0, NB * 3 * 8));
            if (load) for (int q = 1; q < 4; q++) hipLaunchKernelGGL(hammer, dim3(2048), dim3(256), 0, s[q], 20000, sink);
            hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s[0], G, 2000, dv, dout);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(o.data(), dout, NB * 3 * 8, hipMemcpyDeviceToHost));
            unsigned long long bo = 0, br = 0, bx = 0;
            for (int b = 0; b < NB; b++) { bo += o[3 * b]; br += o[3 * b + 1]; }
            if (!load) { first_alone.resize(NB); for (int b = 0; b < NB; b++) first_alone[b] = o[3 * b + 2]; }
            else for (int b = 0; b < NB; b++) bx += o[3 * b + 2] != first_alone[b];
            printf("G=%2d %-10s groups x repetitions %lld: differ from the lane-order sum %llu, differ from the first repetition %llu%s\n", G, load ? "under load" : "alone",
                   (long long)NB * (64 / G) * 2000, bo, br, load ? (bx ? " ; first results differ from the unloaded run" : " ; first results = unloaded run") : "");
        }
    }
    return 0;
}
