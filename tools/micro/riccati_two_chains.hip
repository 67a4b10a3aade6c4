// micro-benchmark (round 6): what interleaving TWO (or three) independent Riccati chains in one wavefront would buy -- the execution pattern a two-partition recursion
//  would have.  One "phase" is what a phase of obca_solver_riccati.h is for a lane: two 4-vectors and two scalars from LDS (4 ds_read_b128 + 2 ds_read_b64),
// a 4-term product with two initial values (two chains of two fma, one add: three dependent operations), one ds_write_b64, a wavefront-scope fence; every read depends on what
// OTHER lanes wrote in the previous phase.  A "stage" is three phases.  NCH chains work on separate LDS regions; their instructions are interleaved in program order, so the
// latency of one chain's round trip and fp64 chain hides behind the other's issue.  Printed: clocks per stage AND CHAIN, one wavefront per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o riccati_two_chains riccati_two_chains.hip && ./riccati_two_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#define FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <int NCH, int EXTRA>
__global__ __launch_bounds__(64, 1) void k(double *out, long long *cyc, int stages, int reps, double *rec) {
    __shared__ __attribute__((aligned(16))) double sh[NCH][512];
    const int lane = threadIdx.x;
    for (int c = 0; c < NCH; c++) for (int i = lane; i < 512; i += 64) sh[c][i] = 1.0 + 1e-3 * ((i * 7 + c) % 11);
    __syncthreads();
    // operand offsets of the lane's item: they rotate with the phase so that a lane reads what other lanes wrote (the item tables of the real sweep)
    const int oa = 4 * ((lane * 5 + 1) & 31), ob = 4 * ((lane * 3 + 2) & 31), oi = 128 + ((lane + 9) & 63), oj = 128 + ((lane + 23) & 63), od = 128 + lane;
    long long t0 = 0;
    double *myrec = rec + (size_t)blockIdx.x * 128 * 80; double pre = 0;
    for (int rep = 0; rep < reps; rep++) {
        if (rep == 1) t0 = clock64();
        for (int s = 0; s < stages; s++) {
#pragma unroll
            for (int ph = 0; ph < 3; ph++) {
                double2 a0[NCH], a1[NCH], b0[NCH], b1[NCH]; double vi[NCH], vj[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {      // all reads of all chains first: the chains' round trips overlap
                    const double2 *pa = (const double2 *)&sh[c][(oa + 8 * ph) & 124], *pb = (const double2 *)&sh[c][(ob + 12 * ph) & 124];
                    a0[c] = pa[0]; a1[c] = pa[1]; b0[c] = pb[0]; b1[c] = pb[1]; vi[c] = sh[c][oi]; vj[c] = sh[c][oj];
                }
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    double v = fma(a0[c].y, b0[c].y, fma(a0[c].x, b0[c].x, vi[c])) + fma(a1[c].y, b1[c].y, fma(a1[c].x, b1[c].x, vj[c]));
                    if ((EXTRA & 1) && ph == 2) {      // the 2 x 2 pivot: three values out of other lanes' registers, determinant, refined reciprocal, the item's last fma
                        const double q00 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 28), __builtin_amdgcn_readlane(__double2loint(v), 28)) + 3.0;
                        const double q10 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 40), __builtin_amdgcn_readlane(__double2loint(v), 40));
                        const double q11 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 41), __builtin_amdgcn_readlane(__double2loint(v), 41)) + 3.0;
                        const double det = fma(q00, q11, -(q10 * q10)); double r = __builtin_amdgcn_rcp(det); const double e = fma(-det, r, 1.0); r = fma(r, fma(e, e, e), r);
                        v = fma(fma(a0[c].x, q10, a1[c].x * q00), r, v);
                    }
                    sh[c][od] = 0.2 * v;      // (a contraction: the values stay bounded)
                    if ((EXTRA & 2) && ph == 2) { double *ro = myrec + (size_t)s * 128; ro[lane] = v; ro[(lane + 5) & 63] = v; ro[64 + (lane & 31)] = 0.5 * v; ro[96 + (lane & 31)] = 0.25 * v; }
                    if ((EXTRA & 4) && ph == 2) { sh[c][256 + lane] = pre; pre = myrec[(size_t)((s + 4) % 80) * 128 + ((lane * 3) & 127)]; }
                }
                FENCE();
            }
        }
    }
    const long long t1 = clock64();
    double acc = 0; for (int c = 0; c < NCH; c++) acc += sh[c][od];
    out[(size_t)blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH, int EXTRA> static int run(double *o, long long *c, int blocks, double *rec) {
    const int stages = 80, reps = 201;
    k<NCH, EXTRA><<<blocks, 64>>>(o, c, stages, reps, rec);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    long long h[1024]; if (hipMemcpy(h, c, blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("%d chain(s), extras %d (1: pivot chain, 2: four record stores, 4: gather + unpack store), %4d wavefronts: %.0f clocks per stage for all chains = %.0f per stage and chain\n", NCH, EXTRA, blocks, h[0] / (double)(stages * (reps - 1)),
           h[0] / (double)(stages * (reps - 1)) / NCH);
    return 0;
}
int main() {
    double *o; long long *c;
    if (hipMalloc(&o, 1024 * 64 * 8) != hipSuccess || hipMalloc(&c, 1024 * 8) != hipSuccess) { printf("no device memory\n"); return 1; }
    double *rec; if (hipMalloc(&rec, (size_t)1024 * 128 * 80 * 8) != hipSuccess || hipMemset(rec, 0, (size_t)1024 * 128 * 80 * 8) != hipSuccess) return 1;
    for (int blocks : {1, 1024}) {
        if (run<1, 0>(o, c, blocks, rec) || run<2, 0>(o, c, blocks, rec) || run<3, 0>(o, c, blocks, rec)) return 1;
        if (run<1, 1>(o, c, blocks, rec) || run<1, 3>(o, c, blocks, rec) || run<1, 7>(o, c, blocks, rec) || run<2, 7>(o, c, blocks, rec)) return 1;
    }
    printf("(the kernel's stage: three such phases + the 2 x 2 pivot chain, the gathers of the stage record and the stores of the Riccati record: 1 050 clocks)\n");
    return 0;
}
