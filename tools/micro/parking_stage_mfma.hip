// micro-benchmark (round 6): what ONE stage of the parking kernel's Riccati sweep would cost on the fp64 matrix cores, against the 1 050 clocks of the three-phase LDS sweep
// (obca_solver_riccati.h).  The parking stage is 6 states + 2 inputs with 6 right-hand sides: [x | u | rhs] = 14 columns, one 16 x 16 tile.  With the value function kept in
// accumulator ("D") layout -- register r of lane (g, j) = C[g + 4 r][j]: register kb IS the B operand of K-block kb and, the tile being symmetric, the A operand too -- a stage is
//     T    = [P | p] applied to [F | off]            K = 6 -> 2 v_mfma_f64_16x16x4_f64          (accumulator starts from [0 | p])
//     Qhat = [H | hc] + F' T                          K = 6 -> 2
//     border constants += off' (T + p)                K = 6 -> 2 (independent of the chain: they fill the pipe)
//     Quu (2 x 2) out of the tile through v_readlane, det, reciprocal (the same four-operation refinement as the kernel), gains per lane
//     [P' | p'] = Qhat_xx - Qhat_xu K                 K = 2 (padded to 4) -> 1, border += ... 1
//     P' made exactly symmetric through LDS (the kernel's LDS sweep stores 21 entries twice; the quadcopter's tile sweep needs this step or round-off flips its pivot test)
// = 8 matrix instructions on a dependent chain of five, one LDS round trip, the pivot chain.  The operands that do not depend on the recursion (F, off, H, hc: the stage record)
// are held in registers here -- the real sweep gathers them from HBM four stages ahead, which costs issue slots this benchmark does not pay: the number printed is a LOWER bound.
// Data are synthetic (a contraction, so that 80 stages stay finite); only the instruction pattern and its dependencies matter.
//   hipcc --offload-arch=gfx950 -O3 -o parking_stage_mfma parking_stage_mfma.hip && ./parking_stage_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double rcp_nr(double d) { double r = __builtin_amdgcn_rcp(d); const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }
template <int SYM>
__global__ __launch_bounds__(64, 1) void k(double *out, long long *cyc, int stages, int reps) {
    __shared__ double tr[16 * 17];
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    // stage operands in MFMA operand layout (one double per lane and K-block): F (A / B operand), off, H, hc
    double FA[2], FB[2], OA[2];
    for (int kb = 0; kb < 2; kb++) { const int r = 4 * kb + g; FA[kb] = (r == j ? 0.9 : 0.0) + 1e-3 * ((r * 7 + j) % 5); FB[kb] = FA[kb]; OA[kb] = j < 2 ? 1e-2 * (r + 1) : 0.0; }
    v4d P = {0, 0, 0, 0}, Bm = {0, 0, 0, 0};
    for (int r = 0; r < 4; r++) { const int i = g + 4 * r; P[r] = (i == j && i < 6) ? 1.0 : 0.0; }
    v4d H; for (int r = 0; r < 4; r++) { const int i = g + 4 * r; H[r] = (i == j && i < 8) ? (i >= 6 ? 2.0 : 0.1) : (j >= 8 && j < 14 && i < 8 ? 1e-3 : 0.0); }
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; rep++) {
        if (rep == 1) t0 = clock64();
        for (int s = 0; s < stages; s++) {
            // T = P F (+ [0 | p] already in the accumulator's rhs columns: here the tile itself)
            v4d T = P;
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], FB[0], T, 0, 0, 0);
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], FB[1], T, 0, 0, 0);
            // Qhat = H + F' T: register kb of T (D layout) is the B operand of K-block kb
            v4d Q = H;
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[0], T[0], Q, 0, 0, 0);
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[1], T[1], Q, 0, 0, 0);
            // border constants: off' (T + p), off the chain
            Bm = __builtin_amdgcn_mfma_f64_16x16x4f64(OA[0], T[0] + P[0], Bm, 0, 0, 0);
            Bm = __builtin_amdgcn_mfma_f64_16x16x4f64(OA[1], T[1] + P[1], Bm, 0, 0, 0);
            // Quu = Qhat[6..7][6..7]: rows 6, 7 = lane groups 2, 3 of register 1
            const double q00 = readlane_f64(Q[1], 16 * 2 + 6), q10 = readlane_f64(Q[1], 16 * 3 + 6), q11 = readlane_f64(Q[1], 16 * 3 + 7);
            const double det = fma(q00, q11, -(q10 * q10)), idet = rcp_nr(det);
            // the u rows of the lane's own column (rows 6, 7 sit in lane groups 2, 3): one exchange across lane groups each
            const double q6 = __shfl(Q[1], 32 + j, 64), q7 = __shfl(Q[1], 48 + j, 64);
            const double k0 = (q10 * q7 - q11 * q6) * idet, k1 = (q10 * q6 - q00 * q7) * idet;          // gains of column j
            // P' = Qhat_xx + Qhat_xu K: A operand = the u columns of Qhat (by symmetry the u rows: lane (k, i) <- Q[6 + k][i]), B operand = the gains (K = 2, padded)
            const double Aq = g == 0 ? q6 : (g == 1 ? q7 : 0.0), Bk = g == 0 ? k0 : (g == 1 ? k1 : 0.0);
            v4d Pn = Q;
            Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Aq, Bk, Pn, 0, 0, 0);
            Bm = __builtin_amdgcn_mfma_f64_16x16x4f64(Aq, Bk, Bm, 0, 0, 0);
            if (SYM) {      // exactly symmetric value function: through LDS, row stride 17 against bank conflicts
                for (int r = 0; r < 4; r++) tr[(g + 4 * r) * 17 + j] = Pn[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int r = 0; r < 4; r++) { const int i = g + 4 * r; const double t = tr[j * 17 + i]; Pn[r] = (i < 6 && j < 6) ? 0.5 * (Pn[r] + t) : Pn[r]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (int r = 0; r < 4; r++) { const int i = g + 4 * r; P[r] = (i < 6 && j < 14) ? 0.5 * Pn[r] + ((i == j) ? 0.5 : 0.0) : 0.0; }      // (kept bounded: a contraction)
        }
    }
    t1 = clock64();
    for (int r = 0; r < 4; r++) out[(size_t)blockIdx.x * 256 + 64 * r + lane] = P[r] + Bm[r];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *o; long long *c;
    if (hipMalloc(&o, 1024 * 256 * 8) != hipSuccess || hipMalloc(&c, 1024 * 8) != hipSuccess) { printf("no device memory\n"); return 1; }
    const int stages = 80, reps = 201;
    for (int sym = 1; sym >= 0; sym--) for (int blocks : {1, 256, 1024}) {
        if (sym) k<1><<<blocks, 64>>>(o, c, stages, reps); else k<0><<<blocks, 64>>>(o, c, stages, reps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        long long h[1024]; if (hipMemcpy(h, c, blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        double worst = 0; for (int i = 0; i < blocks; i++) worst = h[i] > worst ? (double)h[i] : worst;
        printf("parking stage on fp64 MFMA tiles, %s, %4d wavefronts (one per SIMD up to 1024): %.0f clocks per stage (slowest wavefront %.0f)\n",
               sym ? "P symmetrised through LDS" : "no symmetrisation        ", blocks, h[0] / (double)(stages * (reps - 1)), worst / (double)(stages * (reps - 1)));
    }
    printf("(the three-phase LDS sweep of the kernel: 1 050 clocks per stage alone, 1 050-1 150 under load, gathers of the stage record included)\n");
    return 0;
}
