// obca_hip.hip -- HIP kernels and the C ABI of libobca_hip.so (gfx950 only; see include/obca_hip.h).
//
// Kernels
//   obca_parking_ipm_kernel : one 128-thread workgroup (two wavefronts) per problem instance, persistent over the whole
//                             interior-point solve (obca_solver.h).  grid = B, block = 128.
//   obca_quad_ipm_kernel    : the same for the quadcopter NLP (obca_quad_solver.h).
//   obca_dualws_kernel      : one lane per (instance, stage, obstacle) convex sub-problem of DualMultWS (obca_model.h).
// Memory (per instance, fp64, all in HBM; sizes for N=80, 3 obstacles / 5 rows in brackets):
//   prob  header+rx,ry,ryaw   [411]      z  primal-dual iterate [6554]      d  search direction [3804 used]
//   as    assembled stage records (N+1) x 88 [7128]     rs  Riccati records (N+1) x 116 [9396]
//   oc    condensed obstacle records (N+1) x nOb x 12 [2916]    traj (N+2) x 6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "obca_solver.h"
#include "obca_quad_solver.h"
#include "../../include/obca_hip.h"

using namespace obca;

static_assert(sizeof(obca_opts) == sizeof(Opts), "obca_opts must mirror obca::Opts");
static_assert(OBCA_QUAD_NMAX == QNMAX, "ABI limits must match the kernels");
static_assert(OBCA_VMAX == OB_VMAX && OBCA_NOBMAX == OB_NOBMAX && OBCA_NMAX == OB_NMAX, "ABI limits must match the kernels");

struct DevBufs {
    double *prob, *z0, *z, *d, *as, *rs, *oc, *traj, *info, *dws, *prof;
    double *slice;                                   // slice records (SL_SIZE doubles per instance) of the two-launch schedule
    int *order;                                      // B instance indices in dispatch order (-1: nothing left to do), then the class counters
    size_t s_prob, s_z, s_as, s_rs, s_oc, s_traj;   // strides in doubles
};

// One wavefront per SIMD (two instances per CU): each wave may then use 256 VGPRs + 256 AGPRs, and the register-hungry per-lane phases
// keep their spills (and most callee-saved registers) in AGPRs instead of scratch.  Against two waves per SIMD with 256 registers each this
// is faster at every batch size measured: a lone instance is ~10 % quicker per pass, and a full machine no longer streams the scratch
// save areas through HBM.
#ifndef OBCA_IPM_WAVES_PER_EU
#define OBCA_IPM_WAVES_PER_EU 1
#endif
__global__ __launch_bounds__(OB_NT, OBCA_IPM_WAVES_PER_EU) void obca_parking_ipm_kernel(int B, int N, DevBufs b, Opts o, int mode, int budget) {
    // mode 0: fresh solve of instance blockIdx.x (at most `budget` factorisation passes if budget > 0); mode 1: continue the parked solves in
    // the order the ordering kernel chose (workgroups are dispatched in blockIdx order, so the expected stragglers start first)
    const int inst = mode ? b.order[blockIdx.x] : (int)blockIdx.x;
    if (inst < 0 || inst >= B) return;
    if (threadIdx.x == 0) {
        Inst &I = g_sh.inst;
        I.prob = (const gdbl *)(b.prob + (size_t)inst * b.s_prob);
        I.z = (gdbl *)(b.z + (size_t)inst * b.s_z); I.d = (gdbl *)(b.d + (size_t)inst * b.s_z);
        I.as = (gdbl *)(b.as + (size_t)inst * b.s_as); I.rs = (gdbl *)(b.rs + (size_t)inst * b.s_rs);
        I.oc = (gdbl *)(b.oc + (size_t)inst * b.s_oc); I.traj = (gdbl *)(b.traj + (size_t)inst * b.s_traj);
#ifdef OBCA_PROFILE
        I.tlast = clock64();
#endif
    }
#ifdef OBCA_PROFILE
    if (threadIdx.x < 16) g_sh.prof[threadIdx.x] = mode ? b.prof[(size_t)inst * 16 + threadIdx.x] : 0.0;   // counters add up over the slices
#endif
    __syncthreads();
    solve_instance(N, o, b.info + (size_t)inst * 8, (gdbl *)(b.slice + (size_t)inst * SL_SIZE), mode, budget);
#ifdef OBCA_PROFILE
    __syncthreads();
    if (threadIdx.x < 16) b.prof[(size_t)inst * 16 + threadIdx.x] = g_sh.prof[threadIdx.x];
#endif
}

// difficulty class of a parked instance (0..63, higher = dispatched earlier)
__device__ inline int obca_slice_class(const double *st) {
    const int nreg = (int)st[SL_NREG] + (int)st[SL_NREGPREV];
    const double pinf = st[SL_PINF];
    int c = 8 * (nreg < 7 ? nreg : 7);
    // within the same retry count: the constraint violation that is left, one class per decade from 1e-6 up
    int e = (pinf > 1e-30 && pinf < 1e30) ? (int)floor(log10(pinf)) + 7 : (pinf >= 1e30 ? 7 : 0);
    e = e < 0 ? 0 : (e > 7 ? 7 : e);
    return c + e;
}

// Dispatch order of the second launch: parked instances sorted by a difficulty class (descending), finished ones dropped.  The class is what
// the first slice revealed about the instance: every inertia retry so far counts, then the constraint violation still left -- the instances
// that go on to need two or three times the median number of passes are almost all among those that already needed retries (DESIGN.md
// section 3 has the measured ranking quality).  One workgroup, counting sort on 64 classes; the order inside a class is whatever the LDS
// atomics give, which changes timing only, never results.
__global__ __launch_bounds__(1024) void obca_order_kernel(int B, const double *info, const double *slice, int *order) {
    __shared__ int cnt[64], base[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        if ((int)info[(size_t)i * 8] != ST_SUSPENDED) continue;
        const double *st = slice + (size_t)i * SL_SIZE;
        atomicAdd(&cnt[obca_slice_class(st)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int c = 63; c >= 0; c--) { base[c] = acc; acc += cnt[c]; cnt[c] = 0; } base[0] = base[0]; order[B] = acc; }
    __syncthreads();
    const int total = order[B];
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        if ((int)info[(size_t)i * 8] != ST_SUSPENDED) continue;
        const double *st = slice + (size_t)i * SL_SIZE;
        const int c = obca_slice_class(st);
        order[base[c] + atomicAdd(&cnt[c], 1)] = i;
    }
    for (int i = total + threadIdx.x; i < B; i += blockDim.x) order[i] = -1;
}

// one lane per (instance, stage, obstacle); writes lam/mu into the iterate buffer `z` (instance layout) and d into dws
__global__ __launch_bounds__(256) void obca_dualws_kernel(int B, int N, int nObMax, DevBufs b, double *zdst, size_t s_zdst) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)(N + 1) * nObMax;
    if (gid >= (long long)B * per) return;
    const int inst = (int)(gid / per); const int rem = (int)(gid % per);
    const int k = rem / nObMax, j = rem % nObMax;
    const double *p = b.prob + (size_t)inst * b.s_prob;
    const int nOb = (int)p[PH_NOB], M = (int)p[PH_M];
    if (j >= nOb) return;
    const int v = (int)p[PH_VOB + j], r0 = (int)p[PH_ROFF + j];
    double a1[OB_VMAX], a2[OB_VMAX], bj[OB_VMAX], g[4];
#pragma unroll
    for (int i = 0; i < OB_VMAX; i++) { bool on = i < v; a1[i] = on ? p[PH_A + 2 * (r0 + i)] : 0.0; a2[i] = on ? p[PH_A + 2 * (r0 + i) + 1] : 0.0; bj[i] = on ? p[PH_B + r0 + i] : 0.0; }
#pragma unroll
    for (int i = 0; i < 4; i++) g[i] = p[PH_G + i];
    const double rx = p[OB_HDR + k], ry = p[OB_HDR + (N + 1) + k], ryaw = p[OB_HDR + 2 * (N + 1) + k], off = p[PH_OFF];
    double sn, cs; sincos(ryaw, &sn, &cs);
    double lam[OB_VMAX], mu[4], dv;
    dualws_one(v, a1, a2, bj, g, rx + cs * off, ry + sn * off, cs, sn, lam, mu, &dv);
    Lay l; make_layout(N, nOb, M, l);
    double *z = zdst + (size_t)inst * s_zdst;
#pragma unroll
    for (int i = 0; i < OB_VMAX; i++) if (i < v) z[l.lam + k * M + r0 + i] = lam[i];
#pragma unroll
    for (int i = 0; i < 4; i++) z[l.mu + 4 * (k * nOb + j) + i] = mu[i];
    if (b.dws) b.dws[(size_t)inst * per + rem] = dv;
}

// receding-horizon restart (not in the reference): the warm start of the next solve is the previous solution advanced by
// `shift` stages -- x, lambda, mu, the tracking reference (rx, ry, ryaw) and u move up, the tail repeats the terminal stage (standing at the
// goal: acceleration 0), t = 1 and sl = 0 as in ParkingSignedDist.jl:213-222 -- entirely on the device.  One workgroup per instance.
__global__ __launch_bounds__(128) void obca_shift_kernel(int B, int N, int shift, DevBufs b, const double *x0_new /* 4 x B or NULL */) {
    const int inst = blockIdx.x; if (inst >= B) return;
    double *p = b.prob + (size_t)inst * b.s_prob; const double *z = b.z + (size_t)inst * b.s_z; double *w = b.z0 + (size_t)inst * b.s_z;
    double *tmp = b.d + (size_t)inst * b.s_z;                      // the direction buffer is free between solves
    const int nOb = (int)p[PH_NOB], M = (int)p[PH_M], N1 = N + 1;
    Lay l; make_layout(N, nOb, M, l);
    for (int i = threadIdx.x; i < 3 * N1; i += blockDim.x) { const int a = i / N1, k = i % N1, ks = min(k + shift, N); tmp[i] = p[OB_HDR + a * N1 + ks]; }
    for (int i = threadIdx.x; i < l.nprimal; i += blockDim.x) w[i] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * N1; i += blockDim.x) p[OB_HDR + i] = tmp[i];
    for (int i = threadIdx.x; i < 4 * N1; i += blockDim.x) { const int k = i / 4, c = i % 4, ks = min(k + shift, N); w[l.x + i] = z[l.x + 4 * ks + c]; }
    for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) { const int k = i / 2, c = i % 2, ks = min(k + shift, N - 1); w[l.u + i] = (c == 1 && k + shift > N - 1) ? 0.0 : z[l.u + 2 * ks + c]; }
    for (int i = threadIdx.x; i < M * N1; i += blockDim.x) { const int k = i / M, c = i % M, ks = min(k + shift, N); w[l.lam + i] = z[l.lam + ks * M + c]; }
    for (int i = threadIdx.x; i < 4 * nOb * N1; i += blockDim.x) { const int k = i / (4 * nOb), c = i % (4 * nOb), ks = min(k + shift, N); w[l.mu + i] = z[l.mu + ks * 4 * nOb + c]; }
    __syncthreads();
    if (threadIdx.x < 4) { const double v = x0_new ? x0_new[4 * inst + threadIdx.x] : z[l.x + 4 * min(shift, N) + threadIdx.x]; p[PH_X0 + threadIdx.x] = v; w[l.x + threadIdx.x] = v; }
    if (threadIdx.x == 4) w[l.t] = 1.0;
}

// quadcopter path: one 128-thread workgroup per instance, persistent over the interior-point solve (obca_quad_solver.h)
struct QDevBufs {
    double *prob, *z, *d, *as, *rs, *oc, *info, *prof;
    size_t s_prob, s_z, s_d, s_as, s_rs, s_oc;
};
#ifndef OBCA_QUAD_WAVES_PER_EU
#define OBCA_QUAD_WAVES_PER_EU 1      // as for the parking kernel
#endif
__global__ __launch_bounds__(OB_NT, OBCA_QUAD_WAVES_PER_EU) void obca_quad_ipm_kernel(int B, int N, QDevBufs b, Opts o) {
    const int inst = blockIdx.x;
    if (inst >= B) return;
    if (threadIdx.x == 0) {
        quad::QInst &I = quad::gq_sh.inst;
        I.prob = (const gdbl *)(b.prob + (size_t)inst * b.s_prob);
        I.z = (gdbl *)(b.z + (size_t)inst * b.s_z); I.d = (gdbl *)(b.d + (size_t)inst * b.s_d);
        I.as = (gdbl *)(b.as + (size_t)inst * b.s_as); I.rs = (gdbl *)(b.rs + (size_t)inst * b.s_rs);
        I.oc = (gdbl *)(b.oc + (size_t)inst * b.s_oc);
#ifdef OBCA_PROFILE
        quad::gq_sh.tlast = clock64();
#endif
    }
#ifdef OBCA_PROFILE
    if (threadIdx.x < 16) quad::gq_sh.prof[threadIdx.x] = 0;
#endif
    __syncthreads();
    quad::q_solve_instance(N, o, b.info + (size_t)inst * 8);
#ifdef OBCA_PROFILE
    __syncthreads();
    if (threadIdx.x < 16) b.prof[(size_t)inst * 16 + threadIdx.x] = quad::gq_sh.prof[threadIdx.x];
#endif
}

// ------------------------------------------------------------------------------------------------ host side
struct obca_ctx { int device; hipStream_t stream; std::string err; std::string name; int cus; };
static std::string g_create_err;

struct obca_batch {
    obca_ctx *ctx; int B, N, nObMax, MMax, zlen, have_duals, uploaded, dist;
    DevBufs d;
    std::vector<int> nOb, M, obOff, rowOff;
    std::vector<double> Ts; int fixTime;
    hipEvent_t e0, e1, e2;
    long long bytes;
    int sliced;          // slice length (passes) of the last solve if it used the two-launch schedule, else 0
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } } while (0)

extern "C" {

int obca_default_opts(obca_opts *o) {
    if (!o) return -1;
    o->tol = 1e-5; o->max_iter = 200;                 /* ParkingSignedDist.jl:42 */
    o->mu_init = 0.1; o->kappa_eps = 10; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99;
    o->bound_push = 1e-2; o->bound_frac = 1e-2;
    o->dw_min = 1e-12;                                 /* min_hessian_perturbation, :43 */
    o->dw0 = 1e-4; o->dw_max = 1e40; o->kw_inc0 = 100; o->kw_inc = 8; o->kw_dec = 1.0 / 3;
    o->dc_bar = 1e-7;                                  /* jacobian_regularization_value, :43 */
    o->kappa_c = 0.25;
    o->gamma_theta = 1e-5; o->gamma_phi = 1e-8; o->delta = 1; o->s_theta = 1.1; o->s_phi = 2.3;
    o->eta_phi = 1e-8; o->gamma_alpha = 0.05; o->s_max = 100; o->kappa_sigma = 1e10;
    o->constr_viol_tol = 1e-4; o->dual_inf_tol = 1; o->compl_inf_tol = 1e-4; o->rho_term = 1e3;
    return 0;
}

int obca_create(obca_ctx **out, int device) {
    if (!out) return -1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_err = "no HIP device available (libobca_hip has no CPU fallback)"; return -2; }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return -1; }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) { g_create_err = "hipGetDeviceProperties failed"; return -2; }
    if (std::string(pr.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_err = std::string("device is ") + pr.gcnArchName + ", libobca_hip is built for gfx950 only"; return -2;
    }
    obca_ctx *c = new obca_ctx();
    c->device = device; c->name = std::string(pr.name) + " (" + pr.gcnArchName + ")"; c->cus = pr.multiProcessorCount;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_err = "hipStreamCreate failed"; delete c; return -2; }
    // (non-blocking: the context's stream never synchronises implicitly with the legacy default stream, which other libraries in the process may use;
    //  several contexts are meant to run side by side, see INTEGRATION.md "several batches in flight")
    *out = c;
    return 0;
}
int obca_destroy(obca_ctx *c) { if (!c) return -1; hipSetDevice(c->device); hipStreamDestroy(c->stream); delete c; return 0; }
const char *obca_last_error(const obca_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }
int obca_device_name(const obca_ctx *c, char *buf, int n) { if (!c || !buf || n <= 0) return -1; snprintf(buf, n, "%s", c->name.c_str()); return 0; }

int obca_batch_create(obca_ctx *ctx, int B, int N, obca_batch **out) {
    if (!ctx || !out) return -1;
    if (B < 1 || N < 0 || N > OBCA_NMAX) { ctx->err = "obca_batch_create: need B>=1, 0<=N<=OBCA_NMAX"; return -1; }
    obca_batch *bt = new obca_batch();
    bt->ctx = ctx; bt->B = B; bt->N = N; bt->uploaded = 0; bt->have_duals = 0; bt->nObMax = 0; bt->MMax = 0; bt->bytes = 0; bt->dist = 0; bt->sliced = 0;
    memset(&bt->d, 0, sizeof bt->d);
    hipSetDevice(ctx->device);
    HIPCHK(ctx, hipEventCreate(&bt->e0)); HIPCHK(ctx, hipEventCreate(&bt->e1)); HIPCHK(ctx, hipEventCreate(&bt->e2));
    *out = bt;
    return 0;
}
static void free_dev(obca_batch *bt) {
    double **ps[] = {&bt->d.prob, &bt->d.z0, &bt->d.z, &bt->d.d, &bt->d.as, &bt->d.rs, &bt->d.oc, &bt->d.traj, &bt->d.info, &bt->d.dws, &bt->d.prof, &bt->d.slice};
    for (auto p : ps) { if (*p) hipFree(*p); *p = nullptr; }
    if (bt->d.order) hipFree(bt->d.order); bt->d.order = nullptr;
}
int obca_batch_destroy(obca_batch *bt) {
    if (!bt) return -1;
    hipSetDevice(bt->ctx->device);
    free_dev(bt); hipEventDestroy(bt->e0); hipEventDestroy(bt->e1); hipEventDestroy(bt->e2);
    delete bt; return 0;
}
int obca_batch_debug_phase_cycles(obca_batch *bt, double *out /* B x 16 */) {   /* non-zero only in -DOBCA_PROFILE builds */
    if (!bt || !out) return -1;
    HIPCHK(bt->ctx, hipStreamSynchronize(bt->ctx->stream));
    HIPCHK(bt->ctx, hipMemcpy(out, bt->d.prof, (size_t)bt->B * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
int obca_batch_set_formulation(obca_batch *bt, int dist) { if (!bt) return -1; bt->dist = dist ? 1 : 0; return 0; }   /* before obca_batch_upload */
int obca_batch_scratch_bytes(const obca_batch *bt, long long *bytes) { if (!bt || !bytes) return -1; *bytes = bt->bytes; return 0; }

int obca_batch_upload(obca_batch *bt, const double *Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                      const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A, const double *b,
                      const double *rx, const double *ry, const double *ryaw, const double *xWS, const double *uWS,
                      const double *lWS, const double *nWS) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    const int B = bt->B, N = bt->N, N1 = N + 1;
    if (!Ts || !ego || !XYb || !nOb || !vOb || !A || !b || !rx || !ry || !ryaw) { ctx->err = "obca_batch_upload: NULL argument"; return -1; }
    bt->nOb.assign(B, 0); bt->M.assign(B, 0); bt->obOff.assign(B + 1, 0); bt->rowOff.assign(B + 1, 0);
    int nObMax = 0, MMax = 0;
    for (int i = 0; i < B; i++) {
        int n = nOb[i];
        if (n < 1 || n > OBCA_NOBMAX) { ctx->err = "obca_batch_upload: nOb out of range 1..OBCA_NOBMAX"; return -1; }
        int m = 0;
        for (int j = 0; j < n; j++) { int v = vOb[bt->obOff[i] + j]; if (v < 1 || v > OBCA_VMAX) { ctx->err = "obca_batch_upload: vOb out of range 1..OBCA_VMAX"; return -1; } m += v; }
        bt->nOb[i] = n; bt->M[i] = m; bt->obOff[i + 1] = bt->obOff[i] + n; bt->rowOff[i + 1] = bt->rowOff[i] + m;
        if (n > nObMax) nObMax = n; if (m > MMax) MMax = m;
    }
    Lay lmax; make_layout(N, nObMax, MMax, lmax);
    hipSetDevice(ctx->device);
    if (!bt->uploaded || nObMax != bt->nObMax || MMax != bt->MMax) {
        free_dev(bt);
        bt->nObMax = nObMax; bt->MMax = MMax; bt->zlen = lmax.len;
        DevBufs &d = bt->d;
        d.s_prob = OB_HDR + 3 * (size_t)N1; d.s_z = lmax.len; d.s_as = (size_t)N1 * OB_AS; d.s_rs = (size_t)N1 * OB_RS;
        d.s_oc = (size_t)N1 * nObMax * OB_OC; d.s_traj = (size_t)(N + 2) * 6;
        size_t tot = 0;
#define ALLOC(ptr, cnt) do { size_t by_ = (size_t)(cnt) * sizeof(double); HIPCHK(ctx, hipMalloc((void **)&(ptr), by_)); tot += by_; } while (0)
        ALLOC(d.prob, B * d.s_prob); ALLOC(d.z0, B * d.s_z); ALLOC(d.z, B * d.s_z); ALLOC(d.d, B * d.s_z);
        ALLOC(d.as, B * d.s_as); ALLOC(d.rs, B * d.s_rs); ALLOC(d.oc, B * d.s_oc); ALLOC(d.traj, B * d.s_traj);
        ALLOC(d.info, (size_t)B * 8); ALLOC(d.dws, (size_t)B * N1 * nObMax); ALLOC(d.prof, (size_t)B * 16);
        ALLOC(d.slice, (size_t)B * SL_SIZE);
        HIPCHK(ctx, hipMalloc((void **)&d.order, ((size_t)B + 1) * sizeof(int))); tot += ((size_t)B + 1) * sizeof(int);
#undef ALLOC
        bt->bytes = (long long)tot;
    }
    const DevBufs &d = bt->d;
    const size_t W = (size_t)lmax.nprimal;                 // only the primal prefix (x,u,t,lam,mu,...) of the iterate travels over PCIe
    std::vector<double> hp((size_t)B * d.s_prob, 0.0), hz((size_t)B * W, 0.0);
    const double W_ev = ego[1] + ego[3], L_ev = ego[0] + ego[2];     /* ParkingSignedDist.jl:182-188 */
    for (int i = 0; i < B; i++) {
        double *p = hp.data() + (size_t)i * d.s_prob;
        const int n = bt->nOb[i], m = bt->M[i];
        p[PH_TS] = Ts[i]; p[PH_L] = L; p[PH_DIST] = bt->dist ? 1.0 : 0.0;
        p[PH_G] = L_ev / 2; p[PH_G + 1] = W_ev / 2; p[PH_G + 2] = L_ev / 2; p[PH_G + 3] = W_ev / 2;
        p[PH_OFF] = (ego[0] + ego[2]) / 2 - ego[2];
        p[PH_XL] = XYb[0]; p[PH_XL + 1] = XYb[2]; p[PH_XL + 2] = -1e300; p[PH_XL + 3] = -1.0;     /* :104-106 */
        p[PH_XU] = XYb[1]; p[PH_XU + 1] = XYb[3]; p[PH_XU + 2] = 1e300; p[PH_XU + 3] = 2.0;
        for (int q = 0; q < 4; q++) { p[PH_X0 + q] = x0 ? x0[4 * i + q] : 0.0; p[PH_XF + q] = xF ? xF[4 * i + q] : 0.0; }
        p[PH_FIX] = fixTime ? 1 : 0; p[PH_NOB] = n; p[PH_M] = m;
        int ro = 0;
        for (int j = 0; j < n; j++) { int v = vOb[bt->obOff[i] + j]; p[PH_VOB + j] = v; p[PH_ROFF + j] = ro; ro += v; }
        p[PH_ROFF + n] = ro;
        for (int r = 0; r < m; r++) { p[PH_A + 2 * r] = A[2 * (bt->rowOff[i] + r)]; p[PH_A + 2 * r + 1] = A[2 * (bt->rowOff[i] + r) + 1]; p[PH_B + r] = b[bt->rowOff[i] + r]; }
        for (int k = 0; k < N1; k++) { p[OB_HDR + k] = rx[(size_t)i * N1 + k]; p[OB_HDR + N1 + k] = ry[(size_t)i * N1 + k]; p[OB_HDR + 2 * N1 + k] = ryaw[(size_t)i * N1 + k]; }
        Lay l; make_layout(N, n, m, l);
        double *z = hz.data() + (size_t)i * W;
        if (xWS) memcpy(z + l.x, xWS + (size_t)i * 4 * N1, sizeof(double) * 4 * N1);
        if (uWS) memcpy(z + l.u, uWS + (size_t)i * 2 * N, sizeof(double) * 2 * N);
        z[l.t] = 1.0;                                                 /* ParkingSignedDist.jl:214 */
        if (lWS && nWS) {
            memcpy(z + l.lam, lWS + (size_t)bt->rowOff[i] * N1, sizeof(double) * m * N1);
            memcpy(z + l.mu, nWS + (size_t)bt->obOff[i] * 4 * N1, sizeof(double) * 4 * n * N1);
        }
    }
    bt->have_duals = (lWS && nWS) ? 1 : 0; bt->fixTime = fixTime ? 1 : 0;
    bt->Ts.assign(Ts, Ts + B);
    HIPCHK(ctx, hipMemcpyAsync(d.prob, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d.z0, 0, (size_t)B * d.s_z * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(d.z0, d.s_z * sizeof(double), hz.data(), W * sizeof(double), W * sizeof(double), B, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    bt->uploaded = 1;
    return 0;
}

static int launch_dualws(obca_batch *bt, double *zdst) {
    obca_ctx *ctx = bt->ctx;
    long long tot = (long long)bt->B * (bt->N + 1) * bt->nObMax;
    int blocks = (int)((tot + 255) / 256);
    hipLaunchKernelGGL(obca_dualws_kernel, dim3(blocks), dim3(256), 0, ctx->stream, bt->B, bt->N, bt->nObMax, bt->d, zdst, bt->d.s_z);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int obca_batch_solve(obca_batch *bt, const obca_opts *opts) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    if (!bt->uploaded) { ctx->err = "obca_batch_solve: nothing uploaded"; return -1; }
    if (bt->N < 2) { ctx->err = "obca_batch_solve: the NLP needs a horizon N>=2"; return -1; }
    obca_opts o; if (opts) o = *opts; else obca_default_opts(&o);
    Opts ko; memcpy(&ko, &o, sizeof ko);
    hipSetDevice(ctx->device);
    const DevBufs &d = bt->d;
    HIPCHK(ctx, hipMemcpyAsync(d.z, d.z0, (size_t)bt->B * d.s_z * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(bt->e0, ctx->stream));
    if (!bt->have_duals) { int rc = launch_dualws(bt, d.z); if (rc) return rc; }
    HIPCHK(ctx, hipEventRecord(bt->e1, ctx->stream));
    // Two-launch schedule.  The kernel keeps two instances per CU resident; a larger batch is dispatched in blockIdx
    // order as workgroups retire, so an instance that needs three times the median number of passes and happens to sit late in the batch
    // would start late and finish alone.  Instead every instance first runs a short slice (OBCA_SLICE_PASSES factorisation passes, default
    // 6), the parked solves are ranked by what the slice revealed, and a second launch finishes them hardest-first.  No work is repeated
    // and every instance walks through the same iterates as in a single launch.  OBCA_SLICE_PASSES=0 turns it off; OBCA_SLICE_ONLY=1
    // (diagnostic) stops after the first slice.
    int budget = 6;
    if (const char *e = getenv("OBCA_SLICE_PASSES")) budget = atoi(e);
    const bool slice_only = getenv("OBCA_SLICE_ONLY") && atoi(getenv("OBCA_SLICE_ONLY"));
    const int slots = 2 * (ctx->cus > 0 ? ctx->cus : 256);
    bt->sliced = (budget > 0 && (bt->B > slots || slice_only)) ? budget : 0;   // 0: single launch, else the slice length
    if (!bt->sliced) {
        hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), 0, ctx->stream, bt->B, bt->N, bt->d, ko, 0, 0);
        HIPCHK(ctx, hipGetLastError());
    } else {
        hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), 0, ctx->stream, bt->B, bt->N, bt->d, ko, 0, budget);
        HIPCHK(ctx, hipGetLastError());
        if (!slice_only) {
            hipLaunchKernelGGL(obca_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, bt->B, (const double *)d.info, (const double *)d.slice, d.order);
            HIPCHK(ctx, hipGetLastError());
            hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), 0, ctx->stream, bt->B, bt->N, bt->d, ko, 1, 0);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    HIPCHK(ctx, hipEventRecord(bt->e2, ctx->stream));
    return 0;
}
int obca_batch_shift_warm_start(obca_batch *bt, int shift, const double *x0_new) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    if (!bt->uploaded) { ctx->err = "obca_batch_shift_warm_start: nothing uploaded"; return -1; }
    if (shift < 0 || shift > bt->N) { ctx->err = "obca_batch_shift_warm_start: shift out of range 0..N"; return -1; }
    hipSetDevice(ctx->device);
    double *dx0 = nullptr;
    if (x0_new) {
        HIPCHK(ctx, hipMalloc((void **)&dx0, (size_t)bt->B * 4 * sizeof(double)));
        HIPCHK(ctx, hipMemcpyAsync(dx0, x0_new, (size_t)bt->B * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(obca_shift_kernel, dim3(bt->B), dim3(128), 0, ctx->stream, bt->B, bt->N, shift, bt->d, (const double *)dx0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (dx0) hipFree(dx0);
    bt->have_duals = 1;                                 // the shifted multipliers are the dual warm start: DualMultWS is skipped
    return 0;
}
int obca_batch_sync(obca_batch *bt) { if (!bt) return -1; hipSetDevice(bt->ctx->device); HIPCHK(bt->ctx, hipStreamSynchronize(bt->ctx->stream)); return 0; }
int obca_batch_last_schedule(const obca_batch *bt, int *ipm_launches, int *slice_passes) {
    if (!bt) return -1;
    if (ipm_launches) *ipm_launches = bt->sliced ? 2 : 1;
    if (slice_passes) *slice_passes = bt->sliced;
    return 0;
}
int obca_batch_kernel_ms(obca_batch *bt, float *ipm_ms, float *dualws_ms) {
    if (!bt) return -1;
    float a = 0, b = 0;
    HIPCHK(bt->ctx, hipEventElapsedTime(&a, bt->e0, bt->e1)); HIPCHK(bt->ctx, hipEventElapsedTime(&b, bt->e1, bt->e2));
    if (dualws_ms) *dualws_ms = a; if (ipm_ms) *ipm_ms = b;
    return 0;
}

int obca_batch_download(obca_batch *bt, double *xp, double *up, double *ts, int *exitflag, double *lp, double *np, double *slp, double *info) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    const int B = bt->B, N = bt->N, N1 = N + 1;
    const DevBufs &d = bt->d;
    hipSetDevice(ctx->device);
    Lay lmax; make_layout(N, bt->nObMax, bt->MMax, lmax);
    const size_t W = (size_t)lmax.so;                       // outputs are a prefix of the iterate: x, u, t, lam, mu, sl
    std::vector<double> hz((size_t)B * W), hi((size_t)B * 8);
    HIPCHK(ctx, hipMemcpy2DAsync(hz.data(), W * sizeof(double), d.z, d.s_z * sizeof(double), W * sizeof(double), B, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hi.data(), d.info, hi.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < B; i++) {
        Lay l; make_layout(N, bt->nOb[i], bt->M[i], l);
        const double *z = hz.data() + (size_t)i * W;
        if (xp) memcpy(xp + (size_t)i * 4 * N1, z + l.x, sizeof(double) * 4 * N1);
        if (up) memcpy(up + (size_t)i * 2 * N, z + l.u, sizeof(double) * 2 * N);
        if (ts) for (int k = 0; k < N1; k++) ts[(size_t)i * N1 + k] = bt->fixTime ? 1.0 : z[l.t];   /* ParkingSignedDist.jl:304-308 */
        if (lp) memcpy(lp + (size_t)bt->rowOff[i] * N1, z + l.lam, sizeof(double) * bt->M[i] * N1);
        if (np) memcpy(np + (size_t)bt->obOff[i] * 4 * N1, z + l.mu, sizeof(double) * 4 * bt->nOb[i] * N1);
        if (slp) memcpy(slp + (size_t)bt->obOff[i] * N1, z + l.sl, sizeof(double) * bt->nOb[i] * N1);
        if (exitflag) exitflag[i] = (int)hi[(size_t)i * 8 + 7];
        if (info) memcpy(info + (size_t)i * 8, hi.data() + (size_t)i * 8, sizeof(double) * 8);
    }
    return 0;
}

static int parking_batch(obca_ctx *ctx, int dist, int B, int N, const double *Ts, double L, const double ego[4], const double XYb[4],
                         int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                         const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                         const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp,
                         double *up, double *timeScale, int *exitflag, double *lp, double *np, double *slp, double *info) {
    if (!ctx) return -1;
    if (!x0 || !xF || !xWS || !uWS) { ctx->err = "obca_parking_(signed_)dist_batch: NULL argument"; return -1; }
    obca_batch *bt = nullptr;
    int rc = obca_batch_create(ctx, B, N, &bt);
    if (rc) return rc;
    obca_batch_set_formulation(bt, dist);
    rc = obca_batch_upload(bt, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS);
    if (!rc) rc = obca_batch_solve(bt, opts);
    if (!rc) rc = obca_batch_sync(bt);
    if (!rc) rc = obca_batch_download(bt, xp, up, timeScale, exitflag, lp, np, slp, info);
    obca_batch_destroy(bt);
    return rc;
}
int obca_parking_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double L, const double ego[4], const double XYb[4],
                                   int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                                   const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                                   const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp,
                                   double *up, double *timeScale, int *exitflag, double *lp, double *np, double *slp, double *info) {
    return parking_batch(ctx, 0, B, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, opts, xp, up,
                         timeScale, exitflag, lp, np, slp, info);
}
int obca_parking_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double L, const double ego[4], const double XYb[4],
                            int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                            const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                            const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp,
                            double *up, double *timeScale, int *exitflag, double *lp, double *np, double *info) {
    return parking_batch(ctx, 1, B, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, opts, xp, up,
                         timeScale, exitflag, lp, np, nullptr, info);
}

int obca_dualmult_ws_batch(obca_ctx *ctx, int B, int N, const double ego[4], const int *nOb, const int *vOb, const double *A,
                           const double *b, const double *rx, const double *ry, const double *ryaw, double *lWS, double *nWS, double *dd) {
    if (!ctx) return -1;
    if (!lWS || !nWS) { ctx->err = "obca_dualmult_ws_batch: NULL output"; return -1; }
    obca_batch *bt = nullptr;
    int rc = obca_batch_create(ctx, B, N, &bt);
    if (rc) return rc;
    std::vector<double> Ts(B, 1.0); const double XYb[4] = {0, 0, 0, 0};
    rc = obca_batch_upload(bt, Ts.data(), 1.0, ego, XYb, 0, nullptr, nullptr, nOb, vOb, A, b, rx, ry, ryaw, nullptr, nullptr, nullptr, nullptr);
    if (!rc) rc = launch_dualws(bt, bt->d.z);
    if (!rc) rc = obca_batch_sync(bt);
    if (!rc) {
        const int N1 = N + 1;
        std::vector<double> hz((size_t)B * bt->d.s_z), hd((size_t)B * N1 * bt->nObMax);
        if (hipMemcpy(hz.data(), bt->d.z, hz.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hd.data(), bt->d.dws, hd.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "obca_dualmult_ws_batch: copy back failed"; rc = -2; }
        for (int i = 0; i < B && !rc; i++) {
            Lay l; make_layout(N, bt->nOb[i], bt->M[i], l);
            const double *z = hz.data() + (size_t)i * bt->d.s_z;
            memcpy(lWS + (size_t)bt->rowOff[i] * N1, z + l.lam, sizeof(double) * bt->M[i] * N1);
            memcpy(nWS + (size_t)bt->obOff[i] * 4 * N1, z + l.mu, sizeof(double) * 4 * bt->nOb[i] * N1);
            if (dd) for (int k = 0; k < N1; k++) for (int j = 0; j < bt->nOb[i]; j++)
                dd[(size_t)bt->obOff[i] * N1 + (size_t)k * bt->nOb[i] + j] = hd[(size_t)i * N1 * bt->nObMax + (size_t)k * bt->nObMax + j];
        }
    }
    obca_batch_destroy(bt);
    return rc;
}

/* ---------------------------------------------------------------- quadcopter path */
struct obca_quad_batch {
    obca_ctx *ctx; int B, N, uploaded;
    QDevBufs d; hipEvent_t e0, e1; long long bytes;
};
static void qfree_dev(obca_quad_batch *bt) {
    double **ps[] = {&bt->d.prob, &bt->d.z, &bt->d.d, &bt->d.as, &bt->d.rs, &bt->d.oc, &bt->d.info, &bt->d.prof};
    for (auto p : ps) { if (*p) hipFree(*p); *p = nullptr; }
}
int obca_quadcopter_default_opts(obca_opts *o) {
    if (obca_default_opts(o)) return -1;
    o->max_iter = 3000; o->dw_min = 1e-10;            /* QuadcopterSignedDist.jl:28-31: no max_iter (IPOPT default), min_hessian_perturbation 1e-10 */
    return 0;
}
int obca_quad_batch_create(obca_ctx *ctx, int B, int N, obca_quad_batch **out) {
    if (!ctx || !out) return -1;
    if (B < 1 || N < 2 || N > OBCA_QUAD_NMAX) { ctx->err = "obca_quad_batch_create: need B>=1, 2<=N<=OBCA_QUAD_NMAX"; return -1; }
    obca_quad_batch *bt = new obca_quad_batch();
    bt->ctx = ctx; bt->B = B; bt->N = N; bt->uploaded = 0; bt->bytes = 0; memset(&bt->d, 0, sizeof bt->d);
    hipSetDevice(ctx->device);
    quad::QLay l; quad::q_make_layout(N, l);
    QDevBufs &d = bt->d; const size_t N1 = N + 1;
    d.s_prob = QPH_SIZE + QX * N1; d.s_z = l.len; d.s_d = l.n + l.m; d.s_as = N1 * QSR; d.s_rs = N1 * QRR; d.s_oc = N1 * QOB * OB_OC;
    size_t tot = 0;
#define ALLOC(ptr, cnt) do { size_t by_ = (size_t)(cnt) * sizeof(double); if (hipMalloc((void **)&(ptr), by_) != hipSuccess) { ctx->err = "obca_quad_batch_create: hipMalloc failed"; qfree_dev(bt); delete bt; return -2; } tot += by_; } while (0)
    ALLOC(d.prob, B * d.s_prob); ALLOC(d.z, B * d.s_z); ALLOC(d.d, B * d.s_d); ALLOC(d.as, B * d.s_as); ALLOC(d.rs, B * d.s_rs);
    ALLOC(d.oc, B * d.s_oc); ALLOC(d.info, (size_t)B * 8); ALLOC(d.prof, (size_t)B * 16);
#undef ALLOC
    bt->bytes = (long long)tot;
    if (hipEventCreate(&bt->e0) != hipSuccess || hipEventCreate(&bt->e1) != hipSuccess) { ctx->err = "hipEventCreate failed"; qfree_dev(bt); delete bt; return -2; }
    *out = bt;
    return 0;
}
int obca_quad_batch_destroy(obca_quad_batch *bt) {
    if (!bt) return -1;
    hipSetDevice(bt->ctx->device); qfree_dev(bt); hipEventDestroy(bt->e0); hipEventDestroy(bt->e1); delete bt; return 0;
}
int obca_quad_batch_debug_phase_cycles(obca_quad_batch *bt, double *out /* B x 16 */) {   /* non-zero only in -DOBCA_PROFILE builds */
    if (!bt || !out) return -1;
    HIPCHK(bt->ctx, hipStreamSynchronize(bt->ctx->stream));
    HIPCHK(bt->ctx, hipMemcpy(out, bt->d.prof, (size_t)bt->B * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
int obca_quad_batch_scratch_bytes(const obca_quad_batch *bt, long long *bytes) { if (!bt || !bytes) return -1; *bytes = bt->bytes; return 0; }
int obca_quad_batch_upload(obca_quad_batch *bt, const double *Ts, double R, const double *x0, const double *xF, const double *ob,
                           const double *xWS, const double *timeWS, int dual_ws, int dist) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    if (!Ts || !x0 || !xF || !ob || !xWS || !timeWS) { ctx->err = "obca_quad_batch_upload: NULL argument"; return -1; }
    const int B = bt->B, N1 = bt->N + 1; const QDevBufs &d = bt->d;
    std::vector<double> hp((size_t)B * d.s_prob, 0.0);
    for (int i = 0; i < B; i++) {
        double *p = hp.data() + (size_t)i * d.s_prob;
        p[QPH_TS] = Ts[i]; p[QPH_R] = R; p[QPH_TWS] = timeWS[i]; p[QPH_DWS] = dual_ws ? 1.0 : 0.0; p[QPH_DIST] = dist ? 1.0 : 0.0;
        memcpy(p + QPH_X0, x0 + (size_t)QX * i, sizeof(double) * QX); memcpy(p + QPH_XF, xF + (size_t)QX * i, sizeof(double) * QX);
        memcpy(p + QPH_OB, ob + (size_t)QOB * QL * i, sizeof(double) * QOB * QL);
        memcpy(p + QPH_SIZE, xWS + (size_t)QX * N1 * i, sizeof(double) * QX * N1);
    }
    hipSetDevice(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(d.prob, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    bt->uploaded = 1;
    return 0;
}
int obca_quad_batch_solve(obca_quad_batch *bt, const obca_opts *opts) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    if (!bt->uploaded) { ctx->err = "obca_quad_batch_solve: nothing uploaded"; return -1; }
    obca_opts o; if (opts) o = *opts; else obca_quadcopter_default_opts(&o);
    Opts ko; memcpy(&ko, &o, sizeof ko);
    hipSetDevice(ctx->device);
    HIPCHK(ctx, hipEventRecord(bt->e0, ctx->stream));
    hipLaunchKernelGGL(obca_quad_ipm_kernel, dim3(bt->B), dim3(OB_NT), 0, ctx->stream, bt->B, bt->N, bt->d, ko);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(bt->e1, ctx->stream));
    return 0;
}
int obca_quad_batch_sync(obca_quad_batch *bt) { if (!bt) return -1; hipSetDevice(bt->ctx->device); HIPCHK(bt->ctx, hipStreamSynchronize(bt->ctx->stream)); return 0; }
int obca_quad_batch_kernel_ms(obca_quad_batch *bt, float *ipm_ms) {
    if (!bt || !ipm_ms) return -1;
    HIPCHK(bt->ctx, hipEventElapsedTime(ipm_ms, bt->e0, bt->e1));
    return 0;
}
int obca_quad_batch_download(obca_quad_batch *bt, double *xp, double *up, double *ts, int *exitflag, double *lp, double *slp, double *info) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx; const int B = bt->B, N = bt->N, N1 = N + 1; const QDevBufs &d = bt->d;
    quad::QLay l; quad::q_make_layout(N, l);
    hipSetDevice(ctx->device);
    std::vector<double> hz((size_t)B * d.s_z), hi((size_t)B * 8);
    HIPCHK(ctx, hipMemcpyAsync(hz.data(), d.z, hz.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hi.data(), d.info, hi.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < B; i++) {
        const double *z = hz.data() + (size_t)i * d.s_z;
        if (xp) memcpy(xp + (size_t)i * QX * N1, z + l.x, sizeof(double) * QX * N1);
        if (up) memcpy(up + (size_t)i * QU * N, z + l.u, sizeof(double) * QU * N);
        if (ts) for (int k = 0; k < N1; k++) ts[(size_t)i * N1 + k] = z[l.t];                       /* QuadcopterSignedDist.jl:293 */
        if (lp) memcpy(lp + (size_t)i * QL * QOB * N1, z + l.lam, sizeof(double) * QL * QOB * N1);    /* [l1;..;l5] stacked, :295 */
        if (slp) memcpy(slp + (size_t)i * QOB * N1, z + l.s, sizeof(double) * QOB * N1);
        if (exitflag) exitflag[i] = (int)hi[(size_t)i * 8 + 7];
        if (info) memcpy(info + (size_t)i * 8, hi.data() + (size_t)i * 8, sizeof(double) * 8);
    }
    return 0;
}
static int quadcopter_batch(obca_ctx *ctx, int dist, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                            const double *ob, const double *xWS, const double *timeWS, int dual_ws, const obca_opts *opts, double *xp,
                            double *up, double *timeScale, int *exitflag, double *lp, double *slp, double *info) {
    if (!ctx) return -1;
    obca_quad_batch *bt = nullptr;
    int rc = obca_quad_batch_create(ctx, B, N, &bt);
    if (rc) return rc;
    rc = obca_quad_batch_upload(bt, Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, dist);
    if (!rc) rc = obca_quad_batch_solve(bt, opts);
    if (!rc) rc = obca_quad_batch_sync(bt);
    if (!rc) rc = obca_quad_batch_download(bt, xp, up, timeScale, exitflag, lp, slp, info);
    obca_quad_batch_destroy(bt);
    return rc;
}
int obca_quadcopter_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                                      const double *ob, const double *xWS, const double *uWS, const double *timeWS, int dual_ws,
                                      const obca_opts *opts, double *xp, double *up, double *timeScale, int *exitflag, double *lp,
                                      double *slp, double *info) {
    (void)uWS;                                         /* the reference ignores it too: inputs start at the hover speed, :202 */
    return quadcopter_batch(ctx, 0, B, N, Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, opts, xp, up, timeScale, exitflag, lp, slp, info);
}
int obca_quadcopter_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                               const double *ob, const double *xWS, const double *uWS, const double *timeWS, int dual_ws,
                               const obca_opts *opts, double *xp, double *up, double *timeScale, int *exitflag, double *lp, double *info) {
    (void)uWS;                                         /* QuadcopterDist.jl:196 */
    return quadcopter_batch(ctx, 1, B, N, Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, opts, xp, up, timeScale, exitflag, lp, nullptr, info);
}

}  // extern "C"
