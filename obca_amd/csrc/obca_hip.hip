// obca_hip.hip -- HIP kernels and the C ABI of libobca_hip.so (gfx950 only; see include/obca_hip.h).
//
// Kernels
//   obca_parking_ipm_kernel : one wavefront (64-thread workgroup) per problem instance, persistent over the whole
//                             interior-point solve (obca_solver.h).  grid = B, block = 64; four instances per CU.
//   obca_quad_ipm_kernel    : the same for the quadcopter NLP (obca_quad_solver.h).
//   obca_dualws_kernel      : one lane per (instance, stage, obstacle) convex sub-problem of DualMultWS (obca_model.h).
// Memory (per instance, fp64, all in HBM; sizes for N=80, 3 obstacles / 5 rows in brackets):
//   prob  header+rx,ry,ryaw   [495]      z, zn  primal-dual iterate and the line search's
//   trial point (they swap) [6797 each]      d  stage part of the search direction [~650 used]
//   as    assembled stage records (N+1) x 60 [4860]     rs  Riccati records N x 74 [5920]      slice  state of a parked solve [480]
//   (the condensed obstacle sums, the forward-sweep trajectory and the composed stage-pair maps live in LDS; `oc` is used by the quadcopter kernel only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <thread>
#include <atomic>
#include <algorithm>
#include "obca_solver.h"
#include "obca_quad_solver.h"
#include "../../include/obca_hip.h"

using namespace obca;
#ifdef OBCA_POISON
#ifndef OBCA_LDS_GUARD
#define OBCA_LDS_GUARD 1024      // doubles (8 KB) of NaN guard behind the dynamic LDS block of the poisoned build
#endif
#endif

static_assert(sizeof(obca_opts) == sizeof(OptsAbi) && offsetof(obca_opts, max_soc) == offsetof(OptsAbi, max_soc), "obca_opts must mirror obca::OptsAbi");
static_assert(OBCA_QUAD_NMAX == QNMAX, "ABI limits must match the kernels");
static_assert(OBCA_VMAX == OB_VMAX && OBCA_NOBMAX == OB_NOBMAX && OBCA_NMAX == OB_NMAX && OBCA_MMAX == OB_MMAX, "ABI limits must match the kernels");

struct DevBufs {
    double *prob, *z0, *z, *zn, *d, *as, *rs, *oc, *info, *dws, *prof;     // zn: the second iterate buffer of the fused line search (obca_solver.h)
    double *slice;                                   // slice records (SL_SIZE doubles per instance) of the two-launch schedule
    // second-order correction (opts.max_soc > 0): the corrected right-hand-side rows of every instance; allocated at the first such solve
    double *csoc; size_t s_csoc, o_csoc;      // per instance: [ direction of a correction, o_csoc doubles (indexed like d: the layout's offsets below zxL) | the rows c_soc ]
    int *order;                                      // B instance indices in dispatch order (-1: nothing left to do), then the class counters
    size_t s_prob, s_z, s_as, s_rs, s_oc;   // strides in doubles
};

// One wavefront per SIMD (four one-wavefront instances per CU): each wave may then use 256 VGPRs + 256 AGPRs, and the register-hungry per-lane phases
// keep their spills (and most callee-saved registers) in AGPRs instead of scratch.  Against two waves per SIMD with 256 registers each this
// is faster at every batch size measured: a lone instance is ~10 % quicker per pass, and a full machine no longer streams the scratch
// save areas through HBM.
#ifndef OBCA_IPM_WAVES_PER_EU
#define OBCA_IPM_WAVES_PER_EU 1
#endif
#define OBCA_RESIDENT_PER_CU (4 * OBCA_IPM_WAVES_PER_EU)   // parking instances (one wavefront each) resident per CU
__global__ __launch_bounds__(OB_NT, OBCA_IPM_WAVES_PER_EU) void obca_parking_ipm_kernel(int B, int N, DevBufs b, Opts o, int mode, int budget, int max_soc, int recalc_y, int lsq_init, int restoration) {
    // mode 0: fresh solve of instance blockIdx.x (at most `budget` factorisation passes if budget > 0); mode 1: continue the parked solves in
    // the order the ordering kernel chose (workgroups are dispatched in blockIdx order, so the expected stragglers start first)
    const int inst = mode == 1 ? b.order[blockIdx.x] : (int)blockIdx.x;
    if (inst < 0 || inst >= B) return;
#ifdef OBCA_POISON      // diagnostic build: whatever the previous workgroup on this CU left in LDS is replaced by NaNs before anything is initialised
    {
#ifndef OBCA_POISON_VALUE      // (NaN hides behind fmax / fmin and every comparison: build with -DOBCA_POISON_VALUE=1e30 as well)
#define OBCA_POISON_VALUE __longlong_as_double(-1LL)
#endif
        const double nan_ = OBCA_POISON_VALUE;
        double *w = (double *)&g_sh;
#ifndef OBCA_POISON_PARTS
#define OBCA_POISON_PARTS 15      // bit 0: HBM work buffers, 1: the static LDS block, 2: the dynamic LDS block, 3: the guard behind it (bisecting builds set a subset)
#endif
        if (OBCA_POISON_PARTS & 2) for (int i = threadIdx.x; i < (int)(sizeof(Shared) / sizeof(double)); i += OB_NT) w[i] = nan_;
        if (OBCA_POISON_PARTS & 4) for (int i = threadIdx.x; i < (int)OB_DYN_LDS_DOUBLES(N); i += OB_NT) g_traj[i] = nan_;
        if (OBCA_POISON_PARTS & 8) for (int i = (int)OB_DYN_LDS_DOUBLES(N) + threadIdx.x; i < (int)OB_DYN_LDS_DOUBLES(N) + OBCA_LDS_GUARD; i += OB_NT) g_traj[i] = nan_;
        __syncthreads();
    }
#endif
    if (threadIdx.x == 0) {
        Inst &I = g_sh.inst;
        I.prob = (const gdbl *)(b.prob + (size_t)inst * b.s_prob);
        I.z = (gdbl *)(b.z + (size_t)inst * b.s_z); I.zn = (gdbl *)(b.zn + (size_t)inst * b.s_z); I.d = (gdbl *)(b.d + (size_t)inst * b.s_z);
        I.as = (gdbl *)(b.as + (size_t)inst * b.s_as); I.rs = (gdbl *)(b.rs + (size_t)inst * b.s_rs);
        I.oc = nullptr;
        g_sh.soc.dsoc = b.csoc ? (gdbl *)(b.csoc + (size_t)inst * b.s_csoc) : nullptr;
        g_sh.soc.csoc = b.csoc ? (gdbl *)(b.csoc + (size_t)inst * b.s_csoc + b.o_csoc) : nullptr;
#ifdef OBCA_PROFILE
        I.tlast = clock64();
#endif
    }
#ifdef OBCA_PROFILE
    if (threadIdx.x < 16) g_sh.prof[threadIdx.x] = mode ? b.prof[(size_t)inst * 16 + threadIdx.x] : 0.0;   // counters add up over the slices
#endif
    __syncthreads();
#if defined(OBCA_PROFILE) && !defined(OBCA_PROFILE_FINE)      // slots 11, 12 (the per-stage counters of the FINE build): when the workgroup started and ended on the constant 100 MHz clock (tools/load_profile.py: residency, shader clock rate)
    if (threadIdx.x == 0 && mode == 0) g_sh.prof[11] = (double)wall_clock64();
#endif
    solve_instance(N, o, b.info + (size_t)inst * 8, (gdbl *)(b.slice + (size_t)inst * SL_SIZE), mode, budget, max_soc, recalc_y, lsq_init, restoration);
#ifdef OBCA_PROFILE
    __syncthreads();
#ifndef OBCA_PROFILE_FINE
    if (threadIdx.x == 0 && mode == 0) {      // slot 12: resident time + (which SIMD of which CU of which XCD: XCC_ID << 12 | HW_ID[15:4]) / 65536 in the fraction
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g_sh.prof[12] = ((double)wall_clock64() - g_sh.prof[11]) + (double)(((xcc & 15u) << 12) | ((hw >> 4) & 0xfffu)) / 65536.0;
    }
    __syncthreads();
#endif
    if (threadIdx.x < 16) b.prof[(size_t)inst * 16 + threadIdx.x] = g_sh.prof[threadIdx.x];
#endif
}

// instances resident per CU: registers allow 4 x OBCA_IPM_WAVES_PER_EU, LDS (static Shared + the horizon-sized dynamic part) may allow fewer -- ask the runtime
static int parking_resident_per_cu(int N) {
    // 0 = not asked yet.  Worker lanes of several devices call this concurrently: atomics (the answer depends on the
    static std::atomic<int> cache[OB_NMAX + 1];
                                                         // code object and the horizon only -- every device of a context is
                                                         // a gfx950 with 160 KB of LDS per CU, obca_create_multi checks)
    if (N < 0 || N > OB_NMAX) return OBCA_RESIDENT_PER_CU;
    int n = cache[N].load(std::memory_order_relaxed);
    if (!n) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, obca_parking_ipm_kernel, OB_NT, OB_DYN_LDS_DOUBLES(N) * sizeof(double)) != hipSuccess || n < 1) n = OBCA_RESIDENT_PER_CU;
        cache[N].store(n, std::memory_order_relaxed);
    }
    return n;
}

// difficulty class of a parked instance (0..63, higher = dispatched earlier)
__device__ inline int obca_slice_class(const double *st) {
    // inertia rungs so far + the correction / re-estimate passes of the IPOPT switches
    const int nreg = (int)st[SL_NREG] + (int)st[SL_NREGPREV] + (int)st[SL_XPASS];
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
// (four wavefronts per SIMD for the 2-row class -- 128 registers, 35 spilled -- was measured
// in round 4: 0.29 ms against 0.25 ms at three wavefronts with 158 registers: not kept)
template <int VM>
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
    double a1[VM], a2[VM], bj[VM], g[4];
#pragma unroll
    for (int i = 0; i < VM; i++) { bool on = i < v; a1[i] = on ? p[PH_A + 2 * (r0 + i)] : 0.0; a2[i] = on ? p[PH_A + 2 * (r0 + i) + 1] : 0.0; bj[i] = on ? p[PH_B + r0 + i] : 0.0; }
#pragma unroll
    for (int i = 0; i < 4; i++) g[i] = p[PH_G + i];
    const double rx = p[OB_HDR + k], ry = p[OB_HDR + (N + 1) + k], ryaw = p[OB_HDR + 2 * (N + 1) + k], off = p[PH_OFF];
    double sn, cs; sincos(ryaw, &sn, &cs);
    double lam[VM], mu[4], dv;
    dualws_one<VM>(v, a1, a2, bj, g, rx + cs * off, ry + sn * off, cs, sn, lam, mu, &dv);
    Lay l; make_layout(N, nOb, M, l);
    double *z = zdst + (size_t)inst * s_zdst;
#pragma unroll
    for (int i = 0; i < VM; i++) if (i < v) z[l.lam + k * M + r0 + i] = lam[i];
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
__global__ __launch_bounds__(QNT, OBCA_QUAD_WAVES_PER_EU) void obca_quad_ipm_kernel(int B, int N, QDevBufs b, Opts o, int max_soc, int lsq_init, int obj_scaling) {
    const int inst = blockIdx.x;
    if (inst >= B) return;
#ifdef OBCA_POISON
    {
        const double nan_ = OBCA_POISON_VALUE;
        double *w = (double *)&quad::gq_sh;
        for (int i = threadIdx.x; i < (int)(sizeof(quad::QShared) / sizeof(double)); i += QNT) w[i] = nan_;
        for (int i = threadIdx.x; i < (N + 2) * (QS + QU); i += QNT) quad::gq_traj[i] = nan_;
        __syncthreads();
    }
#endif
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
    quad::q_solve_instance(N, o, b.info + (size_t)inst * 8, max_soc, lsq_init, obj_scaling);
#ifdef OBCA_PROFILE
    __syncthreads();
    if (threadIdx.x < 16) b.prof[(size_t)inst * 16 + threadIdx.x] = quad::gq_sh.prof[threadIdx.x];
#endif
}

// rows of a strided device array <-> a dense staging array (one contiguous PCIe transfer per direction instead of a 2-D copy):
//   scatter: dst[i * ds + j] = j < W ? src[i * W + j] : 0   for j < zero_to   (upload: primal prefix of the iterate, rest of the row cleared)
//   gather : dst[i * W + j] = src[i * ss + j]                                   (download: the output prefix of the iterate)
__global__ __launch_bounds__(256) void obca_scatter_rows_kernel(double *dst, size_t ds, const double *src, size_t W, size_t zero_to) {
    const size_t i = blockIdx.x;                     // row (instance) in grid.x: no 65 535 limit on the batch size
    for (size_t j = (size_t)blockIdx.y * blockDim.x + threadIdx.x; j < zero_to; j += (size_t)gridDim.y * blockDim.x)
        dst[i * ds + j] = j < W ? src[i * W + j] : 0.0;
}
__global__ __launch_bounds__(256) void obca_gather_rows_kernel(double *dst, size_t W, const double *src, size_t ss) {
    const size_t i = blockIdx.x;
    for (size_t j = (size_t)blockIdx.y * blockDim.x + threadIdx.x; j < W; j += (size_t)gridDim.y * blockDim.x)
        dst[i * W + j] = src[i * ss + j];
}


// ------------------------------------------------------------------------------------------------ host side
// A context drives one or several devices.  Every device has OBCA_SLOTS worker lanes ("slots": a HIP stream, a cached chunk-sized batch
// with pinned staging buffers); the host-pointer entry points cut a call's batch into chunks and a host thread per slot pulls chunks from a
// shared counter (work queue, SURVEY 8e: solve times are heavy-tailed, a static slice per device would wait for the unluckiest one):
// pack into pinned memory -> H2D -> DualMultWS + interior point -> D2H of the outputs only -> unpack into the caller's arrays.  With several
// slots per device the transfers and the host-side (un)packing of one chunk overlap the solves of the others, and the few hard instances
// at the end of one chunk overlap the bulk of the next.  Instances are independent: no collective touches the data path.
struct obca_batch;
struct obca_quad_batch;
struct Slot { int device; hipStream_t stream; obca_batch *pb; obca_quad_batch *qb; int cus; };
struct obca_ctx {
    int device; hipStream_t stream;         // primary device / stream (= slots[0]): the device-resident obca_batch_* API runs here
    std::vector<int> devices; std::vector<Slot> slots;
    std::string err; std::string name; int cus;
};
static std::string g_create_err;

struct obca_batch {
    obca_ctx *ctx; int device; hipStream_t stream; std::string err;
    int B, cap, N, nObMax, MMax, zlen, have_duals, uploaded, dist, vmax;   // vmax: most rows of one obstacle in the uploaded instances
    DevBufs d; double *stage;                               // stage: dense device staging of the PCIe transfers
    double *h_prob, *h_zin, *h_zout, *h_info; size_t hcap_prob, hcap_zin, hcap_zout, hcap_info, dcap_stage;   // pinned host staging
    std::vector<int> nOb, M, obOff, rowOff;                 // per instance; offsets into the caller's packed obstacle arrays
    // |a_r| of every half-space row of the uploaded instances (index: row offset - rowOff[0]), see batch_upload_range
    std::vector<double> rowLen;
    int fixTime;
    hipEvent_t e0, e1, e2;
    long long bytes;
    int sliced;          // slice length (passes) of the last solve if it used the two-launch schedule, else 0
};

#define HIPCHK(bt, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (bt)->err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } } while (0)
// The device index of a context / batch was validated when the context was created (obca_create: hipSetDevice checked there); later calls only select it again,
// and whatever they then launch or copy reports its own error.  Release paths ignore the status of hipFree & co. on purpose -- `(void)` says so at each site
// (the library builds with -Wall -Wextra -Werror: an ignored status is a compile error).
static inline void use_device(int device) { (void)hipSetDevice(device); }
static inline int fin(obca_batch *bt, int rc) { if (rc) bt->ctx->err = bt->err; return rc; }
// Device work buffers never carry what a previous owner of the memory left in them:
// every allocation is filled once, on the stream of the batch it belongs to (the
// lanes' streams do not synchronise with the null stream).  The product build clears
// them; -DOBCA_POISON (diagnostic build, tools/determinism_ragged.py) fills them with
// the all-ones NaN pattern instead -- and the kernels then also poison their LDS at entry
// -- so that any read of a value nothing has written yet shows up as NaN in the results.
#if defined(OBCA_POISON) && (!defined(OBCA_POISON_PARTS) || (OBCA_POISON_PARTS & 1))
#define OBCA_FILL_BYTE 0xFF
#else
#define OBCA_FILL_BYTE 0
#endif
static hipError_t dev_alloc(void **p, size_t bytes, hipStream_t stream) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { *p = nullptr; return e; }
    return hipMemsetAsync(*p, OBCA_FILL_BYTE, bytes, stream);
}

static int pinned_reserve(std::string &err, double **p, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    if (hipHostMalloc((void **)p, need * sizeof(double), hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return -2; }
    *cap = need;
    return 0;
}

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
    o->max_soc = 0; o->recalc_y = 0; o->lsq_init = 0; o->obj_scaling = 0; o->restoration = 0;
                     /* throughput defaults: the three IPOPT switches off (obca_reference_opts switches them on; obca_hip.h has the numbers behind the choice) */
    return 0;
}

int obca_reference_opts(obca_opts *o) {            /* the reference's IPOPT configuration as far as the kernels carry it */
    if (obca_default_opts(o)) return -1;
    o->max_soc = 4;                                    /* IPOPT default max_soc */
    o->recalc_y = 1;                                   /* recalc_y = "yes", ParkingSignedDist.jl:41 / ParkingDist.jl:41 */
    o->lsq_init = 1;                                   /* IPOPT default: least-squares initial multipliers, constr_mult_init_max = 1e3 */
    o->restoration = 1;                                /* IPOPT has a restoration phase; the kernels carry a block feasibility restoration in its place (obca_hip.h) */
    return 0;
}

int obca_visible_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

/* devices == NULL or ndev <= 0: every visible device */
int obca_create_multi(obca_ctx **out, const int *devices, int ndev) {
    if (!out) return -1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_err = "no HIP device available (libobca_hip has no CPU fallback)"; return -2; }
    std::vector<int> devs;
    if (!devices || ndev <= 0) { for (int i = 0; i < n; i++) devs.push_back(i); }
    else devs.assign(devices, devices + ndev);
    int nslot = 8;                                       // worker lanes per device (OBCA_SLOTS; 4 until round 6: with 16 hardware queues 8 lanes give +2.6 % on the host-pointer call)
    if (const char *ev = getenv("OBCA_SLOTS")) { nslot = atoi(ev); if (nslot < 1) nslot = 1; if (nslot > 16) nslot = 16; }
    // every device is validated before anything is created, so that a bad index cannot leave a half-built context (or a leaked stream) behind
    std::vector<hipDeviceProp_t> props(devs.size());
    for (size_t di = 0; di < devs.size(); di++) {
        const int dv = devs[di];
        if (dv < 0 || dv >= n) { g_create_err = "device index out of range"; return -1; }
        if (hipGetDeviceProperties(&props[di], dv) != hipSuccess) { g_create_err = "hipGetDeviceProperties failed"; return -2; }
        if (std::string(props[di].gcnArchName).find("gfx950") == std::string::npos) {
            g_create_err = std::string("device is ") + props[di].gcnArchName + ", libobca_hip is built for gfx950 only"; return -2;
        }
    }
    obca_ctx *c = new obca_ctx();
    c->devices = devs;
    c->name = std::string(props[0].name) + " (" + props[0].gcnArchName + ")"; c->cus = props[0].multiProcessorCount;
    for (int s = 0; s < nslot; s++) for (size_t di = 0; di < devs.size(); di++) {       // slot order interleaves the devices
        Slot sl; sl.device = devs[di]; sl.pb = nullptr; sl.qb = nullptr; sl.cus = props[di].multiProcessorCount; sl.stream = nullptr;
        // Non-blocking: the streams never synchronise implicitly with the legacy default stream, which other libraries in the process may use.
        // Only the primary lane gets its stream now; the others are created when a call first needs them: HIP spreads streams over a handful of
        // hardware queues in creation order, and a process that keeps several single-lane contexts busy at once (bench.py) would otherwise find
        // its four primary streams on ONE queue, serialised (measured: 103 k instead of 132 k solves/s).
        if (s == 0 && di == 0 && (hipSetDevice(sl.device) != hipSuccess || hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess)) {
            g_create_err = "hipStreamCreate failed"; delete c; return -2;
        }
        c->slots.push_back(sl);
    }
    c->device = c->slots[0].device; c->stream = c->slots[0].stream;
    use_device(c->device);
    *out = c;
    return 0;
}
int obca_create(obca_ctx **out, int device) { return obca_create_multi(out, &device, 1); }
int obca_device_count(const obca_ctx *c) { return c ? (int)c->devices.size() : -1; }
int obca_batch_destroy(obca_batch *bt);
int obca_quad_batch_destroy(obca_quad_batch *bt);
int obca_destroy(obca_ctx *c) {
    if (!c) return -1;
    for (auto &s : c->slots) {
        if (s.pb) obca_batch_destroy(s.pb);
        if (s.qb) obca_quad_batch_destroy(s.qb);
        if (s.stream) { use_device(s.device); (void)hipStreamDestroy(s.stream); }
    }
    delete c; return 0;
}
const char *obca_last_error(const obca_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }
int obca_device_name(const obca_ctx *c, char *buf, int n) { if (!c || !buf || n <= 0) return -1; snprintf(buf, n, "%s", c->name.c_str()); return 0; }

}  // extern "C"

static int batch_create_on(obca_ctx *ctx, int device, hipStream_t stream, int B, int N, obca_batch **out, std::string &err) {
    if (B < 1 || N < 0 || N > OBCA_NMAX) { err = "obca_batch_create: need B>=1, 0<=N<=OBCA_NMAX"; return -1; }
    obca_batch *bt = new obca_batch();
    bt->ctx = ctx; bt->device = device; bt->stream = stream;
    bt->B = B; bt->cap = B; bt->N = N; bt->uploaded = 0; bt->have_duals = 0; bt->nObMax = 0; bt->MMax = 0; bt->bytes = 0; bt->dist = 0; bt->sliced = 0;
    bt->zlen = 0; bt->fixTime = 0; bt->vmax = 0;
    memset(&bt->d, 0, sizeof bt->d); bt->stage = nullptr; bt->dcap_stage = 0;
    bt->h_prob = bt->h_zin = bt->h_zout = bt->h_info = nullptr; bt->hcap_prob = bt->hcap_zin = bt->hcap_zout = bt->hcap_info = 0;
    use_device(device);
    if (hipEventCreate(&bt->e0) != hipSuccess || hipEventCreate(&bt->e1) != hipSuccess || hipEventCreate(&bt->e2) != hipSuccess) { err = "hipEventCreate failed"; delete bt; return -2; }
    *out = bt;
    return 0;
}
static void free_dev(obca_batch *bt) {
    double **ps[] = {&bt->d.prob, &bt->d.z0, &bt->d.z, &bt->d.zn, &bt->d.d, &bt->d.as, &bt->d.rs, &bt->d.oc, &bt->d.info, &bt->d.dws, &bt->d.prof, &bt->d.slice, &bt->d.csoc, &bt->stage};
    for (auto p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (bt->d.order) (void)hipFree(bt->d.order); bt->d.order = nullptr;
    bt->dcap_stage = 0;
}

// everything a parking call hands over (host pointers of the whole call; a chunk is instances lo .. lo+n-1 of it)
struct ParkIn {
    const double *Ts; double L; const double *ego, *XYb; int fixTime;
    const double *x0, *xF; const int *nOb, *vOb; const double *A, *b, *rx, *ry, *ryaw, *xWS, *uWS, *lWS, *nWS;
    std::vector<int> obOff, rowOff;                    // running sums over the call's instances
};
struct ParkOut { double *xp, *up, *ts; int *exitflag; double *lp, *np, *slp, *info; double *lWS, *nWS, *dd; };

static int park_prefix(std::string &err, int B, const int *nOb, const int *vOb, ParkIn &in) {
    in.obOff.assign(B + 1, 0); in.rowOff.assign(B + 1, 0);
    for (int i = 0; i < B; i++) {
        const int n = nOb[i];
        if (n < 1 || n > OBCA_NOBMAX) { err = "nOb out of range 1..OBCA_NOBMAX"; return -1; }
        int m = 0;
        for (int j = 0; j < n; j++) { const int v = vOb[in.obOff[i] + j]; if (v < 1 || v > OBCA_VMAX) { err = "vOb out of range 1..OBCA_VMAX"; return -1; } m += v; }
        if (m > OBCA_MMAX) { err = "more than OBCA_MMAX half-space rows in one instance"; return -1; }
        in.obOff[i + 1] = in.obOff[i] + n; in.rowOff[i + 1] = in.rowOff[i] + m;
    }
    return 0;
}

// instances lo .. lo+n-1 of the call become instances 0 .. n-1 of the batch (n <= capacity)
static int batch_upload_range(obca_batch *bt, const ParkIn &in, int lo, int n) {
    const int N = bt->N, N1 = N + 1;
    if (n < 1 || n > bt->cap) { bt->err = "obca_batch_upload: more instances than the batch was created for"; return -1; }
    bt->B = n;
    bt->nOb.assign(n, 0); bt->M.assign(n, 0); bt->obOff.assign(n + 1, 0); bt->rowOff.assign(n + 1, 0);
    int nObMax = 0, MMax = 0; bt->vmax = 0;
    for (int i = 0; i < n; i++) {
        const int g = lo + i;
        for (int j = in.obOff[g]; j < in.obOff[g + 1]; j++) bt->vmax = std::max(bt->vmax, in.vOb[j]);
        bt->nOb[i] = in.obOff[g + 1] - in.obOff[g]; bt->M[i] = in.rowOff[g + 1] - in.rowOff[g];
        bt->obOff[i] = in.obOff[g]; bt->rowOff[i] = in.rowOff[g];
        nObMax = std::max(nObMax, bt->nOb[i]); MMax = std::max(MMax, bt->M[i]);
    }
    bt->obOff[n] = in.obOff[lo + n]; bt->rowOff[n] = in.rowOff[lo + n];
    bt->rowLen.assign((size_t)(bt->rowOff[n] - bt->rowOff[0]), 1.0);
    use_device(bt->device);
    const size_t B = bt->cap;
    if (!bt->uploaded || nObMax > bt->nObMax || MMax > bt->MMax) {       // (a cached batch keeps the largest shape it has seen)
        // (re)allocation: the batch counts as empty until every buffer exists -- a failed hipMalloc must leave a state the next call can recover from,
        // not a batch that still claims to be uploaded with null device pointers
        free_dev(bt);
        bt->uploaded = 0; bt->bytes = 0;
        bt->nObMax = std::max(nObMax, bt->nObMax); bt->MMax = std::max(MMax, bt->MMax);
        Lay lmax; make_layout(N, bt->nObMax, bt->MMax, lmax);
        bt->zlen = lmax.len;
        DevBufs &d = bt->d;
        d.s_prob = OB_HDR + 3 * (size_t)N1; d.s_z = lmax.len; d.s_as = (size_t)N1 * OB_AS; d.s_rs = (size_t)N1 * OB_RS;
        d.s_oc = (size_t)N1 * bt->nObMax * OB_OC; d.o_csoc = (size_t)lmax.zxL; d.s_csoc = d.o_csoc + (size_t)(lmax.zxL - lmax.pi);
        size_t tot = 0;
#define ALLOC(ptr, cnt) do { size_t by_ = (size_t)(cnt) * sizeof(double); hipError_t e_ = dev_alloc((void **)&(ptr), by_, bt->stream); if (e_ != hipSuccess) { bt->err = std::string("hipMalloc(" #ptr "): ") + hipGetErrorString(e_); free_dev(bt); return -2; } tot += by_; } while (0)
        ALLOC(d.prob, B * d.s_prob); ALLOC(d.z0, B * d.s_z); ALLOC(d.z, B * d.s_z); ALLOC(d.zn, B * d.s_z); ALLOC(d.d, B * d.s_z);
        ALLOC(d.as, B * d.s_as); ALLOC(d.rs, B * d.s_rs);      // (d.oc stays null: the condensed obstacle sums of the parking kernel live in LDS since round 4)
        ALLOC(d.info, B * 8); ALLOC(d.dws, B * N1 * bt->nObMax); ALLOC(d.prof, B * 16);
        ALLOC(d.slice, B * SL_SIZE);
        ALLOC(d.csoc, B * d.s_csoc);      // (rows + direction of a second-order correction: with the other buffers since round 6 -- allocated by the first solve that asked for corrections, it put a 60 MB hipMalloc into that solve: profiles/r06_first_solve_costs.txt)
        bt->dcap_stage = B * (size_t)lmax.nprimal;                       // nprimal >= the output prefix
        ALLOC(bt->stage, bt->dcap_stage);
        if (dev_alloc((void **)&d.order, (B + 1) * sizeof(int), bt->stream) != hipSuccess) { bt->err = "hipMalloc(order) failed"; free_dev(bt); return -2; }
        tot += (B + 1) * sizeof(int);
#undef ALLOC
        bt->bytes = (long long)tot;
    }
    const DevBufs &d = bt->d;
    Lay lmax; make_layout(N, bt->nObMax, bt->MMax, lmax);
    const bool duals = in.lWS && in.nWS;
    // only what the caller provides travels over PCIe: x, u, t (the same offsets in every instance's layout) and, with a dual warm start, lam and mu;
    // the scatter kernel clears the rest of each row
    const size_t W = duals ? (size_t)lmax.sl : (size_t)lmax.lam;
    if (pinned_reserve(bt->err, &bt->h_prob, &bt->hcap_prob, B * d.s_prob) || pinned_reserve(bt->err, &bt->h_zin, &bt->hcap_zin, B * W)) return -2;
    const double *ego = in.ego, *XYb = in.XYb;
    const double W_ev = ego[1] + ego[3], L_ev = ego[0] + ego[2];     /* ParkingSignedDist.jl:182-188 */
    for (int i = 0; i < n; i++) {
        const int g = lo + i;
        double *p = bt->h_prob + (size_t)i * d.s_prob;
        memset(p, 0, sizeof(double) * OB_HDR);
        const int no = bt->nOb[i], m = bt->M[i];
        p[PH_TS] = in.Ts[g]; p[PH_L] = in.L; p[PH_DIST] = bt->dist ? 1.0 : 0.0;
        p[PH_G] = L_ev / 2; p[PH_G + 1] = W_ev / 2; p[PH_G + 2] = L_ev / 2; p[PH_G + 3] = W_ev / 2;
        p[PH_OFF] = (ego[0] + ego[2]) / 2 - ego[2];
        p[PH_XL] = XYb[0]; p[PH_XL + 1] = XYb[2]; p[PH_XL + 2] = -1e300; p[PH_XL + 3] = -1.0;     /* :104-106 */
        p[PH_XU] = XYb[1]; p[PH_XU + 1] = XYb[3]; p[PH_XU + 2] = 1e300; p[PH_XU + 3] = 2.0;
        for (int q = 0; q < 4; q++) { p[PH_X0 + q] = in.x0 ? in.x0[4 * (size_t)g + q] : 0.0; p[PH_XF + q] = in.xF ? in.xF[4 * (size_t)g + q] : 0.0; }
        p[PH_FIX] = in.fixTime ? 1 : 0; p[PH_NOB] = no; p[PH_M] = m;
        int ro = 0;
        for (int j = 0; j < no; j++) { const int v = in.vOb[bt->obOff[i] + j]; p[PH_VOB + j] = v; p[PH_ROFF + j] = ro; ro += v; }
        p[PH_ROFF + no] = ro;
        const size_t r0 = bt->rowOff[i];
        // The solve runs on unit-length half-space rows a_r / |a_r|, b_r / |a_r| (the
        // same obstacle; lambda_r scales with |a_r|, A'lam and b'lam do not change) and
        // hands lambda back in the caller's scaling.  obstHrep.jl:57-86 leaves the rows of a sloped edge unnormalised ([-s 1]: |a| up to 1e3 for a steep edge);
        // IPOPT's default gradient-based NLP scaling stands between such rows and the
        // reference's solves.  Without either 1.2-1.9 % of the config-5 instances -- all of
        // them with a row of |a| > 100 -- failed and the iteration counts had a tail up
        // to 400; with unit rows all solve in at most 80.  The
        // reference's own scenarios have rows of length 1: nothing changes for them, bit for bit.
        double *rl = bt->rowLen.data() + (r0 - (size_t)bt->rowOff[0]);
        for (int r = 0; r < m; r++) {
            const double a1 = in.A[2 * (r0 + r)], a2 = in.A[2 * (r0 + r) + 1]; double nr = hypot(a1, a2);
            if (!(nr > 0)) nr = 1.0;
            rl[r] = nr; p[PH_A + 2 * r] = a1 / nr; p[PH_A + 2 * r + 1] = a2 / nr; p[PH_B + r] = in.b[r0 + r] / nr;
        }
        memcpy(p + OB_HDR, in.rx + (size_t)g * N1, sizeof(double) * N1);
        memcpy(p + OB_HDR + N1, in.ry + (size_t)g * N1, sizeof(double) * N1);
        memcpy(p + OB_HDR + 2 * N1, in.ryaw + (size_t)g * N1, sizeof(double) * N1);
        Lay l; make_layout(N, no, m, l);
        double *z = bt->h_zin + (size_t)i * W;
        if (in.xWS) memcpy(z + l.x, in.xWS + (size_t)g * 4 * N1, sizeof(double) * 4 * N1); else memset(z + l.x, 0, sizeof(double) * 4 * N1);
        if (in.uWS) memcpy(z + l.u, in.uWS + (size_t)g * 2 * N, sizeof(double) * 2 * N); else memset(z + l.u, 0, sizeof(double) * 2 * N);
        z[l.t] = 1.0;                                                 /* ParkingSignedDist.jl:214 */
        if (duals) {
            // caller's row scaling -> unit rows
            for (int k = 0; k < N1; k++) for (int r = 0; r < m; r++) z[l.lam + k * m + r] = in.lWS[r0 * N1 + (size_t)k * m + r] * rl[r];
            memcpy(z + l.mu, in.nWS + (size_t)bt->obOff[i] * 4 * N1, sizeof(double) * 4 * no * N1);
            if ((size_t)l.sl < W) memset(z + l.sl, 0, sizeof(double) * (W - l.sl));      // a smaller instance's layout ends before the widest one's
        }
    }
    bt->have_duals = duals ? 1 : 0; bt->fixTime = in.fixTime ? 1 : 0;
    HIPCHK(bt, hipMemcpyAsync(d.prob, bt->h_prob, (size_t)n * d.s_prob * sizeof(double), hipMemcpyHostToDevice, bt->stream));
    HIPCHK(bt, hipMemcpyAsync(bt->stage, bt->h_zin, (size_t)n * W * sizeof(double), hipMemcpyHostToDevice, bt->stream));
    hipLaunchKernelGGL(obca_scatter_rows_kernel, dim3(n, (unsigned)((d.s_z + 1023) / 1024)), dim3(256), 0, bt->stream, d.z0, d.s_z, (const double *)bt->stage, W, d.s_z);
    HIPCHK(bt, hipGetLastError());
    bt->uploaded = 1;
    return 0;
}

static int launch_dualws(obca_batch *bt, double *zdst) {
    long long tot = (long long)bt->B * (bt->N + 1) * bt->nObMax;
    int blocks = (int)((tot + 255) / 256);
    // (the sub-problem is sized by the template: the reference's scenarios have <= 2 rows per obstacle)
    if (bt->vmax <= 2) hipLaunchKernelGGL(obca_dualws_kernel<2>, dim3(blocks), dim3(256), 0, bt->stream, bt->B, bt->N, bt->nObMax, bt->d, zdst, bt->d.s_z);
    else if (bt->vmax <= OB_VMID) hipLaunchKernelGGL(obca_dualws_kernel<OB_VMID>, dim3(blocks), dim3(256), 0, bt->stream, bt->B, bt->N, bt->nObMax, bt->d, zdst, bt->d.s_z);
    else hipLaunchKernelGGL(obca_dualws_kernel<OB_VMAX>, dim3(blocks), dim3(256), 0, bt->stream, bt->B, bt->N, bt->nObMax, bt->d, zdst, bt->d.s_z);
    HIPCHK(bt, hipGetLastError());
    return 0;
}

// asynchronous on the batch's stream: warm start -> (DualMultWS) -> interior point; dualws_only: stop after DualMultWS
static int batch_solve(obca_batch *bt, const obca_opts *opts, int dualws_only) {
    if (!bt->uploaded) { bt->err = "obca_batch_solve: nothing uploaded"; return -1; }
    if (!dualws_only && bt->N < 2) { bt->err = "obca_batch_solve: the NLP needs a horizon N>=2"; return -1; }
    obca_opts o; if (opts) o = *opts; else obca_default_opts(&o);
    Opts ko; memcpy(&ko, &o, sizeof ko);
    use_device(bt->device);
    if (o.max_soc < 0 || o.max_soc > 16) { bt->err = "opts.max_soc must be in 0 .. 16"; return -1; }
    if (o.restoration < 0 || o.restoration > 2) { bt->err = "opts.restoration must be 0, 1 or 2"; return -1; }
    // IPOPT's objective scaling is 1 on the parking NLP only at the reference's own start
    // (gradient = the slack penalty 1e2); a caller's start with a larger gradient would
    // be scaled by IPOPT and is not by these kernels: the switch is refused here rather
    // than silently ignored (the quadcopter entry points refuse recalc_y the same way)
    if (o.obj_scaling != 0) { bt->err = "opts.obj_scaling is carried by the quadcopter kernel only (obca_quadcopter_reference_opts); the parking entry points take 0"; return -1; }
    if (o.max_soc > 0 && !bt->d.csoc) {      // the second-order correction keeps its right-hand-side rows per instance
        hipError_t e_ = dev_alloc((void **)&bt->d.csoc, (size_t)bt->cap * bt->d.s_csoc * sizeof(double), bt->stream);
        if (e_ != hipSuccess) { bt->d.csoc = nullptr; bt->err = std::string("hipMalloc(csoc): ") + hipGetErrorString(e_); return -2; }
        bt->bytes += (long long)((size_t)bt->cap * bt->d.s_csoc * sizeof(double));
    }
    const DevBufs &d = bt->d;
    // a solve starts from the uploaded iterate: reset z <- z0 (device-to-device), DualMultWS writes its multipliers into z
    HIPCHK(bt, hipMemcpyAsync(d.z, d.z0, (size_t)bt->B * d.s_z * sizeof(double), hipMemcpyDeviceToDevice, bt->stream));
    HIPCHK(bt, hipEventRecord(bt->e0, bt->stream));
    if (!bt->have_duals || dualws_only) { int rc = launch_dualws(bt, d.z); if (rc) return rc; }
    HIPCHK(bt, hipEventRecord(bt->e1, bt->stream));
    if (dualws_only) { HIPCHK(bt, hipEventRecord(bt->e2, bt->stream)); return 0; }
    // Two-launch schedule.  The kernel keeps a fixed number of instances resident; a larger batch is dispatched in blockIdx
    // order as workgroups retire, so an instance that needs three times the median number of passes and happens to sit late in the batch
    // would start late and finish alone.  Instead every instance first runs a short slice (OBCA_SLICE_PASSES factorisation passes, default
    // 6), the parked solves are ranked by what the slice revealed, and a second launch finishes them hardest-first.  No work is repeated
    // and every instance walks through the same iterates as in a single launch.  OBCA_SLICE_PASSES=0 turns it off; OBCA_SLICE_ONLY=1
    // (diagnostic) stops after the first slice.
    int budget = 6;
    if (const char *e = getenv("OBCA_SLICE_PASSES")) budget = atoi(e);
    const bool slice_only = getenv("OBCA_SLICE_ONLY") && atoi(getenv("OBCA_SLICE_ONLY"));
    // forward-sweep trajectory + stage buffers / pair maps, sized for the horizon (obca_solver.h)
#ifdef OBCA_POISON      // the poisoned build adds a guard of OBCA_LDS_GUARD doubles behind the dynamic block (NaNs, written at entry, never again): a read beyond the block finds it
    const size_t dyn_lds = (OB_DYN_LDS_DOUBLES(bt->N) + OBCA_LDS_GUARD) * sizeof(double);
#else
    const size_t dyn_lds = OB_DYN_LDS_DOUBLES(bt->N) * sizeof(double);
#endif
    const int slots = parking_resident_per_cu(bt->N) * (bt->ctx->cus > 0 ? bt->ctx->cus : 256);
    // OBCA_SLICE_ALWAYS=1 (tuning knob): the two-launch schedule also for a batch that fits the machine -- with several batches in flight the workgroups of a launch start as
    // slots free up, in blockIdx order, and the ranking lets the long solves of a batch start first
    const bool slice_always = getenv("OBCA_SLICE_ALWAYS") && atoi(getenv("OBCA_SLICE_ALWAYS"));
    bt->sliced = (budget > 0 && (bt->B > slots || slice_only || slice_always)) ? budget : 0;   // 0: single launch, else the slice length
    if (!bt->sliced) {
        hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), dyn_lds, bt->stream, bt->B, bt->N, bt->d, ko, 0, 0, o.max_soc, o.recalc_y != 0, o.lsq_init != 0, o.restoration);
        HIPCHK(bt, hipGetLastError());
    } else {
        hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), dyn_lds, bt->stream, bt->B, bt->N, bt->d, ko, 0, budget, o.max_soc, o.recalc_y != 0, o.lsq_init != 0, o.restoration);
        HIPCHK(bt, hipGetLastError());
        if (!slice_only) {
            hipLaunchKernelGGL(obca_order_kernel, dim3(1), dim3(1024), 0, bt->stream, bt->B, (const double *)d.info, (const double *)d.slice, d.order);
            HIPCHK(bt, hipGetLastError());
            hipLaunchKernelGGL(obca_parking_ipm_kernel, dim3(bt->B), dim3(OB_NT), dyn_lds, bt->stream, bt->B, bt->N, bt->d, ko, 1, 0, o.max_soc, o.recalc_y != 0, o.lsq_init != 0, o.restoration);
            HIPCHK(bt, hipGetLastError());
        }
    }
    HIPCHK(bt, hipEventRecord(bt->e2, bt->stream));
    return 0;
}

// outputs of the batch's instances into the caller's arrays (instance i of the batch = instance lo + i of the call; packed obstacle outputs
// go to the offsets recorded at upload).  Synchronises the batch's stream.
static int batch_download_range(obca_batch *bt, const ParkOut &o, int lo) {
    const int B = bt->B, N = bt->N, N1 = N + 1;
    const DevBufs &d = bt->d;
    use_device(bt->device);
    Lay lmax; make_layout(N, bt->nObMax, bt->MMax, lmax);
    const size_t W = (size_t)lmax.so;                       // outputs are a prefix of the iterate: x, u, t, lam, mu, sl
    const bool want_z = o.xp || o.up || o.ts || o.lp || o.np || o.slp || o.lWS || o.nWS;
    if (pinned_reserve(bt->err, &bt->h_zout, &bt->hcap_zout, (size_t)bt->cap * std::max(W, (size_t)N1 * bt->nObMax)) ||
        pinned_reserve(bt->err, &bt->h_info, &bt->hcap_info, (size_t)bt->cap * 8)) return -2;
    if (want_z) {
        hipLaunchKernelGGL(obca_gather_rows_kernel, dim3(B, (unsigned)((W + 1023) / 1024)), dim3(256), 0, bt->stream, bt->stage, W, (const double *)d.z, d.s_z);
        HIPCHK(bt, hipGetLastError());
        HIPCHK(bt, hipMemcpyAsync(bt->h_zout, bt->stage, (size_t)B * W * sizeof(double), hipMemcpyDeviceToHost, bt->stream));
    }
    HIPCHK(bt, hipMemcpyAsync(bt->h_info, d.info, (size_t)B * 8 * sizeof(double), hipMemcpyDeviceToHost, bt->stream));
    HIPCHK(bt, hipStreamSynchronize(bt->stream));
    for (int i = 0; i < B; i++) {
        const size_t g = (size_t)lo + i;
        Lay l; make_layout(N, bt->nOb[i], bt->M[i], l);
        const double *z = bt->h_zout + (size_t)i * W;
        if (o.xp) memcpy(o.xp + g * 4 * N1, z + l.x, sizeof(double) * 4 * N1);
        if (o.up) memcpy(o.up + g * 2 * N, z + l.u, sizeof(double) * 2 * N);
        if (o.ts) for (int k = 0; k < N1; k++) o.ts[g * N1 + k] = bt->fixTime ? 1.0 : z[l.t];   /* ParkingSignedDist.jl:304-308 */
        const double *rl = bt->rowLen.data() + (bt->rowOff[i] - bt->rowOff[0]); const int m = bt->M[i];      // lambda back in the caller's row scaling
        if (o.lp) for (int k = 0; k < N1; k++) for (int r = 0; r < m; r++) o.lp[(size_t)bt->rowOff[i] * N1 + (size_t)k * m + r] = z[l.lam + k * m + r] / rl[r];
        if (o.np) memcpy(o.np + (size_t)bt->obOff[i] * 4 * N1, z + l.mu, sizeof(double) * 4 * bt->nOb[i] * N1);
        if (o.slp) memcpy(o.slp + (size_t)bt->obOff[i] * N1, z + l.sl, sizeof(double) * bt->nOb[i] * N1);
        if (o.lWS) for (int k = 0; k < N1; k++) for (int r = 0; r < m; r++) o.lWS[(size_t)bt->rowOff[i] * N1 + (size_t)k * m + r] = z[l.lam + k * m + r] / rl[r];
        if (o.nWS) memcpy(o.nWS + (size_t)bt->obOff[i] * 4 * N1, z + l.mu, sizeof(double) * 4 * bt->nOb[i] * N1);
        if (o.exitflag) o.exitflag[g] = (int)bt->h_info[(size_t)i * 8 + 7];
        if (o.info) memcpy(o.info + g * 8, bt->h_info + (size_t)i * 8, sizeof(double) * 8);
    }
    if (o.dd) {   // DualMultWS distances: nOb_i x (N+1) packed
        HIPCHK(bt, hipMemcpyAsync(bt->h_zout, d.dws, (size_t)B * N1 * bt->nObMax * sizeof(double), hipMemcpyDeviceToHost, bt->stream));
        HIPCHK(bt, hipStreamSynchronize(bt->stream));
        for (int i = 0; i < B; i++) for (int k = 0; k < N1; k++) for (int j = 0; j < bt->nOb[i]; j++)
            o.dd[(size_t)bt->obOff[i] * N1 + (size_t)k * bt->nOb[i] + j] = bt->h_zout[(size_t)i * N1 * bt->nObMax + (size_t)k * bt->nObMax + j];
    }
    return 0;
}

// ---- chunked execution of a host-pointer call over the slots of the context (work queue)
template <typename F>
static int run_chunks(obca_ctx *ctx, int B, int chunk, F &&fn /* int(Slot &, int lo, int n, std::string &err) */) {
    const int nchunks = (B + chunk - 1) / chunk;
    const int nw = std::min<int>(nchunks, (int)ctx->slots.size());
    std::atomic<int> next(0);
    // OBCA_CHUNK_PERM = s (diagnostic, tests/test_gpu_determinism.py): the t-th ticket of
    // the queue is chunk (t + s) mod nchunks for even s, the chunks in descending order
    // from there for odd s -- which lane (stream, cached batch, staging buffers) solves which chunk must not matter to a single bit of the results
    int perm = 0; if (const char *e = getenv("OBCA_CHUNK_PERM")) perm = atoi(e) > 0 ? atoi(e) : 0;
    std::vector<int> rcs(nw, 0); std::vector<std::string> errs(nw);
    for (int w = 0; w < nw; w++) {      // lanes get their stream on first use (see obca_create_multi)
        Slot &s = ctx->slots[w];
        if (!s.stream && (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess)) { ctx->err = "hipStreamCreate failed"; return -2; }
    }
    auto work = [&](int w) {
        Slot &s = ctx->slots[w];
        use_device(s.device);
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= nchunks) break;
            const int c = !perm ? t : ((perm & 1) ? ((nchunks - 1 - t) + perm) % nchunks : (t + perm) % nchunks);
            const int lo = c * chunk, n = std::min(chunk, B - lo);
            const int rc = fn(s, lo, n, errs[w]);
            if (rc) { rcs[w] = rc; next.store(nchunks); break; }
        }
    };
    if (nw <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int w = 1; w < nw; w++) th.emplace_back(work, w);
        work(0);
        for (auto &t : th) t.join();
    }
    use_device(ctx->device);
    for (int w = 0; w < nw; w++) if (rcs[w]) { ctx->err = errs[w]; return rcs[w]; }
    return 0;
}
static int pick_chunk(const obca_ctx *ctx, int B, int resident_per_cu) {
    // twice the instances resident on one GPU per chunk: measured best on config-2 batches (PCIe-inclusive, 4 lanes: 92-98 k solves/s against 75 k
    // with 256-instance chunks) -- what counts is the number of instances in flight (lanes x chunk), which must cover the heavy tail of the
    // solve times several times over; chunks beyond the resident capacity use the two-launch schedule of batch_solve
    int chunk = (resident_per_cu >= 4 ? 1 : 2) * resident_per_cu * (ctx->cus > 0 ? ctx->cus : 256);      // 1024 on a 256-CU part for both paths
    if (const char *e = getenv("OBCA_CHUNK")) { const int v = atoi(e); if (v > 0) chunk = v; }
    // small calls: still give every slot something to do once there is enough work to hide a transfer behind
    const int ns = (int)ctx->slots.size();
    if (B < chunk * ns && B >= 64 * ns) chunk = (B + ns - 1) / ns;
    return std::max(1, std::min(chunk, B));
}
static int slot_parking_batch(obca_ctx *ctx, Slot &s, int n, int N, int dist, std::string &err) {
    if (s.pb && (s.pb->cap < n || s.pb->N != N)) { obca_batch_destroy(s.pb); s.pb = nullptr; }
    if (!s.pb) { int rc = batch_create_on(ctx, s.device, s.stream, n, N, &s.pb, err); if (rc) return rc; }
    if (s.pb->dist != (dist ? 1 : 0)) { s.pb->dist = dist ? 1 : 0; }
    return 0;
}

static int parking_call(obca_ctx *ctx, int dist, int dualws_only, int B, int N, ParkIn &in, const obca_opts *opts, const ParkOut &out) {
    if (!ctx) return -1;
    if (B < 1 || N < 0 || N > OBCA_NMAX) { ctx->err = "need B>=1, 0<=N<=OBCA_NMAX"; return -1; }
    if (!in.Ts || !in.ego || !in.XYb || !in.nOb || !in.vOb || !in.A || !in.b || !in.rx || !in.ry || !in.ryaw) { ctx->err = "NULL argument"; return -1; }
    if (!dualws_only && N < 2) { ctx->err = "the NLP needs a horizon N>=2"; return -1; }
    if (int rc = park_prefix(ctx->err, B, in.nOb, in.vOb, in)) return rc;
    const int chunk = pick_chunk(ctx, B, 4);
    return run_chunks(ctx, B, chunk, [&](Slot &s, int lo, int n, std::string &err) -> int {
        int rc = slot_parking_batch(ctx, s, std::min(chunk, B), N, dist, err);
        if (rc) return rc;
        obca_batch *bt = s.pb;
        rc = batch_upload_range(bt, in, lo, n);
        if (!rc) rc = batch_solve(bt, opts, dualws_only);
        if (!rc) rc = batch_download_range(bt, out, lo);
        if (rc) err = bt->err;
        return rc;
    });
}

extern "C" {

int obca_batch_create(obca_ctx *ctx, int B, int N, obca_batch **out) {
    if (!ctx || !out) return -1;
    return batch_create_on(ctx, ctx->device, ctx->stream, B, N, out, ctx->err);
}
int obca_batch_destroy(obca_batch *bt) {
    if (!bt) return -1;
    use_device(bt->device);
    free_dev(bt); (void)hipEventDestroy(bt->e0); (void)hipEventDestroy(bt->e1); (void)hipEventDestroy(bt->e2);
    double **hs[] = {&bt->h_prob, &bt->h_zin, &bt->h_zout, &bt->h_info};
    for (auto p : hs) if (*p) (void)hipHostFree(*p);
    delete bt; return 0;
}
#ifdef OBCA_PROFILE      /* the per-phase clocks exist in the profiling build only (libobca_hip_prof.so, tools/phase_profile.py): not an entry point of the product */
int obca_batch_debug_phase_cycles(obca_batch *bt, double *out /* B x 16 */) {   
    if (!bt || !out) return -1;
    use_device(bt->device);
    if (hipStreamSynchronize(bt->stream) != hipSuccess ||
        hipMemcpy(out, bt->d.prof, (size_t)bt->B * 16 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { bt->ctx->err = "obca_batch_debug_phase_cycles: copy failed";
        return -2; }
    return 0;
}
#endif
int obca_batch_set_formulation(obca_batch *bt, int dist) { if (!bt) return -1; bt->dist = dist ? 1 : 0; return 0; }   /* before obca_batch_upload */
int obca_batch_scratch_bytes(const obca_batch *bt, long long *bytes) { if (!bt || !bytes) return -1; *bytes = bt->bytes; return 0; }

int obca_batch_upload(obca_batch *bt, const double *Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                      const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A, const double *b,
                      const double *rx, const double *ry, const double *ryaw, const double *xWS, const double *uWS,
                      const double *lWS, const double *nWS) {
    if (!bt) return -1;
    if (!Ts || !ego || !XYb || !nOb || !vOb || !A || !b || !rx || !ry || !ryaw) { bt->ctx->err = "obca_batch_upload: NULL argument"; return -1; }
    ParkIn in = {Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, {}, {}};
    if (int rc = park_prefix(bt->err, bt->cap, nOb, vOb, in)) { bt->ctx->err = "obca_batch_upload: " + bt->err; return rc; }
    int rc = batch_upload_range(bt, in, 0, bt->cap);
    if (!rc && hipStreamSynchronize(bt->stream) != hipSuccess) { bt->err = "obca_batch_upload: stream sync failed"; rc = -2; }
    return fin(bt, rc);
}
int obca_batch_solve(obca_batch *bt, const obca_opts *opts) { if (!bt) return -1; return fin(bt, batch_solve(bt, opts, 0)); }
int obca_batch_shift_warm_start(obca_batch *bt, int shift, const double *x0_new) {
    if (!bt) return -1;
    obca_ctx *ctx = bt->ctx;
    if (!bt->uploaded) { ctx->err = "obca_batch_shift_warm_start: nothing uploaded"; return -1; }
    if (shift < 0 || shift > bt->N) { ctx->err = "obca_batch_shift_warm_start: shift out of range 0..N"; return -1; }
    use_device(bt->device);
    double *dx0 = nullptr;
    if (x0_new) {   // staged through the (idle) device staging buffer of the batch: nothing to free on the error paths
        if (pinned_reserve(bt->err, &bt->h_info, &bt->hcap_info, (size_t)bt->cap * 8)) return fin(bt, -2);
        memcpy(bt->h_info, x0_new, (size_t)bt->B * 4 * sizeof(double));
        dx0 = bt->stage;
        if (hipMemcpyAsync(dx0, bt->h_info, (size_t)bt->B * 4 * sizeof(double), hipMemcpyHostToDevice, bt->stream) != hipSuccess) { ctx->err = "obca_batch_shift_warm_start: H2D failed"; return -2; }
    }
    hipLaunchKernelGGL(obca_shift_kernel, dim3(bt->B), dim3(128), 0, bt->stream, bt->B, bt->N, shift, bt->d, (const double *)dx0);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(bt->stream) != hipSuccess) { ctx->err = "obca_batch_shift_warm_start: kernel failed"; return -2; }
    bt->have_duals = 1;                                 // the shifted multipliers are the dual warm start: DualMultWS is skipped
    return 0;
}
int obca_batch_sync(obca_batch *bt) {
    if (!bt) return -1;
    use_device(bt->device);
    if (hipStreamSynchronize(bt->stream) != hipSuccess) { bt->ctx->err = "obca_batch_sync: hipStreamSynchronize failed"; return -2; }
    return 0;
}
int obca_batch_last_schedule(const obca_batch *bt, int *ipm_launches, int *slice_passes) {
    if (!bt) return -1;
    if (ipm_launches) *ipm_launches = bt->sliced ? 2 : 1;
    if (slice_passes) *slice_passes = bt->sliced;
    return 0;
}
int obca_batch_kernel_ms(obca_batch *bt, float *ipm_ms, float *dualws_ms) {
    if (!bt) return -1;
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, bt->e0, bt->e1) != hipSuccess || hipEventElapsedTime(&b, bt->e1, bt->e2) != hipSuccess) { bt->ctx->err = "obca_batch_kernel_ms: events not ready"; return -2; }
    if (dualws_ms) *dualws_ms = a; if (ipm_ms) *ipm_ms = b;
    return 0;
}
int obca_batch_download(obca_batch *bt, double *xp, double *up, double *ts, int *exitflag, double *lp, double *np, double *slp, double *info) {
    if (!bt) return -1;
    if (!bt->uploaded) { bt->ctx->err = "obca_batch_download: nothing uploaded"; return -1; }
    ParkOut o = {xp, up, ts, exitflag, lp, np, slp, info, nullptr, nullptr, nullptr};
    return fin(bt, batch_download_range(bt, o, 0));
}

int obca_parking_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double L, const double ego[4], const double XYb[4],
                                   int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                                   const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                                   const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp,
                                   double *up, double *timeScale, int *exitflag, double *lp, double *np, double *slp, double *info) {
    if (!ctx) return -1;
    if (!x0 || !xF || !xWS || !uWS) { ctx->err = "obca_parking_signed_dist_batch: NULL argument"; return -1; }
    ParkIn in = {Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, {}, {}};
    ParkOut o = {xp, up, timeScale, exitflag, lp, np, slp, info, nullptr, nullptr, nullptr};
    return parking_call(ctx, 0, 0, B, N, in, opts, o);
}
int obca_parking_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double L, const double ego[4], const double XYb[4],
                            int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                            const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                            const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp,
                            double *up, double *timeScale, int *exitflag, double *lp, double *np, double *info) {
    if (!ctx) return -1;
    if (!x0 || !xF || !xWS || !uWS) { ctx->err = "obca_parking_dist_batch: NULL argument"; return -1; }
    ParkIn in = {Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, {}, {}};
    ParkOut o = {xp, up, timeScale, exitflag, lp, np, nullptr, info, nullptr, nullptr, nullptr};
    return parking_call(ctx, 1, 0, B, N, in, opts, o);
}

int obca_dualmult_ws_batch(obca_ctx *ctx, int B, int N, const double ego[4], const int *nOb, const int *vOb, const double *A,
                           const double *b, const double *rx, const double *ry, const double *ryaw, double *lWS, double *nWS, double *dd) {
    if (!ctx) return -1;
    if (!lWS || !nWS) { ctx->err = "obca_dualmult_ws_batch: NULL output"; return -1; }
    if (B < 1) { ctx->err = "obca_dualmult_ws_batch: need B>=1"; return -1; }
    std::vector<double> Ts(B, 1.0); const double XYb[4] = {0, 0, 0, 0};
    ParkIn in = {Ts.data(), 1.0, ego, XYb, 0, nullptr, nullptr, nOb, vOb, A, b, rx, ry, ryaw, nullptr, nullptr, nullptr, nullptr, {}, {}};
    ParkOut o = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lWS, nWS, dd};
    return parking_call(ctx, 0, 1, B, N, in, nullptr, o);
}

}  // extern "C"

/* ---------------------------------------------------------------- quadcopter path */
struct obca_quad_batch {
    obca_ctx *ctx; int device; hipStream_t stream; std::string err;
    int B, cap, N, uploaded;
    QDevBufs d; double *stage; hipEvent_t e0, e1; long long bytes;
    double *h_prob, *h_z, *h_info; size_t hcap_prob, hcap_z, hcap_info;       // pinned host staging
};
#define QCHK(bt, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (bt)->err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } } while (0)
static inline int qfin(obca_quad_batch *bt, int rc) { if (rc) bt->ctx->err = bt->err; return rc; }
static void qfree_dev(obca_quad_batch *bt) {
    double **ps[] = {&bt->d.prob, &bt->d.z, &bt->d.d, &bt->d.as, &bt->d.rs, &bt->d.oc, &bt->d.info, &bt->d.prof};
    for (auto p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (bt->stage) (void)hipFree(bt->stage); bt->stage = nullptr;
}
static int quad_batch_create_on(obca_ctx *ctx, int device, hipStream_t stream, int B, int N, obca_quad_batch **out, std::string &err) {
    if (B < 1 || N < 2 || N > OBCA_QUAD_NMAX) { err = "obca_quad_batch_create: need B>=1, 2<=N<=OBCA_QUAD_NMAX"; return -1; }
    obca_quad_batch *bt = new obca_quad_batch();
    bt->ctx = ctx; bt->device = device; bt->stream = stream; bt->B = B; bt->cap = B; bt->N = N; bt->uploaded = 0; bt->bytes = 0;
    memset(&bt->d, 0, sizeof bt->d); bt->stage = nullptr;
    bt->h_prob = bt->h_z = bt->h_info = nullptr; bt->hcap_prob = bt->hcap_z = bt->hcap_info = 0;
    use_device(device);
    quad::QLay l; quad::q_make_layout(N, l);
    QDevBufs &d = bt->d; const size_t N1 = N + 1;
    d.s_prob = QPH_SIZE + QX * N1; d.s_z = l.len; d.s_d = QDIR_DOUBLES(l);
         /* two direction buffers (dv | dy) + the rows of a second-order correction, see QCS */ d.s_as = N1 * QSP; d.s_rs = N1 * QRR; d.s_oc = N1 * QOB * OB_OC;
    size_t tot = 0;
#define ALLOC(ptr, cnt) do { size_t by_ = (size_t)(cnt) * sizeof(double); if (dev_alloc((void **)&(ptr), by_, stream) != hipSuccess) { err = "obca_quad_batch_create: hipMalloc failed"; qfree_dev(bt); delete bt; return -2; } tot += by_; } while (0)
    ALLOC(d.prob, B * d.s_prob); ALLOC(d.z, B * d.s_z); ALLOC(d.d, B * d.s_d); ALLOC(d.as, B * d.s_as); ALLOC(d.rs, B * d.s_rs);
    ALLOC(d.oc, B * d.s_oc); ALLOC(d.info, (size_t)B * 8); ALLOC(d.prof, (size_t)B * 16); ALLOC(bt->stage, (size_t)B * l.so);
#undef ALLOC
    bt->bytes = (long long)tot;
    if (hipEventCreate(&bt->e0) != hipSuccess || hipEventCreate(&bt->e1) != hipSuccess) { err = "hipEventCreate failed"; qfree_dev(bt); delete bt; return -2; }
    *out = bt;
    return 0;
}
struct QuadIn { const double *Ts; double R; const double *x0, *xF, *ob, *xWS, *timeWS; int dual_ws, dist; };
struct QuadOut { double *xp, *up, *ts; int *exitflag; double *lp, *slp, *info; };
static int quad_upload_range(obca_quad_batch *bt, const QuadIn &in, int lo, int n) {
    if (n < 1 || n > bt->cap) { bt->err = "obca_quad_batch_upload: more instances than the batch was created for"; return -1; }
    bt->B = n;
    const int N1 = bt->N + 1; const QDevBufs &d = bt->d;
    if (pinned_reserve(bt->err, &bt->h_prob, &bt->hcap_prob, (size_t)bt->cap * d.s_prob)) return -2;
    for (int i = 0; i < n; i++) {
        const size_t g = (size_t)lo + i;
        double *p = bt->h_prob + (size_t)i * d.s_prob;
        memset(p, 0, sizeof(double) * QPH_SIZE);
        p[QPH_TS] = in.Ts[g]; p[QPH_R] = in.R; p[QPH_TWS] = in.timeWS[g]; p[QPH_DWS] = in.dual_ws ? 1.0 : 0.0; p[QPH_DIST] = in.dist ? 1.0 : 0.0;
        memcpy(p + QPH_X0, in.x0 + QX * g, sizeof(double) * QX); memcpy(p + QPH_XF, in.xF + QX * g, sizeof(double) * QX);
        memcpy(p + QPH_OB, in.ob + (size_t)QOB * QL * g, sizeof(double) * QOB * QL);
        memcpy(p + QPH_SIZE, in.xWS + (size_t)QX * N1 * g, sizeof(double) * QX * N1);
    }
    use_device(bt->device);
    QCHK(bt, hipMemcpyAsync(d.prob, bt->h_prob, (size_t)n * d.s_prob * sizeof(double), hipMemcpyHostToDevice, bt->stream));
    bt->uploaded = 1;
    return 0;
}
static int quad_solve(obca_quad_batch *bt, const obca_opts *opts);
static int quad_download_range(obca_quad_batch *bt, const QuadOut &o, int lo) {
    const int B = bt->B, N = bt->N, N1 = N + 1; const QDevBufs &d = bt->d;
    quad::QLay l; quad::q_make_layout(N, l);
    use_device(bt->device);
    const size_t W = (size_t)l.so;                          // outputs are a prefix of the iterate: x, u, t, lam, s
    if (pinned_reserve(bt->err, &bt->h_z, &bt->hcap_z, (size_t)bt->cap * W) || pinned_reserve(bt->err, &bt->h_info, &bt->hcap_info, (size_t)bt->cap * 8)) return -2;
    hipLaunchKernelGGL(obca_gather_rows_kernel, dim3(B, (unsigned)((W + 1023) / 1024)), dim3(256), 0, bt->stream, bt->stage, W, (const double *)d.z, d.s_z);
    QCHK(bt, hipGetLastError());
    QCHK(bt, hipMemcpyAsync(bt->h_z, bt->stage, (size_t)B * W * sizeof(double), hipMemcpyDeviceToHost, bt->stream));
    QCHK(bt, hipMemcpyAsync(bt->h_info, d.info, (size_t)B * 8 * sizeof(double), hipMemcpyDeviceToHost, bt->stream));
    QCHK(bt, hipStreamSynchronize(bt->stream));
    for (int i = 0; i < B; i++) {
        const size_t g = (size_t)lo + i;
        const double *z = bt->h_z + (size_t)i * W;
        if (o.xp) memcpy(o.xp + g * QX * N1, z + l.x, sizeof(double) * QX * N1);
        if (o.up) memcpy(o.up + g * QU * N, z + l.u, sizeof(double) * QU * N);
        if (o.ts) for (int k = 0; k < N1; k++) o.ts[g * N1 + k] = z[l.t];                                 /* QuadcopterSignedDist.jl:293 */
        if (o.lp) memcpy(o.lp + g * QL * QOB * N1, z + l.lam, sizeof(double) * QL * QOB * N1);            /* [l1;..;l5] stacked, :295 */
        if (o.slp) memcpy(o.slp + g * QOB * N1, z + l.s, sizeof(double) * QOB * N1);
        if (o.exitflag) o.exitflag[g] = (int)bt->h_info[(size_t)i * 8 + 7];
        if (o.info) memcpy(o.info + g * 8, bt->h_info + (size_t)i * 8, sizeof(double) * 8);
    }
    return 0;
}

extern "C" {

int obca_quadcopter_default_opts(obca_opts *o) {
    if (obca_default_opts(o)) return -1;
    o->max_iter = 3000; o->dw_min = 1e-10;            /* QuadcopterSignedDist.jl:28-31: no max_iter (IPOPT default), min_hessian_perturbation 1e-10 */
    return 0;
}
int obca_quadcopter_reference_opts(obca_opts *o) {
    if (obca_quadcopter_default_opts(o)) return -1;
    o->max_soc = 4;                                    /* IPOPT default max_soc; recalc_y stays 0: QuadcopterSignedDist.jl:29 sets recalc_y = "no" */
    o->lsq_init = 1;                                   /* IPOPT default: least-squares initial multipliers, constr_mult_init_max = 1e3 */
    o->obj_scaling = 1;                                /* IPOPT default: gradient-based scaling; on this NLP it is the objective factor 100 / 2 100 */
    return 0;
}
int obca_quad_batch_create(obca_ctx *ctx, int B, int N, obca_quad_batch **out) {
    if (!ctx || !out) return -1;
    return quad_batch_create_on(ctx, ctx->device, ctx->stream, B, N, out, ctx->err);
}
int obca_quad_batch_destroy(obca_quad_batch *bt) {
    if (!bt) return -1;
    use_device(bt->device); qfree_dev(bt); (void)hipEventDestroy(bt->e0); (void)hipEventDestroy(bt->e1);
    double **hs[] = {&bt->h_prob, &bt->h_z, &bt->h_info};
    for (auto p : hs) if (*p) (void)hipHostFree(*p);
    delete bt; return 0;
}
#ifdef OBCA_PROFILE
int obca_quad_batch_debug_phase_cycles(obca_quad_batch *bt, double *out /* B x 16 */) {   
    if (!bt || !out) return -1;
    use_device(bt->device);
    if (hipStreamSynchronize(bt->stream) != hipSuccess ||
        hipMemcpy(out, bt->d.prof, (size_t)bt->B * 16 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { bt->ctx->err = "obca_quad_batch_debug_phase_cycles: copy failed";
        return -2; }
    return 0;
}
#endif
int obca_quad_batch_scratch_bytes(const obca_quad_batch *bt, long long *bytes) { if (!bt || !bytes) return -1; *bytes = bt->bytes; return 0; }
int obca_quad_batch_upload(obca_quad_batch *bt, const double *Ts, double R, const double *x0, const double *xF, const double *ob,
                           const double *xWS, const double *timeWS, int dual_ws, int dist) {
    if (!bt) return -1;
    if (!Ts || !x0 || !xF || !ob || !xWS || !timeWS) { bt->ctx->err = "obca_quad_batch_upload: NULL argument"; return -1; }
    QuadIn in = {Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, dist};
    int rc = quad_upload_range(bt, in, 0, bt->cap);
    if (!rc && hipStreamSynchronize(bt->stream) != hipSuccess) { bt->err = "obca_quad_batch_upload: stream sync failed"; rc = -2; }
    return qfin(bt, rc);
}
}  // extern "C"
static int quad_solve(obca_quad_batch *bt, const obca_opts *opts) {
    if (!bt->uploaded) { bt->err = "obca_quad_batch_solve: nothing uploaded"; return -1; }
    obca_opts o; if (opts) o = *opts; else obca_quadcopter_default_opts(&o);
    if (o.max_soc < 0 || o.max_soc > 16) { bt->err = "opts.max_soc must be in 0 .. 16"; return -1; }
    if (o.restoration < 0 || o.restoration > 2) { bt->err = "opts.restoration must be 0, 1 or 2"; return -1; }
    if (o.recalc_y != 0) { bt->err = "quadcopter solve: recalc_y is a switch of the parking kernels only (the reference's quadcopter call sets recalc_y = \"no\", QuadcopterSignedDist.jl:29); the quadcopter kernel would ignore it -- refusing instead"; return -1; }
    Opts ko; memcpy(&ko, &o, sizeof ko);
    use_device(bt->device);
    QCHK(bt, hipEventRecord(bt->e0, bt->stream));
    hipLaunchKernelGGL(obca_quad_ipm_kernel, dim3(bt->B), dim3(QNT), (size_t)(bt->N + 2) * (QS + QU) * sizeof(double), bt->stream, bt->B, bt->N, bt->d, ko, o.max_soc, o.lsq_init != 0, o.obj_scaling != 0);
    QCHK(bt, hipGetLastError());
    QCHK(bt, hipEventRecord(bt->e1, bt->stream));
    return 0;
}
static int quadcopter_call(obca_ctx *ctx, int B, int N, const QuadIn &in, const obca_opts *opts, const QuadOut &out) {
    if (!ctx) return -1;
    if (B < 1 || N < 2 || N > OBCA_QUAD_NMAX) { ctx->err = "need B>=1, 2<=N<=OBCA_QUAD_NMAX"; return -1; }
    if (!in.Ts || !in.x0 || !in.xF || !in.ob || !in.xWS || !in.timeWS) { ctx->err = "NULL argument"; return -1; }
    const int chunk = pick_chunk(ctx, B, (QNT == 64 ? 4 : 2) * OBCA_QUAD_WAVES_PER_EU);
    return run_chunks(ctx, B, chunk, [&](Slot &s, int lo, int n, std::string &err) -> int {
        if (s.qb && (s.qb->cap < n || s.qb->N != N)) { obca_quad_batch_destroy(s.qb); s.qb = nullptr; }
        if (!s.qb) { int rc = quad_batch_create_on(ctx, s.device, s.stream, std::min(chunk, B), N, &s.qb, err); if (rc) return rc; }
        obca_quad_batch *bt = s.qb;
        int rc = quad_upload_range(bt, in, lo, n);
        if (!rc) rc = quad_solve(bt, opts);
        if (!rc) rc = quad_download_range(bt, out, lo);
        if (rc) err = bt->err;
        return rc;
    });
}
extern "C" {
int obca_quad_batch_solve(obca_quad_batch *bt, const obca_opts *opts) { if (!bt) return -1; return qfin(bt, quad_solve(bt, opts)); }
int obca_quad_batch_sync(obca_quad_batch *bt) {
    if (!bt) return -1;
    use_device(bt->device);
    if (hipStreamSynchronize(bt->stream) != hipSuccess) { bt->ctx->err = "obca_quad_batch_sync: hipStreamSynchronize failed"; return -2; }
    return 0;
}
int obca_quad_batch_kernel_ms(obca_quad_batch *bt, float *ipm_ms) {
    if (!bt || !ipm_ms) return -1;
    if (hipEventElapsedTime(ipm_ms, bt->e0, bt->e1) != hipSuccess) { bt->ctx->err = "obca_quad_batch_kernel_ms: events not ready"; return -2; }
    return 0;
}
int obca_quad_batch_download(obca_quad_batch *bt, double *xp, double *up, double *ts, int *exitflag, double *lp, double *slp, double *info) {
    if (!bt) return -1;
    QuadOut o = {xp, up, ts, exitflag, lp, slp, info};
    return qfin(bt, quad_download_range(bt, o, 0));
}
int obca_quadcopter_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                                      const double *ob, const double *xWS, const double *uWS, const double *timeWS, int dual_ws,
                                      const obca_opts *opts, double *xp, double *up, double *timeScale, int *exitflag, double *lp,
                                      double *slp, double *info) {
    (void)uWS;                                         /* the reference ignores it too: inputs start at the hover speed, :202 */
    QuadIn in = {Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, 0};
    QuadOut o = {xp, up, timeScale, exitflag, lp, slp, info};
    return quadcopter_call(ctx, B, N, in, opts, o);
}
int obca_quadcopter_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                               const double *ob, const double *xWS, const double *uWS, const double *timeWS, int dual_ws,
                               const obca_opts *opts, double *xp, double *up, double *timeScale, int *exitflag, double *lp, double *info) {
    (void)uWS;                                         /* QuadcopterDist.jl:196 */
    QuadIn in = {Ts, R, x0, xF, ob, xWS, timeWS, dual_ws, 1};
    QuadOut o = {xp, up, timeScale, exitflag, lp, nullptr, info};
    return quadcopter_call(ctx, B, N, in, opts, o);
}

}  // extern "C"
