// obca_solver.h -- one OBCA parking NLP instance solved by ONE wavefront (workgroup of OB_NT = 64 threads, gfx950), persistent over the whole
// interior-point solve; the kernel runs one wavefront per SIMD (256 VGPRs + 256 AGPRs per wave), FOUR instances per CU (6.3 KB of static LDS +
// OB_DYN_LDS_DOUBLES(N) * 8 bytes sized for the horizon at launch: 24 KB at N = 80).
//
// Programming model: code outside a PAR(lane){...} region is wave-uniform (every lane computes the same scalars; what must survive a phase call lives
// in LDS: Shared::drv / sol / o); PAR regions distribute work items over the 64 lanes; data crosses lanes through LDS (`Shared`, g_traj), the per-instance
// records in HBM, or DPP / ds_bpermute / v_readlane.
//   * (stage, obstacle) blocks  -> one lane per block        (condensation / back-substitution, obca_model.h)
//   * stages                    -> one lane per stage        (bicycle model derivatives, costs, bounds)
//   * Riccati backward sweep    -> sequential in the stage index; per stage three short LDS phases, one structure-aware item per lane, 16-byte LDS operands
//   * forward sweep             -> two stages per dependent step: pair maps composed into LDS, state broadcast with v_readlane
//   * reductions (norms, step lengths, objective) -> 64-lane register butterfly (DPP inside a row of 16 lanes, ds_bpermute across rows)
//   * line search               -> fused into the next assembly (assemble_obs / assemble_stage <FUSED = 1>): the trial point goes to the second iterate buffer
// Phases are non-inlined device functions (ph_*), each with its own register allocation; a solve can be parked right after an accepted trial and
// resumed by a later launch with the assembly at hand (Slice, two-launch schedule).  The algorithm is the primal-dual interior-point method stated in
// DESIGN.md (IPOPT's Algorithm A with the option values of ParkingSignedDist.jl:41-43).
//
// The same source is compiled by tests/emu (g++, -DOBCA_EMU) where PAR is a plain loop over the lanes: that build exists only so that
// the kernel logic can be unit-tested on a machine without a GPU.  It is not linked into the product.
#pragma once
#ifndef OB_NT
#define OB_NT 64     // threads per problem instance: one wavefront
#endif
#ifdef OBCA_EMU
#define OBCA_FN static inline
#define OBCA_HD static inline
#define OBCA_PHASE static
#define PAR(lane) for (int lane = 0; lane < OB_NT; ++lane)
#define PAR64(lane) for (int lane = 0; lane < 64; ++lane)       // inside a WAVE0 section
#define WAVE0_BEGIN {
#define WAVE0_END }
#define SYNC() ((void)0)
#define LANE0 1
#define LDS_SYNC() ((void)0)
#define VM_DRAIN() ((void)0)
#define LDS_BARRIER() ((void)0)
#define OBCA_NLT OB_NT
#define UNIFORM(x) (x)
#define UNIFORM_D(x) (x)
#define OPAQUE(x) ((void)0)
#define SEAM(x) ((void)0)
#define OBCA_NL 64          // per-lane variables that live across a SYNC are arrays over the lanes in the emulation
#define LI(lane) (lane)
#else
#define OBCA_FN __device__ __forceinline__
#define OBCA_HD __host__ __device__ inline
// Phase entry points are real (non-inlined) device functions: each gets its own register allocation, so the unrolled
// per-lane model code of one phase cannot force spills into the latency-critical sequential sweeps of another.
#define OBCA_PHASE static __device__ __noinline__
#define PAR(lane) for (int lane = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
#define PAR64(lane) for (int lane = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
#define WAVE0_BEGIN if (threadIdx.x < 64) {      // sequential sweeps run on the first wavefront; the others wait at the next SYNC()
#define WAVE0_END }
#define SYNC() __syncthreads()
#define LANE0 (threadIdx.x == 0)
// The workgroup is ONE wavefront: its LDS operations execute in program order, so lanes only need the compiler to keep that
// order (wavefront-scope fences emit no instruction).  Unlike __syncthreads() this does not drain outstanding global loads,
// which lets the software-pipelined HBM gathers of the sequential sweeps stay in flight across phases.  Use it only where the
// cross-lane traffic of the surrounding phases goes through LDS.
#define LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)   // s_waitcnt vmcnt(0): all outstanding global loads / stores of this wave
// workgroup barrier for phases that exchange data through LDS only: unlike __syncthreads() it does not drain the global-memory counter
#define LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#define OBCA_NLT 1
#define OPAQUE(x) asm volatile("" : "+v"(x))   // hide a loop-invariant register from LICM: what is derived from it is recomputed, not kept live
// SEAM: a value formed by the fused line search enters the assembly as if it had been loaded from the iterate -- the compiler must not contract its producing
// expression into the consumers, or a resumed solve (which assembles the stored point) would walk through different bits than an uninterrupted one
#define SEAM(x) asm volatile("" : "+v"(x))
#define UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // value known to be wave-uniform: keep it in an SGPR (scalar branches, scalar loop counters)
#define UNIFORM_D(x) __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)))   // the same for a double: two SGPRs instead of two VGPRs
#define OBCA_NL 1
#define LI(lane) 0
#endif
#include "obca_model.h"
// Pointers into the per-instance HBM buffers carry the global address space explicitly: they are kept in LDS (Shared::inst),
// and a pointer loaded from memory would otherwise be "generic" -> flat_load/flat_store, whose completion is tied to the LDS
// counter (lgkmcnt) and would serialise every LDS read behind the outstanding HBM gathers.
#ifdef OBCA_EMU
typedef double gdbl;
#else
typedef __attribute__((address_space(1))) double gdbl;
#endif

namespace obca {

#define OB_NC 6      // Riccati right-hand sides: main, t, nu1..nu4
#define OB_NMAX 128  // longest horizon: the forward sweep gives ONE lane to every pair of stages (direction_main: 64 pairs per wavefront); LDS would allow more (5.7 KB + (27 N + 54) x 8 bytes)
#define OB_AS 60     // doubles per assembled stage record (only the entries that can be non-zero are kept: as_h / as_df below)
#define OB_RS 74     // doubles per Riccati stage record
#define OB_OC 12     // doubles per condensed obstacle record
// problem header (doubles, in front of rx, ry, ryaw): 26 scalars, then the obstacle set -- row counts, row offsets, the rows themselves.  It lives in LDS for the whole solve
// (Shared::hdr): 8 bytes per scalar, 16 per obstacle, 24 per half-space row = 2.0 KB at the limits below, of which a 3-obstacle / 5-row instance uses 0.4 KB.
#define PH_TS 0
#define PH_L 1
#define PH_G 2
#define PH_OFF 6
#define PH_XL 7
#define PH_XU 11
#define PH_X0 15
#define PH_XF 19
#define PH_FIX 23
#define PH_NOB 24
#define PH_M 25
#define PH_VOB 26                               // OB_NOBMAX row counts
#define PH_ROFF (PH_VOB + OB_NOBMAX)            // OB_NOBMAX + 1 row offsets
#define PH_DIST (PH_ROFF + OB_NOBMAX + 1)       // 1: ParkingDist.jl formulation
#define PH_A (PH_DIST + 1)                      // 2 x OB_MMAX: (a1, a2) of every row (unit length)
#define PH_B (PH_A + 2 * OB_MMAX)               // OB_MMAX
#define OB_HDR (PH_B + OB_MMAX)                 // doubles of problem header in front of rx, ry, ryaw
// stage record
#define AS_H 0       // 19 entries of the symmetric 8 x 8 stage Hessian (variables X, Y, psi, v, w0, w1, delta, a): slot as_h(i, j)
#define AS_HB 19     // gradient (8)
#define AS_HT 27     // d/dt column of the rows psi .. a (6: row i at AS_HT + i - 2; rows X, Y are zero)
#define AS_DF 33     // 16 entries of the bicycle Jacobian d(F - x)/d(psi, v, delta, a, t) (4 x 5): slot as_df(i, j)
#define AS_DD 49     // residual (4)
#define AS_SIG 53
#define AS_RG 54
#define AS_GG 55
#define AS_DSS 58
#define AS_RSS 59
// Riccati record
#define RS_K 0
#define RS_KF 12
#define RS_PX 24
#define RS_PV 48
#define RS_PAD 72    // unused slot: target of the dummy stores of lanes without an item
#define OB_FILT 224

struct Opts {
    double tol; int max_iter;
    double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
    double dw_min, dw0, dw_max, kw_inc0, kw_inc, kw_dec, dc_bar, kappa_c;
    double gamma_theta, gamma_phi, delta, s_theta, s_phi, eta_phi, gamma_alpha, s_max, kappa_sigma;
    double constr_viol_tol, dual_inf_tol, compl_inf_tol, rho_term;
};
struct OptsAbi { Opts o; int max_soc, recalc_y, lsq_init, obj_scaling; };      // obca_opts of the C ABI: the interior-point options + the three IPOPT switches (max_soc: second-order correction trials per iteration,
                                                          // IPOPT's default 4; recalc_y; lsq_init; all 0 = off by default, as in the checker).  Kept apart so that the options' place in LDS (Shared::o) is what the phases were tuned with.

struct Lay {
    int x, u, t, lam, mu, sl, so, ss, pi, nu, yg, yo, zxL, zxU, zuL, zuU, ztL, ztU, zlam, zmu, zso, zssL, zssU, zs1, nprimal, len;   // zs1: multiplier of the norm-row slack (ParkingDist only)
};
OBCA_HD void make_layout(int N, int nOb, int M, Lay &l) {
    int N1 = N + 1, o = 0;
    l.x = o; o += 4 * N1; l.u = o; o += 2 * N; l.t = o; o += 1;
    l.lam = o; o += M * N1; l.mu = o; o += 4 * nOb * N1; l.sl = o; o += nOb * N1;
    l.so = o; o += nOb * N1; l.ss = o; o += N; l.nprimal = o;
    l.pi = o; o += 4 * N; l.nu = o; o += 4; l.yg = o; o += N; l.yo = o; o += 4 * nOb * N1;
    l.zxL = o; o += 4 * N1; l.zxU = o; o += 4 * N1; l.zuL = o; o += 2 * N; l.zuU = o; o += 2 * N;
    l.ztL = o; o += 1; l.ztU = o; o += 1; l.zlam = o; o += M * N1; l.zmu = o; o += 4 * nOb * N1;
    l.zso = o; o += nOb * N1; l.zssL = o; o += N; l.zssU = o; o += N; l.zs1 = o; o += nOb * N1; l.len = o;
}

struct AsmOut { int ok; double dinf, pinf, cinf0, cmin, cmax, sumy, sumz, f, th1, bar, Htt, gtb; int nb, nm; };   // cmin, cmax: extreme complementarity products
template <class P> OBCA_FN void asm_pack(P *o, const AsmOut &A) {
    o[0] = A.ok; o[1] = A.dinf; o[2] = A.pinf; o[3] = A.cinf0; o[4] = A.cmin; o[5] = A.cmax; o[6] = A.sumy; o[7] = A.sumz; o[8] = A.f; o[9] = A.th1; o[10] = A.bar; o[11] = A.Htt; o[12] = A.gtb;
    o[13] = A.nb; o[14] = A.nm;
}
template <class P> OBCA_FN void asm_unpack(AsmOut &A, const P *o) {
    A.ok = (int)o[0]; A.dinf = o[1]; A.pinf = o[2]; A.cinf0 = o[3]; A.cmin = o[4]; A.cmax = o[5]; A.sumy = o[6]; A.sumz = o[7]; A.f = o[8]; A.th1 = o[9]; A.bar = o[10]; A.Htt = o[11]; A.gtb = o[12];
    A.nb = (int)o[13]; A.nm = (int)o[14];
}
OBCA_HD double cinf_mu(const AsmOut &A, double mu) { return fmax(fabs(A.cmax - mu), fabs(A.cmin - mu)); }      // complementarity error w.r.t. the barrier parameter mu: max |s z - mu|
struct StepOut { int ok; double ap, az, gd, gr; };   // gr: rate-cost part of d phi / d t (summed in the stage back-substitution, used with dt at the end)
struct Consts; struct Lay;
struct Inst {              // uniform: pointers of this instance
    const gdbl *prob;      // header + rx, ry, ryaw
    gdbl *z, *zn, *d, *as, *rs, *oc;   // z: the current iterate; zn: the buffer the line search writes its trial point to (the two swap when a trial is accepted);
                                              // d: stage part of the search direction (u, ss, pi, yg; the obstacle part is recomputed where it is needed, x lives in LDS)
    mutable long long tlast;           // diagnostic builds (-DOBCA_PROFILE): time stamp of the previous phase boundary
};

enum { SL_ATT = 0, SL_IT, SL_NF, SL_NREG, SL_MU, SL_DWLAST, SL_THMIN, SL_THMAX, SL_ITPREV, SL_NREGPREV, SL_PINF, SL_HAVE, SL_XPASS, SL_ASM = 16, SL_FILT = 32, SL_SIZE = SL_FILT + 2 * OB_FILT };
// SL_XPASS: full passes the slice spent outside iterations and inertia rungs (second-order corrections, rebuilds after rejected ones, multiplier re-estimates): the ordering kernel ranks by them too
// SL_HAVE / SL_ASM: a solve parked right after an accepted trial keeps that trial's assembly -- the scalars here, the stage records in the instance's own buffers, which
// outlive the launch -- so the resumed solve continues from exactly the state an uninterrupted one has at that point, without assembling again
struct Slice {
    gdbl *st;        // slice record of this instance (never null: the filter's overflow entries live there too)
    int resume;      // 1: the next ipm_attempt continues from the record instead of starting at the warm start
    int budget;      // factorisation passes (iterations + inertia retries) this launch may spend; 0 = no limit
    int used;        // passes spent so far in this launch
};
struct Result { int status, iters, nreg; double obj, pinf, dinf, mu; };
struct Sol { gdbl *home; Slice sl; Result R; int att, it_prev, nreg_prev, ef, iters, nreg, retry; };      // state of solve_instance (wave-uniform, in LDS)
struct Drv {                // state of the interior-point driver (wave-uniform; see ipm_attempt)
    double mu, tau, dw, dw_last, dc_mu, dc_val, th_min, th_max, f, pinf, dinf, sd, sc, cm, th, phi, gd, az, pw_th, pw_gd, amin, alpha;
    int nf, it, nreg, status, p_start, have_asm, mu_changed, ok, tr, acc;
};
struct Soc {                // state of the three IPOPT switches (second-order correction, recalc_y, least-squares initial multipliers: cold paths); at the END of Shared, so that
                            // nothing the phases of the default path address moves (their code is instruction-for-instruction that of the build without the switches)
    gdbl *csoc;             // c_soc = alpha c(z) + c(z + alpha d) of the instance, layout pi | nu | yg | yo as in the iterate (null unless max_soc > 0)
    int max_soc, nsoc, nsoc_acc;      // option; corrections tried / accepted in this attempt (diagnostic)
    int recalc_y, nrecalc;            // option recalc_y = "yes"; multiplier re-estimates in this attempt (diagnostic)
    int lsq_init;                     // option: least-squares initial multipliers (IPOPT's default initialisation, constr_mult_init_max = 1e3)
    int nrebuild;                     // Newton systems rebuilt after a rejected correction in this attempt (a full pass each: counted against the slice budget)
};
#define OB_FILT_LDS 32     // filter entries kept in LDS; the (rare) rest lives in the instance's slice record

struct alignas(16) Shared {
    double hdr[OB_HDR];
    alignas(16) double Bm[36], coef[8];                         // border constants (left by the backward sweep for the border solve), (dt, nu)
    double filt[OB_FILT_LDS][2];
    Drv drv; Sol sol; Opts o;      // (the options too: as kernel arguments they would sit in ~60 SGPRs that are spilled around every phase call)
    int roff[OB_NOBMAX + 1], vOb[OB_NOBMAX], ric_ok, upl[OB_NT], ucn[3][OB_NT];      // upl, ucn: which positions of the unpacked stage data a lane serves (init_unpack_table)
    Consts c; Lay l;
    double prof[16];           // diagnostic per-phase cycle counters (-DOBCA_PROFILE)
    Inst inst; AsmOut A, A2, An, Ap; StepOut S; int vm2, vmc;   // vmc: row class of the instance's widest obstacle (0: <= 2, 1: <= OB_VMID, 2: <= OB_VMAX)   // phase inputs/outputs (wave-uniform, exchanged through LDS)
    Soc soc;
    int ft_ok, ft_done;      // ft_ok: the instance's (stage, obstacle) items fit OB_KEEP rounds (first-trial block part merged into the direction phase); ft_done: that part has run for the direction at hand
};

// Dynamic LDS behind `Shared`, sized for the horizon at launch (OB_DYN_LDS_DOUBLES).  Three layouts share it, one per phase of a pass (they never overlap in time):
//   forward sweep / line search : [ trajectory (N + 2) x 6 | composed closed-loop maps of the stage pairs (N / 2 + 1) x 42 ]      s_k = (dx_k, dw_k): the x part of the search
//                                 direction lives in the trajectory and nowhere else (direction_*, the fused line search read it)
//   assembly                    : [ trajectory (still the direction the trial point is formed along) | condensed obstacle sums (N + 1) x 12 ]
//   backward sweep              : [ per-stage border data N x RIC_BD | two unpacked stage buffers 2 x OB_STG (SG_* offsets, + a pad slot) | operands RicLds ]   -- the trajectory is dead
//                                 by then (a backward sweep always starts a new direction), so the sweep uses the region from its start
// Round 3 kept the sweep's operands (2.4 KB) in the static block; with them here the block is 5 KB and seven instances fit a CU's 160 KB instead of six.
#define OB_STG 200
struct alignas(16) RicLds {
    // Riccati backward sweep.  Every operand of its dot products is a CONTIGUOUS, 16-byte aligned 6-vector (P rows, the rows of the transposed FA', T', p'), read as three
    // ds_read_b128: a lone wavefront per SIMD issues 8-byte LDS reads at a fifth of the LDS rate but 16-byte reads at the full rate (MI355X_MICROARCH.md, LDS).
    alignas(16) double Pn[36], pn[6 * OB_NC], Qhat[8 * 14];     // P (row-major, symmetric), p' (pn[c * 6 + a]: right-hand side c, state a), Qhat (8 x 14 row-major)
    alignas(16) double sB[24], TT[14 * 6];                      // static parts of the border constants, T' (TT[cc * 6 + a])
    alignas(16) double zero6[6]; double zero, dump, dump4[4];   // constant 0 and a write-only slot: operand / destination of the lanes without an item in the Riccati phases
};
#define OB_RICLDS_DOUBLES (sizeof(RicLds) / sizeof(double))
#define OB_MAX2(a, b) ((a) > (b) ? (a) : (b))
#define OB_DYN_LDS_DOUBLES(N) OB_MAX2((size_t)((N) + 2) * 6 + OB_MAX2((size_t)((N) / 2 + 1) * 42, (size_t)((N) + 1) * OB_OC), (size_t)16 * (N) + 2 * OB_STG + OB_RICLDS_DOUBLES)
#ifdef OBCA_EMU
static Shared g_sh;
alignas(16) static double g_traj[OB_DYN_LDS_DOUBLES(OB_NMAX)];
#else
__shared__ Shared g_sh;     // the static LDS block of the workgroup (= one wavefront = one problem instance)
extern __shared__ __attribute__((aligned(16))) double g_traj[];
#endif

OBCA_FN double *stg_base(const Shared &sh) { return g_traj + (size_t)(sh.c.N + 2) * 6; }     // pair maps of the forward sweep / condensed obstacle sums of the assembly: behind the trajectory
OBCA_FN double *ric_sg0(const Shared &sh) { return g_traj + (size_t)sh.c.N * 16; }            // backward sweep: the two stage buffers, behind the per-stage border data (RIC_BD = 16 doubles per stage)
OBCA_FN RicLds &ric_lds(const Shared &sh) { return *(RicLds *)(ric_sg0(sh) + 2 * OB_STG); }   // backward sweep: its operands, behind the stage buffers

// phase ids of the diagnostic cycle counters
enum { PF_INIT = 0, PF_ASM_OBS, PF_ASM_STAGE, PF_RIC_BWD, PF_BORDER_CL, PF_FWD_SEQ, PF_BS_STAGE, PF_BS_OBS, PF_TRIAL, PF_APPLY, PF_OTHER, PF_RIC_P1, PF_RIC_P2, PF_N };
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)
#define PROF(I, id) do { if (LANE0) { long long now_ = clock64(); sh.prof[id] += (double)(now_ - (I).tlast); (I).tlast = now_; } } while (0)   /* one writer: lane 0 */
#else
#define PROF(I, id) ((void)0)
#endif
#ifdef OBCA_PROFILE_FINE          // per-stage counters inside the Riccati sweep: they cost ~20 % of a stage, off by default even in profile builds
#define PROF_FINE(I, id) PROF(I, id)
#else
#define PROF_FINE(I, id) ((void)0)
#endif

// ---------------------------------------------------------------- reductions over the lanes of the instance
// (Two-wavefront instances -- the quadcopter kernel, NT = 128 -- first fold the second wavefront's slots onto the first.)  A 64-lane butterfly in ASCENDING distance (1, 2, 4, 8, 16, 32).  The first
// four exchanges stay inside a row of 16 lanes and run as DPP moves on the vector ALU (quad permutes, then half-row and row mirrors:
// once every lane of a quad / half-row holds the same partial result, the mirrored partner carries exactly what the xor partner
// would); only distances 16 and 32 cross rows and go through ds_bpermute.  An LDS exchange costs a ~100-clock round trip that the
// compiler serialises per reduction, and a pass holds some thirty reductions.  The emulation pairs lanes i and i^o in the same
// order, so its results are bit-identical (sum and max are commutative).
#ifdef OBCA_EMU
#define RED_IMPL(NAME, COMB)                                                                                     \
    template <int NT> OBCA_FN double NAME(const double *r) {                                                     \
        double a[64], b[64];                                                                                     \
        for (int i = 0; i < 64; i++) { a[i] = r[i]; if (NT > 64) { double w = r[i + 64 * (NT > 64)], v = a[i]; a[i] = COMB; } } \
        for (int o = 1; o < 64; o <<= 1) {                                                                       \
            for (int i = 0; i < 64; i++) { double v = a[i], w = a[i ^ o]; b[i] = COMB; }                         \
            for (int i = 0; i < 64; i++) a[i] = b[i];                                                            \
        }                                                                                                        \
        return a[0];                                                                                             \
    }
#else
OBCA_FN double readlane_f64(double v, const int l) {   // value of lane l (a constant) as a wave-uniform scalar
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
OBCA_FN double dpp_f64(double v) {   // every lane active (the reductions are called from uniform control flow)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define RED_IMPL(NAME, COMB)                                                                                     \
    template <int NT> OBCA_FN double NAME(const double *r) {                                                     \
        double v = r[threadIdx.x & 63], w;                                                                       \
        if (NT > 64) { w = r[(threadIdx.x & 63) + 64 * (NT > 64)]; v = COMB; }                                   \
        w = dpp_f64<0xB1>(v); v = COMB;          /* quad_perm [1,0,3,2]  : i ^ 1 */                               \
        w = dpp_f64<0x4E>(v); v = COMB;          /* quad_perm [2,3,0,1]  : i ^ 2 */                               \
        w = dpp_f64<0x141>(v); v = COMB;         /* row_half_mirror      : stands in for i ^ 4 */                 \
        w = dpp_f64<0x140>(v); v = COMB;         /* row_mirror           : stands in for i ^ 8 */                 \
        w = __shfl_xor(v, 16, 64); v = COMB;                                                                     \
        w = __shfl_xor(v, 32, 64); v = COMB;                                                                     \
        return v;                                                                                                \
    }
#endif
// the same butterflies on per-lane REGISTER values (one-wavefront instances: nothing goes through LDS).  In the host emulation a per-lane value that lives
// across the lanes' loop is an array over the lanes (OBCA_NL = 64), on the GPU it is one register (OBCA_NL = 1).
#ifdef OBCA_EMU
#define WRED_IMPL(NAME, COMB)                                                                                    \
    OBCA_FN double NAME(const double (&r)[OBCA_NL]) {                                                            \
        double a[64], b[64];                                                                                     \
        for (int i = 0; i < 64; i++) a[i] = r[i];                                                                \
        for (int o = 1; o < 64; o <<= 1) {                                                                       \
            for (int i = 0; i < 64; i++) { double v = a[i], w = a[i ^ o]; b[i] = COMB; }                         \
            for (int i = 0; i < 64; i++) a[i] = b[i];                                                            \
        }                                                                                                        \
        return a[0];                                                                                             \
    }
#else
#define WRED_IMPL(NAME, COMB)                                                                                    \
    OBCA_FN double NAME(const double (&r)[OBCA_NL]) {                                                            \
        double v = r[0], w;                                                                                      \
        w = dpp_f64<0xB1>(v); v = COMB;                                                                          \
        w = dpp_f64<0x4E>(v); v = COMB;                                                                          \
        w = dpp_f64<0x141>(v); v = COMB;                                                                         \
        w = dpp_f64<0x140>(v); v = COMB;                                                                         \
        w = __shfl_xor(v, 16, 64); v = COMB;                                                                     \
        w = __shfl_xor(v, 32, 64); v = COMB;                                                                     \
        return v;                                                                                                \
    }
#endif
// sum over each quad of lanes (4 q .. 4 q + 3), left in all four of them: two DPP exchanges
#ifdef OBCA_EMU
OBCA_FN void wquad_sum(const double (&r)[OBCA_NL], double (&out)[OBCA_NL]) {
    double a[64];
    for (int i = 0; i < 64; i++) a[i] = r[i] + r[i ^ 1];
    for (int i = 0; i < 64; i++) out[i] = a[i] + a[i ^ 2];
}
#else
OBCA_FN void wquad_sum(const double (&r)[OBCA_NL], double (&out)[OBCA_NL]) { double v = r[0]; v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); out[0] = v; }
#endif
// value of lane l (a constant) of a per-lane variable as a wave-uniform scalar (v_readlane; the emulation keeps per-lane variables as arrays over the lanes)
#ifdef OBCA_EMU
#define WV_READLANE(x, l) ((x)[l])
#else
#define WV_READLANE(x, l) readlane_f64((x)[0], (l))
#endif
WRED_IMPL(wred_sum, (v + w))
WRED_IMPL(wred_max, ((w > v || w != w) ? w : v))      // NaN-propagating max
WRED_IMPL(wred_min, ((w < v) ? w : v))
RED_IMPL(red_sum_t, (v + w))
RED_IMPL(red_max_t, ((w > v || w != w) ? w : v))      // NaN-propagating max
RED_IMPL(red_min_t, ((w < v) ? w : v))
OBCA_FN double red_sum(const double *r) { return red_sum_t<OB_NT>(r); }
OBCA_FN double red_max(const double *r) { return red_max_t<OB_NT>(r); }
OBCA_FN double red_min(const double *r) { return red_min_t<OB_NT>(r); }

// Sum of the condensed contributions (12 doubles) of the obstacles of a stage, in the order of the obstacles -- the sums are the same bits in every run, on every box.
// Items are stage-major (item = k nOb + j), so the lanes of one round that belong to a stage are neighbours: position p = min(j, lane) within the stage's run of lanes.
// A running sum walks down the run, one lane per step (wave_shr:1 moves it to the next lane; nOb - 1 uniform steps, all lanes take part in the moves, only the lane whose
// turn it is adds); the last lane of the run stores the 12 sums.  A stage whose obstacles straddle two rounds is continued: lane 0 of the next round starts from the stored
// partial sums (LDS traffic of one wavefront is in order, the rounds in program order).  The first obstacle of a stage starts from +0, as the emulation's cleared cell does.
// (Round 4 used ds_add_f64 here: up to 16 lanes of one instruction on one address, relying on the hardware serving them in lane order -- nothing documents that, fp64 addition
// is not associative, and the driver's round-4 GPU run saw two runs of the same inputs differ.  tools/micro/lds_atomic_order.hip probes the order)
#ifdef OBCA_EMU
OBCA_FN void obs_sum_ordered(double *ocs, const ObsCond &cd, int k, int j, bool on, int nOb, int lane) {
    (void)j; (void)nOb; (void)lane;
    if (!on) return;
    double *o = ocs + (size_t)k * OB_OC;
    for (int i = 0; i < 6; i++) o[i] += cd.Hpp[i];
    for (int i = 0; i < 3; i++) { o[6 + i] += cd.gz[i]; o[9 + i] += cd.gcorr[i]; }
}
#else
OBCA_FN void obs_sum_ordered(double *ocs, const ObsCond &cd, int k, int j, bool on, int nOb, int lane) {
    double c[OB_OC], run[OB_OC];
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = cd.Hpp[i];
#pragma unroll
    for (int i = 0; i < 3; i++) { c[6 + i] = cd.gz[i]; c[9 + i] = cd.gcorr[i]; }
    double *o = ocs + (size_t)k * OB_OC;
    const int p = j < lane ? j : lane;
    LDS_SYNC();                                        // the partial sums the previous round stored are visible (and the compiler keeps the order)
    const bool cont = on && lane == 0 && j > 0;        // the stage began in the previous round
#pragma unroll
    for (int i = 0; i < OB_OC; i++) run[i] = 0.0;
    if (cont) {
#pragma unroll
        for (int i = 0; i < OB_OC; i++) run[i] = o[i];
    }
#pragma unroll
    for (int i = 0; i < OB_OC; i++) run[i] += c[i];
    for (int s = 1; s < nOb; s++) {                    // uniform
        double t[OB_OC];
#pragma unroll
        for (int i = 0; i < OB_OC; i++) t[i] = dpp_f64<0x138>(run[i]);      // wave_shr:1 -- lane l receives lane l - 1's running sum
        if (on && p == s) {
#pragma unroll
            for (int i = 0; i < OB_OC; i++) run[i] = t[i] + c[i];
        }
    }
    if (on && (j == nOb - 1 || lane == OB_NT - 1)) {   // end of the stage's run in this round (the last item of all is a last obstacle)
#pragma unroll
        for (int i = 0; i < OB_OC; i++) o[i] = run[i];
    }
}
#endif
// the per-instance constants the (stage, obstacle) block code reads, copied into scalar registers (as LDS reads they would sit in vector registers for the whole item loop)
OBCA_FN void obs_consts(const Consts &s_, Consts &c) {
    c.N = UNIFORM(s_.N); c.nOb = UNIFORM(s_.nOb); c.M = UNIFORM(s_.M); c.dist = UNIFORM(s_.dist); c.fixTime = UNIFORM(s_.fixTime);
    c.off = UNIFORM_D(s_.off);
#pragma unroll
    for (int i = 0; i < 4; i++) c.g[i] = UNIFORM_D(s_.g[i]);
}
template <int VM>
OBCA_FN void load_obs(const Inst &I, const Shared &sh, const gdbl *z, int k, int j, ObsIn<VM> &in) {
    const Lay &l = sh.l; const int nOb = sh.c.nOb, M = sh.c.M;
    const int r0 = sh.roff[j], v = sh.vOb[j], bo = k * nOb + j;
    in.v = v;
#pragma unroll
    for (int i = 0; i < VM; i++) {
        bool on = i < v;
        in.a1[i] = on ? sh.hdr[PH_A + 2 * (r0 + i)] : 0.0; in.a2[i] = on ? sh.hdr[PH_A + 2 * (r0 + i) + 1] : 0.0;
        in.b[i] = on ? sh.hdr[PH_B + r0 + i] : 0.0;
        in.lam[i] = on ? z[l.lam + k * M + r0 + i] : 1.0; in.zl[i] = on ? z[l.zlam + k * M + r0 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { in.mu[i] = z[l.mu + 4 * bo + i]; in.zm[i] = z[l.zmu + 4 * bo + i]; in.y[i] = z[l.yo + 4 * bo + i]; }
    in.sl = z[l.sl + bo]; in.so = z[l.so + bo]; in.zso = z[l.zso + bo]; in.zs1 = z[l.zs1 + bo];
    in.X = z[l.x + 4 * k]; in.Y = z[l.x + 4 * k + 1]; in.psi = z[l.x + 4 * k + 2];
}

struct B2 { double Sig, gz, gb; };
// (the largest |s z| is not tracked: it is max(|smallest product|, |largest product|), formed once from cmn / cmx where the assembly ends)
OBCA_FN B2 bound2(double v, double lo, double hi, double zL, double zU, double mu, double mult, double &cmn, double &cmx, double &sumz) {
    const double dL = v - lo, dU = hi - v, iL = rcp_nr(dL), iU = rcp_nr(dU);
    B2 r; r.Sig = mult * (zL * iL + zU * iU); r.gz = mult * (-zL + zU); r.gb = mult * mu * (iU - iL);
    double c1 = dL * zL, c2 = dU * zU;
    cmn = fmin(cmn, fmin(c1, c2)); cmx = fmax(cmx, fmax(c1, c2));
    sumz += fabs(zL) + fabs(zU);
    return r;
}
// the same with the running max of |s z| kept by the caller (the quadcopter kernel's stage assembly, obca_quad_solver.h)
OBCA_FN B2 bound2(double v, double lo, double hi, double zL, double zU, double mu, double mult, double &c0, double &cmn, double &cmx, double &sumz) {
    const B2 r = bound2(v, lo, hi, zL, zU, mu, mult, cmn, cmx, sumz);
    c0 = fmax(c0, fmax(fabs((v - lo) * zL), fabs((hi - v) * zU)));
    return r;
}
// Barrier sums.  sum_i log(d_i) is evaluated as log(prod_i d_i) over groups of at most G distances: a double-precision log is a ~2k-clock
// dependent chain for a lone wavefront and there are a dozen per stage / obstacle item, while the product of twelve distances in
// [1e-25, 1e25] stays inside the double range.  A non-positive distance poisons its group (NaN), as its own log would.  The assembly and
// the trial evaluation use the same groups in the same order, so the same point gives the same bits in both.
template <int NN, int G = 12>
OBCA_FN double log_prod(const double (&dd)[NN]) {
    double s_ = 0;
#pragma unroll
    for (int g = 0; g < NN; g += G) {
        double p0 = 1, p1 = 1, mn = 1;
#pragma unroll
        for (int i = g; i < g + G && i < NN; i++) { if (i & 1) p1 *= dd[i]; else p0 *= dd[i]; mn = fmin(mn, dd[i]); }
        const double lg = log(p0 * p1);
        s_ += mn > 0 ? lg : NAN;
    }
    return s_;
}
// the same with running accumulators (two product chains, lower / upper distances) for code that meets its bounds one at a time
struct BarAcc { double p0, p1, mn; };
OBCA_FN void bar_init(BarAcc &a) { a.p0 = a.p1 = a.mn = 1.0; }
OBCA_FN void bar_mul(BarAcc &a, double dlo, double dhi) { a.p0 *= dlo; a.p1 *= dhi; a.mn = fmin(a.mn, fmin(dlo, dhi)); }
OBCA_FN double bar_log(const BarAcc &a) { const double lg = log(a.p0 * a.p1); return a.mn > 0 ? lg : NAN; }
OBCA_FN int hidx(int i, int j) { int a_ = i < j ? i : j, b_ = i < j ? j : i; return a_ * 8 - a_ * (a_ - 1) / 2 + (b_ - a_); }
// Which entries of a stage record exist.  Hessian: pose block (obstacles, tracking), the (psi, v, delta, a) block of the bicycle model, the rate terms (w, u) and the
// steering row (w0, delta) -- 19 of 36; everything else is structurally zero and neither stored nor gathered by the backward sweep (-1).
OBCA_FN int as_h(int i, int j) {
    const int a_ = i < j ? i : j, b_ = i < j ? j : i;
    switch (a_ * 8 + b_) {
        case 0: return 0; case 1: return 1; case 2: return 2; case 9: return 3; case 10: return 4; case 18: return 5; case 19: return 6; case 22: return 7; case 23: return 8;
        case 27: return 9; case 30: return 10; case 31: return 11; case 36: return 12; case 38: return 13; case 45: return 14; case 47: return 15; case 54: return 16;
        case 55: return 17; case 63: return 18; default: return -1;
    }
}
// Jacobian: F_psi does not depend on psi itself beyond the identity, F_v only on a and t
OBCA_FN int as_df(int i, int j) { return i < 2 ? 5 * i + j : (i == 2 ? (j >= 1 ? 9 + j : -1) : (j >= 3 ? 11 + j : -1)); }


// ---------------------------------------------------------------- accepting a step: new bound multipliers
template <int RS = 1>
OBCA_FN double clampz(double zz, double dist, double mu, double ks) { const double q = mu * rcp_nr<RS>(dist), lo = q * rcp_nr<RS>(ks), hi = ks * q; return zz < lo ? lo : (zz > hi ? hi : zz); }
// bound-multiplier step for a lower bound at distance `dist` (upper bound: pass -dv):  z += az (mu/dist - z - z/dist dv)
template <int RS = 1>
OBCA_FN double zstep(double zz, double dist, double dv, double mu, double az) { const double id = rcp_nr<RS>(dist); return zz + az * (mu * id - zz - zz * id * dv); }


// ---------------------------------------------------------------- assemble the condensed Newton system
// The line search is fused into the assembly (FUSED = 1): the trial point z + alpha d is formed on the fly, written to the second iterate buffer
// (Inst::zn) and assembled right there -- objective, constraint norm and barrier of the trial point ARE the f / th1 / bar of its assembly, and when the
// trial is accepted (the first one, as a rule) the two buffers swap and the next iteration starts with its Newton system already assembled.  Against
// separate trial / accept / assemble phases (round 2) the iterate is read once instead of three times per iteration and the obstacle part of the search
// direction is never stored: a (stage, obstacle) block recomputes its step from the pose step (obs_block<1>, the same code direction_obs ran).
// The obstacle part of the search direction (d lambda, d mu, d sl, d so, d y per (stage, obstacle) block) is needed twice: for the step lengths (direction_obs) and for
// the trial point (fused assembly).  OBCA_STORE_DOBS = 0 (default): the fused assembly recomputes it (obs_block<1>: ~25 % of that phase's arithmetic, no traffic);
// 1: direction_obs writes it to `d` and the fused assembly loads it with the block's iterate.  Measured on MI355X (config 2, profiles/r03_ab_obstacle_steps.txt): the same
// pipelined rate (212.8 k / 214.6 k solves/s), storing is 5 % quicker per pass for a lone instance and moves 10 % more HBM bytes (15.9 against 14.4 GB per launch).
#ifndef OBCA_STORE_DOBS
#define OBCA_STORE_DOBS 0
#endif
struct FuseArgs { double alpha, ay, az, ks, dw_dir; };   // step lengths (primal, equality multipliers, bound multipliers), kappa_sigma, delta_w of the factorisation that gave d
// part (a): one lane per (stage, obstacle) block; partial results go to sh.Ap
// SOC = 1: the system of a second-order correction step -- FUSED = 0: condensation with c_soc on the right-hand side; FUSED = 1: the block steps of the trial are those of
// the correction direction (recomputed with c_soc), the assembly at the trial point is the ordinary one.
// KEEP = 1 (with FUSED = 1): the block steps were kept in registers by direction_obs<KEEP = 1> of the same phase call (ph_direction2_trial: the first trial of a line search);
// item lane + 64 r finds its step in keep[r] and is not factorised a second time at the old point.
#define OB_KEEP 4          // rounds of (stage, obstacle) items whose steps a lane keeps: (N + 1) nOb <= 64 OB_KEEP (N = 80, 3 obstacles: 243 items)
template <int VM, int FUSED, int SOC = 0, int LSQ = 0, int KEEP = 0>      // LSQ = 1 (with FUSED = 0): the blocks of the least-squares multiplier system (obs_block)
OBCA_FN void assemble_obs(const Inst &I, Shared &sh, double mu_, double dw_, double dc_, const FuseArgs &fa_, const ObsStep<VM> (*keep)[OBCA_NL] = nullptr) {
    const Lay &l = sh.l;
    Consts c; obs_consts(sh.c, c);            // the constants the block code uses, in scalar registers (see assemble_stage)
    const double mu = UNIFORM_D(mu_), dw = UNIFORM_D(dw_), dc = UNIFORM_D(dc_);
    const FuseArgs fa = {UNIFORM_D(fa_.alpha), UNIFORM_D(fa_.ay), UNIFORM_D(fa_.az), UNIFORM_D(fa_.ks), UNIFORM_D(fa_.dw_dir)};
    const int N = c.N, nOb = c.nOb, M = c.M;
    constexpr int RS_ = VM <= 2 ? 1 : 0;       // which reciprocal form (rcp_nr, obca_model.h)
    const gdbl *z = I.z; gdbl *zn = I.zn;
    double red[11][OBCA_NL];                 // per-lane partial results, reduced over the wavefront in registers
    double *ocs = stg_base(sh);              // 12 condensed sums per stage (LDS: the region of the sweeps' buffers, idle during the assembly)
#ifdef OBCA_EMU
    PAR(lane) { for (int i = lane; i < (N + 1) * OB_OC; i += OB_NT) ocs[i] = 0.0; }      // (on the GPU the first obstacle of a stage starts the sum)
#endif
    LDS_SYNC();
    // ---- (a) obstacle blocks: one lane per (stage, obstacle)
    PAR(lane) {
        ObsStats st; st.dmax = st.pmax = st.sumz = st.sumy = 0; st.cmin = 1e300; st.cmax = -1e300; st.bad = 0;
        double fsl = 0, th = 0, bar = 0;
        const int nit = (N + 1) * nOb;
#pragma unroll
        for (int rr = 0; rr < (KEEP ? OB_KEEP : 1); rr++)      // KEEP: the rounds are unrolled so that keep[rr] is a fixed set of registers; otherwise one pass of the plain item loop
        // (the round loop is UNIFORM -- its bounds sit in scalar registers -- and a lane without an item in the last round skips the item code: the ordered sum of the
        // condensed contributions below exchanges registers between the lanes and needs all of them)
        for (int it0 = KEEP ? rr * OB_NT : 0; it0 < nit; it0 += (KEEP ? nit : OB_NT)) {
            const int it = it0 + lane; const bool on = it < nit;
            int k = 0, j = 0;
            ObsIn<VM> in; ObsCond cd;
            if (on) {
            k = it / nOb; j = it - k * nOb;
            load_obs<VM>(I, sh, z, k, j, in);
            double crs[4] = {0, 0, 0, 0};
            if (SOC) {
#pragma unroll
                for (int r = 0; r < 4; r++) crs[r] = sh.soc.csoc[(l.yo - l.pi) + 4 * it + r];
            }
            if (FUSED) {
                const double dp[3] = {g_traj[(size_t)k * 6], g_traj[(size_t)k * 6 + 1], g_traj[(size_t)k * 6 + 2]};      // pose step of the stage (x_0 is fixed: s_0 = 0)
                ObsStep<VM> sp;
                const int r0 = sh.roff[j];
                if (OBCA_STORE_DOBS) {
                    const gdbl *d = I.d;
#pragma unroll
                    for (int i = 0; i < VM; i++) sp.dlam[i] = i < in.v ? d[l.lam + k * M + r0 + i] : 0.0;
#pragma unroll
                    for (int i = 0; i < 4; i++) { sp.dmu[i] = d[l.mu + 4 * it + i]; sp.dy[i] = d[l.yo + 4 * it + i]; }
                    sp.dsl = d[l.sl + it]; sp.dso = d[l.so + it];
                } else if (KEEP) sp = keep[rr][LI(lane)];
                else obs_block<1, VM, SOC>(c, in, mu, fa.dw_dir, dc, nullptr, nullptr, dp, &sp, crs);
#pragma unroll
                for (int i = 0; i < VM; i++) if (i < in.v) {
                    const double v1 = fma(fa.alpha, sp.dlam[i], in.lam[i]), z1 = zstep<RS_>(in.zl[i], in.lam[i], sp.dlam[i], mu, fa.az);
                    in.lam[i] = v1; in.zl[i] = clampz<RS_>(z1, v1, mu, fa.ks);
                    zn[l.lam + k * M + r0 + i] = in.lam[i]; zn[l.zlam + k * M + r0 + i] = in.zl[i];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double v1 = fma(fa.alpha, sp.dmu[i], in.mu[i]), z1 = zstep<RS_>(in.zm[i], in.mu[i], sp.dmu[i], mu, fa.az);
                    in.mu[i] = v1; in.zm[i] = clampz<RS_>(z1, v1, mu, fa.ks); in.y[i] = fma(fa.ay, sp.dy[i], in.y[i]);
                    zn[l.mu + 4 * it + i] = in.mu[i]; zn[l.zmu + 4 * it + i] = in.zm[i]; zn[l.yo + 4 * it + i] = in.y[i];
                }
                {
                    const double v1 = fma(fa.alpha, sp.dso, in.so), z1 = zstep<RS_>(in.zso, in.so, sp.dso, mu, fa.az);
                    in.so = v1; in.zso = clampz<RS_>(z1, v1, mu, fa.ks);
                    const double s1 = fma(fa.alpha, sp.dsl, in.sl);
                    if (c.dist) in.zs1 = clampz<RS_>(zstep<RS_>(in.zs1, in.sl, sp.dsl, mu, fa.az), s1, mu, fa.ks);
                    in.sl = s1;
                    zn[l.so + it] = in.so; zn[l.zso + it] = in.zso; zn[l.sl + it] = in.sl; zn[l.zs1 + it] = in.zs1;
                }
                in.X = fma(fa.alpha, dp[0], in.X); in.Y = fma(fa.alpha, dp[1], in.Y); in.psi = fma(fa.alpha, dp[2], in.psi);      // (explicit fma: the stage part forms the same values from the same operands, bit for bit)
#pragma unroll
                for (int i = 0; i < VM; i++) { SEAM(in.lam[i]); SEAM(in.zl[i]); }
#pragma unroll
                for (int i = 0; i < 4; i++) { SEAM(in.mu[i]); SEAM(in.zm[i]); SEAM(in.y[i]); }
                SEAM(in.so); SEAM(in.zso); SEAM(in.sl); SEAM(in.zs1); SEAM(in.X); SEAM(in.Y); SEAM(in.psi);
            }
            obs_block<0, VM, (SOC && !FUSED) ? 1 : 0, LSQ>(c, in, mu, dw, dc, &cd, &st, nullptr, nullptr, crs);
            }
            // the condensed contribution goes into the stage's 12 sums in LDS (rounds 1-3 wrote a record per (stage, obstacle) to HBM and the stage part read nOb of them back:
            // 72 doubles of traffic per stage and pass), summed over the obstacles in a FIXED order: obs_sum_ordered
            obs_sum_ordered(ocs, cd, k, j, on, nOb, lane);
            if (on) {
            if (!c.dist) fsl += 1e2 * in.sl + 1e4 * in.sl * in.sl;
            double r[4]; obs_rows<VM>(c, in, r);
            th += fabs(r[0]) + fabs(r[1]) + fabs(r[2]) + fabs(r[3]);
            {
                double dd[VM + 6];
#pragma unroll
                for (int i = 0; i < VM; i++) dd[i] = i < in.v ? in.lam[i] : 1.0;
#pragma unroll
                for (int i = 0; i < 4; i++) dd[VM + i] = in.mu[i];
                dd[VM + 4] = in.so; dd[VM + 5] = c.dist ? in.sl : 1.0;
                bar += log_prod(dd);
            }
            }
        }
        red[0][LI(lane)] = st.dmax; red[1][LI(lane)] = st.pmax; red[3][LI(lane)] = st.cmin; red[10][LI(lane)] = st.cmax;
        red[4][LI(lane)] = st.sumz; red[5][LI(lane)] = st.sumy; red[6][LI(lane)] = fsl; red[7][LI(lane)] = th;
        red[8][LI(lane)] = bar; red[9][LI(lane)] = st.bad ? 1.0 : 0.0;
    }
    AsmOut &P = sh.Ap;
    P.dinf = wred_max(red[0]); P.pinf = wred_max(red[1]); P.cinf0 = 0; P.cmin = wred_min(red[3]); P.cmax = wred_max(red[10]);
    P.sumz = wred_sum(red[4]); P.sumy = wred_sum(red[5]); P.f = wred_sum(red[6]); P.th1 = wred_sum(red[7]);
    P.bar = wred_sum(red[8]);
    P.ok = !(wred_max(red[9]) > 0.5);
    SYNC();
    PROF(I, FUSED ? PF_TRIAL : PF_ASM_OBS);      // (diagnostic counters: the fused line-search step is booked under the former trial / apply slots)
}

// part (b): one lane per stage; combines with the partial results of part (a)
// The stage item is written in SECTIONS -- state x_k | condensed obstacle sums | inputs, rate cost, steering row | dynamics | finish -- each of which loads what it needs,
// folds it into the few accumulators of the stage record and stores what is final, with a scheduling barrier in between: the live set stays below the 256 registers a
// wavefront has when TWO of them share a SIMD (rounds 1-3 issued every load of the stage up front and kept ~430 registers alive, which fixed the kernel at one wavefront per
// SIMD).  The loads of a section are issued one section ahead, so a section's arithmetic runs in the shadow of the next one's memory round trip.
#ifdef OBCA_EMU
#define SECTION() ((void)0)
#else
#define SECTION() __builtin_amdgcn_sched_barrier(0)
#endif
template <int FUSED, int SOC = 0, int LSQ = 0>      // SOC = 1 (with FUSED = 0): steering and dynamics rows enter the right-hand side with c_soc;  LSQ = 1 (with FUSED = 0, mu = dw = dc = 0):
// the least-squares multiplier system -- unit Hessian, no second derivatives, zero constraint right-hand side, gradients in their z-form
OBCA_FN void assemble_stage(const Inst &I, Shared &sh, double mu_, double dw_, double dc_, const FuseArgs &fa_, AsmOut &out) {
    const Consts &c = sh.c; const Lay &l = sh.l;
    const int N = UNIFORM(c.N), nOb = c.nOb, M = c.M;
    const gdbl *z = I.z, *d = I.d; gdbl *zn = I.zn;
    // what is the same for every lane lives in scalar registers (as function arguments and LDS reads these values would each hold two of the 256 vector registers)
    const double mu = UNIFORM_D(mu_), dw = UNIFORM_D(dw_), dc = UNIFORM_D(dc_);
    const FuseArgs fa = {UNIFORM_D(fa_.alpha), UNIFORM_D(fa_.ay), UNIFORM_D(fa_.az), UNIFORM_D(fa_.ks), 0.0};
    const double cTs = UNIFORM_D(c.Ts), ciL = UNIFORM_D(c.iL), cwpsi = UNIFORM_D(c.wpsi), cwa = UNIFORM_D(c.wa);
    const double xl0 = UNIFORM_D(c.xl[0]), xl1 = UNIFORM_D(c.xl[1]), xl3 = UNIFORM_D(c.xl[3]), xu0 = UNIFORM_D(c.xu[0]), xu1 = UNIFORM_D(c.xu[1]), xu3 = UNIFORM_D(c.xu[3]);
    const int fixT = UNIFORM(c.fixTime);
    // time scale: uniform.  FUSED: the trial value and its bound multipliers, stored by lane 0 below
    double t = z[l.t], ztL = z[l.ztL], ztU = z[l.ztU];
    if (FUSED && !fixT) {
        const double dt = sh.coef[0], dL = t - OB_TL, dU = OB_TU - t;
        const double zL = zstep(ztL, dL, dt, mu, fa.az), zU = zstep(ztU, dU, -dt, mu, fa.az);
        t = fma(fa.alpha, dt, t);
        ztL = clampz(zL, t - OB_TL, mu, fa.ks); ztU = clampz(zU, OB_TU - t, mu, fa.ks);
        SEAM(t); SEAM(ztL); SEAM(ztU);
    }
    t = UNIFORM_D(t);
    const double q = t * cTs;
    const double iq = UNIFORM_D(1.0 / q), it_ = UNIFORM_D(1.0 / t);          // uniform: one division each, the stage code multiplies
    double dinf = sh.Ap.dinf, pinf = sh.Ap.pinf, cmn = sh.Ap.cmin, cmx = sh.Ap.cmax, sumz = sh.Ap.sumz, sumy = sh.Ap.sumy, f = sh.Ap.f,
           th1 = sh.Ap.th1, bar = sh.Ap.bar;
    const int ok = sh.Ap.ok;
    const double *ocs = stg_base(sh);      // condensed obstacle sums of every stage, 12 doubles each (accumulated by part (a) in LDS)
    double red[13][OBCA_NL];
    // ---- (b) stages: one lane per stage
    PAR(lane) {
        double dmax = 0, pmax = 0, lcmn = 1e300, lcmx = -1e300, lsz = 0, lsy = 0, lf = 0, lth = 0, lbar = 0, lHtt = 0, lgtb = 0, lgtz = 0;
        if (FUSED && lane == 0) {
            zn[l.t] = t; zn[l.ztL] = ztL; zn[l.ztU] = ztU;
#pragma unroll
            for (int i = 0; i < 4; i++) zn[l.nu + i] = fma(fa.ay, sh.coef[1 + i], z[l.nu + i]);
        }
        for (int k = lane; k <= N; k += OB_NT) {
            BarAcc ba; bar_init(ba);                  // barrier distances of the stage: x (3 pairs), u (2), steering rate (1)
            const int kc = k < N ? k : N - 1, km = k >= 1 ? k - 1 : 0, kn = k + 1 < N ? k + 1 : kc;
            gdbl *rec = I.as + (size_t)k * OB_AS;
            double hz[8], hb[8];                      // gradient of the Lagrangian w.r.t. (X, Y, psi, v, w0, w1, delta, a): z-form (dual infeasibility) and barrier form (right-hand side)
            // ================================================================ section 1: the state x_k
            double x[4], zxL[4], zxU[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] = z[l.x + 4 * k + i]; zxL[i] = z[l.zxL + 4 * k + i]; zxU[i] = z[l.zxU + 4 * k + i]; }
            const double rx = I.prob[OB_HDR + k], ry = I.prob[OB_HDR + (N + 1) + k], ryaw = I.prob[OB_HDR + 2 * (N + 1) + k];
            // (issued one section ahead) section 3: inputs, steering slack, their multipliers
            double u[2] = {z[l.u + 2 * kc], z[l.u + 2 * kc + 1]}, um[2] = {z[l.u + 2 * km], z[l.u + 2 * km + 1]};
            double zuL[2] = {z[l.zuL + 2 * kc], z[l.zuL + 2 * kc + 1]}, zuU[2] = {z[l.zuU + 2 * kc], z[l.zuU + 2 * kc + 1]};
            double ss = z[l.ss + kc], yg = z[l.yg + kc], zssL = z[l.zssL + kc], zssU = z[l.zssU + kc];
            double du[2] = {0, 0}, dum[2] = {0, 0}, dss = 0, dyg = 0;
            if (FUSED) { du[0] = d[l.u + 2 * kc]; du[1] = d[l.u + 2 * kc + 1]; dum[0] = d[l.u + 2 * km]; dum[1] = d[l.u + 2 * km + 1]; dss = d[l.ss + kc]; dyg = d[l.yg + kc]; }
            if (FUSED) {
                // the trial point of this stage (steps: x in LDS, the rest in d).  Explicit fma wherever a trial value is formed: neighbouring stages (and the obstacle blocks) form
                // the same value again and a parked solve reads the stored one -- all of them must be the same bits
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double dx = g_traj[(size_t)k * 6 + i];
                    const double v = fma(fa.alpha, dx, x[i]);
                    if (i != 2 && k >= 1) {
                        const double xlo = i == 0 ? xl0 : (i == 1 ? xl1 : xl3), xhi = i == 0 ? xu0 : (i == 1 ? xu1 : xu3);
                        const double zL = zstep(zxL[i], x[i] - xlo, dx, mu, fa.az), zU = zstep(zxU[i], xhi - x[i], -dx, mu, fa.az);
                        zxL[i] = clampz(zL, v - xlo, mu, fa.ks); zxU[i] = clampz(zU, xhi - v, mu, fa.ks);
                    }
                    x[i] = v;
                    zn[l.x + 4 * k + i] = x[i]; zn[l.zxL + 4 * k + i] = zxL[i]; zn[l.zxU + 4 * k + i] = zxU[i];
                    SEAM(x[i]); SEAM(zxL[i]); SEAM(zxU[i]);
                }
            }
            lf += 1e-4 * x[3] * x[3] + 1e-3 * (x[0] - rx) * (x[0] - rx) + 1e-3 * (x[1] - ry) * (x[1] - ry) + cwpsi * (x[2] - ryaw) * (x[2] - ryaw);
            double Hd[4];                             // diagonal of the state block: tracking cost + bound barrier + delta_w
            {
                const double gx[4] = {2e-3 * (x[0] - rx), 2e-3 * (x[1] - ry), 2 * cwpsi * (x[2] - ryaw), 2e-4 * x[3]};
                const double hx[4] = {2e-3, 2e-3, 2 * cwpsi, 2e-4};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    hz[i] = gx[i]; hb[i] = gx[i];
                    double Sig = 0;
                    if (i != 2 && k >= 1) {
                        const double xlo = i == 0 ? xl0 : (i == 1 ? xl1 : xl3), xhi = i == 0 ? xu0 : (i == 1 ? xu1 : xu3);
                        B2 b = bound2(x[i], xlo, xhi, zxL[i], zxU[i], mu, 1, lcmn, lcmx, lsz);
                        Sig = b.Sig; hz[i] += b.gz; hb[i] += LSQ ? b.gz : b.gb;
                        bar_mul(ba, x[i] - xlo, xhi - x[i]);
                    }
                    Hd[i] = LSQ ? 1.0 : hx[i] + Sig + dw;
                }
            }
            SECTION();
            // ================================================================ section 2: condensed obstacle contributions of this stage (summed over the obstacles by part (a))
            double H00, H01, H02, H11, H12, H22;
            {
                double oc_[12]; const double *os = ocs + (size_t)k * OB_OC;
#pragma unroll
                for (int i = 0; i < 12; i++) oc_[i] = os[i];
                H00 = Hd[0] + oc_[0]; H01 = oc_[1]; H02 = oc_[2]; H11 = Hd[1] + oc_[3]; H12 = oc_[4]; H22 = Hd[2] + oc_[5];
                rec[AS_H + 0] = H00; rec[AS_H + 1] = H01; rec[AS_H + 2] = H02; rec[AS_H + 3] = H11; rec[AS_H + 4] = H12;      // final: stored now, not carried through the dynamics
#pragma unroll
                for (int i = 0; i < 3; i++) { hz[i] += oc_[6 + i]; hb[i] += oc_[6 + i] - oc_[9 + i]; }
            }
            double H33 = Hd[3];
            if (k == N) {
                // ---- terminal stage: x_N = xF with multiplier nu, costate pi_{N-1}
                double pi[4], nu4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { pi[i] = z[l.pi + 4 * kc + i]; nu4[i] = z[l.nu + i]; }
                if (FUSED) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { pi[i] = fma(fa.ay, (double)d[l.pi + 4 * kc + i], pi[i]); nu4[i] = fma(fa.ay, sh.coef[1 + i], nu4[i]); SEAM(pi[i]); SEAM(nu4[i]); }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double e = fabs(x[i] - c.xF[i]); pmax = fmax(pmax, e); lth += e;
                    const double r = pi[i] + nu4[i];
                    hz[i] += r; hb[i] += r;
                    dmax = fmax(dmax, fabs(hz[i]));
                    lsy += fabs(nu4[i]);
                }
                lbar += bar_log(ba);
                rec[AS_H + 5] = H22; rec[AS_H + 9] = H33;
                rec[AS_H + 6] = 0.0; rec[AS_H + 7] = 0.0; rec[AS_H + 8] = 0.0;
#pragma unroll
                for (int i = 10; i < 19; i++) rec[AS_H + i] = 0.0;
#pragma unroll
                for (int i = 0; i < 8; i++) { rec[AS_HB + i] = i < 4 ? hb[i] : 0.0; if (i >= 2) rec[AS_HT + i - 2] = 0.0; }
                continue;
            }
            // (issued one section ahead) section 4: costates and the next state
            double pi[4], pim[4], xn[4], dpi[4], dpim[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { pi[i] = z[l.pi + 4 * kc + i]; pim[i] = z[l.pi + 4 * km + i]; xn[i] = z[l.x + 4 * (kc + 1) + i]; dpi[i] = 0; dpim[i] = 0; }
            if (FUSED) {
#pragma unroll
                for (int i = 0; i < 4; i++) { dpi[i] = d[l.pi + 4 * kc + i]; dpim[i] = d[l.pi + 4 * km + i]; }
            }
            SECTION();
            // ================================================================ section 3: inputs u_k, their copy w_k = u_{k-1}, rate cost, bounds, steering-rate row
            if (FUSED) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const double lo = i ? OB_UL1 : OB_UL0, hi = i ? OB_UU1 : OB_UU0;
                    const double zL = zstep(zuL[i], u[i] - lo, du[i], mu, fa.az), zU = zstep(zuU[i], hi - u[i], -du[i], mu, fa.az), v = fma(fa.alpha, du[i], u[i]);
                    u[i] = v; zuL[i] = clampz(zL, v - lo, mu, fa.ks); zuU[i] = clampz(zU, hi - v, mu, fa.ks);
                    um[i] = fma(fa.alpha, dum[i], um[i]);
                }
                {
                    const double zL = zstep(zssL, ss + OB_SSB, dss, mu, fa.az), zU = zstep(zssU, OB_SSB - ss, -dss, mu, fa.az), v = fma(fa.alpha, dss, ss);
                    ss = v; zssL = clampz(zL, v + OB_SSB, mu, fa.ks); zssU = clampz(zU, OB_SSB - v, mu, fa.ks);
                }
                yg = fma(fa.ay, dyg, yg);
#pragma unroll
                for (int i = 0; i < 2; i++) { zn[l.u + 2 * k + i] = u[i]; zn[l.zuL + 2 * k + i] = zuL[i]; zn[l.zuU + 2 * k + i] = zuU[i]; }
                zn[l.ss + k] = ss; zn[l.zssL + k] = zssL; zn[l.zssU + k] = zssU; zn[l.yg + k] = yg;
#pragma unroll
                for (int i = 0; i < 2; i++) { SEAM(u[i]); SEAM(um[i]); SEAM(zuL[i]); SEAM(zuU[i]); }
                SEAM(ss); SEAM(yg); SEAM(zssL); SEAM(zssU);
            }
            double H44, H46, H55, H57, H66, H77, Ht4, Ht5, Ht6, Ht7;
            {
                const double w[2] = {k ? um[0] : 0.0, k ? um[1] : 0.0};
                const double cu[2] = {0.01, cwa};
                const double rr = 0.1 * (iq * iq), e1 = u[0] - w[0], e2 = u[1] - w[1], rv = rr * (e1 * e1 + e2 * e2);
                lf += 0.01 * u[0] * u[0] + cwa * u[1] * u[1] + rv;
                double Huu[2], Hww[2] = {0, 0}, Hwu[2] = {0, 0}, Htu[2] = {0, 0}, Htw[2] = {0, 0};
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const double ei = i ? e2 : e1, lo = i ? OB_UL1 : OB_UL0, hi = i ? OB_UU1 : OB_UU0;
                    const double gu = 2 * cu[i] * u[i] + 2 * rr * ei;
                    hz[6 + i] = gu; hb[6 + i] = gu; hz[4 + i] = -2 * rr * ei; hb[4 + i] = -2 * rr * ei;
                    B2 b = bound2(u[i], lo, hi, zuL[i], zuU[i], mu, 1, lcmn, lcmx, lsz);
                    hz[6 + i] += b.gz; hb[6 + i] += LSQ ? b.gz : b.gb;
                    bar_mul(ba, u[i] - lo, hi - u[i]);
                    Huu[i] = LSQ ? 1.0 : 2 * cu[i] + 2 * rr + b.Sig + dw;
                    if (!LSQ) { Hww[i] = 2 * rr; Hwu[i] = -2 * rr; }
                    if (!fixT && !LSQ) { Htu[i] = -4 * rr * ei * it_; Htw[i] = 4 * rr * ei * it_; }
                }
                if (!fixT) { lgtz += -2 * rv * it_; lgtb += -2 * rv * it_; if (!LSQ) lHtt += 6 * rv * (it_ * it_); }
                H44 = Hww[0]; H55 = Hww[1]; H46 = Hwu[0]; H57 = Hwu[1]; H66 = Huu[0]; H77 = Huu[1]; Ht4 = Htw[0]; Ht5 = Htw[1]; Ht6 = Htu[0]; Ht7 = Htu[1];
                {   // steering-rate row  g=(w0-delta)/(t Ts) - ss = 0, |ss|<=0.6   (ParkingSignedDist.jl:157-174)
                    const double g = (w[0] - u[0]) * iq;
                    const double gg[3] = {iq, -iq, fixT ? 0.0 : -g * it_};
                    B2 b = bound2(ss, -OB_SSB, OB_SSB, zssL, zssU, mu, 1, lcmn, lcmx, lsz);
                    bar_mul(ba, ss + OB_SSB, OB_SSB - ss);
                    lsy += fabs(yg);
                    const double rz = -yg + b.gz, rb = LSQ ? rz : -yg + b.gb;
                    dmax = fmax(dmax, fabs(rz));
                    const double res = g - ss; pmax = fmax(pmax, fabs(res)); lth += fabs(res);
                    const double Dss = LSQ ? 1.0 : b.Sig + dw, iDss = rcp_nr(Dss), sig = rcp_nr(iDss + dc), rg = (LSQ ? 0.0 : (SOC ? (double)sh.soc.csoc[(l.yg - l.pi) + k] : res)) + rb * iDss;
                    rec[AS_SIG] = sig; rec[AS_RG] = rg; rec[AS_GG] = gg[0]; rec[AS_GG + 1] = gg[1]; rec[AS_GG + 2] = gg[2];
                    rec[AS_DSS] = Dss; rec[AS_RSS] = rb;
                    hz[4] += gg[0] * yg; hb[4] += gg[0] * (yg + sig * rg); hz[6] += gg[1] * yg; hb[6] += gg[1] * (yg + sig * rg);
                    H44 += sig * gg[0] * gg[0]; H46 += sig * gg[0] * gg[1]; H66 += sig * gg[1] * gg[1];
                    if (!fixT) {
                        Ht4 += sig * gg[0] * gg[2] + (LSQ ? 0.0 : yg * -(iq * it_)); Ht6 += sig * gg[1] * gg[2] + (LSQ ? 0.0 : yg * (iq * it_));
                        lgtz += gg[2] * yg; lgtb += gg[2] * (yg + sig * rg); lHtt += sig * gg[2] * gg[2] + (LSQ ? 0.0 : yg * 2 * g * (it_ * it_));
                    }
                }
            }
            rec[AS_H + 12] = H44; rec[AS_H + 13] = H46; rec[AS_H + 14] = H55; rec[AS_H + 15] = H57; rec[AS_HT + 2] = Ht4; rec[AS_HT + 3] = Ht5; rec[AS_HB + 4] = hb[4]; rec[AS_HB + 5] = hb[5];      // final
            // (issued one section ahead) section 5: the next stage's inputs and steering multiplier, for the part of u_k's dual infeasibility that lives in stage k + 1
            double un[2] = {z[l.u + 2 * kn], z[l.u + 2 * kn + 1]}, ygn = z[l.yg + kn], dun[2] = {0, 0}, dygn = 0;
            if (FUSED) { dun[0] = d[l.u + 2 * kn]; dun[1] = d[l.u + 2 * kn + 1]; dygn = d[l.yg + kn]; }
            SECTION();
            // ================================================================ section 4: dynamics x_{k+1} - F(x_k,u_k,t) = 0, multiplier pi_k   (ParkingSignedDist.jl:139-155)
            if (FUSED) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    xn[i] = fma(fa.alpha, g_traj[(size_t)(kc + 1) * 6 + i], xn[i]); pi[i] = fma(fa.ay, dpi[i], pi[i]); pim[i] = fma(fa.ay, dpim[i], pim[i]);
                    zn[l.pi + 4 * k + i] = pi[i];
                    SEAM(xn[i]); SEAM(pi[i]); SEAM(pim[i]);
                }
            }
            double H23, H26, H27, H36, H37, H67, Ht2, Ht3;
            {
                Dyn dy; dyn_derivs(cTs, ciL, x, u, t, pi, dy);
                const bool ft = fixT;
#pragma unroll
                for (int j = 0; j < 5; j++) { rec[AS_DF + as_df(0, j)] = (j == 4 && ft) ? 0.0 : dy.dX[j]; rec[AS_DF + as_df(1, j)] = (j == 4 && ft) ? 0.0 : dy.dY[j]; }
#pragma unroll
                for (int j = 1; j < 5; j++) rec[AS_DF + as_df(2, j)] = (j == 4 && ft) ? 0.0 : dy.dP[j - 1];
                rec[AS_DF + as_df(3, 3)] = dy.dVa; rec[AS_DF + as_df(3, 4)] = ft ? 0.0 : dy.dVt;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double r = xn[i] - dy.F[i];
                    rec[AS_DD + i] = LSQ ? 0.0 : (SOC ? -(double)sh.soc.csoc[4 * k + i] : -r); pmax = fmax(pmax, fabs(r)); lth += fabs(r);
                    lsy += fabs(pi[i]);
                }
                H23 = 0; H26 = 0; H27 = 0; H36 = 0; H37 = 0; H67 = 0; Ht2 = 0; Ht3 = 0;
                if (!LSQ) {      // variables (psi, v, delta, a) = positions (2, 3, 6, 7) of the stage vector
                    H22 += -dy.h00; H23 = -dy.h01; H26 = -dy.h02; H27 = -dy.h03; H33 += -dy.h11; H36 = -dy.h12; H37 = -dy.h13; H66 += -dy.h22; H67 = -dy.h23;
                    if (!ft) { Ht2 = -dy.h04; Ht3 = -dy.h14; Ht6 += -dy.h24; Ht7 += -dy.h34; lHtt += -dy.h44; }
                }
                // J^T pi: x_k rows get +pi_{k-1} - A_k^T pi_k ; u_k rows -B_k^T pi_k ; t gets -Ft^T pi
                const double ATpi[4] = {pi[0], pi[1], (pi[2] + dy.dX[0] * pi[0]) + dy.dY[0] * pi[1], ((pi[3] + dy.dX[1] * pi[0]) + dy.dY[1] * pi[1]) + dy.dP[0] * pi[2]};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double r = (k >= 1 ? pim[i] : 0.0) - ATpi[i];
                    hz[i] += r; hb[i] += r;
                    if (k >= 1 && fabs(hz[i]) > dmax) dmax = fabs(hz[i]);
                }
                {
                    const double r6 = -((dy.dX[2] * pi[0] + dy.dY[2] * pi[1]) + dy.dP[1] * pi[2]);
                    const double r7 = -(((dy.dX[3] * pi[0] + dy.dY[3] * pi[1]) + dy.dP[2] * pi[2]) + dy.dVa * pi[3]);
                    hz[6] += r6; hb[6] += r6; hz[7] += r7; hb[7] += r7;
                }
                if (!ft) {
                    const double r = ((dy.dX[4] * pi[0] + dy.dY[4] * pi[1]) + dy.dP[3] * pi[2]) + dy.dVt * pi[3];
                    lgtz -= r; lgtb -= r;
                }
            }
            SECTION();
            // ================================================================ section 5: dual infeasibility of u_k (own part + copy part living in stage k+1), barrier, stores
            {
                if (FUSED) { un[0] = fma(fa.alpha, dun[0], un[0]); un[1] = fma(fa.alpha, dun[1], un[1]); ygn = fma(fa.ay, dygn, ygn); SEAM(un[0]); SEAM(un[1]); SEAM(ygn); }
                const double rr = 0.1 * (iq * iq);
                double wn[2] = {0, 0};
                if (k + 1 < N) {
                    wn[0] = -2 * rr * (un[0] - u[0]) + iq * ygn;
                    wn[1] = -2 * rr * (un[1] - u[1]);
                }
#pragma unroll
                for (int i = 0; i < 2; i++) { const double tot = hz[6 + i] + wn[i]; dmax = fmax(dmax, fabs(tot)); }
            }
            lbar += bar_log(ba);
            rec[AS_H + 5] = H22; rec[AS_H + 6] = H23; rec[AS_H + 7] = H26; rec[AS_H + 8] = H27; rec[AS_H + 9] = H33; rec[AS_H + 10] = H36; rec[AS_H + 11] = H37;
            rec[AS_H + 16] = H66; rec[AS_H + 17] = H67; rec[AS_H + 18] = H77;
            rec[AS_HB + 0] = hb[0]; rec[AS_HB + 1] = hb[1]; rec[AS_HB + 2] = hb[2]; rec[AS_HB + 3] = hb[3]; rec[AS_HB + 6] = hb[6]; rec[AS_HB + 7] = hb[7];
            rec[AS_HT + 0] = Ht2; rec[AS_HT + 1] = Ht3; rec[AS_HT + 4] = Ht6; rec[AS_HT + 5] = Ht7;
        }
        red[0][LI(lane)] = dmax; red[1][LI(lane)] = pmax; red[3][LI(lane)] = lcmn; red[12][LI(lane)] = lcmx;
        red[4][LI(lane)] = lsz; red[5][LI(lane)] = lsy; red[6][LI(lane)] = lf; red[7][LI(lane)] = lth;
        red[8][LI(lane)] = lbar; red[9][LI(lane)] = lHtt; red[10][LI(lane)] = lgtb; red[11][LI(lane)] = lgtz;
    }
    dinf = fmax(dinf, wred_max(red[0])); pinf = fmax(pinf, wred_max(red[1])); cmn = fmin(cmn, wred_min(red[3])); cmx = fmax(cmx, wred_max(red[12]));
    sumz += wred_sum(red[4]); sumy += wred_sum(red[5]); f += wred_sum(red[6]); th1 += wred_sum(red[7]);
    bar += wred_sum(red[8]);
    double Htt = wred_sum(red[9]), gtb = wred_sum(red[10]), gtz = wred_sum(red[11]);
    SYNC();
    int nb = 6 * N + 4 * N + 2 * N + (M + (c.dist ? 6 : 5) * nOb) * (N + 1);
    int nm = 4 * N + 4 + N + 4 * nOb * (N + 1);
    if (!c.fixTime) {
        double d2 = 0;
        B2 b = bound2(t, OB_TL, OB_TU, ztL, ztU, mu, N + 1, cmn, cmx, d2);
        sumz += (N + 1) * (fabs(ztL) + fabs(ztU));
        nb += 2 * (N + 1);
        double gf = (N + 1) * (0.5 + 2 * t);
        Htt += LSQ ? (double)(N + 1) : 2.0 * (N + 1) + b.Sig + dw;      // (least-squares system: t stands for the N + 1 timeScale variables of the reference's model, N + 1 unit diagonal entries)
        gtb += gf + (LSQ ? b.gz : b.gb); gtz += gf + b.gz;
        f += (N + 1) * (0.5 * t + t * t);
        bar += (N + 1) * log((t - OB_TL) * (OB_TU - t));
        dinf = fmax(dinf, fabs(gtz));
    } else { Htt = 1.0; gtb = 0; }
    out.ok = ok; out.dinf = dinf; out.pinf = pinf; out.cinf0 = fmax(fabs(cmn), fabs(cmx)); out.cmin = cmn; out.cmax = cmx;      // (largest |s z| of all complementarity pairs = the larger of the two extreme products in magnitude) out.sumy = sumy; out.sumz = sumz;
    out.f = f; out.th1 = th1; out.bar = bar; out.Htt = Htt; out.gtb = gtb; out.nb = nb; out.nm = nm;
    PROF(I, FUSED ? PF_APPLY : PF_ASM_STAGE);
}

// ---------------------------------------------------------------- Riccati backward sweep
// Stage k is condensed onto (x_k, w_k=u_{k-1}); six right-hand sides (main, t, nu1..4) ride along as extra columns and the
// bilinear constants B(a,b) of the cost-to-go give every entry of the 5x5 (t, nu) border without a forward pass per column.
// Returns 1 if every 2x2 input block is positive definite.
OBCA_FN void pair_of(int p, int &a_, int &b_) {   // p-th pair (a<=b) of the 6 columns, row-major upper triangle
    a_ = (p >= 6) + (p >= 11) + (p >= 15) + (p >= 18) + (p >= 20);
    b_ = p - (6 * a_ - a_ * (a_ - 1) / 2) + a_;
}

// unpacked stage data in LDS (one of two buffers): H (8x8 full), FA' = [Fm | off]' (14x6: FA'[cc * 6 + a]), hc (8x6)
#define SG_H 0
#define SG_FA 64
#define SG_HC 148
#define SG_SIZE 196
// Staged values of a stage: 196, of which 92 are constants of the layout (identity / zero pattern of FA, unused right-hand-side columns of hc):
// those are written ONCE per sweep into both buffers (stage_unpack_constants); the 104 that change with the stage -- H (64, the symmetric entries
// twice), the 24 bicycle-model entries of FA, the 16 gradient / time columns of hc -- of which 64 can be non-zero (as_h, as_df) -- are gathered per stage, ONE per lane
// (value = kc + rec[idx]); which position a lane serves is tabulated once per solve (Shared::upos).
#define SG_NVAR 64
struct UnpackPlan { int idx, dst; double kc; };
OBCA_FN void stage_unpack_item(int it, int &idx, int &dst, double &fl, double &kc) {     // all 196 positions: what is stored where (fl = 0: the constant kc)
    idx = AS_DD; fl = 0.0; kc = 0.0; dst = SG_SIZE;              // default: harmless gather, store to the pad slot behind the buffer
    if (it < 64) { dst = SG_H + it; if (as_h(it >> 3, it & 7) >= 0) { idx = AS_H + as_h(it >> 3, it & 7); fl = 1.0; } }
    else if (it < 64 + 84) {
        const int e = it - 64, a_ = e / 14, cc = e % 14; dst = SG_FA + cc * 6 + a_;      // FA is staged TRANSPOSED: row cc of FA' = column cc of FA, contiguous
        if (cc < 8) {
            if (a_ < 4) {
                if (cc < 4) kc = (a_ == cc) ? 1.0 : 0.0;
                const int jc = cc == 2 ? 0 : (cc == 3 ? 1 : (cc == 6 ? 2 : (cc == 7 ? 3 : -1)));
                if (jc >= 0 && as_df(a_, jc) >= 0) { idx = AS_DF + as_df(a_, jc); fl = 1.0; }
            } else kc = (cc == a_ + 2) ? 1.0 : 0.0;
        } else if (a_ < 4) { const int col = cc - 8; if (col == 0) { idx = AS_DD + a_; fl = 1.0; } if (col == 1) { idx = AS_DF + as_df(a_, 4); fl = 1.0; } }
    } else if (it < SG_SIZE) {
        const int e = it - 148, i = e / OB_NC, cc = e % OB_NC; dst = SG_HC + e;
        if (cc == 0) { idx = AS_HB + i; fl = 1.0; }
        if (cc == 1 && i >= 2) { idx = AS_HT + i - 2; fl = 1.0; }
    }
}
// The item maps above are irregular (a divergent switch per position), so they are evaluated ONCE per solve into two small LDS tables; a sweep only reads them:
//   upl[lane]    = (idx << 8) | dst | one << 16 : the stage-dependent position this lane gathers (there are exactly SG_NVAR = OB_NT of them)
//   ucn[r][lane] = dst | one << 16, or -1       : the constant positions (+ the pad slot) this lane rewrites at the start of a sweep (the stage buffers share
//                                                  their LDS with the forward sweep's pair maps)
#define SG_NCONST_ROUNDS 3      // (SG_SIZE + 1 - SG_NVAR = 133 constant positions over 64 lanes)
OBCA_FN void init_unpack_table(Shared &sh) {
    PAR(lane) {
        int nv_ = 0, nc_ = 0;
        for (int r = 0; r < SG_NCONST_ROUNDS; r++) sh.ucn[r][lane] = -1;
        for (int it = 0; it <= SG_SIZE; it++) {
            int idx, dst; double fl, kc; stage_unpack_item(it, idx, dst, fl, kc);
            const int one = kc != 0.0 ? (1 << 16) : 0;
            if (fl != 0.0) { if (nv_ == lane) sh.upl[lane] = (idx << 8) | dst | one; nv_++; }
            else { if (nc_ % OB_NT == lane && nc_ / OB_NT < SG_NCONST_ROUNDS) sh.ucn[nc_ / OB_NT][lane] = dst | one; nc_++; }
        }
    }
}
OBCA_FN void stage_unpack_plan(const Shared &sh, int lane, UnpackPlan &p) { const int w = sh.upl[lane]; p.idx = (w >> 8) & 0xff; p.dst = w & 0xff; p.kc = (w >> 16) & 1 ? 1.0 : 0.0; }
OBCA_FN void stage_unpack_constants(const Shared &sh, double *sg, int lane) {     // once per sweep, both buffers (+ the pad slot)
#pragma unroll
    for (int r = 0; r < SG_NCONST_ROUNDS; r++) {
        const int w = sh.ucn[r][lane];
        if (w >= 0) { const double kc = (w >> 16) & 1 ? 1.0 : 0.0; sg[w & 0xffff] = kc; sg[OB_STG + (w & 0xffff)] = kc; }
    }
}
OBCA_FN void stage_unpack_load(const gdbl *rec, const UnpackPlan &p, double &v) { v = rec[p.idx]; }      // an independent, branch-free gather; the raw value is only touched at store time
OBCA_FN void stage_unpack_store(double *sg, const UnpackPlan &p, const double v) { sg[p.dst] = p.kc + v; }

// A dependent fp64 operation costs ~45 clock ticks when an instance runs alone on its CU (one wavefront per SIMD: nothing fills the pipeline;
// tools/micro/lds_barrier_latency.hip), so the short dot products of the sequential sweeps are summed as a tree (depth 4 instead of 7).
OBCA_FN double dot6_tree(double init, double a0, double b0, double a1, double b1, double a2, double b2, double a3, double b3, double a4, double b4,
                         double a5, double b5) {
    const double t0 = fma(a1, b1, a0 * b0), t1 = fma(a3, b3, a2 * b2), t2 = fma(a5, b5, fma(a4, b4, init));
    return (t0 + t1) + t2;
}
OBCA_FN double dot4_tree(double init, const double (&a)[4], const double *b) { return fma(a[1], b[1], a[0] * b[0]) + fma(a[3], b[3], fma(a[2], b[2], init)); }
// NV contiguous, 16-byte aligned doubles from LDS as ds_read_b128
template <int NV>
OBCA_FN void ldv(const double *q, double (&v)[NV]) {
#ifdef OBCA_EMU
    for (int i = 0; i < NV; i++) v[i] = q[i];
#else
    const double2 *q2 = (const double2 *)__builtin_assume_aligned(q, 16);
#pragma unroll
    for (int i = 0; i < NV / 2; i++) { const double2 t = q2[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
#endif
}
#ifndef RIC_D
#define RIC_D 4   // stage records are gathered from HBM this many stages before they are needed (memory latency >> one stage of math)
#endif
// One stage of the sweep on the 64 lanes of the wavefront: three short LDS phases (T = P [F|off] + [0|p];  Qhat = [H|hc] + F'T;  eliminate u_k), ONE item per lane and
// phase, wave-local LDS ordering in between (no cross-wavefront synchronisation: the instance IS one wavefront).  Every lane runs the SAME straight-line code in every phase:
// what differs between the item kinds of a phase is only where the operands live, and that is a per-lane table of LDS offsets built once per sweep (RicItem).
// A phase costs what its one wavefront ISSUES (a 16-byte LDS read ~16 clocks, a dependent fp64 operation ~47: profiles/r03_ab_reciprocal_and_early_quu.txt), so the items are
// cut down to the products that are not structure (rounds 1-3 computed all 96 / 124 / 93 entries, two per lane):
//   * FA = [F | off] has the unit columns 0, 1 (X, Y), the zero columns 4, 5 (the input copy w: x+ does not depend on it) and 10..13 (the nu right-hand sides): the columns
//     0, 1, 4, 5, 10..13 of T are columns of P, zero, or columns of p -- phase B reads them where they are; phase A forms the six others (36 items).
//   * rows 4, 5 and columns 4, 5 of Qhat are [H | hc] itself (F has nothing there): phase C reads them from the stage buffer; rows 0, 1 of Qhat are [H | hc] + T rows 0, 1:
//     for the copied columns that is one more phase-A item each (16), for the others a phase-B item with a unit-vector operand (12).  Phase B: rows psi, v, delta, a over the
//     twelve live columns (48) + those 12 + 4.
//   * P is symmetric: phase C forms the 21 entries i <= c once and stores them twice (exactly symmetric, where rounds 1-3 computed both halves), and the 36 entries of p.
//   * the static parts of the bilinear constants ACCUMULATE in their own slots over the stages (the item's initial value is its previous sum); u1[m][b] = off_m . T(8+b)
//     equals u2[m][b] = off_m . p(b) for b >= 2 (T(8+b) = p(b) there) and is not formed.  The dynamic part - Qhat_u(a)' Quu^-1 Qhat_u(b) is not needed before the sweep ends:
//     every stage leaves Qhat_u of its six right-hand sides, Quu and 1 / det in LDS (RIC_BD doubles) and the 21 sums over the stages are formed afterwards, three lanes per pair.
// PIPE = 1: steady state of the software pipeline -- the last phase first retires the gather of stage k-1 (issued RIC_D stages ago into
// nv[..][slot]) into the LDS buffer and re-issues the slot for stage k-1-RIC_D.  Every global load / store is issued unconditionally
// (clamped stage index, dummy slot RS_PAD for the lanes without an item) and the loop has a single exit: with no branch around a
// memory operation the compiler's in-order vmcnt bookkeeping stays exact and old gathers retire without draining the younger ones.
struct RicItem {      // offsets in doubles from the start of Shared
    int a_a, a_b, a_i, a_j, a_d, a_sg;      // phase A: the two operand vectors (4 contiguous doubles each), two initial values, destination; *_sg: bit 0/1/2 = A/B/first initial
    int b_a, b_b, b_i, b_j, b_d, b_sg;      //          value live in the stage buffer (its parity offset is added at run time); phase B likewise
    int c_x6, c_x7, c_q6, c_q7, c_base, c_sg, c_d1, c_d2, c_bd, c_rv, c_rv2, c_rk0, c_rk1;   // phase C: see riccati_stage (c_sg: bits 0..4 = x6, x7, q6, q7, base live in the stage buffer)
};
#define RIC_BD 16       // per stage: Qhat_u (rows 6, 7) of the six right-hand sides, then q00, q10, q11, 1 / det
OBCA_FN int ric_qsrc(int oQ, int oSG, int r, int c, int bit, int &sg) {      // where phase C finds Qhat[r][c]: rows / columns 4, 5 are [H | hc] in the stage buffer
    if (r == 4 || r == 5 || c == 4 || c == 5) { sg |= bit; return oSG + (c < 8 ? SG_H + r * 8 + c : SG_HC + r * OB_NC + (c - 8)); }
    return oQ + r * 14 + c;
}
OBCA_FN void ric_item(const Shared &sh, int lane, RicItem &p) {
    const double *L = (const double *)&sh; const RicLds &rl = ric_lds(sh);
    const int oPn = (int)(rl.Pn - L), opn = (int)(rl.pn - L), oQ = (int)(rl.Qhat - L), osB = (int)(rl.sB - L), oT = (int)(rl.TT - L),
              oSG = (int)(ric_sg0(sh) - L), oZ = (int)(&rl.zero - L), oZ6 = (int)(rl.zero6 - L), oD = (int)(&rl.dump - L), oD4 = (int)(rl.dump4 - L);
    const int S6[6] = {2, 3, 6, 7, 8, 9}, R8[8] = {0, 1, 4, 5, 10, 11, 12, 13}, I4[4] = {2, 3, 6, 7}, C12[12] = {0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13};
    // A: items 0..35 T[a][cc] = [cc >= 8] p[a][cc-8] + P[a][:] . FA[:][cc] for the six live columns (stored as T'[cc][a]);  36..47 u2[m][b] += FA[:][8+m] . p[:][b];
    //    48..63 Qhat[a][cc] = [H | p][a][cc] + P[a][:] . FA[:][cc] for a = 0, 1 and the copied columns (FA[:][cc] is a unit vector or zero there)
    // (F acts through its rows 0..3 only -- the bicycle model -- plus the selector rows w+ = u, whose coefficient is exactly 1: every product is a 4-term dot product with up
    //  to two initial values, and the dependency chain of an item is three operations deep instead of four)
    p.a_a = oZ6; p.a_b = oZ6; p.a_i = oZ; p.a_j = oZ; p.a_d = oD; p.a_sg = 0;
    if (lane < 36) { const int cc = S6[lane / 6], a_ = lane % 6; p.a_a = oPn + a_ * 6; p.a_b = oSG + SG_FA + cc * 6; p.a_sg = 2;
                     p.a_i = cc < 8 ? oZ : opn + (cc - 8) * 6 + a_; p.a_j = cc == 6 ? oPn + a_ * 6 + 4 : (cc == 7 ? oPn + a_ * 6 + 5 : oZ); p.a_d = oT + cc * 6 + a_; }
    else if (lane < 48) { const int m = (lane - 36) / 6, b_ = (lane - 36) % 6; p.a_a = oSG + SG_FA + (8 + m) * 6; p.a_sg = 1; p.a_b = opn + b_ * 6; p.a_i = p.a_d = osB + 12 + m * 6 + b_; }
    else { const int a_ = (lane - 48) / 8, cc = R8[(lane - 48) % 8]; p.a_a = oPn + a_ * 6; p.a_b = oSG + SG_FA + cc * 6; p.a_sg = 2; p.a_d = oQ + a_ * 14 + cc;
           if (cc < 8) { p.a_i = oSG + SG_H + a_ * 8 + cc; p.a_sg |= 4; } else p.a_i = opn + (cc - 8) * 6 + a_; }
    // B: items 0..47 Qhat[i][cc] = [H | hc][i][cc] + FA[:][i] . T[:][cc] for the rows psi, v, delta, a and the twelve live columns;  48..59 the same for rows X, Y and the six
    //    columns phase A formed;  60..63 u1[m][b] += FA[:][8+m] . T[:][8+b], b = 0, 1.  T[:][cc] is read where it lives: a row of P (symmetric), a column of p', or T'
    p.b_a = oZ6; p.b_b = oZ6; p.b_i = oZ; p.b_j = oZ; p.b_d = oD; p.b_sg = 0;
    if (lane < 60) {
        const int i = lane < 48 ? I4[lane / 12] : (lane - 48) / 6, cc = lane < 48 ? C12[lane % 12] : S6[(lane - 48) % 6];
        p.b_a = oSG + SG_FA + i * 6; p.b_sg = 1 | 4;
        p.b_b = cc < 2 ? oPn + cc * 6 : (cc >= 10 ? opn + (cc - 8) * 6 : oT + cc * 6);
        p.b_j = i == 6 ? p.b_b + 4 : (i == 7 ? p.b_b + 5 : oZ);                 // the selector rows of F: + T[4][cc] for the delta row, + T[5][cc] for the a row
        p.b_i = oSG + (cc < 8 ? SG_H + i * 8 + cc : SG_HC + i * OB_NC + (cc - 8)); p.b_d = oQ + i * 14 + cc;
    } else { const int m = (lane - 60) / 2, b_ = (lane - 60) % 2; p.b_a = oSG + SG_FA + (8 + m) * 6; p.b_sg = 1; p.b_b = oT + (8 + b_) * 6; p.b_i = p.b_d = osB + m * 6 + b_; }
    // C: value = base + (X6 n0 + X7 n1) / det with (n0, n1) = adj(Quu) applied to rows 6, 7 of the item's column of Qhat
    //    items 0..20 P[i][cc], i <= cc (stored twice);  21..56 p[i][c] (stored transposed);  the items (0, c) carry the gains of their column
    p.c_x6 = oZ; p.c_x7 = oZ; p.c_q6 = oZ; p.c_q7 = oZ; p.c_base = oZ; p.c_sg = 0; p.c_d1 = oD; p.c_d2 = oD; p.c_bd = -1; p.c_rv = RS_PAD; p.c_rv2 = RS_PAD; p.c_rk0 = RS_PAD; p.c_rk1 = RS_PAD;
    (void)oD4;
    if (lane < 57) {
        int r, i, cc;
        if (lane < 21) { r = 0; pair_of(lane, i, cc); } else { r = 1; i = (lane - 21) / 6; cc = (lane - 21) % 6; }
        const int qc = r ? cc + 8 : cc;
        p.c_x6 = ric_qsrc(oQ, oSG, i, 6, 1, p.c_sg); p.c_x7 = ric_qsrc(oQ, oSG, i, 7, 2, p.c_sg);
        p.c_q6 = ric_qsrc(oQ, oSG, 6, qc, 4, p.c_sg); p.c_q7 = ric_qsrc(oQ, oSG, 7, qc, 8, p.c_sg); p.c_base = ric_qsrc(oQ, oSG, i, qc, 16, p.c_sg);
        if (r) { p.c_d1 = opn + cc * 6 + i; if (i < 4) p.c_rv = RS_PV + i * 6 + cc; if (i == 0) { p.c_rk0 = RS_KF + cc; p.c_rk1 = RS_KF + OB_NC + cc; p.c_bd = 2 * cc; } }
        else {
            p.c_d1 = oPn + i * 6 + cc; if (i != cc) p.c_d2 = oPn + cc * 6 + i;
            if (i < 4) p.c_rv = RS_PX + i * 6 + cc;                              // rows 0..3 of P go to HBM (the costate recovery reads them)
            if (cc < 4 && i != cc) p.c_rv2 = RS_PX + cc * 6 + i;
            if (i == 0) { p.c_rk0 = RS_K + cc; p.c_rk1 = RS_K + 6 + cc; }
        }
    }
}
// four contiguous, 16-byte aligned doubles from LDS: two ds_read_b128
OBCA_FN void ld4(const double *q, double (&v)[4]) {
#ifdef OBCA_EMU
    for (int i = 0; i < 4; i++) v[i] = q[i];
#else
    const double2 *q2 = (const double2 *)__builtin_assume_aligned(q, 16);
    const double2 a = q2[0], b = q2[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
#endif
}
// i1 + i2 + a . b over four terms, three dependent operations deep (two chains of two fma, one add)
OBCA_FN double dot4_two(double i1, double i2, const double (&a)[4], const double (&b)[4]) { return fma(a[1], b[1], fma(a[0], b[0], i1)) + fma(a[3], b[3], fma(a[2], b[2], i2)); }
// six contiguous, 16-byte aligned doubles from LDS: three ds_read_b128
OBCA_FN void ld6(const double *q, double (&v)[6]) {
#ifdef OBCA_EMU
    for (int i = 0; i < 6; i++) v[i] = q[i];
#else
    const double2 *q2 = (const double2 *)__builtin_assume_aligned(q, 16);
    const double2 a = q2[0], b = q2[1], c = q2[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
#endif
}
template <int PIPE>
OBCA_FN int riccati_stage(const Inst &I, Shared &sh, const int k, const UnpackPlan (&plan)[OBCA_NLT], const RicItem (&rp)[OBCA_NLT],
                          double (&nv)[OBCA_NLT][RIC_D], const int slot, double *sg0) {
    double *L = (double *)&sh;
    const int sgo = (k & 1) * OB_STG;         // which of the two stage buffers holds stage k
    // In every phase all LDS reads are issued before the first LDS write of the phase (a write may alias a later read as far as the compiler
    // knows; reads that follow a write would wait for their own round trip).
    PAR(lane) {   // phase A
        const RicItem &p = rp[LI(lane)];
        double A[4], B[4]; ld4(L + p.a_a + ((p.a_sg & 1) ? sgo : 0), A); ld4(L + p.a_b + ((p.a_sg & 2) ? sgo : 0), B);
        L[p.a_d] = dot4_two(L[p.a_i + ((p.a_sg & 4) ? sgo : 0)], L[p.a_j], A, B);
    }
    LDS_SYNC();
    double vB[OBCA_NLT];
    PAR(lane) {   // phase B
        const RicItem &p = rp[LI(lane)];
        double A[4], B[4]; ld4(L + p.b_a + ((p.b_sg & 1) ? sgo : 0), A); ld4(L + p.b_b, B);
        vB[LI(lane)] = dot4_two(L[p.b_i + ((p.b_sg & 4) ? sgo : 0)], L[p.b_j], A, B);
        L[p.b_d] = vB[LI(lane)];
    }
    // Quu = [q00 q10; q10 q11] must be positive definite (q00 > 0, det > 0).  Its inverse is adj(Quu) / det: ONE division.  The three entries come straight out of the
    // registers of the lanes that formed them (phase-B items (delta, delta), (a, delta), (a, a) = lanes 28, 40, 41), so that the pivot test and the division run in the
    // shadow of phase B's LDS round trip instead of behind it
    const double q00 = WV_READLANE(vB, 28), q10 = WV_READLANE(vB, 40), q11 = WV_READLANE(vB, 41);
    LDS_SYNC();
    PROF_FINE(I, PF_RIC_P1);
    const double det = q00 * q11 - q10 * q10;
    const int ok = UNIFORM((q00 > 0) && (det > 0) ? 1 : 0);        // (no early exit; after a failed pivot the rest of the group runs on garbage)
    const double idet = rcp_nr(det);
    gdbl *ro = I.rs + (size_t)k * OB_RS;
    double *bd = g_traj + (size_t)k * RIC_BD;      // per-stage border data: at the start of the dynamic block (the trajectory is dead during the sweep)
    PAR(lane) {   // phase C
        const RicItem &p = rp[LI(lane)];
        const double x6 = L[p.c_x6 + ((p.c_sg & 1) ? sgo : 0)], x7 = L[p.c_x7 + ((p.c_sg & 2) ? sgo : 0)], q6 = L[p.c_q6 + ((p.c_sg & 4) ? sgo : 0)],
                     q7 = L[p.c_q7 + ((p.c_sg & 8) ? sgo : 0)], ba = L[p.c_base + ((p.c_sg & 16) ? sgo : 0)];
        const double n0 = fma(q10, q7, -(q11 * q6)), n1 = fma(q10, q6, -(q00 * q7));       // det * gains of this column
        const double v = fma(fma(x6, n0, x7 * n1), idet, ba);
        if (PIPE) {
            const int kp = k > 0 ? k - 1 : 0, kl = k - 1 - RIC_D > 0 ? k - 1 - RIC_D : 0;
            stage_unpack_store(sg0 + (kp & 1) * OB_STG, plan[LI(lane)], nv[LI(lane)][slot]);
            stage_unpack_load(I.as + (size_t)kl * OB_AS, plan[LI(lane)], nv[LI(lane)][slot]);
        }
        L[p.c_d1] = v; L[p.c_d2] = v;
        if (p.c_bd >= 0) { bd[p.c_bd] = q6; bd[p.c_bd + 1] = q7; if (p.c_bd == 0) { bd[12] = q00; bd[13] = q10; bd[14] = q11; bd[15] = idet; } }
        ro[p.c_rv] = v; ro[p.c_rv2] = v; ro[p.c_rk0] = (double)(n0 * idet); ro[p.c_rk1] = (double)(n1 * idet);
    }
    LDS_SYNC();
    PROF_FINE(I, PF_RIC_P2);
    return ok;
}

template <int SOC = 0>      // SOC = 1: the terminal row enters with c_soc
OBCA_FN int riccati_body(const Inst &I, Shared &sh, double rho) {   // all lanes
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = UNIFORM(c.N);
    const gdbl *z = I.z;
    double nv[OBCA_NLT][RIC_D];   // software pipeline, RIC_D stages deep; the slot of a stage is fixed by the unrolled loop below
    double *sg0 = ric_sg0(sh);       // the two stage buffers (dynamic LDS; in front of them the per-stage border data, behind them the operands)
    RicLds &rl = ric_lds(sh);
    UnpackPlan plan[OBCA_NLT]; RicItem rp[OBCA_NLT];
    PAR(lane) {   // terminal cost-to-go
        stage_unpack_plan(sh, lane, plan[LI(lane)]); ric_item(sh, lane, rp[LI(lane)]);
        stage_unpack_constants(sh, sg0, lane);
        if (lane == 0) { rl.zero = 0.0; rl.dump = 0.0; }
        if (lane < 6) rl.zero6[lane] = 0.0;
        if (lane < 24) rl.sB[lane] = 0.0;                      // the static parts of the bilinear constants accumulate here
        const gdbl *rec = I.as + (size_t)N * OB_AS;
        if (lane < 36) {
            int i = lane / 6, j = lane % 6;
            double v = as_h(i, j) >= 0 ? rec[AS_H + as_h(i, j)] : 0.0;
            if (i == j && i < 4) v += rho;
            rl.Pn[lane] = v;
        }
        if (lane < 6) {
            double e = lane < 4 ? (SOC ? -(double)sh.soc.csoc[(l.nu - l.pi) + lane] : -(z[l.x + 4 * N + lane] - c.xF[lane])) : 0.0;
            rl.pn[0 * 6 + lane] = rec[AS_HB + lane] - (lane < 4 ? rho * e : 0.0);      // (p is kept transposed: pn[c * 6 + a])
            rl.pn[1 * 6 + lane] = lane >= 2 ? rec[AS_HT + lane - 2] : 0.0;
            for (int cc = 0; cc < 4; cc++) rl.pn[(2 + cc) * 6 + lane] = (lane == cc) ? 1.0 : 0.0;
        }
    }
    LDS_SYNC();
    // head: N mod RIC_D stages with synchronous gathers, so that the pipelined loop below runs whole groups of RIC_D stages
    int k = N - 1, ok = 1;
    for (; k >= 0 && (k + 1) % RIC_D != 0 && ok; k--) {
        PAR(lane) { double v; stage_unpack_load(I.as + (size_t)k * OB_AS, plan[LI(lane)], v); stage_unpack_store(sg0 + (k & 1) * OB_STG, plan[LI(lane)], v); }
        LDS_SYNC();
        ok = riccati_stage<0>(I, sh, k, plan, rp, nv, 0, sg0);
    }
    if (ok && k >= 0) {
        PAR(lane) {   // unpack stage k; start the gathers of stages k-1 .. k-RIC_D; enter the loop with nothing in flight
            double v; stage_unpack_load(I.as + (size_t)k * OB_AS, plan[LI(lane)], v);
            stage_unpack_store(sg0 + (k & 1) * OB_STG, plan[LI(lane)], v);
#pragma unroll
            for (int j = 0; j < RIC_D; j++) { const int st = k - 1 - j > 0 ? k - 1 - j : 0; stage_unpack_load(I.as + (size_t)st * OB_AS, plan[LI(lane)], nv[LI(lane)][(j + 1) % RIC_D]); }
#ifndef OBCA_EMU
#pragma unroll
            for (int j = 0; j < RIC_D; j++) asm volatile("" : "+v"(nv[0][j]));
#endif
        }
        LDS_SYNC();
        for (int kb = k; kb >= RIC_D - 1 && ok; kb -= RIC_D) {
#pragma unroll
            for (int ju = 0; ju < RIC_D; ju++) ok &= riccati_stage<1>(I, sh, kb - ju, plan, rp, nv, (ju + 1) % RIC_D, sg0);
        }
    }
    if (!ok) { PROF(I, PF_RIC_BWD); return 0; }
    // bilinear constants: B(a,b) = sum over the stages of Qhat_u(a) . (adj(Quu) Qhat_u(b)) / det + the accumulated static parts; lane 3 p + q sums every third stage of pair p
    PAR(lane) {
        if (lane < 63) {
            int a_, b_; pair_of(lane / 3, a_, b_);
            const double *bd = g_traj;
            double acc = 0;
            for (int kk = lane % 3; kk < N; kk += 3) {
                const double *r = bd + (size_t)kk * RIC_BD;
                const double q6a = r[2 * a_], q7a = r[2 * a_ + 1], q6b = r[2 * b_], q7b = r[2 * b_ + 1], q00 = r[12], q10 = r[13], q11 = r[14], idet = r[15];
                const double n0 = fma(q10, q7b, -(q11 * q6b)), n1 = fma(q10, q6b, -(q00 * q7b));
                acc = fma(fma(q6a, n0, q7a * n1), idet, acc);
            }
            rl.TT[lane] = acc;
        }
    }
    LDS_SYNC();
    PAR(lane) {
        if (lane < 21) {
            int a_, b_; pair_of(lane, a_, b_);
            double v = (rl.TT[3 * lane] + rl.TT[3 * lane + 1]) + rl.TT[3 * lane + 2];
            if (a_ < 2) v += b_ < 2 ? rl.sB[a_ * 6 + b_] : rl.sB[12 + a_ * 6 + b_];      // off_a . (P off_b + p_b): u1[a][b], = u2[a][b] for the right-hand sides b >= 2
            if (b_ < 2) v += rl.sB[12 + b_ * 6 + a_];                                     // off_b . p_a
            sh.Bm[a_ * 6 + b_] = v; sh.Bm[b_ * 6 + a_] = v;
        }
    }
    LDS_SYNC();
    PROF(I, PF_RIC_BWD);
    return 1;
}

// ---------------------------------------------------------------- wave-level matrix-core helpers (used by the quadcopter sweep, obca_quad_solver.h)
//   lane = 16 g + j.  wv_mfma(C, a, b): C[i][n] += sum_{k<4} a(lane (k, i)) * b(lane (k, n)); the f64 accumulator layout is register r of lane (g, j)
//   = C[g + 4r][j] (checked on the hardware by tools/micro/mfma_f64_layout.hip).  The parking blocks (6 x 14, 8 x 14) fill a third of a tile and were
//   measured slower on the matrix cores than in the three-phase LDS sweep above (round 2), so the parking sweep does not use them.
#ifdef OBCA_EMU
OBCA_FN void wv_mfma(double (&acc)[4][OBCA_NLT], const double (&a)[OBCA_NLT], const double (&b)[OBCA_NLT]) {
    double out[4][64];
    for (int l = 0; l < 64; l++) { const int g = l >> 4, n = l & 15;
        for (int r = 0; r < 4; r++) { const int i = g + 4 * r; double s_ = acc[r][l]; for (int k = 0; k < 4; k++) s_ = fma(a[16 * k + i], b[16 * k + n], s_); out[r][l] = s_; } }
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) acc[r][l] = out[r][l];
}
OBCA_FN void wv_shfl_group(double (&out)[OBCA_NLT], const double (&in)[OBCA_NLT], int grp) { for (int l = 0; l < 64; l++) out[l] = in[16 * grp + (l & 15)]; }   // lane (grp, j) -> every lane (g, j)
OBCA_FN void wv_shfl_xor(double (&out)[OBCA_NLT], const double (&in)[OBCA_NLT], int m) { for (int l = 0; l < 64; l++) out[l] = in[l ^ m]; }
#else
typedef double v4d_t __attribute__((ext_vector_type(4)));
OBCA_FN void wv_mfma(double (&acc)[4][1], const double (&a)[1], const double (&b)[1]) {
    v4d_t c = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
    acc[0][0] = c[0]; acc[1][0] = c[1]; acc[2][0] = c[2]; acc[3][0] = c[3];
}
OBCA_FN void wv_shfl_group(double (&out)[1], const double (&in)[1], int grp) { out[0] = __shfl(in[0], 16 * grp + ((int)threadIdx.x & 15), 64); }
OBCA_FN void wv_shfl_xor(double (&out)[1], const double (&in)[1], int m) { out[0] = __shfl_xor(in[0], m, 64); }
#endif

template <int SOC = 0>
OBCA_FN int riccati_backward(const Inst &I, Shared &sh, double rho) {
    const int ok = riccati_body<SOC>(I, sh, rho);
    PAR(lane) { if (lane == 0) sh.ric_ok = ok; }
    SYNC();
    return sh.ric_ok;
}

// ---------------------------------------------------------------- border solve + forward sweep + back-substitution

// part 1: border, closed loop, forward sweep, stage-parallel back-substitution; leaves partial (ap, az, gd) and (dt, nu) in LDS
template <int SOC = 0, int LSQ = 0>      // SOC = 1: the terminal row enters with c_soc;  LSQ = 1: with zero (least-squares multiplier system; call with mu = dw = dc = rho = 0)
OBCA_FN void direction_main(const Inst &I, Shared &sh, const AsmOut &A, double mu, double dw, double dc, double rho, double tau, StepOut &so) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb, M = c.M;
    const gdbl *z = I.z; gdbl *d = I.d;
    int ok = 1;     // kept in a register and stored ONCE: both wavefronts write the shared slot, so it must never hold an intermediate value
    // ---- 5x5 border in (dt, nu): all entries are bilinear constants of the Riccati value function
    double dt, nu[4];
    {
        const double *B = sh.Bm;
        double e[4];
        for (int i = 0; i < 4; i++) e[i] = LSQ ? 0.0 : (SOC ? -(double)sh.soc.csoc[(l.nu - l.pi) + i] : -(z[l.x + 4 * N + i] - c.xF[i]));
        double att = A.Htt + B[1 * 6 + 1], rt = -A.gtb - B[1 * 6 + 0];
        double S[16], col[4], colr[4];
        for (int a_ = 0; a_ < 4; a_++) {
            for (int b_ = 0; b_ < 4; b_++) S[a_ * 4 + b_] = -B[(2 + a_) * 6 + (2 + b_)];
            col[a_] = -B[(2 + a_) * 6 + 1]; colr[a_] = -(e[a_] - B[(2 + a_) * 6 + 0]);
        }
        if (ldl_fact<4>(4, S)) ok = 0;
        ldl_solve<4>(4, S, col); ldl_solve<4>(4, S, colr);
        double piv = att, rr = rt;
        for (int a_ = 0; a_ < 4; a_++) { piv -= B[1 * 6 + 2 + a_] * col[a_]; rr -= B[1 * 6 + 2 + a_] * colr[a_]; }
        if (c.fixTime) { dt = 0; for (int a_ = 0; a_ < 4; a_++) nu[a_] = colr[a_]; }
        else {
            if (!(piv > 0)) ok = 0;
            dt = rr / piv;
            for (int a_ = 0; a_ < 4; a_++) nu[a_] = colr[a_] - col[a_] * dt;
        }
    }
    so.ok = ok;
    if (!ok) return;
    const double coef[OB_NC] = {1.0, dt, nu[0], nu[1], nu[2], nu[3]};
    // ---- forward recursion s_{k+1} = Acl_k s_k + bcl_k with the closed-loop maps Acl = [A + B K ; K] (6x6), bcl = [B kf + off ; kf].  The recursion is a chain of N
    // dependent steps (~280 clocks each: a 4-deep fp64 dependency plus the broadcast), so it runs TWO stages per step.  One lane per stage pair j builds the maps
    // of stages 2j and 2j+1 from the Riccati gains and the stage records, composes them (Pm_j = Acl_{2j+1} Acl_{2j}, pb_j = Acl_{2j+1} bcl_{2j} + bcl_{2j+1}) into
    // LDS and KEEPS the plain map of stage 2j in registers; the sequential loop then produces the even states s_{2j+2} from the composed maps (rows read from LDS
    // one step ahead, the state itself in scalar registers via v_readlane), and afterwards every pair lane fills in its odd state s_{2j+1} = Acl_{2j} s_{2j} + bcl_{2j}.
    // (Round 2 wrote the 42-double closed-loop map of every stage to the Riccati record and the composed maps to a second HBM buffer, and the sequential loop
    // gathered both back through a ring of registers: 0.1 MB of traffic per pass and a loop whose step time followed the memory latency under load.)
    const int NP = UNIFORM(N / 2), NH = UNIFORM((N + 1) / 2);     // pairs; pair lanes incl. the single last stage of an odd horizon
    double *pm = stg_base(sh);                                    // composed maps: NP x 42 doubles behind the trajectory (the backward sweep's stage buffers are dead by now)
    double M0[OBCA_NL][42];
    PAR(lane) {
        const int L_ = LI(lane);
        if (lane < NH) {
            double M1[42];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = 2 * lane + h < N ? 2 * lane + h : 2 * lane;      // (clamped: the second stage of the last lane may not exist)
                const gdbl *rec = I.as + (size_t)k * OB_AS, *ro = I.rs + (size_t)k * OB_RS;
                double K0[6], K1[6], kf0 = 0, kf1 = 0, b0[4], b1[4], a2[4], a3[4], dd[4], ft[4];
#pragma unroll
                for (int j = 0; j < 6; j++) { K0[j] = ro[RS_K + j]; K1[j] = ro[RS_K + 6 + j]; }
#pragma unroll
                for (int cc = 0; cc < OB_NC; cc++) { kf0 += ro[RS_KF + cc] * coef[cc]; kf1 += ro[RS_KF + OB_NC + cc] * coef[cc]; }
#pragma unroll
                for (int i = 0; i < 4; i++) { b0[i] = as_df(i, 2) >= 0 ? rec[AS_DF + as_df(i, 2)] : 0.0; b1[i] = rec[AS_DF + as_df(i, 3)]; a2[i] = as_df(i, 0) >= 0 ? rec[AS_DF + as_df(i, 0)] : 0.0;
                                              a3[i] = as_df(i, 1) >= 0 ? rec[AS_DF + as_df(i, 1)] : 0.0; dd[i] = rec[AS_DD + i]; ft[i] = rec[AS_DF + as_df(i, 4)]; }
                double *cm = h ? M1 : M0[L_];
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        double a_ = (j < 4 && i == j) ? 1.0 : 0.0;
                        if (j == 2) a_ += a2[i];
                        if (j == 3) a_ += a3[i];
                        cm[i * 6 + j] = a_ + b0[i] * K0[j] + b1[i] * K1[j];
                    }
                    cm[36 + i] = dd[i] + dt * ft[i] + b0[i] * kf0 + b1[i] * kf1;
                }
#pragma unroll
                for (int j = 0; j < 6; j++) { cm[24 + j] = K0[j]; cm[30 + j] = K1[j]; }
                cm[40] = kf0; cm[41] = kf1;
            }
            if (lane < NP) {
                double *po = pm + (size_t)lane * 42; const double *m0 = M0[L_];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double *m1r = M1 + r * 6;
#pragma unroll
                    for (int cI = 0; cI < 6; cI++)
                        po[r * 6 + cI] = dot6_tree(0.0, m1r[0], m0[cI], m1r[1], m0[6 + cI], m1r[2], m0[12 + cI], m1r[3], m0[18 + cI], m1r[4], m0[24 + cI], m1r[5], m0[30 + cI]);
                    po[36 + r] = dot6_tree(M1[36 + r], m1r[0], m0[36], m1r[1], m0[37], m1r[2], m0[38], m1r[3], m0[39], m1r[4], m0[40], m1r[5], m0[41]);
                }
            }
        }
        if (lane < 6) g_traj[lane] = 0.0;                         // s_0 = 0 (x_0 is fixed)
    }
    LDS_SYNC();
    PROF(I, PF_BORDER_CL);
    WAVE0_BEGIN
        {
            double fw_s[6] = {0, 0, 0, 0, 0, 0};               // s_2j, wave-uniform (scalar registers)
            double cr[OBCA_NL][7], nx[OBCA_NL][7], v[OBCA_NL];
            PAR64(lane) {
                const int L_ = LI(lane), r = lane < 6 ? lane : 0; const double *row = pm + r * 6;
#pragma unroll
                for (int e = 0; e < 6; e++) cr[L_][e] = row[e];
                cr[L_][6] = pm[36 + r];
            }
            for (int j = 0; j < NP; j++) {
                PAR64(lane) {
                    const int L_ = LI(lane), r = lane < 6 ? lane : 0;
                    const double *pn_ = pm + (size_t)(j + 1 < NP ? j + 1 : j) * 42;          // the next step's row: its LDS reads are in flight during this step's arithmetic
#pragma unroll
                    for (int e = 0; e < 6; e++) nx[L_][e] = pn_[r * 6 + e];
                    nx[L_][6] = pn_[36 + r];
                    v[L_] = dot6_tree(cr[L_][6], cr[L_][0], fw_s[0], cr[L_][1], fw_s[1], cr[L_][2], fw_s[2], cr[L_][3], fw_s[3], cr[L_][4], fw_s[4], cr[L_][5], fw_s[5]);
                    if (lane < 6) g_traj[(size_t)(2 * j + 2) * 6 + lane] = v[L_];
#pragma unroll
                    for (int e = 0; e < 7; e++) cr[L_][e] = nx[L_][e];
                }
#pragma unroll
                for (int e = 0; e < 6; e++) fw_s[e] = WV_READLANE(v, e);
            }
        }
    WAVE0_END
    LDS_SYNC();
    PAR(lane) {     // odd states (and the last state of an odd horizon) from the plain maps kept in registers
        const int L_ = LI(lane);
        if (lane < NH) {
            double s_[6]; const double *m0 = M0[L_];
#pragma unroll
            for (int e = 0; e < 6; e++) s_[e] = g_traj[(size_t)(2 * lane) * 6 + e];
#pragma unroll
            for (int r = 0; r < 6; r++)
                g_traj[(size_t)(2 * lane + 1) * 6 + r] = dot6_tree(m0[36 + r], m0[r * 6 + 0], s_[0], m0[r * 6 + 1], s_[1], m0[r * 6 + 2], s_[2], m0[r * 6 + 3], s_[3], m0[r * 6 + 4], s_[4], m0[r * 6 + 5], s_[5]);
        }
    }
    LDS_SYNC();
    PROF(I, PF_FWD_SEQ);
    // ---- stage-parallel: primal steps of x,u; costates; bound terms of x,u ; steering rows
    double red[4][OBCA_NL];
    PAR(lane) {
        double ap = 1.0, az = 1.0, gd = 0, gr = 0, cc_;
#define FTBP(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < ap) ap = cc_; }
#define FTBZ(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < az) az = cc_; }
        const double t = z[l.t], q = t * c.Ts, rr_t = 0.1 / (q * q);
        for (int k = lane; k <= N; k += OB_NT) {
            // Every load of the stage first, every store last: d, z and the records may alias as far as the compiler knows, so a load behind a store waits for
            // its own round trip (the stage used to take seven of them; a lone wavefront per SIMD has nothing to hide them with).
            double s[6], sn[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { s[i] = g_traj[(size_t)k * 6 + i]; sn[i] = g_traj[(size_t)(k < N ? k + 1 : N) * 6 + i]; }
            const int ku = k < N ? k : N - 1;                                    // (clamped: the loads of the last stage's absent input part are unused)
            const double rx = I.prob[OB_HDR + k], ry = I.prob[OB_HDR + (N + 1) + k], ryaw = I.prob[OB_HDR + 2 * (N + 1) + k];
            double x[4], zxL[4], zxU[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] = z[l.x + 4 * k + i]; zxL[i] = z[l.zxL + 4 * k + i]; zxU[i] = z[l.zxU + 4 * k + i]; }
            const gdbl *rec = I.as + (size_t)ku * OB_AS;
            const double u[2] = {z[l.u + 2 * ku], z[l.u + 2 * ku + 1]};
            const double w[2] = {ku ? z[l.u + 2 * ku - 2] : 0.0, ku ? z[l.u + 2 * ku - 1] : 0.0};
            const double zuL[2] = {z[l.zuL + 2 * ku], z[l.zuL + 2 * ku + 1]}, zuU[2] = {z[l.zuU + 2 * ku], z[l.zuU + 2 * ku + 1]};
            const double gg0 = rec[AS_GG], gg1 = rec[AS_GG + 1], gg2 = rec[AS_GG + 2], sig = rec[AS_SIG], rg = rec[AS_RG], rss = rec[AS_RSS], dssd = rec[AS_DSS];
            const double ss = z[l.ss + ku], zsL = z[l.zssL + ku], zsU = z[l.zssU + ku];
            double dpi[4] = {0, 0, 0, 0};
            if (k + 1 < N) {   // costate of x_{k+1} - F_k: -(Px_{k+1} s_{k+1} + pv_{k+1} . coef)
                const gdbl *r1 = I.rs + (size_t)(k + 1) * OB_RS;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    double a_ = 0;
#pragma unroll
                    for (int cc = 0; cc < OB_NC; cc++) a_ += r1[RS_PV + i * OB_NC + cc] * coef[cc];
#pragma unroll
                    for (int j = 0; j < 6; j++) a_ += r1[RS_PX + i * 6 + j] * sn[j];
                    dpi[i] = -a_;
                }
            } else if (k < N) {   // terminal cost-to-go: P_N = H_N(+rho), p_N = (hb_N - rho e, Ht_N, e_i)   (rs[N] is not written)
                const gdbl *rN = I.as + (size_t)N * OB_AS;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    double e = SOC ? -(double)sh.soc.csoc[(l.nu - l.pi) + i] : -(z[l.x + 4 * N + i] - c.xF[i]);
                    double a_ = (rN[AS_HB + i] - rho * e) + (i >= 2 ? rN[AS_HT + i - 2] : 0.0) * dt + nu[i];
#pragma unroll
                    for (int j = 0; j < 6; j++) a_ += ((as_h(i, j) >= 0 ? rN[AS_H + as_h(i, j)] : 0.0) + ((i == j) ? rho : 0.0)) * sn[j];
                    dpi[i] = -a_;
                }
            }
            // ---- arithmetic and stores
#pragma unroll
            for (int i = 0; i < 4; i++) d[l.x + 4 * k + i] = s[i];
            gd += 2e-3 * (x[0] - rx) * s[0] + 2e-3 * (x[1] - ry) * s[1] + 2 * c.wpsi * (x[2] - ryaw) * s[2] + 2e-4 * x[3] * s[3];
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) if (i != 2) {
                    double dL = x[i] - c.xl[i], dU = c.xu[i] - x[i], zL = zxL[i], zU = zxU[i];
                    gd += (-rdiv(mu, dL) + rdiv(mu, dU)) * s[i];
                    FTBP(dL, s[i]); FTBP(dU, -s[i]);
                    FTBZ(zL, rdiv(mu, dL) - zL - rdiv(zL, dL) * s[i]); FTBZ(zU, rdiv(mu, dU) - zU + rdiv(zU, dU) * s[i]);
                }
            }
            if (k < N) {
                // du_k = K_k s_k + kf_k is the input copy dw_{k+1} of the NEXT state: the forward sweep formed it already (rows 4, 5 of the closed-loop map), so it is read from the
                // trajectory instead of being formed again from the gains (rounds 1-3 re-read K and KF here: 24 doubles per stage and pass)
                const double du[2] = {sn[4], sn[5]};
                d[l.u + 2 * k] = du[0]; d[l.u + 2 * k + 1] = du[1];
                if (!c.fixTime) { const double e1 = u[0] - w[0], e2 = u[1] - w[1]; gr += -2 * rr_t * (e1 * e1 + e2 * e2) / t; }
                const double cu[2] = {0.01, c.wa}, iq = 1.0 / q, rr = 0.1 * (iq * iq);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const double ei = u[i] - w[i], lo = i ? OB_UL1 : OB_UL0, hi = i ? OB_UU1 : OB_UU0;
                    const double dL = u[i] - lo, dU = hi - u[i], zL = zuL[i], zU = zuU[i];
                    gd += (2 * cu[i] * u[i] + 2 * rr * ei) * du[i] - 2 * rr * ei * s[4 + i] + (-rdiv(mu, dL) + rdiv(mu, dU)) * du[i];
                    FTBP(dL, du[i]); FTBP(dU, -du[i]);
                    FTBZ(zL, rdiv(mu, dL) - zL - rdiv(zL, dL) * du[i]); FTBZ(zU, rdiv(mu, dU) - zU + rdiv(zU, dU) * du[i]);
                }
                // steering row back-substitution
                const double lin = gg0 * s[4] + gg1 * du[0] + gg2 * dt;
                const double dyg = sig * (lin + rg);
                const double dss = rdiv(dyg - rss, dssd);
                d[l.yg + k] = dyg; d[l.ss + k] = dss;
                const double zL = zsL, zU = zsU, dL = ss + OB_SSB, dU = OB_SSB - ss;
                gd += (-rdiv(mu, dL) + rdiv(mu, dU)) * dss;
                FTBP(dL, dss); FTBP(dU, -dss);
                FTBZ(zL, rdiv(mu, dL) - zL - rdiv(zL, dL) * dss); FTBZ(zU, rdiv(mu, dU) - zU + rdiv(zU, dU) * dss);
#pragma unroll
                for (int i = 0; i < 4; i++) d[l.pi + 4 * k + i] = dpi[i];
            }
        }
        red[0][LI(lane)] = ap; red[1][LI(lane)] = az; red[2][LI(lane)] = gd; red[3][LI(lane)] = gr;
#undef FTBP
#undef FTBZ
    }
    so.ap = wred_min(red[0]); so.az = wred_min(red[1]); so.gd = wred_sum(red[2]); so.gr = wred_sum(red[3]);
    PAR(lane) { if (lane == 0) { sh.coef[0] = dt; sh.coef[1] = nu[0]; sh.coef[2] = nu[1]; sh.coef[3] = nu[2]; sh.coef[4] = nu[3]; } }
    SYNC();
    PROF(I, PF_BS_STAGE);
}

// part 2: obstacle blocks (re-factorised instead of stored), then t / nu and the step-length and descent scalars
template <int VM, int DBG, int SOC = 0, int LSQ = 0, int KEEP = 0>      // DBG = 1 (host emulation tests, least-squares multipliers): the obstacle part of the direction is also written to d; SOC = 1: block right-hand sides with c_soc
// KEEP = 1: the block steps stay in the caller's registers (keep[r] = step of item lane + 64 r) for the first trial of the line search, see assemble_obs<KEEP = 1>
OBCA_FN void direction_obs(const Inst &I, Shared &sh, double mu_, double dw_, double dc_, double tau_, StepOut &so, ObsStep<VM> (*keep)[OBCA_NL] = nullptr) {
    constexpr int RS_ = VM <= 2 ? 1 : 0;       // which reciprocal form (rcp_nr, obca_model.h)
    const Lay &l = sh.l;
    Consts c; obs_consts(sh.c, c);
    const double mu = UNIFORM_D(mu_), dw = UNIFORM_D(dw_), dc = UNIFORM_D(dc_), tau = UNIFORM_D(tau_);
    const int N = c.N, nOb = c.nOb, M = c.M;
    const gdbl *z = I.z; gdbl *d = I.d;
    double ap = so.ap, az = so.az, gd = so.gd;
    const double dt = sh.coef[0], nu[4] = {sh.coef[1], sh.coef[2], sh.coef[3], sh.coef[4]};
    // ---- obstacle blocks: back-substitution (the block is re-factorised instead of being stored)
    double red[3][OBCA_NL];
    PAR(lane) {
        double lap = 1.0, laz = 1.0, lgd = 0, cc_;
#define FTBP(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr<RS_>(dv) : 1e300; if (cc_ < lap) lap = cc_; }
#define FTBZ(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr<RS_>(dv) : 1e300; if (cc_ < laz) laz = cc_; }
        const int nit = (N + 1) * nOb;
#pragma unroll
        for (int rr = 0; rr < (KEEP ? OB_KEEP : 1); rr++)
        for (int it = lane + (KEEP ? rr * OB_NT : 0); it < nit; it += (KEEP ? nit : OB_NT)) {
            int k = it / nOb, j = it - k * nOb;
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, j, in);
            const double dp[3] = {g_traj[(size_t)k * 6], g_traj[(size_t)k * 6 + 1], g_traj[(size_t)k * 6 + 2]};
            ObsStep<VM> st;
            double crs[4] = {0, 0, 0, 0};
            if (SOC) {
#pragma unroll
                for (int r = 0; r < 4; r++) crs[r] = sh.soc.csoc[(l.yo - l.pi) + 4 * it + r];
            }
            obs_block<1, VM, SOC, LSQ>(c, in, mu, dw, dc, nullptr, nullptr, dp, &st, crs);
            if (KEEP) keep[rr][LI(lane)] = st;
            const int r0 = sh.roff[j];
#pragma unroll
            for (int i = 0; i < VM; i++) if (i < in.v) {
                if (DBG || OBCA_STORE_DOBS) d[l.lam + k * M + r0 + i] = st.dlam[i];
                lgd -= rdiv<RS_>(mu, in.lam[i]) * st.dlam[i];
                FTBP(in.lam[i], st.dlam[i]); FTBZ(in.zl[i], rdiv<RS_>(mu, in.lam[i]) - in.zl[i] - rdiv<RS_>(in.zl[i], in.lam[i]) * st.dlam[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (DBG || OBCA_STORE_DOBS) { d[l.mu + 4 * it + i] = st.dmu[i]; d[l.yo + 4 * it + i] = st.dy[i]; }
                lgd -= rdiv<RS_>(mu, in.mu[i]) * st.dmu[i];
                FTBP(in.mu[i], st.dmu[i]); FTBZ(in.zm[i], rdiv<RS_>(mu, in.mu[i]) - in.zm[i] - rdiv<RS_>(in.zm[i], in.mu[i]) * st.dmu[i]);
            }
            if (DBG || OBCA_STORE_DOBS) { d[l.sl + it] = st.dsl; d[l.so + it] = st.dso; }
            lgd += (c.dist ? -rdiv<RS_>(mu, in.sl) : 1e2 + 2e4 * in.sl) * st.dsl - rdiv<RS_>(mu, in.so) * st.dso;
            if (c.dist) { FTBP(in.sl, st.dsl); FTBZ(in.zs1, rdiv<RS_>(mu, in.sl) - in.zs1 - rdiv<RS_>(in.zs1, in.sl) * st.dsl); }
            FTBP(in.so, st.dso); FTBZ(in.zso, rdiv<RS_>(mu, in.so) - in.zso - rdiv<RS_>(in.zso, in.so) * st.dso);
        }
        red[0][LI(lane)] = lap; red[1][LI(lane)] = laz; red[2][LI(lane)] = lgd;
#undef FTBP
#undef FTBZ
    }
    ap = fmin(ap, wred_min(red[0])); az = fmin(az, wred_min(red[1])); gd += wred_sum(red[2]);
    // ---- t and nu (uniform)
    if (!c.fixTime) {
        const double t = z[l.t], dL = t - OB_TL, dU = OB_TU - t, zL = z[l.ztL], zU = z[l.ztU];
        double cc_;
        cc_ = dt < 0 ? -tau * dL * rcp_nr<RS_>(dt) : 1e300; if (cc_ < ap) ap = cc_;
        cc_ = -dt < 0 ? tau * dU * rcp_nr<RS_>(dt) : 1e300; if (cc_ < ap) ap = cc_;
        double dzL = rdiv<RS_>(mu, dL) - zL - rdiv<RS_>(zL, dL) * dt, dzU = rdiv<RS_>(mu, dU) - zU + rdiv<RS_>(zU, dU) * dt;
        cc_ = dzL < 0 ? -tau * zL * rcp_nr<RS_>(dzL) : 1e300; if (cc_ < az) az = cc_;
        cc_ = dzU < 0 ? -tau * zU * rcp_nr<RS_>(dzU) : 1e300; if (cc_ < az) az = cc_;
        // d phi / d t: rate cost (so.gr, summed over the stages by the back-substitution above) + time cost + barrier of its bounds
        const double gt = so.gr + (N + 1) * (0.5 + 2 * t) + (N + 1) * (-mu / (t - OB_TL) + mu / (OB_TU - t));
        gd += gt * dt;
    }
    if (DBG) { PAR(lane) { if (lane < 4) d[l.nu + lane] = nu[lane]; if (lane == 4) d[l.t] = dt; } }
    if (DBG || OBCA_STORE_DOBS) SYNC();      // the stored steps are read back by the fused assembly
    so.ap = ap; so.az = az; so.gd = gd;
    PROF(I, PF_BS_OBS);
}

// ---------------------------------------------------------------- starting point (IPOPT sec. 3.6: push into the bounds, z=1, y=0)
OBCA_FN double push2(double v, double lo, double hi, double k1, double k2) {
    double pl = fmin(k1 * fmax(1.0, fabs(lo)), k2 * (hi - lo)), pu = fmin(k1 * fmax(1.0, fabs(hi)), k2 * (hi - lo));
    if (v < lo + pl) v = lo + pl;
    if (v > hi - pu) v = hi - pu;
    return v;
}
struct PushOpts { double bound_push, bound_frac; };
template <int VM>
OBCA_FN void init_point(const Inst &I, Shared &sh, const PushOpts &o) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb, M = c.M;
    gdbl *z = I.z;
    PAR(lane) {
        if (lane < 4) z[l.x + lane] = c.x0[lane];
        if (lane == 4 && c.fixTime) z[l.t] = 1.0;
        for (int i = l.pi + lane; i < l.zxL; i += OB_NT) z[i] = 0.0;
        for (int i = l.zxL + lane; i < l.len; i += OB_NT) z[i] = 1.0;
    }
    SYNC();
    const double q = z[l.t] * c.Ts;
    PAR(lane) {   // slacks take the row values at the (un-pushed) warm start
        for (int k = lane; k < N; k += OB_NT) z[l.ss + k] = ((k ? z[l.u + 2 * k - 2] : 0.0) - z[l.u + 2 * k]) / q;
        for (int it = lane; it < (N + 1) * nOb; it += OB_NT) {
            int k = it / nOb, j = it - k * nOb;
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, j, in);
            in.so = 0; if (c.dist) in.sl = 0;
            double r[4]; obs_rows<VM>(c, in, r);
            z[l.so + it] = r[3];
            if (c.dist) z[l.sl + it] = -r[0];          // slack of |A'lam|^2 <= 1 takes the row value
        }
    }
    SYNC();
    PAR(lane) {   // push into the interior
        for (int k = lane; k <= N; k += OB_NT) {
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) if (i != 2) z[l.x + 4 * k + i] = push2(z[l.x + 4 * k + i], c.xl[i], c.xu[i], o.bound_push, o.bound_frac);
            }
            if (k < N) {
                z[l.u + 2 * k] = push2(z[l.u + 2 * k], OB_UL0, OB_UU0, o.bound_push, o.bound_frac);
                z[l.u + 2 * k + 1] = push2(z[l.u + 2 * k + 1], OB_UL1, OB_UU1, o.bound_push, o.bound_frac);
                z[l.ss + k] = push2(z[l.ss + k], -OB_SSB, OB_SSB, o.bound_push, o.bound_frac);
            }
        }
        if (lane == 4 && !c.fixTime) z[l.t] = push2(z[l.t], OB_TL, OB_TU, o.bound_push, o.bound_frac);
        for (int i = lane; i < M * (N + 1); i += OB_NT) z[l.lam + i] = fmax(z[l.lam + i], o.bound_push);
        for (int i = lane; i < 4 * nOb * (N + 1); i += OB_NT) z[l.mu + i] = fmax(z[l.mu + i], o.bound_push);
        for (int i = lane; i < nOb * (N + 1); i += OB_NT) { z[l.so + i] = fmax(z[l.so + i], o.bound_push); if (c.dist) z[l.sl + i] = fmax(z[l.sl + i], o.bound_push); }
    }
    SYNC();
}

// ---------------------------------------------------------------- phase entry points (non-inlined; state lives in g_sh)
// the per-lane (stage, obstacle) code exists in three sizes (VM = 2, OB_VMID, OB_VMAX rows); an instance uses the smallest that holds its widest obstacle
#define VM_CALL(F, ...) do { if (g_sh.vmc == 0) F<2>(__VA_ARGS__); else if (g_sh.vmc == 1) F<OB_VMID>(__VA_ARGS__); else F<OB_VMAX>(__VA_ARGS__); } while (0)
OBCA_PHASE void ph_init(double bound_push, double bound_frac) {
    Shared &sh = g_sh; PushOpts po = {bound_push, bound_frac}; PROF(sh.inst, PF_OTHER);
    VM_CALL(init_point, sh.inst, sh, po);
    PROF(sh.inst, PF_INIT);
}
// Assembly of the Newton system at the current iterate (`which` = 0 -> sh.A, 1 -> sh.A2), and the fused line-search step (ph_fused: trial point -> Inst::zn,
// assembled -> sh.An).  The (stage, obstacle) part and the stage part of the common (<= 2 rows per obstacle) case share ONE non-inlined function.
#define OB_NOFUSE FuseArgs{0.0, 0.0, 0.0, 0.0, 0.0}
OBCA_PHASE void ph_assemble_obs2(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<2, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_PHASE void ph_assemble_obs4(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<OB_VMID, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_PHASE void ph_assemble_obs8(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<OB_VMAX, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_FN void ph_assemble_obs(double mu, double dw, double dc) { if (g_sh.vmc == 0) ph_assemble_obs2(mu, dw, dc); else if (g_sh.vmc == 1) ph_assemble_obs4(mu, dw, dc); else ph_assemble_obs8(mu, dw, dc); }
OBCA_PHASE void ph_assemble_stage(double mu, double dw, double dc, int second) { Shared &sh = g_sh; assemble_stage<0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, second ? sh.A2 : sh.A); }
OBCA_PHASE void ph_assemble2(double mu, double dw, double dc, int second) { Shared &sh = g_sh; assemble_obs<2, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); assemble_stage<0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, second ? sh.A2 : sh.A); }
OBCA_FN void ph_assemble(double mu, double dw, double dc, int second) { if (g_sh.vm2) ph_assemble2(mu, dw, dc, second); else { ph_assemble_obs(mu, dw, dc); ph_assemble_stage(mu, dw, dc, second); } }
OBCA_PHASE void ph_fused_obs4(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_obs<OB_VMID, 1>(sh.inst, sh, mu, 0.0, dc, fa); }
OBCA_PHASE void ph_fused_obs8(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_obs<OB_VMAX, 1>(sh.inst, sh, mu, 0.0, dc, fa); }
OBCA_PHASE void ph_fused_stage(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An); }
OBCA_PHASE void ph_fused2(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {
    Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd};
    assemble_obs<2, 1>(sh.inst, sh, mu, 0.0, dc, fa); assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An);
}
OBCA_FN void ph_fused(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {
    if (g_sh.vm2) { ph_fused2(mu, dc, alpha, ay, az, ks, dwd); return; }
    if (g_sh.vmc == 1) ph_fused_obs4(mu, dc, alpha, ay, az, ks, dwd); else ph_fused_obs8(mu, dc, alpha, ay, az, ks, dwd);
    ph_fused_stage(mu, dc, alpha, ay, az, ks, dwd);
}
OBCA_PHASE int ph_riccati(double rho) { Shared &sh = g_sh; return riccati_backward(sh.inst, sh, rho); }
OBCA_PHASE void ph_direction_main(double mu, double dw, double dc, double rho, double tau) { Shared &sh = g_sh; direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S); }
OBCA_PHASE void ph_direction_obs2(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<2, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_PHASE void ph_direction_obs4(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<OB_VMID, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_PHASE void ph_direction_obs8(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<OB_VMAX, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_FN void ph_direction_obs(double mu, double dw, double dc, double tau) { if (g_sh.vmc == 0) ph_direction_obs2(mu, dw, dc, tau); else if (g_sh.vmc == 1) ph_direction_obs4(mu, dw, dc, tau); else ph_direction_obs8(mu, dw, dc, tau); }
OBCA_PHASE void ph_direction2(double mu, double dw, double dc, double rho, double tau) {   // both parts in one call, see ph_assemble2
    Shared &sh = g_sh;
    direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (sh.S.ok) direction_obs<2, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S);
}
// Search direction AND the block part of the first trial of the line search in one call (round 4).  The first trial always takes the fraction-to-the-boundary step lengths, which
// are known the moment the block back-substitution has been reduced over the wavefront -- so the blocks' steps stay in the lanes' registers (48 doubles for the 4 rounds of a
// 3-obstacle instance) and the trial point's block part is formed and condensed right away.  Until round 4 the fused line search factorised every block a second time at the old
// point just to get that step back (a fifth of a pass).  The stage part of the trial follows as ph_fused_stage once the driver has set up the line search; later (backtracking)
// trials and everything on the cold paths recompute as before.  Bit for bit the numbers of the two-call sequence.
OBCA_PHASE void ph_direction2_trial(double mu, double dw, double dc, double rho, double tau, double ks) {
    Shared &sh = g_sh;
    direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (!sh.S.ok) return;
    ObsStep<2> keep[OB_KEEP][OBCA_NL];
    direction_obs<2, 0, 0, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S, keep);
    const FuseArgs fa = {sh.S.ap, fmin(sh.S.ap, sh.S.az), sh.S.az, ks, dw};
    assemble_obs<2, 1, 0, 0, 1>(sh.inst, sh, mu, 0.0, dc, fa, keep);
    PAR(lane) { if (lane == 0) sh.ft_done = 1; }
    LDS_SYNC();
}
OBCA_FN void ph_direction(double mu, double dw, double dc, double rho, double tau, double ks_first_trial = 0.0) {      // ks_first_trial > 0: also the block part of the first trial (main path only)
    if (g_sh.vm2 && g_sh.ft_ok && ks_first_trial > 0) { ph_direction2_trial(mu, dw, dc, rho, tau, ks_first_trial); return; }
    if (g_sh.vm2) { ph_direction2(mu, dw, dc, rho, tau); return; }
    ph_direction_main(mu, dw, dc, rho, tau);
    if (g_sh.S.ok) ph_direction_obs(mu, dw, dc, tau);
}
// ---- second-order correction (IPOPT A-5.5..A-5.9; Opts::max_soc > 0; cold path: one non-inlined function per step, every obstacle width inside)
// c_soc <- asoc * (first ? c(z) : c_soc) + c(zn)   (zn: the rejected trial point; rows as the assembly forms them: dynamics x_{k+1} - F, terminal x_N - xF, steering, obstacle rows)
template <int VM>
OBCA_FN void soc_accumulate(const Inst &I, Shared &sh, double asoc, int first) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb; gdbl *cs = sh.soc.csoc;
    PAR(lane) {
        for (int k = lane; k < N; k += OB_NT) {
            double v[2][5];
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const gdbl *z = w ? I.zn : I.z; const double t = z[l.t];
                double x[4], u[2], F[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = z[l.x + 4 * k + i];
                u[0] = z[l.u + 2 * k]; u[1] = z[l.u + 2 * k + 1];
                dyn_value(c, x, u, t, F);
#pragma unroll
                for (int i = 0; i < 4; i++) v[w][i] = z[l.x + 4 * (k + 1) + i] - F[i];
                v[w][4] = ((k ? z[l.u + 2 * k - 2] : 0.0) - u[0]) / (t * c.Ts) - z[l.ss + k];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) cs[4 * k + i] = asoc * (first ? v[0][i] : (double)cs[4 * k + i]) + v[1][i];
            cs[(l.yg - l.pi) + k] = asoc * (first ? v[0][4] : (double)cs[(l.yg - l.pi) + k]) + v[1][4];
        }
        if (lane < 4) { const int o_ = (l.nu - l.pi) + lane; cs[o_] = asoc * (first ? I.z[l.x + 4 * N + lane] - c.xF[lane] : (double)cs[o_]) + (I.zn[l.x + 4 * N + lane] - c.xF[lane]); }
        for (int it = lane; it < (N + 1) * nOb; it += OB_NT) {
            const int k = it / nOb, j = it - k * nOb; double r0[4], r1[4];
            { ObsIn<VM> in; load_obs<VM>(I, sh, I.z, k, j, in); obs_rows<VM>(c, in, r0); }
            { ObsIn<VM> in; load_obs<VM>(I, sh, I.zn, k, j, in); obs_rows<VM>(c, in, r1); }
#pragma unroll
            for (int r = 0; r < 4; r++) { const int o_ = (l.yo - l.pi) + 4 * it + r; cs[o_] = asoc * (first ? r0[r] : (double)cs[o_]) + r1[r]; }
        }
    }
    SYNC();
}
OBCA_PHASE void ph_soc_accumulate(double asoc, int first) { Shared &sh = g_sh; VM_CALL(soc_accumulate, sh.inst, sh, asoc, first); }
OBCA_PHASE void ph_soc_assemble(double mu, double dw, double dc) {      // the system at z with c_soc on the right-hand side (sh.A keeps the values of the iterate: f, theta, errors use the true rows)
    Shared &sh = g_sh;
    if (sh.vmc == 0) assemble_obs<2, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); else if (sh.vmc == 1) assemble_obs<OB_VMID, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); else assemble_obs<OB_VMAX, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE);
    assemble_stage<0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, sh.A);
}
OBCA_PHASE int ph_soc_riccati(double rho) { Shared &sh = g_sh; return riccati_backward<1>(sh.inst, sh, rho); }
OBCA_PHASE void ph_soc_direction(double mu, double dw, double dc, double rho, double tau) {
    Shared &sh = g_sh;
    direction_main<1>(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (!sh.S.ok) return;
    if (sh.vmc == 0) direction_obs<2, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S); else if (sh.vmc == 1) direction_obs<OB_VMID, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S); else direction_obs<OB_VMAX, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S);
}
OBCA_PHASE void ph_soc_fused(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {      // trial point along the correction direction, assembled there as usual
    Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd};
    if (sh.vmc == 0) assemble_obs<2, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa); else if (sh.vmc == 1) assemble_obs<OB_VMID, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa); else assemble_obs<OB_VMAX, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa);
    assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An);
}

// ---- recalc_y = "yes" (ParkingSignedDist.jl:41; IPOPT recalc_y_feas_tol = 1e-6): once the iterate is (nearly) feasible its equality multipliers are replaced by the
// least-squares estimate -- the same structured solve with H := I, zero constraint right-hand side, gradients in their z-form; only the multiplier part of the solution is used.
// Cold path: one non-inlined function, every obstacle width inside.  1 = the multipliers were replaced (the assembly at hand is then stale).
#ifdef OBCA_EMU
static int g_emu_recalc_fail = 0;      // host test hook: every estimate is attempted (at every accepted iterate) and thrown away
#define OB_RECALC_FEAS_TOL (g_emu_recalc_fail ? 1e300 : 1e-6)
#else
#define OB_RECALC_FEAS_TOL 1e-6        // IPOPT recalc_y_feas_tol
#endif
OBCA_PHASE int ph_recalc_y(int init) {      // init = 1: IPOPT's initial multipliers (least-squares estimate at the starting point, kept only if its max-norm is <= constr_mult_init_max = 1e3)
    Shared &sh = g_sh; const Inst &I = sh.inst; const Lay &l = sh.l;
    if (sh.vmc == 0) assemble_obs<2, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE); else if (sh.vmc == 1) assemble_obs<OB_VMID, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE); else assemble_obs<OB_VMAX, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE);
    assemble_stage<0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE, sh.A2);
    if (!riccati_backward(I, sh, 0.0)) return 0;
    direction_main<0, 1>(I, sh, sh.A2, 0.0, 0.0, 0.0, 0.0, 0.99, sh.S);
    if (!sh.S.ok) return 0;
    if (sh.vmc == 0) direction_obs<2, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S); else if (sh.vmc == 1) direction_obs<OB_VMID, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S); else direction_obs<OB_VMAX, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S);
    double red[1][OBCA_NL];
    PAR(lane) { double w = 0; for (int i = l.pi + lane; i < l.zxL; i += OB_NT) { const double v = I.d[i], y1 = fabs(I.z[i] + v); w = (v == v && fabs(v) <= 1e300 && w <= 1e300) ? fmax(w, y1) : 1e301; } red[0][LI(lane)] = w; }
    const double ymax = wred_max(red[0]);
#ifdef OBCA_EMU
    if (g_emu_recalc_fail && !init) return 0;                                // (host test hook: the estimate is discarded AFTER the records were overwritten)
#endif
    if (ymax > 1e300 || (init && ymax > 1e3)) return 0;                      // a non-finite entry (or, at the start, an estimate beyond constr_mult_init_max): keep the multipliers
    PAR(lane) { for (int i = l.pi + lane; i < l.zxL; i += OB_NT) I.z[i] += I.d[i]; }
    SYNC();
    if (!init) sh.soc.nrecalc++;
    return 1;
}

// the iterate the solve ends with (or is parked at) must sit in the instance's own buffer `home`: copy it over if the last accepted trial left it in the other one
OBCA_PHASE void ph_bring_home() {
    Shared &sh = g_sh; Inst &I = sh.inst;
    PAR(lane) { for (int i = lane; i < sh.l.len; i += OB_NT) I.zn[i] = I.z[i]; }
    SYNC();
    PAR(lane) { if (lane == 0) { gdbl *t_ = I.z; I.z = I.zn; I.zn = t_; } }
    SYNC();
}

// ---------------------------------------------------------------- the interior-point driver
enum { ST_OPTIMAL = 0, ST_USERLIMIT = 1, ST_ERROR = 2, ST_SUSPENDED = 3 };

// Time slicing ("two-launch schedule").  A solve may be cut at the top of an interior-point iteration and continued by
// a later launch: everything the iteration loop carries across iterations besides the iterate itself (which lives in HBM anyway) is a
// handful of scalars and the filter, saved in the instance's slice record.  A resumed solve recomputes the assembly at the same point, so
// the sequence of iterates is bit-identical to an uninterrupted solve.  Record layout (doubles):
OBCA_FN double filt_get(const Shared &sh, const gdbl *st, int i, int c) { return i < OB_FILT_LDS ? sh.filt[i][c] : st[SL_FILT + 2 * i + c]; }

// The reference's acceptance test on the current iterate, with its quirks (ParkingConstraints.jl:29-149): in variable-time
// mode only the speed row of the dynamics is kept (:76-79), only the LAST obstacle's rows survive (:108-130), the separation row is
// evaluated without any slack, the steering rate divides by timeScale[1].  1 = every class <= 5e-5.  Cold path (failed attempts only).
template <int VM>
OBCA_FN int ref_constraints(const Inst &I, Shared &sh, int sd) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb, M = c.M; const gdbl *z = I.z;
    const double t = c.fixTime ? 1.0 : z[l.t];
    double red[1][OBCA_NL];
    PAR(lane) {
        double w = -1e300;                                     // running max of every "should be <= 0" quantity
        for (int i = lane; i < M * (N + 1); i += OB_NT) w = fmax(w, -z[l.lam + i]);
        for (int i = lane; i < 4 * nOb * (N + 1); i += OB_NT) w = fmax(w, -z[l.mu + i]);
        for (int k = lane; k < N; k += OB_NT) {
            double x[4], u[2], F[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = z[l.x + 4 * k + i];
            u[0] = z[l.u + 2 * k]; u[1] = z[l.u + 2 * k + 1];
            w = fmax(w, fmax(fabs(u[0]) - 0.6, fabs(u[1]) - 0.4));
            dyn_value(c, x, u, t, F);
            if (c.fixTime) {
#pragma unroll
                for (int i = 0; i < 4; i++) w = fmax(w, fabs(z[l.x + 4 * (k + 1) + i] - F[i]));
            } else w = fmax(w, fabs(z[l.x + 4 * (k + 1) + 3] - F[3]));
            w = fmax(w, fabs(u[0] - (k ? z[l.u + 2 * k - 2] : 0.0)) / (t * c.Ts) - 0.6);
        }
        if (lane < 4) w = fmax(w, fabs(z[l.x + 4 * N + lane] - c.xF[lane]));
        if (lane == 4) w = fmax(w, fabs(t - 1) - 0.2);
        if (nOb > 0) for (int k = lane; k <= N; k += OB_NT) {
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, nOb - 1, in);
            in.sl = 0; in.so = 0;
            double p1 = 0, p2 = 0, beta = 0;
#pragma unroll
            for (int i = 0; i < VM; i++) if (i < in.v) { p1 += in.a1[i] * in.lam[i]; p2 += in.a2[i] * in.lam[i]; beta += in.b[i] * in.lam[i]; }
            double sn, cs; sincos_bounded(in.psi, &sn, &cs);
            const double r0 = p1 * p1 + p2 * p2 - 1;
            const double r1 = in.mu[0] - in.mu[2] + cs * p1 + sn * p2, r2 = in.mu[1] - in.mu[3] - sn * p1 + cs * p2;
            const double r3 = -(c.g[0] * in.mu[0] + c.g[1] * in.mu[1] + c.g[2] * in.mu[2] + c.g[3] * in.mu[3]) + (in.X + cs * c.off) * p1 +
                              (in.Y + sn * c.off) * p2 - beta - OB_DMIN;
            w = fmax(w, fmax(sd ? fabs(r0 + 1) - 1 : r0, fmax(fmax(fabs(r1), fabs(r2)), -r3)));
        }
        red[0][LI(lane)] = w;
    }
    const double worst = wred_max(red[0]);
    return worst <= 5e-5;
}
OBCA_PHASE int ph_ref_constraints(int sd) { Shared &sh = g_sh; return sh.vmc == 0 ? ref_constraints<2>(sh.inst, sh, sd) : (sh.vmc == 1 ? ref_constraints<OB_VMID>(sh.inst, sh, sd) : ref_constraints<OB_VMAX>(sh.inst, sh, sd)); }

// What the iteration loop carries lives in LDS (Shared::drv), not in registers: the phases are non-inlined calls that use the whole register file, so every
// value the driver kept in a register was spilled to scratch -- i.e. to HBM -- before each call and fetched back after it (~500 spill instructions in round 2's
// kernel body, a memory round trip behind every phase).  An LDS slot costs a ~100-clock read where the value is needed and nothing at a call.  (Measured and not
// kept: the state in registers between the calls and copied to / from LDS around each call -- the register allocator then spills MORE, 277 scratch stores / 571
// loads in the kernel body against 96 / 279.)
#define PH(call) call
// Second-order correction after the FIRST trial step of an iteration was rejected without reducing theta (IPOPT A-5.5..A-5.9, kappa_soc = 0.99): up to max_soc steps that
// solve the system of the iterate again with c_soc = alpha c(z) + c(trial) on the right-hand side, each tested like a trial step (with the ORIGINAL alpha in the switching and
// Armijo conditions).  1 = accepted: the trial buffer holds z + asoc d_soc with its assembly, D.alpha / D.az are those of the correction.  0: the Newton direction of the
// iteration is rebuilt (the correction overwrote it) and the backtracking goes on.  The phases are fused (factorise + solve), so a correction costs a full pass.
OBCA_PHASE int ph_soc_try(double tht_first) {
    Shared &sh = g_sh; Drv &D = sh.drv; const Opts &o = sh.o; gdbl *const st = sh.sol.sl.st;
    const double alpha = D.alpha, th = D.th, phi = D.phi, gd = D.gd;
    double th_old = 0, th_tr = tht_first, asoc = alpha, azs = D.az; int acc = 0;
    for (int ps = 0; ps < sh.soc.max_soc && !acc && (ps == 0 || th_tr <= 0.99 * th_old); ps++) {
        th_old = th_tr;
        ph_soc_accumulate(asoc, ps == 0);
        ph_soc_assemble(D.mu, D.dw, D.dc_val);
        int a_ = sh.A.ok;
        if (a_) a_ = ph_soc_riccati(o.rho_term);
        if (a_) { ph_soc_direction(D.mu, D.dw, D.dc_val, o.rho_term, D.tau); a_ = sh.S.ok; }
        if (!a_) break;
        asoc = sh.S.ap; azs = sh.S.az;
        ph_soc_fused(D.mu, D.dc_val, asoc, fmin(asoc, azs), azs, o.kappa_sigma, D.dw);
        sh.soc.nsoc++;
        const double ft = sh.An.f, tht = sh.An.th1, pht = ft - D.mu * sh.An.bar;
        if (!(ft == ft && tht == tht)) break;
        th_tr = tht;
        if (pht == pht && tht < D.th_max) {
            int okf = 1; const int nf = D.nf;
            for (int i = 0; i < nf && okf; i++) if (!(tht < filt_get(sh, st, i, 0) || pht < filt_get(sh, st, i, 1))) okf = 0;
            if (okf) {
                const int sw = gd < 0 && alpha * D.pw_gd > o.delta * D.pw_th;
                const int armijo = pht <= phi + o.eta_phi * alpha * gd;
                if (th <= D.th_min && sw) { if (armijo) acc = 1; }
                else if (tht <= (1 - o.gamma_theta) * th || pht <= phi - o.gamma_phi * th) {
                    acc = 1;
                    if (!(sw && armijo) && nf < OB_FILT) {
                        PAR(lane) { if (lane == 0) { const double f0 = (1 - o.gamma_theta) * th, f1 = phi - o.gamma_phi * th;
                                                     if (nf < OB_FILT_LDS) { sh.filt[nf][0] = f0; sh.filt[nf][1] = f1; } else { st[SL_FILT + 2 * nf] = f0; st[SL_FILT + 2 * nf + 1] = f1; } } }
                        SYNC();
                        D.nf = nf + 1;
                    }
                }
            }
        }
    }
    if (acc) { D.alpha = asoc; D.az = azs; sh.soc.nsoc_acc++; return 1; }
    // not accepted: the records and d hold a correction system -- rebuild the Newton system and direction of this iteration (same point, same delta_w: the same numbers)
    sh.soc.nrebuild++;
    ph_assemble(D.mu, D.dw, D.dc_val, 0);
    if (sh.A.ok && ph_riccati(o.rho_term)) ph_direction(D.mu, D.dw, D.dc_val, o.rho_term, D.tau);
    return 0;
}
OBCA_FN void ipm_attempt(const Opts &o, Result &R, Slice &sl) {
    Shared &sh = g_sh; Drv &D = sh.drv;
    gdbl *const st = sl.st;
    const AsmOut &A = sh.A;
    sh.soc.nsoc = 0; sh.soc.nsoc_acc = 0; sh.soc.nrecalc = 0; sh.soc.nrebuild = 0;
    D.mu = o.mu_init; D.dw_last = 0; D.nf = 0; D.it = 0; D.nreg = 0; D.th_min = 0; D.th_max = 0; D.f = 0; D.pinf = 0; D.dinf = 0; D.status = ST_USERLIMIT;
    if (sl.resume) {
        D.it = (int)st[SL_IT]; D.nf = (int)st[SL_NF]; D.nreg = (int)st[SL_NREG]; D.mu = st[SL_MU]; D.dw_last = st[SL_DWLAST]; D.th_min = st[SL_THMIN]; D.th_max = st[SL_THMAX];
        D.pinf = st[SL_PINF];
        if ((int)st[SL_HAVE]) { PAR(lane) { if (lane == 0) asm_unpack(sh.A, st + SL_ASM); } }
        PAR(lane) { const int nl = D.nf < OB_FILT_LDS ? D.nf : OB_FILT_LDS; for (int i = lane; i < 2 * nl; i += OB_NT) (&sh.filt[0][0])[i] = st[SL_FILT + i]; }
        SYNC();
        D.have_asm = (int)st[SL_HAVE];
        sl.resume = 0;
    } else { PH(ph_init(o.bound_push, o.bound_frac)); D.have_asm = 0; if (sh.soc.lsq_init) ph_recalc_y(1); }      // (IPOPT's default initial multipliers, an option here: Opts lsq_init)
    D.tau = fmax(o.tau_min, 1 - D.mu);
    D.p_start = D.it + D.nreg;
    D.dc_mu = -1.0; D.dc_val = 0;
    // D.have_asm = 1: sh.A already holds the assembly of the current iterate, left behind by the accepted trial of the previous iteration (ph_fused)
    for (;;) {
        if (sl.budget > 0 && sl.used + (D.it + D.nreg - D.p_start) + sh.soc.nsoc + sh.soc.nrebuild + sh.soc.nrecalc >= sl.budget) {   // out of budget: park the loop state, a later launch continues
            PAR(lane) {
                if (lane == 0) { st[SL_IT] = D.it; st[SL_NF] = D.nf; st[SL_NREG] = D.nreg; st[SL_MU] = D.mu; st[SL_DWLAST] = D.dw_last; st[SL_THMIN] = D.th_min; st[SL_THMAX] = D.th_max; st[SL_PINF] = D.pinf; st[SL_HAVE] = D.have_asm; st[SL_XPASS] = sh.soc.nsoc + sh.soc.nrebuild + sh.soc.nrecalc; if (D.have_asm) asm_pack(st + SL_ASM, sh.A); }
                const int nl = D.nf < OB_FILT_LDS ? D.nf : OB_FILT_LDS;                 // (entries beyond the LDS part are in the record already)
                for (int i = lane; i < 2 * nl; i += OB_NT) st[SL_FILT + i] = (&sh.filt[0][0])[i];
            }
            D.status = ST_SUSPENDED; break;
        }
        if (D.mu != D.dc_mu) { D.dc_val = o.dc_bar * pow(D.mu, o.kappa_c); D.dc_mu = D.mu; }   // a pow is a ~3k-clock dependent chain: keep it while mu stays
        PROF(sh.inst, PF_OTHER); if (!D.have_asm) PH(ph_assemble(D.mu, 0.0, D.dc_val, 0));
        D.have_asm = 0;
        if (D.it == 0) { D.th_min = 1e-4 * fmax(1.0, A.th1); D.th_max = 1e4 * fmax(1.0, A.th1); }
        D.f = A.f; D.pinf = A.pinf; D.dinf = A.dinf;
        {
            const double sd = fmax(o.s_max, (A.sumy + A.sumz) / (A.nm + A.nb)) / o.s_max;
            const double sc = fmax(o.s_max, A.sumz / A.nb) / o.s_max;
            const double E0 = fmax(A.dinf / sd, fmax(A.pinf, A.cinf0 / sc));
            if (E0 <= o.tol && A.pinf <= o.constr_viol_tol && A.dinf <= o.dual_inf_tol && A.cinf0 <= o.compl_inf_tol) { D.status = ST_OPTIMAL; break; }
            if (D.it >= o.max_iter) { D.status = ST_USERLIMIT; break; }
            if (!(A.f == A.f) || !(A.pinf == A.pinf) || !(A.dinf == A.dinf)) { D.status = ST_ERROR; break; }
            D.sd = sd; D.sc = sc;
        }
        // barrier update: mu <- max(tol/10, min(kappa_mu mu, mu^theta_mu)) while the barrier problem is solved to kappa_eps mu
        D.mu_changed = 0;
        D.cm = cinf_mu(A, D.mu);
        for (;;) {
            const double Emu = fmax(D.dinf / D.sd, fmax(D.pinf, D.cm / D.sc));
            if (Emu <= o.kappa_eps * D.mu && D.mu > o.tol / 10) {
                D.mu = fmax(o.tol / 10, fmin(o.kappa_mu * D.mu, pow(D.mu, o.theta_mu)));
                D.tau = fmax(o.tau_min, 1 - D.mu); D.nf = 0; D.mu_changed = 1;
                D.dc_val = o.dc_bar * pow(D.mu, o.kappa_c); D.dc_mu = D.mu;
                D.cm = cinf_mu(A, D.mu);      // complementarity error w.r.t. the new mu: from the extreme products of the assembly at hand (round 2 re-assembled for it)
            } else break;
        }
        // search direction with inertia correction (IPOPT Algorithm IC)
        D.dw = 0; D.ok = 0;
        for (D.tr = 0; D.tr < 60; D.tr++) {
            PROF(sh.inst, PF_OTHER); if (D.tr > 0 || D.mu_changed) PH(ph_assemble(D.mu, D.dw, D.dc_val, 0));
            int a_ = A.ok;
            PROF(sh.inst, PF_OTHER); if (a_) { PH(a_ = ph_riccati(o.rho_term)); }
            sh.ft_done = 0;
            PROF(sh.inst, PF_OTHER); if (a_) { PH(ph_direction(D.mu, D.dw, D.dc_val, o.rho_term, D.tau, o.kappa_sigma)); a_ = sh.S.ok; }
            if (a_) { D.ok = 1; break; }
            D.nreg++;
            if (D.dw == 0) D.dw = D.dw_last == 0 ? o.dw0 : fmax(o.dw_min, o.kw_dec * D.dw_last);
            else D.dw *= (D.dw_last == 0 ? o.kw_inc0 : o.kw_inc);
            if (D.dw > o.dw_max) break;
        }
        if (!D.ok) { D.status = ST_ERROR; break; }
        if (D.dw > 0) D.dw_last = D.dw;
        {
            const double th = A.th1, gd = sh.S.gd;
            D.th = th; D.phi = A.f - D.mu * A.bar; D.gd = gd; D.az = sh.S.az; D.pw_th = 0; D.pw_gd = 0;
            double amin;
            if (gd < 0) {
                amin = fmin(o.gamma_theta, o.gamma_phi * th / (-gd));
                D.pw_th = pow(th, o.s_theta); D.pw_gd = pow(-gd, o.s_phi);      // once per iteration (also the switching condition of every trial)
                if (th <= D.th_min) amin = fmin(amin, o.delta * D.pw_th / D.pw_gd);
            } else amin = o.gamma_theta;
            D.amin = amin * o.gamma_alpha;
        }
        D.alpha = sh.S.ap; D.acc = 0;
        while (D.alpha >= D.amin) {
            // the trial point z + alpha d goes to the second iterate buffer together with its assembly (mu as is, delta_w = 0: what the next iteration starts from)
            PROF(sh.inst, PF_OTHER);
            if (sh.ft_done) { sh.ft_done = 0; PH(ph_fused_stage(D.mu, D.dc_val, D.alpha, fmin(D.alpha, D.az), D.az, o.kappa_sigma, D.dw)); }      // first trial: its block part ran with the direction (ph_direction2_trial)
            else PH(ph_fused(D.mu, D.dc_val, D.alpha, fmin(D.alpha, D.az), D.az, o.kappa_sigma, D.dw));
            const double ft = sh.An.f, tht = sh.An.th1, pht = ft - D.mu * sh.An.bar, alpha = D.alpha, th = D.th, phi = D.phi, gd = D.gd;
            if (ft == ft && tht == tht && pht == pht && tht < D.th_max) {
                int okf = 1; const int nf = D.nf;
                for (int i = 0; i < nf && okf; i++) if (!(tht < filt_get(sh, st, i, 0) || pht < filt_get(sh, st, i, 1))) okf = 0;
                if (okf) {
                    const int sw = gd < 0 && alpha * D.pw_gd > o.delta * D.pw_th;
                    const int armijo = pht <= phi + o.eta_phi * alpha * gd;
                    if (th <= D.th_min && sw) { if (armijo) { D.acc = 1; break; } }
                    else if (tht <= (1 - o.gamma_theta) * th || pht <= phi - o.gamma_phi * th) {
                        D.acc = 1;
                        if (!(sw && armijo) && nf < OB_FILT) {
                            PAR(lane) { if (lane == 0) { const double f0 = (1 - o.gamma_theta) * th, f1 = phi - o.gamma_phi * th;
                                                         if (nf < OB_FILT_LDS) { sh.filt[nf][0] = f0; sh.filt[nf][1] = f1; } else { st[SL_FILT + 2 * nf] = f0; st[SL_FILT + 2 * nf + 1] = f1; } } }
                            SYNC();
                            D.nf = nf + 1;
                        }
                        break;
                    }
                }
            }
            if (sh.soc.max_soc > 0 && alpha == sh.S.ap && ft == ft && tht == tht && tht >= th) {      // second-order correction: first trial step only (alpha is still the full step sh.S.ap), and only if it did not reduce theta
                if (ph_soc_try(tht)) { D.acc = 1; break; }
            }
            D.alpha = 0.5 * alpha;
        }
        if (!D.acc) { D.status = ST_ERROR; break; }   // IPOPT would enter restoration here
        // accepted: the trial buffer becomes the iterate, its assembly the current one
        PAR(lane) { if (lane == 0) { Inst &I = sh.inst; gdbl *t_ = I.z; I.z = I.zn; I.zn = t_; sh.A = sh.An; } }
        LDS_SYNC();
        D.have_asm = 1;
        if (sh.soc.recalc_y && sh.A.pinf < OB_RECALC_FEAS_TOL) { ph_recalc_y(0); D.have_asm = 0; }      // recalc_y = "yes": least-squares multipliers at a (nearly) feasible iterate.  Whether the estimate is kept or not, the
                                                                                                // call overwrote the stage / obstacle / Riccati records with the least-squares system: the next iteration assembles afresh
        D.it++;
    }
    sl.used += D.it + D.nreg - D.p_start + sh.soc.nsoc + sh.soc.nrebuild + sh.soc.nrecalc;      // (a correction, the rebuild after a rejected one and a multiplier re-estimate are full passes each)
    R.status = D.status; R.iters = D.it; R.nreg = D.nreg; R.obj = D.f; R.pinf = D.pinf; R.dinf = D.dinf; R.mu = D.mu;
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)      // diagnostic counters of the IPOPT switches (slots behind the phase clocks): corrections tried / accepted, rebuilds, multiplier re-estimates
    if (LANE0) { sh.prof[13] += sh.soc.nsoc; sh.prof[14] += sh.soc.nrebuild + 1e-3 * sh.soc.nsoc_acc; sh.prof[15] += sh.soc.nrecalc; }
#endif
}

// Full solve of one instance (pointers already in g_sh.inst): first attempt, and on Error/UserLimit one re-solve from the last
// iterate (ParkingSignedDist.jl:256-290).  info[8] = {status, iterations, objective, pinf, dinf, mu, #regularisations, exitflag}
// Slicing: `st` is the instance's slice record, mode 1 resumes from it, budget > 0 limits the passes of this launch (info[0] = 3 when the
// solve was parked; the iterate buffer then holds the point to continue from).
OBCA_FN void solve_instance(int N, const Opts &o_arg, double *info, gdbl *st = nullptr, int mode = 0, int budget = 0, int max_soc = 0, int recalc_y = 0, int lsq_init = 0) {
    Shared &sh = g_sh;
    PAR(lane) {
        for (int i = lane; i < OB_HDR; i += OB_NT) sh.hdr[i] = sh.inst.prob[i];
        if (lane == 0) { sh.o = o_arg; sh.soc.max_soc = sh.soc.csoc ? max_soc : 0; sh.soc.recalc_y = recalc_y; sh.soc.lsq_init = lsq_init; }      // (Shared::soc.csoc is set by the caller, like the pointers of Shared::inst)
    }
    SYNC();
    PAR(lane) {
        if (lane <= OB_NOBMAX) sh.roff[lane] = (int)sh.hdr[PH_ROFF + lane];
        if (lane < OB_NOBMAX) sh.vOb[lane] = (int)sh.hdr[PH_VOB + lane];
        if (lane == 0) {
            Consts &c = sh.c;
            c.N = N; c.Ts = sh.hdr[PH_TS]; c.L = sh.hdr[PH_L]; c.iL = 1.0 / c.L; c.off = sh.hdr[PH_OFF];
            for (int i = 0; i < 4; i++) { c.g[i] = sh.hdr[PH_G + i]; c.xl[i] = sh.hdr[PH_XL + i]; c.xu[i] = sh.hdr[PH_XU + i]; c.x0[i] = sh.hdr[PH_X0 + i]; c.xF[i] = sh.hdr[PH_XF + i]; }
            c.fixTime = (int)sh.hdr[PH_FIX]; c.nOb = (int)sh.hdr[PH_NOB]; c.M = (int)sh.hdr[PH_M];
            c.dist = (int)sh.hdr[PH_DIST];
            c.wa = (c.fixTime || c.dist) ? 0.5 : 0.1; c.wpsi = c.fixTime ? 1e-2 : 1e-4;      // ParkingDist.jl:87
            make_layout(c.N, c.nOb, c.M, sh.l);
            int vmx = 0; for (int j = 0; j < c.nOb; j++) { int v = (int)sh.hdr[PH_VOB + j]; if (v > vmx) vmx = v; }
            sh.vm2 = vmx <= 2; sh.vmc = vmx <= 2 ? 0 : (vmx <= OB_VMID ? 1 : 2);
            sh.ft_ok = (c.N + 1) * c.nOb <= OB_KEEP * OB_NT; sh.ft_done = 0;
        }
    }
    init_unpack_table(sh);
    SYNC();
    // exit flag: ParkingSignedDist.jl:256-290 (Optimal -> 1; else one retry from the last iterate; if that fails too the reference's own
    // acceptance test decides) and ParkingDist.jl:245-289 (the test runs before the retry; after a failed retry it is inverted)
    // (this function's own state lives in LDS as well -- Shared::sol -- for the reason given at ipm_attempt)
    Sol &X = sh.sol; const Opts &o = sh.o;
    X.home = sh.inst.z;
    X.sl.st = st; X.sl.resume = mode == 1; X.sl.budget = budget; X.sl.used = 0;
    X.att = 0; X.it_prev = 0; X.nreg_prev = 0;
    if (mode == 1) { X.att = (int)st[SL_ATT]; X.it_prev = (int)st[SL_ITPREV]; X.nreg_prev = (int)st[SL_NREGPREV]; }
    X.R.status = ST_ERROR; X.R.iters = 0; X.R.nreg = 0; X.R.obj = X.R.pinf = X.R.dinf = X.R.mu = 0;
    X.ef = 0; X.iters = 0; X.nreg = 0; X.retry = X.att;
    if (X.att == 0) {
        ipm_attempt(o, X.R, X.sl);
        X.iters = X.R.iters; X.nreg = X.R.nreg;
        if (X.R.status != ST_SUSPENDED) {
            X.ef = (X.R.status == ST_OPTIMAL); X.retry = !X.ef;
            if (X.retry && sh.c.dist && ph_ref_constraints(0)) { X.ef = 1; X.retry = 0; }
            if (X.retry) { X.att = 1; X.it_prev = X.R.iters; X.nreg_prev = X.R.nreg; }
        }
    }
    if (X.retry && X.R.status != ST_SUSPENDED) {
        ipm_attempt(o, X.R, X.sl);
        X.iters = X.it_prev + X.R.iters; X.nreg = X.nreg_prev + X.R.nreg;
        if (X.R.status == ST_OPTIMAL) X.ef = 1;
        else if (X.R.status != ST_SUSPENDED) { const int feas = ph_ref_constraints(sh.c.dist ? 0 : 1); X.ef = sh.c.dist ? !feas : feas; }
    }
    if (X.R.status == ST_SUSPENDED) {
        PAR(lane) { if (lane == 0) { st[SL_ATT] = X.att; st[SL_ITPREV] = X.it_prev; st[SL_NREGPREV] = X.nreg_prev; } }
        X.ef = 0;
    }
    if (sh.inst.z != X.home) ph_bring_home();      // the accepted trial points alternate between the two iterate buffers; results and parked solves live in the instance's own
    PAR(lane) {
        if (lane == 0) { info[0] = X.R.status; info[1] = X.iters; info[2] = X.R.obj; info[3] = X.R.pinf; info[4] = X.R.dinf; info[5] = X.R.mu; info[6] = X.nreg; info[7] = X.ef; }
    }
    SYNC();
}

}  // namespace obca
