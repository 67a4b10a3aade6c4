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
#ifdef OBCA_EMU_RACE      // hazard-detecting build of the emulation (tests/test_emu_sanitize.py): PAR publishes the lane at work, the synchronisation points count epochs, and every
                          // access to a per-instance HBM buffer goes through `gdbl` below -- a word that one lane writes and another lane reads or writes before the next point
                          // at which the wavefront's global stores are known to have landed (SYNC, VM_DRAIN) is reported
namespace race { extern int lane; void sync(int drains_global_memory); void rd(const void *p); void wr(const void *p); }
#define PAR(lane) for (int lane = 0; (race::lane = lane) < OB_NT; ++lane)
#define PAR64(lane) for (int lane = 0; (race::lane = lane) < 64; ++lane)
#define SYNC() race::sync(1)
#define LDS_SYNC() race::sync(0)
#define VM_DRAIN() race::sync(1)
#else
#define PAR(lane) for (int lane = 0; lane < OB_NT; ++lane)
#define PAR64(lane) for (int lane = 0; lane < 64; ++lane)       // inside a WAVE0 section
#define SYNC() ((void)0)
#define LDS_SYNC() ((void)0)
#define VM_DRAIN() ((void)0)
#endif
#define WAVE0_BEGIN {
#define WAVE0_END }
#define LANE0 1
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
#define PAR(lane) for ([[maybe_unused]] int lane = (int)threadIdx.x, once_ = 1; once_; once_ = 0)      // (a region that only uses LI(lane) = 0 leaves `lane` unused)
#define PAR64(lane) for ([[maybe_unused]] int lane = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
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
#if defined(OBCA_EMU) && defined(OBCA_EMU_RACE)
struct gdbl {      // a double in a per-instance HBM buffer whose loads and stores are logged with the lane that issued them
    double v;
    operator double() const { race::rd(this); return v; }
    gdbl &operator=(double x) { race::wr(this); v = x; return *this; }
    gdbl &operator=(const gdbl &o) { const double x = o; return *this = x; }
    gdbl &operator+=(double x) { const double y = *this; return *this = y + x; }
    gdbl &operator-=(double x) { const double y = *this; return *this = y - x; }
    gdbl &operator*=(double x) { const double y = *this; return *this = y * x; }
};
#elif defined(OBCA_EMU)
typedef double gdbl;
#else
typedef __attribute__((address_space(1))) double gdbl;
#endif

namespace obca {

#define OB_NC 6      // Riccati right-hand sides: main, t, nu1..nu4
#define OB_NMAX 128  // longest horizon: the forward sweep gives ONE lane to every pair of stages (direction_main: 64 pairs per wavefront); LDS would allow more (5.7 KB + (27 N + 54) x 8 bytes)
#define OB_RIT_FIELDS 25
#ifdef OBCA_EMU
typedef int ob_rit_t;        // (host emulation: static and dynamic LDS are two host arrays, their distance does not fit 16 bits)
#else
typedef short ob_rit_t;      // offsets in doubles within the workgroup's LDS (< 8 192)
#endif
#define OB_AS 60     // doubles per assembled stage record (only the entries that can be non-zero are kept: as_h / as_df below)
#define OB_RS 74     // doubles per Riccati stage record
#define OB_OC 12     // doubles per condensed obstacle record
// problem header (doubles, in front of rx, ry, ryaw): 26 scalars, then the obstacle set
// -- row counts, row offsets, the rows themselves.  It lives in LDS for the whole solve
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
// obca_opts of the C ABI: the interior-point options + the three IPOPT switches (max_soc: second-order correction trials per iteration,
struct OptsAbi { Opts o; int max_soc, recalc_y, lsq_init, obj_scaling, restoration; };
                                                          // IPOPT's default 4; recalc_y; lsq_init; all 0 = off by default, as in the checker).  Kept
                                                          // apart so that the options' place in LDS (Shared::o) is what the phases were tuned with.

struct Lay {
    // zs1: multiplier of the norm-row slack (ParkingDist only)
    int x, u, t, lam, mu, sl, so, ss, pi, nu, yg, yo, zxL, zxU, zuL, zuU, ztL, ztU, zlam, zmu, zso, zssL, zssU, zs1, nprimal, len;
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

// cmin, cmax: extreme complementarity products
struct AsmOut { int ok; double dinf, pinf, cinf0, cmin, cmax, sumy, sumz, f, th1, bar, Htt, gtb; int nb, nm; };
template <class P> OBCA_FN void asm_pack(P *o, const AsmOut &A) {
    o[0] = A.ok; o[1] = A.dinf; o[2] = A.pinf; o[3] = A.cinf0; o[4] = A.cmin; o[5] = A.cmax; o[6] = A.sumy; o[7] = A.sumz; o[8] = A.f; o[9] = A.th1;
    o[10] = A.bar; o[11] = A.Htt; o[12] = A.gtb;
    o[13] = A.nb; o[14] = A.nm;
}
template <class P> OBCA_FN void asm_unpack(AsmOut &A, const P *o) {
    A.ok = (int)o[0]; A.dinf = o[1]; A.pinf = o[2]; A.cinf0 = o[3]; A.cmin = o[4]; A.cmax = o[5]; A.sumy = o[6]; A.sumz = o[7]; A.f = o[8]; A.th1 = o[9];
    A.bar = o[10]; A.Htt = o[11]; A.gtb = o[12];
    A.nb = (int)o[13]; A.nm = (int)o[14];
}
// complementarity error w.r.t. the barrier parameter mu: max |s z - mu|
OBCA_HD double cinf_mu(const AsmOut &A, double mu) { return fmax(fabs(A.cmax - mu), fabs(A.cmin - mu)); }
struct StepOut { int ok; double ap, az, gd, gr; };   // gr: rate-cost part of d phi / d t (summed in the stage back-substitution, used with dt at the end)
struct Consts; struct Lay;
struct Inst {              // uniform: pointers of this instance
    const gdbl *prob;      // header + rx, ry, ryaw
    // z: the current iterate; zn: the buffer the line search writes its trial point to (the two swap when a trial is accepted);
    gdbl *z, *zn, *d, *as, *rs, *oc;
                                              // d: stage part of the search direction (u, ss, pi, yg; the
                                              // obstacle part is recomputed where it is needed, x lives in LDS)
    mutable long long tlast;           // diagnostic builds (-DOBCA_PROFILE): time stamp of the previous phase boundary
};

enum { SL_ATT = 0, SL_IT, SL_NF, SL_NREG, SL_MU, SL_DWLAST, SL_THMIN, SL_THMAX, SL_ITPREV, SL_NREGPREV, SL_PINF, SL_HAVE, SL_XPASS, SL_NREST, SL_ASM = 16, SL_FILT = 32, SL_SIZE = SL_FILT + 2 * OB_FILT };
// SL_XPASS: full passes the slice spent outside iterations and inertia rungs (second-order corrections,
// rebuilds after rejected ones, multiplier re-estimates): the ordering kernel ranks by them too
// SL_NREST: block restorations of this attempt so far (+ 16 if the filter thresholds are to be re-initialised by the next assembly)
// SL_HAVE / SL_ASM: a solve parked right after an accepted trial keeps that trial's
// assembly -- the scalars here, the stage records in the instance's own buffers, which
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
// state of the three IPOPT switches (second-order correction, recalc_y, least-squares initial multipliers: cold paths); at the END of Shared, so that
struct Soc {
                            // nothing the phases of the default path address moves (their code is
                            // instruction-for-instruction that of the build without the switches)
    gdbl *csoc;             // c_soc = alpha c(z) + c(z + alpha d) of the instance, layout pi | nu | yg | yo as in the iterate (null unless max_soc > 0)
    int max_soc, nsoc, nsoc_acc;      // option; corrections tried / accepted in this attempt (diagnostic)
    int recalc_y, nrecalc;            // option recalc_y = "yes"; multiplier re-estimates in this attempt (diagnostic)
    int lsq_init;                     // option: least-squares initial multipliers (IPOPT's default initialisation, constr_mult_init_max = 1e3)
    int nrebuild;                     // corrections rejected in this attempt (rounds 4-5 rebuilt the Newton system after each: a full pass; since round 6 the direction is kept)
    int xpass0;                       // correction / rebuild / re-estimate passes of EARLIER slices of this attempt (SL_XPASS is cumulative like SL_NREG).  (Kept here, at the end of
                                      // Shared: one more int in Drv moved everything behind it by 8 bytes, off the 16-byte boundaries the phases read `c`, `A*`, `inst` at -- 3.5 %
                                      // of `value`, profiles/r05_ab_lds_alignment.txt)
    // option restoration (block feasibility restoration: restore_blocks, obca_solver_ipm.h); restorations of this attempt; 1: the next assembly re-initialises theta_min / theta_max
    int restoration, nrest, reset_th;
    // a correction writes ITS direction to a buffer of its own (dsoc) so that a rejected one leaves the iteration's direction where the backtracking goes on with it
    // (rounds 4-5 let it overwrite d and rebuilt the Newton system afterwards: a full pass per rejected correction); what the correction's phases overwrite besides: kept here
    gdbl *dsoc; double coef_keep[5]; StepOut S_keep;
};
#define OB_FILT_LDS 32     // filter entries kept in LDS; the (rare) rest lives in the instance's slice record

struct alignas(16) Shared {
    double hdr[OB_HDR];
    alignas(16) double Bm[36], coef[8];                         // border constants (left by the backward sweep for the border solve), (dt, nu)
    double filt[OB_FILT_LDS][2];
    // (Layout note, round 5, profiles/r05_ab_lds_alignment.txt: one more int in Drv -- `sol`, `o`, `roff` .. `Ap` 8 bytes further back -- cost 3.5 % of `value`; so did putting every
    //  member on a 16-byte boundary; where the dynamic block behind `Shared` starts (0-240 bytes of padding) made no difference.  New wave-uniform state goes to the END, into Soc.)
    Drv drv; Sol sol; Opts o;      // (the options too: as kernel arguments they would sit in ~60 SGPRs that are spilled around every phase call)
    // upl, ucn: which positions of the unpacked stage data a lane serves (init_unpack_table)
    int roff[OB_NOBMAX + 1], vOb[OB_NOBMAX], ric_ok; int upl[OB_NT], ucn[3][OB_NT];
    // the Riccati sweep's per-lane operand table (RicItem, obca_solver_riccati.h): the same for every sweep of a solve, built once by solve_instance (round 6: it was rebuilt
    // by every sweep, ~1 100 instructions of index arithmetic per lane); [field][lane], so that a field is one conflict-free read
    ob_rit_t rit[OB_RIT_FIELDS][OB_NT];
    Consts c; Lay l;
    double prof[16];           // diagnostic per-phase cycle counters (-DOBCA_PROFILE)
    // vmc: row class of the instance's widest obstacle (0: <= 2, 1: <= OB_VMID, 2: <= OB_VMAX)   // phase inputs/outputs (wave-uniform, exchanged through LDS)
    Inst inst; AsmOut A; AsmOut A2; AsmOut An; AsmOut Ap; StepOut S; int vm2, vmc;
    Soc soc;
    // ft_ok: the instance's (stage, obstacle) items fit OB_KEEP rounds (first-trial block part
    // merged into the direction phase); ft_done: that part has run for the direction at hand
    int ft_ok, ft_done;
};

// Dynamic LDS behind `Shared`, sized for the horizon at launch (OB_DYN_LDS_DOUBLES).
// Three layouts share it, one per phase of a pass (they never overlap in time):
//   forward sweep / line search : [ trajectory (N + 2) x 6 | composed closed-loop maps of
//   the stage pairs (N / 2 + 1) x 42 ]      s_k = (dx_k, dw_k): the x part of the search
//                                 direction lives in the trajectory and nowhere else (direction_*, the fused line search read it)
//   assembly                    : [ trajectory (still the direction the trial point is formed along) | condensed obstacle sums (N + 1) x 12 ]
//   backward sweep              : [ per-stage border data N x RIC_BD | two unpacked stage buffers
//   2 x OB_STG (SG_* offsets, + a pad slot) | operands RicLds ]   -- the trajectory is dead
//                                 by then (a backward sweep always starts a new direction), so the sweep uses the region from its start
// Round 3 kept the sweep's operands (2.4 KB) in the static block; with them here the block is 5 KB and seven instances fit a CU's 160 KB instead of six.
#define OB_STG 200
struct alignas(16) RicLds {
    // Riccati backward sweep.  Every operand of its dot products is a CONTIGUOUS, 16-byte
    // aligned 6-vector (P rows, the rows of the transposed FA', T', p'), read as three
    // ds_read_b128: a lone wavefront per SIMD issues 8-byte LDS reads at a fifth of the LDS rate but 16-byte reads at the full rate (MI355X_MICROARCH.md, LDS).
    // P (row-major, symmetric), p' (pn[c * 6 + a]: right-hand side c, state a), Qhat (8 x 14 row-major)
    alignas(16) double Pn[36], pn[6 * OB_NC], Qhat[8 * 14];
    alignas(16) double sB[24], TT[14 * 6];                      // static parts of the border constants, T' (TT[cc * 6 + a])
    // constant 0 and a write-only slot: operand / destination of the lanes without an item in the Riccati phases
    alignas(16) double zero6[6]; double zero, dump, dump4[4];
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

// pair maps of the forward sweep / condensed obstacle sums of the assembly: behind the trajectory
OBCA_FN double *stg_base(const Shared &sh) { return g_traj + (size_t)(sh.c.N + 2) * 6; }
// backward sweep: the two stage buffers, behind the per-stage border data (RIC_BD = 16 doubles per stage)
OBCA_FN double *ric_sg0(const Shared &sh) { return g_traj + (size_t)sh.c.N * 16; }
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

// The solver in the order of a factorisation pass (each file continues namespace obca; the split is by phase, the programming model is stated at the top):
#include "obca_solver_lanes.h"
#include "obca_solver_assemble.h"
#include "obca_solver_riccati.h"
#include "obca_solver_direction.h"
#include "obca_solver_ipm.h"

}  // namespace obca
