// TEST INFRASTRUCTURE ONLY.  Host emulation of the workgroup-per-instance HIP solver: the device source
// obca_amd/csrc/obca_solver.h is compiled with -DOBCA_EMU, which turns every PAR(lane) region into a plain
// loop over 64 lanes.  It lets the CPU test-suite check the kernel logic (Newton direction, full solves) against
// the oracle on a machine without a GPU.  It is never linked into libobca_hip.so.
// Two further builds of this file serve tests/test_emu_sanitize.py:
//   -DOBCA_EMU_RACE               every access to a per-instance HBM buffer is logged with its lane; cross-lane hazards between two drains of the wavefront's global stores are reported
//   -DOBCA_EMU_ASAN -fsanitize=address   the buffers have exactly the sizes the HIP host code allocates per instance and the dynamic LDS block ends where the launch's does
#define OBCA_EMU 1
#ifdef OBCA_EMU_ASAN
#include <sanitizer/asan_interface.h>
#endif
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdio>
#include "../../obca_amd/csrc/obca_solver.h"
#include "../../obca_amd/csrc/obca_quad_solver.h"
using namespace obca;
extern "C" { void emu_race_begin(); void emu_race_buffer(const char *name, const void *base, long doubles, int record, int pad_slot); void emu_race_dummy_load(const void *word); void emu_race_end(); }
#ifndef OBCA_EMU_RACE
void emu_race_begin() {} void emu_race_buffer(const char *, const void *, long, int, int) {} void emu_race_dummy_load(const void *) {} void emu_race_end() {}
#endif
static int g_csoc_len = 0;      // doubles of the c_soc buffer (0: as long as the iterate; the HIP host code allocates zxL - pi of the batch's largest layout)
extern "C" void emu_set_csoc_len(int n) { g_csoc_len = n; }

struct Scratch { double *z, *zn, *d, *as, *rs, *oc, *csoc, *dsoc; };
static void alloc_scratch(int N, int len, Scratch &s) {
#ifdef OBCA_EMU_ASAN      // the dynamic LDS block of a launch ends at OB_DYN_LDS_DOUBLES(N): whatever lies behind it in the emulation's static array is out of bounds
    ASAN_UNPOISON_MEMORY_REGION(g_traj, sizeof g_traj); memset(g_traj, 0, sizeof g_traj);
    ASAN_POISON_MEMORY_REGION(g_traj + OB_DYN_LDS_DOUBLES(N), sizeof g_traj - OB_DYN_LDS_DOUBLES(N) * sizeof(double));
#endif
    s.z = (double *)calloc(len, 8); s.zn = (double *)calloc(len, 8); s.d = (double *)calloc(len, 8);
    s.as = (double *)calloc((size_t)(N + 1) * OB_AS, 8); s.rs = (double *)calloc((size_t)(N + 1) * OB_RS, 8);
    s.oc = (double *)calloc((size_t)(N + 1) * OB_NOBMAX * OB_OC, 8); s.csoc = (double *)calloc(g_csoc_len ? g_csoc_len : len, 8); s.dsoc = (double *)calloc(len, 8);      // (c_soc: the equality rows, fewer than len)
}
static void free_scratch(Scratch &s) { free(s.z); free(s.zn); free(s.d); free(s.as); free(s.rs); free(s.oc); free(s.csoc); free(s.dsoc); }

static void setup(int N, const double *prob, Scratch &s) {
    Shared &sh = g_sh; Inst &I = sh.inst;
    I.prob = (const gdbl *)prob; I.z = (gdbl *)s.z; I.zn = (gdbl *)s.zn; I.d = (gdbl *)s.d; I.as = (gdbl *)s.as; I.rs = (gdbl *)s.rs; I.oc = (gdbl *)s.oc; g_sh.soc.csoc = (gdbl *)s.csoc; g_sh.soc.dsoc = (gdbl *)s.dsoc;
    for (int i = 0; i < OB_HDR; i++) sh.hdr[i] = prob[i];
    for (int i = 0; i <= OB_NOBMAX; i++) sh.roff[i] = (int)sh.hdr[PH_ROFF + i];
    for (int i = 0; i < OB_NOBMAX; i++) sh.vOb[i] = (int)sh.hdr[PH_VOB + i];
    Consts &c = sh.c; c.N = N;
    c.Ts = sh.hdr[PH_TS]; c.L = sh.hdr[PH_L]; c.iL = 1.0 / c.L; c.off = sh.hdr[PH_OFF];
    for (int i = 0; i < 4; i++) { c.g[i] = sh.hdr[PH_G + i]; c.xl[i] = sh.hdr[PH_XL + i]; c.xu[i] = sh.hdr[PH_XU + i]; c.x0[i] = sh.hdr[PH_X0 + i]; c.xF[i] = sh.hdr[PH_XF + i]; }
    c.fixTime = (int)sh.hdr[PH_FIX]; c.nOb = (int)sh.hdr[PH_NOB]; c.M = (int)sh.hdr[PH_M];
    c.dist = (int)sh.hdr[PH_DIST];
    c.wa = (c.fixTime || c.dist) ? 0.5 : 0.1; c.wpsi = c.fixTime ? 1e-2 : 1e-4;
    make_layout(c.N, c.nOb, c.M, sh.l);
    int vmx = 0; for (int j = 0; j < c.nOb; j++) if (sh.vOb[j] > vmx) vmx = sh.vOb[j];
    sh.vm2 = vmx <= 2; sh.vmc = vmx <= 2 ? 0 : (vmx <= OB_VMID ? 1 : 2);
    init_unpack_table(sh);
    init_ric_table(sh);
}

extern "C" {
void emu_sincos(double x, double *s, double *c) { sincos_bounded(x, s, c); }      // the kernels' bounded-range sin / cos (obca_model.h)
double emu_tan(double x) { return tan_bounded(x); }
int emu_opts_size() { return (int)sizeof(OptsAbi); }
int emu_last_recalc() { return g_sh.soc.nrecalc; }
void emu_force_recalc_failure(int on) { g_emu_recalc_fail = on; }      // every recalc_y estimate is thrown away after its system has overwritten the records (tests/test_emu_cpu.py)
int emu_last_soc(int *accepted) { if (accepted) *accepted = g_sh.soc.nsoc_acc; return g_sh.soc.nsoc; }      // second-order corrections of the last attempt of the last solve

// one Newton direction at a full primal-dual point (oracle layout); returns inertia-ok.  alpha >= 0: the fused line-search step is run as well --
// znext = the trial point z + alpha d with the new multipliers (ay = min(alpha, az)), aux[10..15] = f, th1, bar, dinf, pinf, cinf0 of ITS assembly
static int newton_impl(int N, const double *prob, const double *zin, int len, double mu, double dw, double dc, double rho, double tau,
                       double *dout, double *aux, double alpha, double ks, double *znext) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zin, sizeof(double) * len);
    setup(N, prob, s); Shared &sh = g_sh; Inst &I = sh.inst;
    AsmOut A; const FuseArgs nf = {0, 0, 0, 0, 0};
    if (sh.vmc == 0) assemble_obs<2, 0>(I, sh, mu, dw, dc, nf); else if (sh.vmc == 1) assemble_obs<OB_VMID, 0>(I, sh, mu, dw, dc, nf); else assemble_obs<OB_VMAX, 0>(I, sh, mu, dw, dc, nf);
    assemble_stage<0>(I, sh, mu, dw, dc, nf, A);
    int ok = A.ok;
    StepOut S; S.ap = S.az = S.gd = 0;
    if (ok) ok = riccati_backward(I, sh, rho);
    if (ok) { direction_main(I, sh, A, mu, dw, dc, rho, tau, S); ok = S.ok; }
    if (ok) { if (sh.vmc == 0) direction_obs<2, 1>(I, sh, mu, dw, dc, tau, S); else if (sh.vmc == 1) direction_obs<OB_VMID, 1>(I, sh, mu, dw, dc, tau, S); else direction_obs<OB_VMAX, 1>(I, sh, mu, dw, dc, tau, S); }
    memcpy(dout, s.d, sizeof(double) * len);
    aux[0] = A.dinf; aux[1] = A.pinf; aux[2] = A.cinf0; aux[3] = cinf_mu(A, mu); aux[4] = A.f; aux[5] = A.th1; aux[6] = A.bar;
    aux[7] = S.ap; aux[8] = S.az; aux[9] = S.gd;
    if (ok && alpha >= 0 && znext) {
        const FuseArgs fa = {alpha, alpha < S.az ? alpha : S.az, S.az, ks, dw}; AsmOut An;
        if (sh.vmc == 0) assemble_obs<2, 1>(I, sh, mu, 0.0, dc, fa); else if (sh.vmc == 1) assemble_obs<OB_VMID, 1>(I, sh, mu, 0.0, dc, fa); else assemble_obs<OB_VMAX, 1>(I, sh, mu, 0.0, dc, fa);
        assemble_stage<1>(I, sh, mu, 0.0, dc, fa, An);
        memcpy(znext, s.zn, sizeof(double) * len);
        aux[10] = An.f; aux[11] = An.th1; aux[12] = An.bar; aux[13] = An.dinf; aux[14] = An.pinf; aux[15] = An.cinf0;
    }
    free_scratch(s);
    return ok;
}
// the kernels' second-order-correction step at a given point for given constraint values csoc (layout pi | nu | yg | yo): the SOC = 1 instantiations of the phases
int emu_newton_soc(int N, const double *prob, const double *zin, int len, double mu, double dw, double dc, double rho, double tau, const double *csoc, double *dout) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zin, sizeof(double) * len);
    setup(N, prob, s); Shared &sh = g_sh; Inst &I = sh.inst;
    memcpy(s.csoc, csoc, sizeof(double) * (sh.l.zxL - sh.l.pi));
    AsmOut A; const FuseArgs nf = {0, 0, 0, 0, 0};
    if (sh.vmc == 0) assemble_obs<2, 0, 1>(I, sh, mu, dw, dc, nf); else if (sh.vmc == 1) assemble_obs<OB_VMID, 0, 1>(I, sh, mu, dw, dc, nf); else assemble_obs<OB_VMAX, 0, 1>(I, sh, mu, dw, dc, nf);
    assemble_stage<0, 1>(I, sh, mu, dw, dc, nf, A);
    int ok = A.ok;
    StepOut S; S.ap = S.az = S.gd = 0;
    if (ok) ok = riccati_backward<1>(I, sh, rho);
    if (ok) { direction_main<1>(I, sh, A, mu, dw, dc, rho, tau, S); ok = S.ok; }
    if (ok) { if (sh.vmc == 0) direction_obs<2, 1, 1>(I, sh, mu, dw, dc, tau, S); else if (sh.vmc == 1) direction_obs<OB_VMID, 1, 1>(I, sh, mu, dw, dc, tau, S); else direction_obs<OB_VMAX, 1, 1>(I, sh, mu, dw, dc, tau, S); }
    memcpy(dout, s.d, sizeof(double) * len);
    free_scratch(s);
    return ok;
}
// the least-squares multiplier step of the kernels (ph_recalc_y: recalc_y / lsq_init) at a given point: dout = z_after - z_before (non-zero on [pi, zxL) only)
int emu_lsq(int N, const double *prob, const double *zin, int len, double *dout) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zin, sizeof(double) * len);
    setup(N, prob, s);
    const int ok = ph_recalc_y(0);
    for (int i = 0; i < len; i++) dout[i] = s.z[i] - zin[i];
    free_scratch(s);
    return ok;
}
int emu_newton(int N, const double *prob, const double *zin, int len, double mu, double dw, double dc, double rho, double tau,
               double *dout, double *aux /* dinf,pinf,cinf0,cinfmu,f,th1,bar,ap,az,gd */) {
    return newton_impl(N, prob, zin, len, mu, dw, dc, rho, tau, dout, aux, -1.0, 0.0, nullptr);
}
int emu_newton_fused(int N, const double *prob, const double *zin, int len, double mu, double dw, double dc, double rho, double tau, double alpha, double ks,
                     double *dout, double *aux /* 16 */, double *znext) {
    return newton_impl(N, prob, zin, len, mu, dw, dc, rho, tau, dout, aux, alpha, ks, znext);
}

// full solve; zinit holds the primal warm start in the oracle layout (x,u,t,lam,mu; sl=0)
// OBCA_EMU_POISON = mask (environment): before a solve, NaN-fill what the kernels must not read before they write it -- 1: the work buffers in "HBM" (zn, d, as, rs, csoc, the slice
// record), 2: the static LDS block, 4: the dynamic LDS block.  The result must not change (tests/test_emu_cpu.py); round 5 found on the GPU that a build with several of them
// poisoned gave other (finite) results than the product build.
static void emu_poison(int N, int len, Scratch &s, double *st) {
    const char *e = getenv("OBCA_EMU_POISON"); const int m = e ? atoi(e) : 0;
    if (!m) return;
    const char *pv = getenv("OBCA_EMU_POISON_VALUE");      // NaN hides from fmax / fmin and from every comparison: a finite pattern (1e30, -1e30, 0.5 ...) finds what they swallow
    const double nan_ = pv ? atof(pv) : NAN;
    if (m & 1) {
        for (int i = 0; i < len; i++) { s.zn[i] = nan_; s.d[i] = nan_; }
        for (int i = 0; i < (g_csoc_len ? g_csoc_len : len); i++) s.csoc[i] = nan_;
        for (int i = 0; i < len; i++) s.dsoc[i] = nan_;
        for (size_t i = 0; i < (size_t)(N + 1) * OB_AS; i++) s.as[i] = nan_;
        for (size_t i = 0; i < (size_t)(N + 1) * OB_RS; i++) s.rs[i] = nan_;
        for (int i = 0; i < SL_SIZE; i++) st[i] = nan_;
    }
    if (m & 2) {      // OBCA_EMU_POISON_RANGE=lo:hi restricts the fill to the doubles [lo, hi) of the block (to bisect for the member that is read before it is written)
        double *w = (double *)&g_sh; size_t lo = 0, hi = sizeof(Shared) / sizeof(double);
        if (const char *r = getenv("OBCA_EMU_POISON_RANGE")) { unsigned long a = 0, b = hi; if (sscanf(r, "%lu:%lu", &a, &b) == 2) { lo = a; hi = b < hi ? b : hi; } }
        for (size_t i = lo; i < hi; i++) w[i] = nan_;
    }
    if (m & 4) for (size_t i = 0; i < OB_DYN_LDS_DOUBLES(N); i++) g_traj[i] = nan_;
}
static void emu_poison_lds_only(int N) {      // the LDS part of emu_poison (a resumed launch: the HBM buffers carry the parked solve)
    const char *e = getenv("OBCA_EMU_POISON"); const int m = e ? atoi(e) : 0;
    if (!m) return;
    const char *pv = getenv("OBCA_EMU_POISON_VALUE"); const double v = pv ? atof(pv) : NAN;
    if (m & 2) { double *w = (double *)&g_sh; for (size_t i = 0; i < sizeof(Shared) / sizeof(double); i++) w[i] = v; }
    if (m & 4) for (size_t i = 0; i < OB_DYN_LDS_DOUBLES(N); i++) g_traj[i] = v;
}
int emu_solve(int N, const double *prob, const double *zinit, int len, const void *opts, double *zout, double *info) {
    Scratch s; alloc_scratch(N, len, s);
    { double *st0 = (double *)calloc(SL_SIZE, 8); emu_poison(N, len, s, st0); free(st0); }
    memcpy(s.z, zinit, sizeof(double) * len);
    Inst &I = g_sh.inst; I.prob = (const gdbl *)prob; I.z = (gdbl *)s.z; I.zn = (gdbl *)s.zn; I.d = (gdbl *)s.d; I.as = (gdbl *)s.as; I.rs = (gdbl *)s.rs; I.oc = (gdbl *)s.oc; g_sh.soc.csoc = (gdbl *)s.csoc; g_sh.soc.dsoc = (gdbl *)s.dsoc;
    double *st = (double *)calloc(SL_SIZE, 8);
    emu_race_begin();
    emu_race_buffer("iterate buffer A", s.z, len, 0, -1); emu_race_buffer("iterate buffer B", s.zn, len, 0, -1); emu_race_buffer("d", s.d, len, 0, -1);
    emu_race_buffer("as", s.as, (long)(N + 1) * OB_AS, OB_AS, -1); emu_race_buffer("rs", s.rs, (long)(N + 1) * OB_RS, OB_RS, RS_PAD);
    emu_race_buffer("csoc", s.csoc, g_csoc_len ? g_csoc_len : len, 0, -1); emu_race_buffer("dsoc", s.dsoc, len, 0, -1); emu_race_buffer("slice record", st, SL_SIZE, 0, -1);
    struct RaceOff { ~RaceOff() { emu_race_end(); } } race_off_;
    solve_instance(N, ((const OptsAbi *)opts)->o, info, (gdbl *)st, 0, 0, ((const OptsAbi *)opts)->max_soc, ((const OptsAbi *)opts)->recalc_y, ((const OptsAbi *)opts)->lsq_init, ((const OptsAbi *)opts)->restoration);
    free(st);
    memcpy(zout, s.z, sizeof(double) * len);
    free_scratch(s);
    return 0;
}

// the same solve cut into launches of `budget` factorisation passes each (time slicing, obca_solver.h): returns the number of launches
int emu_solve_sliced(int N, const double *prob, const double *zinit, int len, const void *opts, int budget, double *zout, double *info) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zinit, sizeof(double) * len);
    double *st = (double *)calloc(SL_SIZE, 8);
    int launches = 0;
    for (int mode = 0;; mode = 1) {
        memset(&g_sh, 0, sizeof g_sh);                       // nothing survives a launch but HBM: the iterate and the slice record
        if (mode == 0) emu_poison(N, len, s, st); else { Scratch none = s; emu_poison_lds_only(N); (void)none; }      // (OBCA_EMU_POISON: every launch meets a foreign pattern in LDS)
        Inst &I = g_sh.inst; I.prob = (const gdbl *)prob; I.z = (gdbl *)s.z; I.zn = (gdbl *)s.zn; I.d = (gdbl *)s.d; I.as = (gdbl *)s.as; I.rs = (gdbl *)s.rs; I.oc = (gdbl *)s.oc; g_sh.soc.csoc = (gdbl *)s.csoc; g_sh.soc.dsoc = (gdbl *)s.dsoc;
        solve_instance(N, ((const OptsAbi *)opts)->o, info, (gdbl *)st, mode, budget, ((const OptsAbi *)opts)->max_soc, ((const OptsAbi *)opts)->recalc_y, ((const OptsAbi *)opts)->lsq_init, ((const OptsAbi *)opts)->restoration);
        launches++;
        if ((int)info[0] != ST_SUSPENDED || launches > 100000) break;
    }
    memcpy(zout, s.z, sizeof(double) * len);
    free(st); free_scratch(s);
    return launches;
}

int emu_dualws(int v, const double *a1, const double *a2, const double *b, const double *g, double ex, double ey, double cs, double sn,
               double *lam, double *mu, double *d) {
    double l4[OB_VMAX], m4[4];
    dualws_one<OB_VMAX>(v, a1, a2, b, g, ex, ey, cs, sn, l4, m4, d);
    for (int i = 0; i < v; i++) lam[i] = l4[i];
    for (int i = 0; i < 4; i++) mu[i] = m4[i];
    return 0;
}

// ---------------------------------------------------------------- quadcopter path (obca_quad_solver.h)
struct QScratch { double *z, *d, *as, *rs, *oc; };
static void q_alloc(int N, QScratch &s, quad::QLay &l) {
    quad::q_make_layout(N, l);
    s.z = (double *)calloc(l.len, 8); s.d = (double *)calloc(QDIR_DOUBLES(l), 8);      // two direction buffers + the rows of a second-order correction
    s.as = (double *)calloc((size_t)(N + 1) * QSP, 8); s.rs = (double *)calloc((size_t)(N + 1) * QRR, 8);
    s.oc = (double *)calloc((size_t)(N + 1) * QOB * OB_OC, 8);
}
static void q_free(QScratch &s) { free(s.z); free(s.d); free(s.as); free(s.rs); free(s.oc); }
static void q_setup(int N, const double *prob, QScratch &s) {
    quad::QShared &sh = quad::gq_sh; quad::QConsts &c = sh.c;
    sh.inst.prob = (const gdbl *)prob; sh.inst.z = (gdbl *)s.z; sh.inst.d = (gdbl *)s.d; sh.inst.as = (gdbl *)s.as; sh.inst.rs = (gdbl *)s.rs; sh.inst.oc = (gdbl *)s.oc;
    c.N = N; c.dist = (int)prob[QPH_DIST]; c.Ts = prob[QPH_TS]; c.R = prob[QPH_R]; c.wH = sqrt((Q_MASS * Q_GRAV) / (Q_KF * 4));
    for (int i = 0; i < QX; i++) { c.x0[i] = prob[QPH_X0 + i]; c.xF[i] = prob[QPH_XF + i]; }
    for (int i = 0; i < 3; i++) c.gyro[i] = c.x0[9 + i];
    for (int i = 0; i < QOB * QL; i++) sh.ob[i] = prob[QPH_OB + i];
    quad::q_make_layout(N, sh.l); sh.soc_on = 0; sh.inst.d0 = (gdbl *)s.d; c.sf = 1.0;
}
int emu_quad_layout(int N, int *out) { quad::QLay l; quad::q_make_layout(N, l); memcpy(out, &l, sizeof l); return (int)(sizeof l / sizeof(int)); }

// one Newton direction at a full primal-dual point: zin in the device layout v | y | zL | zU, dout = dv | dy
int emu_quad_newton(int N, const double *prob, const double *zin, double mu, double dw, double dc, double rho, double tau, double *dout, double *aux /* 11 */) {
    QScratch s; quad::QLay l; q_alloc(N, s, l);
    memcpy(s.z, zin, sizeof(double) * l.len);
    q_setup(N, prob, s); quad::QShared &sh = quad::gq_sh;
    // constants of the dense stage record (the solver writes them in its init phase)
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < 3; i++) { s.as[(size_t)k * QSP + quad::QR(QSR_F + i * QFC + i)] = 1.0; s.as[(size_t)k * QSP + quad::QR(QSR_F + (6 + i) * QFC + 6 + i)] = 1.0; }
        for (int j = 0; j < QU; j++) s.as[(size_t)k * QSP + quad::QR(QSR_F + (QX + j) * QFC + QX + j)] = 1.0;
    }
    quad::q_assemble_obs<0>(sh, mu, dw, dc);
    AsmOut A; quad::q_assemble_stage<0>(sh, mu, dw, dc, A);
    int ok = A.ok;
    StepOut S; S.ap = S.az = S.gd = 0; S.ok = 1;
    int fail = ok ? 0 : 1;
    if (ok) { ok = quad::q_riccati_backward(sh, rho); if (!ok) fail = 2; }
    if (ok) { quad::q_direction_main(sh, A, mu, dw, dc, rho, tau, S); ok = S.ok; if (!ok) fail = 3; }
    if (ok) quad::q_direction_obs<0>(sh, mu, dw, dc, tau, S);
    aux[10] = fail;
    memcpy(dout, s.d, sizeof(double) * (l.n + l.m));
    aux[0] = A.dinf; aux[1] = A.pinf; aux[2] = A.cinf0; aux[3] = cinf_mu(A, mu); aux[4] = A.f; aux[5] = A.th1; aux[6] = A.bar;
    aux[7] = S.ap; aux[8] = S.az; aux[9] = S.gd;
    q_free(s);
    return ok;
}

// full solve from the problem record (header + xWS); zout in the device layout
int emu_quad_solve(int N, const double *prob, const void *opts, double *zout, double *info) {
    QScratch s; quad::QLay l; q_alloc(N, s, l);
    quad::QShared &sh = quad::gq_sh;
    if (const char *e = getenv("OBCA_EMU_POISON")) {      // as emu_poison: 1 the work buffers in "HBM", 2 the static LDS block of the quadcopter solver
        const int m = atoi(e); const char *pv = getenv("OBCA_EMU_POISON_VALUE"); const double v = pv ? atof(pv) : NAN;
        if (m & 1) { for (size_t i = 0; i < (size_t)QDIR_DOUBLES(l); i++) s.d[i] = v; for (size_t i = 0; i < (size_t)(N + 1) * QSP; i++) s.as[i] = v;
                     for (size_t i = 0; i < (size_t)(N + 1) * QRR; i++) s.rs[i] = v; for (size_t i = 0; i < (size_t)(N + 1) * QOB * OB_OC; i++) s.oc[i] = v; for (int i = 0; i < l.len; i++) s.z[i] = v; }
        if (m & 2) { double *w = (double *)&sh; for (size_t i = 0; i < sizeof(quad::QShared) / sizeof(double); i++) w[i] = v; }
    }
    sh.inst.prob = (const gdbl *)prob; sh.inst.z = (gdbl *)s.z; sh.inst.d = (gdbl *)s.d; sh.inst.as = (gdbl *)s.as; sh.inst.rs = (gdbl *)s.rs; sh.inst.oc = (gdbl *)s.oc;
    emu_race_begin();
    emu_race_buffer("quad z", s.z, l.len, 0, -1); emu_race_buffer("quad d", s.d, QDIR_DOUBLES(l), 0, -1); emu_race_buffer("quad as", s.as, (long)(N + 1) * QSP, QSP, -1);
    emu_race_buffer("quad rs", s.rs, (long)(N + 1) * QRR, QRR, QRR_PAD); emu_race_buffer("quad oc", s.oc, (long)(N + 1) * QOB * OB_OC, OB_OC, -1);
    emu_race_dummy_load(s.z + l.n);      // q_apply_step: the lanes beyond the end of the multiplier block load its first word (clamped index) and discard it
    struct RaceOff { ~RaceOff() { emu_race_end(); } } race_off_;
    quad::q_solve_instance(N, *(const Opts *)opts, info, ((const OptsAbi *)opts)->max_soc, ((const OptsAbi *)opts)->lsq_init, ((const OptsAbi *)opts)->obj_scaling);
    memcpy(zout, s.z, sizeof(double) * l.len);
    q_free(s);
    return 0;
}
}

#ifdef OBCA_EMU_RACE
// ---------------------------------------------------------------- cross-lane hazards through HBM (race build)
// The lanes of an instance exchange data through HBM (assembled stage records, Riccati records, the direction) as well as through LDS.  LDS operations of one wavefront execute
// in order; global loads and stores do not: a load may be served before an earlier store of ANOTHER lane to the same word has landed unless the wavefront has waited for its
// stores in between (SYNC = __syncthreads, VM_DRAIN = s_waitcnt vmcnt(0)).  LDS_SYNC orders LDS traffic only.  This runtime keeps, per word, who wrote and who read it since the
// last drain and reports every word that two different lanes touched in between with at least one store.  Stores of lanes WITHOUT an item to a record's dummy slot (pad_slot,
// never read) are counted apart.  The emulation runs the lanes of a PAR region one after the other, so both orders of a hazard show up as the same report.
#include <unordered_map>
#include <map>
#include <vector>
#include <string>
#include <algorithm>
#include <dlfcn.h>
namespace race {
int lane = 64;      // 64: outside a PAR region (wave-uniform code: every lane issues the access)
static unsigned long long ep = 1, sub = 1;
struct Rec { unsigned long long ep = 0, wsub = 0, rmask = 0; int wl = -1; };
static std::unordered_map<const void *, Rec> mem;
struct Buf { std::string name; const char *base; size_t bytes; int record, pad; };
static std::vector<Buf> bufs;
struct Hit { long count = 0; std::string first; };
static std::map<std::string, Hit> hits;
static long dummy_stores = 0, accesses = 0;
static std::vector<const void *> dummy_loads;
static bool on = false;
void sync(int drains) { if (drains) ep++; sub++; }
static const Buf *find(const void *p) { for (auto &b : bufs) if ((const char *)p >= b.base && (const char *)p < b.base + b.bytes) return &b; return nullptr; }
static void report(const char *kind, const void *p, int other, unsigned long long dsub, void *site) {
    const Buf *b = find(p); const size_t off = b ? ((const char *)p - b->base) / 8 : 0;
    if (b && b->pad >= 0 && (int)(off % b->record) == b->pad && kind[0] == 'W' && kind[1] == 'W') { dummy_stores++; return; }
    Dl_info di; unsigned long long rel = 0; if (dladdr(site, &di) && di.dli_fbase) rel = (unsigned long long)((char *)site - (char *)di.dli_fbase);
    char key[320]; snprintf(key, sizeof key, "%s | buffer %s | code offset 0x%llx | LDS_SYNCs in between %llu%s", kind, b ? b->name.c_str() : "?", rel, dsub > 3 ? 3ULL : dsub, dsub > 3 ? "+" : "");
    Hit &h = hits[key];
    if (!h.count) { char t[160]; snprintf(t, sizeof t, "word %zu%s, lane %d then lane %d", off, b && b->record ? (" (record " + std::to_string(off / b->record) + ", entry " + std::to_string(off % b->record) + ")").c_str() : "", other, lane); h.first = t; }
    h.count++;
}
void rd(const void *p) {
    if (!on) return;
    accesses++;
    Rec &r = mem[p]; if (r.ep != ep) { r.ep = ep; r.wl = -1; r.rmask = 0; }
    if (r.wl >= 0 && r.wl != lane && std::find(dummy_loads.begin(), dummy_loads.end(), p) == dummy_loads.end()) report("RW: load of a word another lane stored since the last drain", p, r.wl, sub - r.wsub, __builtin_return_address(1));
    if (lane < 64) r.rmask |= 1ULL << lane; else r.rmask = ~0ULL;
}
void wr(const void *p) {
    if (!on) return;
    accesses++;
    Rec &r = mem[p]; if (r.ep != ep) { r.ep = ep; r.wl = -1; r.rmask = 0; }
    if (r.wl >= 0 && r.wl != lane) report("WW: store over a word another lane stored since the last drain", p, r.wl, sub - r.wsub, __builtin_return_address(1));
    const unsigned long long others = lane < 64 ? r.rmask & ~(1ULL << lane) : r.rmask;
    if (others) report("WR: store over a word another lane loaded since the last drain", p, __builtin_ctzll(others), 0, __builtin_return_address(1));
    r.wl = lane; r.wsub = sub;
}
}
extern "C" {
void emu_race_begin() { race::mem.clear(); race::bufs.clear(); race::dummy_loads.clear(); race::on = true; }
void emu_race_dummy_load(const void *word) { race::dummy_loads.push_back(word); }
void emu_race_buffer(const char *name, const void *base, long doubles, int record, int pad_slot) { race::bufs.push_back({name, (const char *)base, (size_t)doubles * 8, record, pad_slot}); }
void emu_race_end() { race::on = false; }
// prints one line per distinct hazard (kind, buffer, code offset in this library: addr2line -e libobca_emu_race.so -f -C <offset>); returns their number
int emu_race_report(long *n_accesses, long *n_dummy) {
    for (auto &kv : race::hits) printf("HAZARD %s | x%ld | first: %s\n", kv.first.c_str(), kv.second.count, kv.second.first.c_str());
    fflush(stdout);
    if (n_accesses) *n_accesses = race::accesses; if (n_dummy) *n_dummy = race::dummy_stores;
    return (int)race::hits.size();
}
}
#endif
