// TEST INFRASTRUCTURE ONLY.  Host emulation of the wave-per-instance HIP solver: the device source
// obca_amd/csrc/obca_solver.h is compiled with -DOBCA_EMU, which turns every PAR(lane) region into a plain
// loop over 64 lanes.  It lets the CPU test-suite check the kernel logic (Newton direction, full solves) against
// the oracle on a machine without a GPU.  It is never linked into libobca_hip.so.
#define OBCA_EMU 1
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "../../obca_amd/csrc/obca_solver.h"
using namespace obca;

struct Scratch { double *z, *d, *as, *rs, *oc, *traj; };
static void alloc_scratch(int N, int len, Scratch &s) {
    s.z = (double *)calloc(len, 8); s.d = (double *)calloc(len, 8);
    s.as = (double *)calloc((size_t)(N + 1) * OB_AS, 8); s.rs = (double *)calloc((size_t)(N + 1) * OB_RS, 8);
    s.oc = (double *)calloc((size_t)(N + 1) * OB_NOBMAX * OB_OC, 8); s.traj = (double *)calloc((size_t)(N + 2) * 6, 8);
}
static void free_scratch(Scratch &s) { free(s.z); free(s.d); free(s.as); free(s.rs); free(s.oc); free(s.traj); }

static void setup(int N, const double *prob, Scratch &s) {
    Shared &sh = g_sh; Inst &I = sh.inst;
    I.prob = prob; I.z = s.z; I.d = s.d; I.as = s.as; I.rs = s.rs; I.oc = s.oc; I.traj = s.traj;
    for (int i = 0; i < OB_HDR; i++) sh.hdr[i] = prob[i];
    for (int i = 0; i <= OB_NOBMAX; i++) sh.roff[i] = (int)sh.hdr[PH_ROFF + i];
    for (int i = 0; i < OB_NOBMAX; i++) sh.vOb[i] = (int)sh.hdr[PH_VOB + i];
    Consts &c = sh.c; c.N = N;
    c.Ts = sh.hdr[PH_TS]; c.L = sh.hdr[PH_L]; c.off = sh.hdr[PH_OFF];
    for (int i = 0; i < 4; i++) { c.g[i] = sh.hdr[PH_G + i]; c.xl[i] = sh.hdr[PH_XL + i]; c.xu[i] = sh.hdr[PH_XU + i]; c.x0[i] = sh.hdr[PH_X0 + i]; c.xF[i] = sh.hdr[PH_XF + i]; }
    c.fixTime = (int)sh.hdr[PH_FIX]; c.nOb = (int)sh.hdr[PH_NOB]; c.M = (int)sh.hdr[PH_M];
    c.wa = c.fixTime ? 0.5 : 0.1; c.wpsi = c.fixTime ? 1e-2 : 1e-4;
    make_layout(c.N, c.nOb, c.M, sh.l);
    int vmx = 0; for (int j = 0; j < c.nOb; j++) if (sh.vOb[j] > vmx) vmx = sh.vOb[j];
    sh.vm2 = vmx <= 2;
}

extern "C" {
int emu_opts_size() { return (int)sizeof(Opts); }

// one Newton direction at a full primal-dual point (oracle layout); returns inertia-ok
int emu_newton(int N, const double *prob, const double *zin, int len, double mu, double dw, double dc, double rho, double tau,
               double *dout, double *aux /* dinf,pinf,cinf0,cinfmu,f,th1,bar,ap,az,gd */) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zin, sizeof(double) * len);
    setup(N, prob, s); Shared &sh = g_sh; Inst &I = sh.inst;
    AsmOut A;
    if (sh.vm2) assemble_obs<2>(I, sh, mu, dw, dc); else assemble_obs<OB_VMAX>(I, sh, mu, dw, dc);
    assemble_stage(I, sh, mu, dw, dc, A);
    int ok = A.ok;
    StepOut S; S.ap = S.az = S.gd = 0;
    if (ok) ok = riccati_backward(I, sh, rho);
    if (ok) { direction_main(I, sh, A, mu, dw, dc, rho, tau, S); ok = S.ok; }
    if (ok) { if (sh.vm2) direction_obs<2>(I, sh, mu, dw, dc, tau, S); else direction_obs<OB_VMAX>(I, sh, mu, dw, dc, tau, S); }
    memcpy(dout, s.d, sizeof(double) * len);
    aux[0] = A.dinf; aux[1] = A.pinf; aux[2] = A.cinf0; aux[3] = A.cinfmu; aux[4] = A.f; aux[5] = A.th1; aux[6] = A.bar;
    aux[7] = S.ap; aux[8] = S.az; aux[9] = S.gd;
    free_scratch(s);
    return ok;
}

int emu_eval_trial(int N, const double *prob, const double *zin, const double *din, int len, double alpha, double *out3) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zin, sizeof(double) * len); memcpy(s.d, din, sizeof(double) * len);
    setup(N, prob, s); Shared &sh = g_sh; Inst &I = sh.inst;
    if (sh.vm2) eval_trial<2>(I, sh, alpha, out3[0], out3[1], out3[2]); else eval_trial<OB_VMAX>(I, sh, alpha, out3[0], out3[1], out3[2]);
    free_scratch(s);
    return 0;
}

// full solve; zinit holds the primal warm start in the oracle layout (x,u,t,lam,mu; sl=0)
int emu_solve(int N, const double *prob, const double *zinit, int len, const void *opts, double *zout, double *info) {
    Scratch s; alloc_scratch(N, len, s);
    memcpy(s.z, zinit, sizeof(double) * len);
    Inst &I = g_sh.inst; I.prob = prob; I.z = s.z; I.d = s.d; I.as = s.as; I.rs = s.rs; I.oc = s.oc; I.traj = s.traj;
    solve_instance(N, *(const Opts *)opts, info);
    memcpy(zout, s.z, sizeof(double) * len);
    free_scratch(s);
    return 0;
}

int emu_dualws(int v, const double *a1, const double *a2, const double *b, const double *g, double ex, double ey, double cs, double sn,
               double *lam, double *mu, double *d) {
    double l4[OB_VMAX], m4[4];
    dualws_one(v, a1, a2, b, g, ex, ey, cs, sn, l4, m4, d);
    for (int i = 0; i < v; i++) lam[i] = l4[i];
    for (int i = 0; i < 4; i++) mu[i] = m4[i];
    return 0;
}
}
