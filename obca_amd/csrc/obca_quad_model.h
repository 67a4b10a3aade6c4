// obca_quad_model.h -- per-lane model pieces of the quadcopter signed-distance NLP (gfx950 device code).
//
// Reference: /root/reference/QuadcopterNavigation/QuadcopterSignedDist.jl
//   constants :51-62, Euler dynamics :136-156, box-obstacle rows :162-197 (A = [I;-I], rows |A'lam|^2 == 1 and
//   -b'lam + p'A'lam + 0.01 slack >= R).
// Everything here runs on ONE lane for ONE work item (a stage or a (stage, box) block); loops have compile-time bounds.
#pragma once
#include <math.h>
#include "obca_model.h"   // small dense helpers (ldl_fact / ldl_solve / hh_apply)

namespace obca {
namespace quad {

#define QX 12          // states
#define QU 4           // inputs (rotor speeds)
#define QS (QX + QU)   // Riccati state: x and the copy w = u_{k-1}
#define QZ (QS + QU)   // stage vector (x, w, u)
#define QC (2 + QX)    // right-hand sides: main, t, nu_1..12
#define QOB 5          // boxes
#define QL 6           // multipliers per box
#define QV 10          // local derivative variables: angles x[3..5], rates x[9..11], u[0..3]

#define Q_MASS 0.5
#define Q_GRAV 9.81
#define Q_KF 0.0611
#define Q_KM 0.0015
#define Q_ARM 0.225
#define Q_I1 3.9e-3
#define Q_I2 4.4e-3
#define Q_I3 4.9e-3
#define Q_ULO 1.2
#define Q_UHI 7.8
#define Q_TLO 0.5
#define Q_THI 2.0

struct QConsts {
    double Ts, R, wH, x0[QX], xF[QX], gyro[3];
    double sf;      // objective scaling factor (IPOPT's gradient-based scaling, opts.obj_scaling; 1: none): the algorithm runs on sf * f
    int N, dist;    // dist = 1: QuadcopterDist.jl (no slack variable, x[10] in [-1.5, 3]); 0: QuadcopterSignedDist.jl
};
// :78-94, QuadcopterDist.jl:88
OBCA_FN double q_xlb(int i, int dist = 0) { return i < 3 ? 0.0 : (i == 3 ? -3.0 : (i < 6 ? -0.2 : (dist && i == 9 ? -1.5 : -1.0))); }
OBCA_FN double q_xub(int i, int dist = 0) { return i < 2 ? 10.0 : (i == 2 ? 5.0 : (i == 3 ? 3.0 : (i < 6 ? 0.2 : (dist && i == 9 ? 3.0 : 1.0)))); }
// stage-vector index of the local derivative variables
OBCA_FN int q_vidx(int a) { return a < 3 ? 3 + a : (a < 6 ? 6 + a : QS + (a - 6)); }

// g(x,u) with x+ = x + t Ts g ; rows 0..2 are x7..x9 (linear, handled by the caller)
OBCA_FN void dyn_g_value(const QConsts &c, const double *x, const double *u, double g[QX]) {
    double s4, c4, s5, c5, s6, c6;
    sincos_bounded(x[3], &s4, &c4); sincos_bounded(x[4], &s5, &c5); sincos_bounded(x[5], &s6, &c6);
    const double S4 = rcp_nr(c4), T4 = s4 * S4, r10 = x[9], r11 = x[10], r12 = x[11];
    const double U = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3], kap = Q_KF / Q_MASS, h = s5 * r10 - c5 * r12;
    g[0] = x[6]; g[1] = x[7]; g[2] = x[8];
    g[3] = c5 * r10 + s5 * r12; g[4] = T4 * h + r11; g[5] = -S4 * h;
    g[6] = kap * U * (s4 * c5 * s6 + s5 * c6); g[7] = kap * U * (-s4 * c5 * c6 + s5 * s6); g[8] = kap * U * c4 * c5 - Q_GRAV;
    g[9] = (Q_ARM * Q_KF * (u[1] * u[1] - u[3] * u[3]) - (Q_I3 - Q_I2) * c.gyro[1] * c.gyro[2]) * (1.0 / Q_I1);
    g[10] = (Q_ARM * Q_KF * (u[2] * u[2] - u[0] * u[0]) - (Q_I1 - Q_I3) * c.gyro[0] * c.gyro[2]) * (1.0 / Q_I2);
    g[11] = (Q_KM * (u[0] * u[0] - u[1] * u[1] + u[2] * u[2] - u[3] * u[3]) - (Q_I2 - Q_I1) * c.gyro[0] * c.gyro[1]) * (1.0 / Q_I3);
}

// value, Jacobian rows 3..11 w.r.t. the QV local variables (dg[i-3][a]) and HG = sum_i w_i Hess g_i (upper triangle, packed 55)
OBCA_FN int q_pidx(int a, int b) { int i = a < b ? a : b, j = a < b ? b : a; return i * QV - i * (i - 1) / 2 + (j - i); }
OBCA_FN void dyn_g_derivs(const QConsts &c, const double *x, const double *u, const double *w, double g[QX], double dg[9][QV], double HG[55]) {
    double s4, c4, s5, c5, s6, c6;
    sincos_bounded(x[3], &s4, &c4); sincos_bounded(x[4], &s5, &c5); sincos_bounded(x[5], &s6, &c6);
    const double S4 = rcp_nr(c4), T4 = s4 * S4, r10 = x[9], r11 = x[10], r12 = x[11];
    const double U = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3], kap = Q_KF / Q_MASS;
    const double g4 = c5 * r10 + s5 * r12, h = s5 * r10 - c5 * r12;
    const double E7 = s4 * c5 * s6 + s5 * c6, E8 = -s4 * c5 * c6 + s5 * s6, E9 = c4 * c5;
    g[0] = x[6]; g[1] = x[7]; g[2] = x[8];
    g[3] = g4; g[4] = T4 * h + r11; g[5] = -S4 * h;
    g[6] = kap * U * E7; g[7] = kap * U * E8; g[8] = kap * U * E9 - Q_GRAV;
    g[9] = (Q_ARM * Q_KF * (u[1] * u[1] - u[3] * u[3]) - (Q_I3 - Q_I2) * c.gyro[1] * c.gyro[2]) * (1.0 / Q_I1);
    g[10] = (Q_ARM * Q_KF * (u[2] * u[2] - u[0] * u[0]) - (Q_I1 - Q_I3) * c.gyro[0] * c.gyro[2]) * (1.0 / Q_I2);
    g[11] = (Q_KM * (u[0] * u[0] - u[1] * u[1] + u[2] * u[2] - u[3] * u[3]) - (Q_I2 - Q_I1) * c.gyro[0] * c.gyro[1]) * (1.0 / Q_I3);
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int a = 0; a < QV; a++) dg[i][a] = 0;
    const double T4p = S4 * S4, S4p = S4 * T4;
    dg[0][1] = -h; dg[0][3] = c5; dg[0][5] = s5;                                                            // g4
    dg[1][0] = T4p * h; dg[1][1] = T4 * g4; dg[1][3] = T4 * s5; dg[1][4] = 1; dg[1][5] = -T4 * c5;          // g5 = T4 h + r11
    dg[2][0] = -S4p * h; dg[2][1] = -S4 * g4; dg[2][3] = -S4 * s5; dg[2][5] = S4 * c5;                      // g6 = -S4 h
    const double E7d[3] = {c4 * c5 * s6, -s4 * s5 * s6 + c5 * c6, s4 * c5 * c6 - s5 * s6};
    const double E8d[3] = {-c4 * c5 * c6, s4 * s5 * c6 + c5 * s6, s4 * c5 * s6 + s5 * c6};
    const double E9d[3] = {-s4 * c5, -c4 * s5, 0};
#pragma unroll
    for (int a = 0; a < 3; a++) { dg[3][a] = kap * U * E7d[a]; dg[4][a] = kap * U * E8d[a]; dg[5][a] = kap * U * E9d[a]; }
#pragma unroll
    for (int j = 0; j < 4; j++) { dg[3][6 + j] = 2 * kap * u[j] * E7; dg[4][6 + j] = 2 * kap * u[j] * E8; dg[5][6 + j] = 2 * kap * u[j] * E9; }
    dg[6][6 + 1] = 2 * Q_ARM * Q_KF * u[1] * (1.0 / Q_I1); dg[6][6 + 3] = -2 * Q_ARM * Q_KF * u[3] * (1.0 / Q_I1);
    dg[7][6 + 2] = 2 * Q_ARM * Q_KF * u[2] * (1.0 / Q_I2); dg[7][6 + 0] = -2 * Q_ARM * Q_KF * u[0] * (1.0 / Q_I2);
    dg[8][6 + 0] = 2 * Q_KM * u[0] * (1.0 / Q_I3); dg[8][6 + 1] = -2 * Q_KM * u[1] * (1.0 / Q_I3); dg[8][6 + 2] = 2 * Q_KM * u[2] * (1.0 / Q_I3);
    dg[8][6 + 3] = -2 * Q_KM * u[3] * (1.0 / Q_I3);
#pragma unroll
    for (int i = 0; i < 55; i++) HG[i] = 0;
#define QSYM(i, j, v) HG[q_pidx((i), (j))] += (v)
    QSYM(1, 1, w[3] * (-g4)); QSYM(1, 3, w[3] * (-s5)); QSYM(1, 5, w[3] * c5);
    QSYM(0, 0, w[4] * 2 * T4 * T4p * h); QSYM(0, 1, w[4] * T4p * g4); QSYM(0, 3, w[4] * T4p * s5); QSYM(0, 5, w[4] * (-T4p * c5));
    QSYM(1, 1, w[4] * (-T4 * h)); QSYM(1, 3, w[4] * T4 * c5); QSYM(1, 5, w[4] * T4 * s5);
    QSYM(0, 0, w[5] * (-S4 * (T4 * T4 + S4 * S4) * h)); QSYM(0, 1, w[5] * (-S4p * g4)); QSYM(0, 3, w[5] * (-S4p * s5)); QSYM(0, 5, w[5] * S4p * c5);
    QSYM(1, 1, w[5] * S4 * h); QSYM(1, 3, w[5] * (-S4 * c5)); QSYM(1, 5, w[5] * (-S4 * s5));
    const double E7h[6] = {-s4 * c5 * s6, -c4 * s5 * s6, c4 * c5 * c6, -E7, -s4 * s5 * c6 - c5 * s6, -E7};   // (0,0)(0,1)(0,2)(1,1)(1,2)(2,2)
    const double E8h[6] = {s4 * c5 * c6, c4 * s5 * c6, c4 * c5 * s6, -E8, -s4 * s5 * s6 + c5 * c6, -E8};
    const double E9h[6] = {-c4 * c5, s4 * s5, 0, -c4 * c5, 0, 0};
    { int q = 0;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = a; b < 3; b++) { QSYM(a, b, kap * U * (w[6] * E7h[q] + w[7] * E8h[q] + w[8] * E9h[q])); q++; } }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        QSYM(6 + j, 6 + j, 2 * kap * (w[6] * E7 + w[7] * E8 + w[8] * E9));
#pragma unroll
        for (int a = 0; a < 3; a++) QSYM(a, 6 + j, 2 * kap * u[j] * (w[6] * E7d[a] + w[7] * E8d[a] + w[8] * E9d[a]));
    }
    QSYM(6 + 1, 6 + 1, w[9] * 2 * Q_ARM * Q_KF * (1.0 / Q_I1)); QSYM(6 + 3, 6 + 3, -w[9] * 2 * Q_ARM * Q_KF * (1.0 / Q_I1));
    QSYM(6 + 2, 6 + 2, w[10] * 2 * Q_ARM * Q_KF * (1.0 / Q_I2)); QSYM(6 + 0, 6 + 0, -w[10] * 2 * Q_ARM * Q_KF * (1.0 / Q_I2));
    QSYM(6 + 0, 6 + 0, w[11] * 2 * Q_KM * (1.0 / Q_I3)); QSYM(6 + 1, 6 + 1, -w[11] * 2 * Q_KM * (1.0 / Q_I3));
    QSYM(6 + 2, 6 + 2, w[11] * 2 * Q_KM * (1.0 / Q_I3)); QSYM(6 + 3, 6 + 3, -w[11] * 2 * Q_KM * (1.0 / Q_I3));
#undef QSYM
}

// ---------------------------------------------------------------- one (stage, box) block
struct QObsIn { double b[QL], lam[QL], zl[QL], s, zs, so, zso, y[2], p[3]; };

OBCA_FN void q_obs_rows(const QConsts &c, const QObsIn &in, double r[2], double q[3]) {
    double bl = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = in.lam[i] - in.lam[3 + i];
#pragma unroll
    for (int i = 0; i < QL; i++) bl += in.b[i] * in.lam[i];
    r[0] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] - 1;
    r[1] = -bl + in.p[0] * q[0] + in.p[1] * q[1] + in.p[2] * q[2] + (c.dist ? 0.0 : 0.01 * in.s) - c.R - in.so;
}

struct QObsStats { double dmax, pmax, cmax0, cmin, cmax, sumz, sumy; int bad; };   // cmin / cmax: extreme complementarity products (obca_model.h: ObsStats)
struct QObsStep { double dlam[QL], ds, dso, dy[2]; };

// MODE 0: condense onto the position (cond: Hpp[6] sym 3x3, gz[3] = q*y2, gcorr[3]); MODE 1: back-substitute for the step dp;
// MODE 2: inertia of the block only (st->bad).  crs: the two rows of a second-order correction in place of the constraint values (IPOPT A-5.7), or nullptr.
// LSQ: the block of the least-squares multiplier system (IPOPT's initial multipliers):
// unit Hessian on every variable, gradients with the bound multipliers themselves,
// zero constraint right-hand sides, multipliers taken as zero (call with dw = dc = 0)
template <int MODE, int LSQ = 0>
OBCA_FN void q_obs_block(const QConsts &c, const QObsIn &in, double mu_b, double dw, double dc, ObsCond *cond, QObsStats *st,
                         const double dp[3], QObsStep *step, const double *crs = nullptr) {
    double cr[2], q[3];
    q_obs_rows(c, in, cr, q);
    // right-hand side of the two rows: the constraint values, or the rows of a second-order correction
    const double rhs0 = LSQ ? 0.0 : (crs ? crs[0] : cr[0]), rhs1 = LSQ ? 0.0 : (crs ? crs[1] : cr[1]);
    const double y0_[2] = {0.0, 0.0};
    const double *y = LSQ ? y0_ : in.y;
    double g1[QL], g2[QL], Dl[QL], rl[QL];
#pragma unroll
    for (int i = 0; i < QL; i++) {
        const double sg = i < 3 ? 1.0 : -1.0; const int a = i % 3;
        g1[i] = 2 * sg * q[a]; g2[i] = -in.b[i] + sg * in.p[a];
        const double il = rcp_nr(in.lam[i]), gl = c.sf * 2e-4 * in.lam[i] + g1[i] * y[0] + g2[i] * y[1];
        rl[i] = LSQ ? gl - in.zl[i] : gl - mu_b * il; Dl[i] = LSQ ? 1.0 : c.sf * 2e-4 + in.zl[i] * il + dw;
        if (MODE == 0) {
            double rz = fabs(gl - in.zl[i]); st->dmax = fmax(st->dmax, rz);
            double cc = in.lam[i] * in.zl[i]; st->cmax0 = fmax(st->cmax0, fabs(cc)); st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc);
            st->sumz += fabs(in.zl[i]);
        }
    }
    const double is = rcp_nr(in.s), iso = rcp_nr(in.so);
    const double gs = c.sf * (1e2 + 2e3 * in.s) + 0.01 * y[1], gso = -y[1];
    // QuadcopterDist has no slack variable: it is frozen (1/D_s = 0, no residual), every term below then drops out and ds = 0
    const double r_s = c.dist ? 0.0 : (LSQ ? gs - in.zs : gs - mu_b * is), r_so = LSQ ? gso - in.zso : gso - mu_b * iso;
    const double iDs = c.dist ? 0.0 : (LSQ ? 1.0 : rcp_nr(c.sf * 2e3 + in.zs * is + dw)), iDso = LSQ ? 1.0 : rcp_nr(in.zso * iso + dw);
    if (MODE == 0) {
        double rz = c.dist ? 0.0 : fabs(gs - in.zs); st->dmax = fmax(st->dmax, rz);
        rz = fabs(gso - in.zso); st->dmax = fmax(st->dmax, rz);
        double cc = c.dist ? 0.0 : in.s * in.zs; st->cmax0 = fmax(st->cmax0, fabs(cc));
        if (!c.dist) { st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc); }
        cc = in.so * in.zso; st->cmax0 = fmax(st->cmax0, fabs(cc)); st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc);
        st->sumz += (c.dist ? 0.0 : fabs(in.zs)) + fabs(in.zso);
        st->pmax = fmax(st->pmax, fabs(cr[0])); st->pmax = fmax(st->pmax, fabs(cr[1]));
        st->sumy += fabs(y[0]) + fabs(y[1]);
    }
    // row 2 after eliminating s and so:  g2'dlam + q'dp - T2 dy2 = r2
    const double iT2 = rcp_nr(1e-4 * iDs + iDso + dc);
    const double r2 = -rhs1 + 0.01 * r_s * iDs - r_so * iDso;
    // (lambda, y1) block: Hb = diag(Dl) + 2 y1 D'D + g2 g2'/T2 ; coupling Cp = y2 D' + g2 q'/T2 ; rk
    double Hb[QL * QL], Cp[QL][3], rk[QL + 1];
#pragma unroll
    for (int i = 0; i < QL; i++) {
        const int a = i % 3; const double sg = i < 3 ? 1.0 : -1.0;
#pragma unroll
        for (int m = 0; m < QL; m++) {
            const double sm = m < 3 ? 1.0 : -1.0;
            Hb[i * QL + m] = ((m % 3) == a ? 2 * y[0] * sg * sm : 0.0) + g2[i] * g2[m] * iT2;
        }
        Hb[i * QL + i] += Dl[i];
#pragma unroll
        for (int cI = 0; cI < 3; cI++) Cp[i][cI] = (cI == a ? y[1] * sg : 0.0) + g2[i] * q[cI] * iT2;
        rk[i] = -rl[i] + g2[i] * r2 * iT2;
    }
    rk[QL] = -rhs0;
    // Householder Qh g1 = alpha e1, 2x2 pivot on (lam~_0, y1), LDL of the 5x5 reduced Hessian (must be positive definite)
    double hw[QL], nq = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) nq += g1[i] * g1[i];
    nq = sqrt(nq);
    const double alpha = g1[0] > 0 ? -nq : nq;
    double nw = 0;
#pragma unroll
    for (int i = 0; i < QL; i++) { hw[i] = g1[i] - (i == 0 ? alpha : 0.0); nw += hw[i] * hw[i]; }
    const double hb = nw > 0 ? 2.0 * rcp_nr(nw) : 0.0;       // unnormalised Householder vector, see hh_apply
#pragma unroll
    for (int j = 0; j < QL; j++) {
        double col[QL];
#pragma unroll
        for (int i = 0; i < QL; i++) col[i] = Hb[i * QL + j];
        hh_apply<QL>(QL, hw, hb, col);
#pragma unroll
        for (int i = 0; i < QL; i++) Hb[i * QL + j] = col[i];
    }
#pragma unroll
    for (int i = 0; i < QL; i++) hh_apply<QL>(QL, hw, hb, Hb + i * QL);
    const double a00 = Hb[0], det = a00 * (-dc) - alpha * alpha;
    int bad = !(det < 0);
    const double idet = rcp_nr(det), Mi0 = -dc * idet, Mi1 = -alpha * idet, Mi2 = a00 * idet;
    double hc[QL - 1], Hr[(QL - 1) * (QL - 1)];
#pragma unroll
    for (int i = 0; i < QL - 1; i++) hc[i] = Hb[(i + 1) * QL];
#pragma unroll
    for (int i = 0; i < QL - 1; i++)
#pragma unroll
        for (int j = 0; j < QL - 1; j++) Hr[i * (QL - 1) + j] = Hb[(i + 1) * QL + (j + 1)] - Mi0 * hc[i] * hc[j];
    bad |= ldl_fact<QL - 1>(QL - 1, Hr);
    if (MODE == 2) { st->bad |= bad; return; }
    auto ksolve = [&](double *col /* QL+1 */) {
        hh_apply<QL>(QL, hw, hb, col);
        double g0 = col[0], gy = col[QL];
        const double t0 = Mi0 * g0 + Mi1 * gy;
        double rr[QL - 1];
#pragma unroll
        for (int i = 0; i < QL - 1; i++) rr[i] = col[i + 1] - hc[i] * t0;
        ldl_solve<QL - 1>(QL - 1, Hr, rr);
        double hl = 0;
#pragma unroll
        for (int i = 0; i < QL - 1; i++) hl += hc[i] * rr[i];
        g0 -= hl;
        col[0] = Mi0 * g0 + Mi1 * gy; col[QL] = Mi1 * g0 + Mi2 * gy;
#pragma unroll
        for (int i = 0; i < QL - 1; i++) col[i + 1] = rr[i];
        hh_apply<QL>(QL, hw, hb, col);
    };
    if (MODE == 0) {
        st->bad |= bad;
        double Z[QL + 1][4];
#pragma unroll
        for (int cI = 0; cI < 4; cI++) {
            double col[QL + 1];
#pragma unroll
            for (int i = 0; i < QL; i++) col[i] = cI < 3 ? Cp[i][cI] : rk[i];
            col[QL] = cI < 3 ? 0.0 : rk[QL];
            ksolve(col);
#pragma unroll
            for (int i = 0; i <= QL; i++) Z[i][cI] = col[i];
        }
        int qn = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int b = 0; b < 3; b++) if (b >= a) {
                double s_ = q[a] * q[b] * iT2;
#pragma unroll
                for (int i = 0; i < QL; i++) s_ -= Cp[i][a] * Z[i][b];
                cond->Hpp[qn++] = s_;
            }
            double s_ = q[a] * r2 * iT2;
#pragma unroll
            for (int i = 0; i < QL; i++) s_ -= Cp[i][a] * Z[i][3];
            cond->gcorr[a] = s_;
            cond->gz[a] = q[a] * y[1];
        }
    } else {
        double col[QL + 1];
#pragma unroll
        for (int i = 0; i < QL; i++) col[i] = rk[i] - (Cp[i][0] * dp[0] + Cp[i][1] * dp[1] + Cp[i][2] * dp[2]);
        col[QL] = rk[QL];
        ksolve(col);
        double a_ = -r2;
#pragma unroll
        for (int i = 0; i < QL; i++) a_ += g2[i] * col[i];
#pragma unroll
        for (int i = 0; i < 3; i++) a_ += q[i] * dp[i];
        const double dy2 = a_ * iT2;
        step->dy[0] = col[QL]; step->dy[1] = dy2;
#pragma unroll
        for (int i = 0; i < QL; i++) step->dlam[i] = col[i];
        step->ds = (-r_s - 0.01 * dy2) * iDs;
        step->dso = (dy2 - r_so) * iDso;
    }
}

// closed-form dual warm start of one (position, box) pair (the quadcopter analogue of DualMultWS)
OBCA_FN void q_dual_ws(const double *b /* [hi; -lo] */, const double *p, double lam[QL]) {
    double d[3], n2 = 0, q[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) { const double hi = b[i], lo = -b[3 + i], cl = p[i] < lo ? lo : (p[i] > hi ? hi : p[i]); d[i] = p[i] - cl; n2 += d[i] * d[i]; }
    if (n2 > 1e-16) { const double in = 1 / sqrt(n2);
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = d[i] * in; }
    else {
        int best = 0; double bd = 1e300, sg = 1;
#pragma unroll
        for (int i = 0; i < 3; i++) { const double hi = b[i] - p[i], lo = p[i] + b[3 + i]; if (hi < bd) { bd = hi; best = i; sg = 1; } if (lo < bd) { bd = lo; best = i; sg = -1; } }
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = (i == best) ? sg : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { lam[i] = q[i] > 0 ? q[i] : 0; lam[3 + i] = q[i] < 0 ? -q[i] : 0; }
}

}  // namespace quad
}  // namespace obca
