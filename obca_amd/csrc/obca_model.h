// obca_model.h -- per-lane model pieces of the OBCA parking signed-distance NLP (gfx950 device code).
//
// Everything in this header is executed by ONE lane on ONE work item (a stage, or a (stage, obstacle) block):
// closed-form values, Jacobians and Lagrangian Hessians of
//   * the kinematic-bicycle multiple-shooting map            (reference: AutonomousParking/ParkingSignedDist.jl:139-155)
//   * the dual-variable hyperplane-separation rows           (reference: ParkingSignedDist.jl:182-208)
// and the condensation of one (stage, obstacle) block (lambda_j, mu_j, sl_j, slack, 4 multipliers) onto the pose
// (X, Y, psi) of its stage.  No dynamic indexing: all loops run to the compile-time bounds OB_VMAX / 4 / 3 so that the
// small matrices stay in VGPRs.
#pragma once
#include <math.h>

#ifndef OBCA_FN
#define OBCA_FN static inline
#endif

#define OB_VMAX 8      // max half-space rows per obstacle (polygons with up to 8 edges; obstHrep.jl:31-102 emits one row per edge)
#define OB_VMID 4      // the code of a (stage, obstacle) block is instantiated for <= 2, <= OB_VMID and <= OB_VMAX rows
#define OB_NOBMAX 16     // obstacles per instance: the header carries 2 NOBMAX + 1 small integers (row counts, row offsets)
#define OB_MMAX 64       // half-space rows per instance (all obstacles together): 3 doubles of the header each (a1, a2, b)

namespace obca {

// Reciprocal for the latency-critical pivots: v_rcp_f64 (good to 2^-24.4) refined as r (1 + e + e^2) with e = 1 - d r: FOUR dependent instructions (two Newton
// steps are five, the IEEE division sequence with its scaling and fix-up eleven), and every link of such a chain is exposed on a wavefront that is alone on its SIMD
// (a dependent v_fma_f64 8 clocks, v_rcp_f64 ~20: tools/micro/fp64_dependent_latency.hip, round 6; round 3 had read 47 per link off tools/micro/lds_barrier_latency.hip).  Measured on MI355X over 80 binades: 1.00 ulp, the same as two Newton steps
// (tools/micro/rcp_accuracy.hip, profiles/r03_rcp_accuracy.txt).  For the normal-range, strictly positive pivots it is used on; a zero or NaN pivot gives NaN,
// and those are rejected by the positivity tests next to every use.  The host emulation divides.
// RS = 0: the two-Newton-step form (five dependent operations, the same 1.00 ulp).  The
// (stage, obstacle) code instantiated for more than two rows per obstacle keeps it:
// there the register allocation of the short form costs more than its shorter chain
// gains (same-box A/B on BASELINE config 5: 3.5 % fewer solves/s with the short form in
// those instantiations, while config 2 -- two rows -- gains 3.7 % from it; profiles/r03_ab_reciprocal_and_early_quu.txt).
template <int RS = 1>
OBCA_FN double rcp_nr(double d) {
#ifdef OBCA_EMU
    return 1.0 / d;
#else
    if (RS) {
        const double r = __builtin_amdgcn_rcp(d), e = fma(-d, r, 1.0);      // e = 1 - d r: 2^-24 at most
        return fma(r, fma(e, e, e), r);                                     // r (1 + e + e^2) = (1/d)(1 - e^3)
    }
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
#endif
}

template <int RS = 1>
OBCA_FN double rdiv(double a, double b) { return a * rcp_nr<RS>(b); }   // a / b to an ulp or two, 6 instructions instead of 12

struct Consts {               // uniform per instance
    double Ts, L, iL, g[4], off, xl[4], xu[4], x0[4], xF[4];   // iL = 1 / L (the wheelbase divides a dozen terms per stage)
    int fixTime, nOb, M, N;
    int dist;                 // 1: ParkingDist.jl formulation (next-1 sibling), 0: ParkingSignedDist.jl
    double wa, wpsi;          // ParkingSignedDist.jl:78-92 (fixTime switches the weights)
};

#define OB_UL0 (-0.6)
#define OB_UU0 (0.6)
#define OB_UL1 (-0.4)
#define OB_UU1 (0.4)
#define OB_TL 0.8
#define OB_TU 1.2
#define OB_SSB 0.6
#define OB_DMIN 0.05

// ---------------------------------------------------------------- sin / cos of a bounded angle
// The library's sincos (ocml: Payne-Hanek reduction for arbitrary arguments, table-free
// kernels, ~235 instructions per call on gfx950) is a fifth of the instructions of an
// obstacle item (tools/isa_lines.py: two calls per item in the fused line search, one in
// the back-substitution) and this kernel is within 1.6 x of its instruction-issue roof in
// those phases.  Headings, steering and Euler angles of these
// problems are a few radians: Cody-Waite reduction by pi/2 in three FMA steps (exact to
// 1e-33 |n|) and the fdlibm kernels on [-pi/4, pi/4] -- ~45 instructions, straight-line,
// <= 1.6 ulp for |x| <= 1e5 (the library: <= 1); the quadrant is an int: valid for
// |x| < 3e9, not-a-number and infinity come back as not-a-number.  (No branch to the
// library for larger arguments: the compiler would inline that path into every caller
// and the phases would pay its registers.)
OBCA_FN void sincos_bounded(double x, double *sp, double *cp) {
    const double n = rint(x * 6.36619772367581382433e-01);                 // 2 / pi
    double r = fma(-n, 1.57079632679489655800e+00, x);                     // pi / 2 = P1 + P2 + P3, each the rounded remainder of the previous
    r = fma(-n, 6.12323399573676603587e-17, r);
    r = fma(-n, -1.49738490485916983294e-33, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sr = fma(r * z, ps, r);                                   // sin r = r + r^3 S(z)
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));                   // cos r = 1 - z / 2 + z^2 C(z)
    const int q = (int)n;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *sp = (q & 2) ? -s0 : s0;
    *cp = ((q + 1) & 2) ? -c0 : c0;
}
OBCA_FN double tan_bounded(double x) { double s_, c_; sincos_bounded(x, &s_, &c_); return s_ / c_; }      // (steering angles: |x| <= 0.6, cos >= 0.8)

// ---------------------------------------------------------------- bicycle model, vars (psi, v, delta, a, t)
OBCA_FN void dyn_value(const Consts &c, const double x[4], const double u[2], double t, double F[4]) {
    double tau = c.Ts * t, s = x[3] + 0.5 * tau * u[1], T = tan_bounded(u[0]);
    double phi = x[2] + tau * x[3] * T * (0.5 * c.iL), sn, cs;
    sincos_bounded(phi, &sn, &cs);
    F[0] = x[0] + tau * s * cs; F[1] = x[1] + tau * s * sn; F[2] = x[2] + tau * s * T * c.iL; F[3] = x[3] + tau * u[1];
}

// First derivatives and HL = sum_i w_i Hess(F_i) of the bicycle map in the variables (psi, v, delta, a, t), with the structure written out.
// F is a composition through m = (tau, s, phi) [rows X, Y] and (tau, s, T) [row psi] with
//   dtau = Ts e_t,   ds = e_v + d3 e_a + d4 e_t,   dphi = e_psi + p1 e_v + p2 e_delta + p4 e_t,   dT = Tp e_delta
// so every Jacobian / Hessian entry is a short closed form; the generic triple products over these mostly-zero vectors (rounds 1-3) cost four times the
// instructions -- a product with a literal zero cannot be folded by the compiler (0 x inf) -- and kept ~60 more values alive in the stage assembly.
// Only what the assembly uses is produced: the 16 Jacobian entries that can be non-zero (as_df), and the 14 entries of the symmetric HL that can be non-zero.
struct Dyn {
    double F[4];
    double dX[5], dY[5];      // d(F_X - X), d(F_Y - Y) / d(psi, v, delta, a, t)
    double dP[4];             // d(F_psi - psi) / d(v, delta, a, t)        (no dependence on psi beyond the identity)
    double dVa, dVt;          // d(F_v - v) / d(a, t)
    double h00, h01, h02, h03, h04, h11, h12, h13, h14, h22, h23, h24, h34, h44;      // HL(i, j), i <= j; HL(3, 3) = 0
};
// Ts, iL = 1 / L: the caller's (uniform) copies of Consts::Ts, Consts::iL
OBCA_FN void dyn_derivs(const double Ts, const double iL, const double x[4], const double u[2], double t, const double w[4], Dyn &o) {
    const double i2L = 0.5 * iL, v = x[3], a = u[1];
    const double tau = Ts * t, s = v + 0.5 * tau * a, T = tan_bounded(u[0]), Tp = 1 + T * T;
    const double phi = x[2] + tau * v * T * i2L;
    double sn, cs; sincos_bounded(phi, &sn, &cs);
    o.F[0] = x[0] + tau * s * cs; o.F[1] = x[1] + tau * s * sn; o.F[2] = x[2] + tau * s * T * iL; o.F[3] = v + tau * a;
    const double d3 = 0.5 * tau, d4 = 0.5 * Ts * a;                                   // ds / d(a, t)
    const double p1 = tau * T * i2L, p2 = tau * v * Tp * i2L, p4 = Ts * v * T * i2L;      // dphi / d(v, delta, t)
    const double ts = tau * s;
    const double gX0 = s * cs, gX1 = tau * cs, gX2 = -ts * sn;      // dF_X / d(tau, s, phi)
    const double gY0 = s * sn, gY1 = tau * sn, gY2 = ts * cs;       // dF_Y / d(tau, s, phi)
    const double gP0 = s * T * iL, gP1 = tau * T * iL, gP2 = ts * iL;      // dF_psi / d(tau, s, T)
    o.dX[0] = gX2; o.dX[1] = fma(gX2, p1, gX1); o.dX[2] = gX2 * p2; o.dX[3] = gX1 * d3; o.dX[4] = fma(gX2, p4, fma(gX1, d4, gX0 * Ts));
    o.dY[0] = gY2; o.dY[1] = fma(gY2, p1, gY1); o.dY[2] = gY2 * p2; o.dY[3] = gY1 * d3; o.dY[4] = fma(gY2, p4, fma(gY1, d4, gY0 * Ts));
    o.dP[0] = gP1; o.dP[1] = gP2 * Tp; o.dP[2] = gP1 * d3; o.dP[3] = fma(gP1, d4, gP0 * Ts);
    o.dVa = tau; o.dVt = Ts * a;
    // second derivatives in m-space, weighted: rows X, Y share (tau, s, phi), row psi has (tau, s, T)
    const double wc = fma(w[1], sn, w[0] * cs), wsn = fma(w[1], cs, -(w[0] * sn));      // w0 cs + w1 sn;  -w0 sn + w1 cs
    const double A = wc, Bq = s * wsn, Cq = tau * wsn, Dq = -ts * wc;                   // (tau,s), (tau,phi), (s,phi), (phi,phi)
    const double E = w[2] * T * iL, Fq = w[2] * s * iL, Gq = w[2] * tau * iL;           // (tau,s), (tau,T), (s,T)
    const double gs = tau * fma(w[2], T * iL, wc);                                       // weight of Hess(s)
    const double gp = ts * wsn;                                                          // weight of Hess(phi)
    const double gT = w[2] * ts * iL;                                                    // weight of Hess(T)
    const double CD1 = fma(Dq, p1, Cq);                                                  // Cq + Dq p1
    o.h00 = Dq;
    o.h01 = CD1;
    o.h02 = Dq * p2;
    o.h03 = Cq * d3;
    o.h04 = fma(Dq, p4, fma(Cq, d4, Bq * Ts));
    o.h11 = p1 * (Cq + CD1);
    o.h12 = fma(gp, tau * Tp * i2L, fma(Gq, Tp, p2 * CD1));
    o.h13 = Cq * p1 * d3;
    o.h14 = fma(gp, Ts * T * i2L, fma(E + A, Ts, fma(Bq * Ts, p1, fma(Cq, fma(p1, d4, p4), Dq * p1 * p4))));
    o.h22 = fma(gT, 2 * T * Tp, fma(gp, tau * v * T * Tp * iL, Dq * p2 * p2));
    o.h23 = d3 * fma(Gq, Tp, Cq * p2);
    o.h24 = fma(gp, Ts * v * Tp * i2L, fma(Gq * Tp, d4, fma(Fq * Tp, Ts, p2 * fma(Dq, p4, fma(Cq, d4, Bq * Ts)))));
    o.h34 = fma(w[3], Ts, fma(gs, 0.5 * Ts, d3 * fma(Cq, p4, (A + E) * Ts)));
    o.h44 = fma(Dq * p4, p4, 2 * fma(Cq * d4, p4, fma(Bq * Ts, p4, (A + E) * Ts * d4)));
}

// ---------------------------------------------------------------- small dense helpers (compile-time sizes)
template <int NMAX, int RS = 1>
OBCA_FN int ldl_fact(int n, double *A) {   // A: NMAX x NMAX row-major; lower triangle in; out: L (strict lower), 1/D (diagonal)
    int bad = 0;
    double D[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
        D[j] = 1.0;
        if (j < n) {
            double d = A[j * NMAX + j];
#pragma unroll
            for (int k = 0; k < NMAX; k++) if (k < j) d -= A[j * NMAX + k] * A[j * NMAX + k] * D[k];
            if (!(d > 0)) bad = 1;
            D[j] = d;
            const double id = rcp_nr<RS>(d);
            A[j * NMAX + j] = id;
#pragma unroll
            for (int i = 0; i < NMAX; i++) if (i > j && i < n) {
                double s = A[i * NMAX + j];
#pragma unroll
                for (int k = 0; k < NMAX; k++) if (k < j) s -= A[i * NMAX + k] * A[j * NMAX + k] * D[k];
                A[i * NMAX + j] = s * id;
            }
        }
    }
    return bad;   // 1 if some pivot is not strictly positive (or NaN)
}
template <int NMAX>
OBCA_FN void ldl_solve(int n, const double *A, double *b) {
#pragma unroll
    for (int i = 0; i < NMAX; i++) if (i < n) {
#pragma unroll
        for (int k = 0; k < NMAX; k++) if (k < i) b[i] -= A[i * NMAX + k] * b[k];
    }
#pragma unroll
    for (int i = 0; i < NMAX; i++) if (i < n) b[i] *= A[i * NMAX + i];
#pragma unroll
    for (int ii = 0; ii < NMAX; ii++) {
        int i = NMAX - 1 - ii;
        if (i < n) {
#pragma unroll
            for (int k = 0; k < NMAX; k++) if (k > i && k < n) b[i] -= A[k * NMAX + i] * b[k];
        }
    }
}
OBCA_FN int chol2(double q00, double q10, double q11, double Lc[3]) {   // 2x2 LDL' (no square roots): Lc = {1/d0, l, 1/d1}; 0 unless positive definite
    if (!(q00 > 0)) return 0;
    Lc[0] = rcp_nr(q00); Lc[1] = q10 * Lc[0];
    const double d = q11 - Lc[1] * q10;
    if (!(d > 0)) return 0;
    Lc[2] = rcp_nr(d);
    return 1;
}
OBCA_FN void chol2_solve(const double Lc[3], double &b0, double &b1) {
    b1 = (b1 - Lc[1] * b0) * Lc[2];
    b0 = b0 * Lc[0] - Lc[1] * b1;
}
template <int VM>
OBCA_FN void hh_apply(int v, const double *w, double beta, double *x) {   // x <- (I - beta w w') x, beta = 2 / (w'w): w is NOT normalised
    double s = 0;                                                         // (a square root and a division less on the dependent chain)
    // No predicate on i < v: the callers' w is zero beyond the obstacle's v rows and their
    // x finite there (the rows an obstacle does not have enter every block matrix as
    // identity rows), so those terms are exact zeros -- the same bits as with the predicate,
    // which cost two selects per term (140 of an obstacle item's 2 500 instructions).
    (void)v;
#pragma unroll
    for (int i = 0; i < VM; i++) s += w[i] * x[i];
    s *= beta;
#pragma unroll
    for (int i = 0; i < VM; i++) x[i] -= s * w[i];
}

// ---------------------------------------------------------------- one (stage, obstacle) block
// VM = compile-time bound on the half-space rows per obstacle (2 for the shipped parking scenarios, 4 or 8 in general): all small
// matrices of a block are sized by it, which decides whether the block fits the register file without spilling.
template <int VM>
struct ObsIn {
    int v;
    double a1[VM], a2[VM], b[VM];
    double lam[VM], zl[VM], mu[4], zm[4], y[4];
    double sl, so, zso, X, Y, psi;
    double zs1;              // ParkingDist only: multiplier of the norm-row slack, which lives in `sl`
};

// the four rows c1..c4 (ParkingSignedDist.jl:198-206, c4 has the slack `so` and dmin moved to the left)
template <int VM>
OBCA_FN void obs_rows(const Consts &c, const ObsIn<VM> &in, double r[4]) {
    double p1 = 0, p2 = 0, beta = 0;
#pragma unroll
    // (rows beyond in.v: a = b = 0, lam = 1 -- load_obs -- exact zeros, no predicate needed)
    for (int i = 0; i < VM; i++) { p1 += in.a1[i] * in.lam[i]; p2 += in.a2[i] * in.lam[i]; beta += in.b[i] * in.lam[i]; }
    double sn, cs;
    sincos_bounded(in.psi, &sn, &cs);
    r[0] = p1 * p1 + p2 * p2 - 1 + (c.dist ? in.sl : 0.0);          // ParkingDist.jl:200: <= 1 (its slack is kept in the sl slot)
    r[1] = in.mu[0] - in.mu[2] + cs * p1 + sn * p2;
    r[2] = in.mu[1] - in.mu[3] - sn * p1 + cs * p2;
    r[3] = -(c.g[0] * in.mu[0] + c.g[1] * in.mu[1] + c.g[2] * in.mu[2] + c.g[3] * in.mu[3]) + (in.X + cs * c.off) * p1 +
           (in.Y + sn * c.off) * p2 - beta + (c.dist ? 0.0 : in.sl) - OB_DMIN - in.so;   // ParkingDist.jl:207-208: no slack
}

// cmin / cmax: smallest / largest complementarity product s z (the error w.r.t. ANY barrier
struct ObsStats { double dmax, pmax, cmin, cmax, sumz, sumy; int bad; };
                                                                                   // parameter follows from the two: max |s z - mu| =
                                                                                   // max(|cmax - mu|, |cmin - mu|), rounding is monotone)

struct ObsCond {          // result of the condensation onto the pose
    double Hpp[6];        // symmetric 3x3: 00 01 02 11 12 22
    double gz[3];         // Jp^T y   (part of grad L w.r.t. the pose)
    double gcorr[3];      // condensed right-hand-side correction (subtract from the barrier-form gradient)
};
template <int VM>
struct ObsStep { double dlam[VM], dmu[4], dsl, dso, dy[4]; };

// MODE 0: condense (fills cond, stats) ; MODE 1: back-substitute for a given pose step dp (fills step)
// SOC = 1 (second-order correction, IPOPT A-5.5..A-5.9): the right-hand side takes
// the four row values from crs (c_soc = alpha c(z) + c(z + alpha d)) instead of
// the rows at z; the violation statistics keep the true rows.
// LSQ = 1 (least-squares multipliers, IPOPT recalc_y / eq. (36) of Waechter & Biegler):
// Hessian := identity on every variable of the block, no second derivatives, no
// regularisation, zero constraint right-hand side, stationarity residuals in their
// z-form (bound multipliers instead of mu / distance); call with mu_b = dw = dc = 0.
template <int MODE, int VM, int SOC = 0, int LSQ = 0>
OBCA_FN void obs_block(const Consts &c, const ObsIn<VM> &in, double mu_b, double dw, double dc, ObsCond *cond, ObsStats *st,
                       const double dp[3], ObsStep<VM> *step, const double *crs = nullptr) {
    const int v = in.v;
    constexpr int RS_ = VM <= 2 ? 1 : 0;       // which reciprocal form (rcp_nr)
    double p1 = 0, p2 = 0, beta = 0;
#pragma unroll
    // (rows beyond v: a = b = 0, lam = 1 -- load_obs)
    for (int i = 0; i < VM; i++) { p1 += in.a1[i] * in.lam[i]; p2 += in.a2[i] * in.lam[i]; beta += in.b[i] * in.lam[i]; }
    double sn, cs;
    sincos_bounded(in.psi, &sn, &cs);
    const double off = c.off;
    double cr[4];
    cr[0] = p1 * p1 + p2 * p2 - 1 + (c.dist ? in.sl : 0.0);
    cr[1] = in.mu[0] - in.mu[2] + cs * p1 + sn * p2;
    cr[2] = in.mu[1] - in.mu[3] - sn * p1 + cs * p2;
    cr[3] = -(c.g[0] * in.mu[0] + c.g[1] * in.mu[1] + c.g[2] * in.mu[2] + c.g[3] * in.mu[3]) + (in.X + cs * off) * p1 +
            (in.Y + sn * off) * p2 - beta + (c.dist ? 0.0 : in.sl) - OB_DMIN - in.so;
    const double *y = in.y;
    // Jacobians
    double Jl[4][VM];
#pragma unroll
    for (int i = 0; i < VM; i++) {
        double a1 = in.a1[i], a2 = in.a2[i];                                   // (zero beyond the obstacle's v rows: load_obs)
        Jl[0][i] = 2 * (p1 * a1 + p2 * a2);
        Jl[1][i] = cs * a1 + sn * a2;
        Jl[2][i] = -sn * a1 + cs * a2;
        Jl[3][i] = (in.X + cs * off) * a1 + (in.Y + sn * off) * a2 - in.b[i];
    }
    // rows 2..4 w.r.t. pose (X,Y,psi): only these entries are non-zero
    const double jp2 = -sn * p1 + cs * p2, jp3 = -cs * p1 - sn * p2;
    const double Jp[3][3] = {{0, 0, jp2}, {0, 0, jp3}, {p1, p2, off * jp2}};
    // d rows 2..4 / d mu
    const double Jmu[3][4] = {{1, 0, -1, 0}, {0, 1, 0, -1}, {-c.g[0], -c.g[1], -c.g[2], -c.g[3]}};
    // local stationarity residuals, diagonals
    const double iso = rcp_nr<RS_>(in.so);
    // sl: the free penetration slack of ParkingSignedDist (cost 1e2 sl + 1e4 sl^2, enters row 4) or, in the ParkingDist formulation,
    // the slack s1 >= 0 of the norm row 1 (no cost, barrier, multiplier zs1); either way a diagonal pivot
    const double isl = c.dist ? rcp_nr<RS_>(in.sl) : 0.0;
    const double iDso = LSQ ? 1.0 : rcp_nr<RS_>(in.zso * iso + dw), iDsl = LSQ ? 1.0 : rcp_nr<RS_>((c.dist ? in.zs1 * isl : 2e4) + dw);
    const double iDs4 = c.dist ? 0.0 : iDsl, iDs1 = c.dist ? iDsl : 0.0;          // where the pivot lands: row 4 or row 1
    double r_so = LSQ ? -y[3] - in.zso : -y[3] - mu_b * iso, r_sl = c.dist ? (LSQ ? y[0] - in.zs1 : y[0] - mu_b * isl) : 1e2 + 2e4 * in.sl + y[3];
    double iDmu[4], r_mu[4], Dlam[VM], r_lam[VM];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // Jmu' y with the 0 / +-1 entries of Jmu written out (a product with a literal
        const double jy = (i == 0 ? y[1] : (i == 1 ? y[2] : (i == 2 ? -y[1] : -y[2]))) + Jmu[2][i] * y[3];
                                                                                                                    // zero cannot be folded by the compiler --
                                                                                                                    // 0 x inf -- and cost two operations each)
        if (LSQ) { r_mu[i] = jy - in.zm[i]; iDmu[i] = 1.0; }
        else { const double im = rcp_nr<RS_>(in.mu[i]); r_mu[i] = jy - mu_b * im; iDmu[i] = rcp_nr<RS_>(in.zm[i] * im + dw); }
        if (MODE == 0) {
            double rz = fabs(jy - in.zm[i]); st->dmax = fmax(st->dmax, rz);
            double cc = in.mu[i] * in.zm[i]; st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc);
            st->sumz += fabs(in.zm[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < VM; i++) {
        if (i < v) {
            double jy = Jl[0][i] * y[0] + Jl[1][i] * y[1] + Jl[2][i] * y[2] + Jl[3][i] * y[3];
            if (LSQ) { r_lam[i] = jy - in.zl[i]; Dlam[i] = 1.0; }
            else { const double il = rcp_nr<RS_>(in.lam[i]); r_lam[i] = jy - mu_b * il; Dlam[i] = in.zl[i] * il + dw; }
            if (MODE == 0) {
                double rz = fabs(jy - in.zl[i]); st->dmax = fmax(st->dmax, rz);
                double cc = in.lam[i] * in.zl[i]; st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc);
                st->sumz += fabs(in.zl[i]);
            }
        } else { r_lam[i] = 0; Dlam[i] = 1; }
    }
    if (MODE == 0) {
        double rz = fabs(-y[3] - in.zso); st->dmax = fmax(st->dmax, rz);
        { const double rzs = c.dist ? fabs(y[0] - in.zs1) : fabs(r_sl); st->dmax = fmax(st->dmax, rzs); }
        if (c.dist) { const double c1 = in.sl * in.zs1; st->cmin = fmin(st->cmin, c1); st->cmax = fmax(st->cmax, c1); st->sumz += fabs(in.zs1); }
        double cc = in.so * in.zso; st->cmin = fmin(st->cmin, cc); st->cmax = fmax(st->cmax, cc);
        st->sumz += fabs(in.zso);
#pragma unroll
        for (int r = 0; r < 4; r++) { st->pmax = fmax(st->pmax, fabs(cr[r])); st->sumy += fabs(y[r]); }
    }
    // rows 2..4 after eliminating so, sl, mu:   Jl dlam + Jp dpose - T dy = r234
    // T = Jmu diag(1 / D_mu) Jmu' + delta_c I with Jmu = [1 0 -1 0; 0 1 0 -1; -g'] written
    // out (same terms in the same order as the triple loop over its entries; the loop
    // multiplied by the literal zeros, which the compiler may not fold: 62 + 40 operations per block against 25)
    double Tm[9];
    {
        const double g0 = Jmu[2][0], g1 = Jmu[2][1], g2 = Jmu[2][2], g3 = Jmu[2][3];      // (= -c.g[i])
        Tm[0] = (iDmu[0] + iDmu[2]) + dc; Tm[1] = Tm[3] = 0.0;
        Tm[2] = Tm[6] = g0 * iDmu[0] + (-g2) * iDmu[2];
        Tm[4] = (iDmu[1] + iDmu[3]) + dc;
        Tm[5] = Tm[7] = g1 * iDmu[1] + (-g3) * iDmu[3];
        Tm[8] = ((((g0 * g0) * iDmu[0] + (g1 * g1) * iDmu[1]) + (g2 * g2) * iDmu[2]) + (g3 * g3) * iDmu[3]) + dc;
    }
    Tm[8] += iDso + iDs4;
    double r234[3];
    {
        const double c1 = LSQ ? 0.0 : -(SOC ? crs[1] : cr[1]), c2 = LSQ ? 0.0 : -(SOC ? crs[2] : cr[2]), c3 = LSQ ? 0.0 : -(SOC ? crs[3] : cr[3]);
        r234[0] = (c1 + r_mu[0] * iDmu[0]) + (-r_mu[2]) * iDmu[2];
        r234[1] = (c2 + r_mu[1] * iDmu[1]) + (-r_mu[3]) * iDmu[3];
        r234[2] = (((c3 + (Jmu[2][0] * r_mu[0]) * iDmu[0]) + (Jmu[2][1] * r_mu[1]) * iDmu[1]) + (Jmu[2][2] * r_mu[2]) * iDmu[2]) + (Jmu[2][3] * r_mu[3]) * iDmu[3];
    }
    r234[2] += -r_so * iDso + r_sl * iDs4;
    int bad = ldl_fact<3, RS_>(3, Tm);
    // W = T^{-1} [Jl234 | Jp234 | r234]
    double W[3][VM + 4];
#pragma unroll
    for (int cI = 0; cI < VM + 4; cI++) {
        double col[3];
#pragma unroll
        for (int r = 0; r < 3; r++) col[r] = cI < VM ? Jl[r + 1][cI] : (cI < VM + 3 ? Jp[r][cI - VM] : r234[r]);
        ldl_solve<3>(3, Tm, col);
#pragma unroll
        for (int r = 0; r < 3; r++) W[r][cI] = col[r];
    }
    // (lambda, y1) block in the null space of q = Jl[0]
    double Kb[VM * VM], Cp[VM][3], rk[VM + 1];
#pragma unroll
    for (int i = 0; i < VM; i++) {
        double a1 = in.a1[i], a2 = in.a2[i];
#pragma unroll
        for (int m = 0; m < VM; m++) {
            double b1 = in.a1[m], b2 = in.a2[m];
            double a_ = LSQ ? 0.0 : y[0] * 2 * (a1 * b1 + a2 * b2);
#pragma unroll
            for (int r = 0; r < 3; r++) a_ += Jl[r + 1][i] * W[r][m];
            Kb[i * VM + m] = a_;
        }
        Kb[i * VM + i] += Dlam[i];
        const double Hlp[3] = {LSQ ? 0.0 : y[3] * a1, LSQ ? 0.0 : y[3] * a2,
                               LSQ ? 0.0 : y[1] * (-sn * a1 + cs * a2) + y[2] * (-cs * a1 - sn * a2) + y[3] * off * (-sn * a1 + cs * a2)};
#pragma unroll
        for (int cI = 0; cI < 3; cI++) {
            double a_ = Hlp[cI];
#pragma unroll
            for (int r = 0; r < 3; r++) a_ += Jl[r + 1][i] * W[r][VM + cI];
            Cp[i][cI] = i < v ? a_ : 0.0;
        }
        double a_ = -r_lam[i];
#pragma unroll
        for (int r = 0; r < 3; r++) a_ += Jl[r + 1][i] * W[r][VM + 3];
        rk[i] = i < v ? a_ : 0.0;
    }
    rk[VM] = (LSQ ? 0.0 : -(SOC ? crs[0] : cr[0])) + r_sl * iDs1;      // (the eliminated norm-row slack of ParkingDist stays in the least-squares system too)
    const double dc1 = (LSQ ? 0.0 : dc) + iDs1;            // (y1, y1) pivot: -(delta_c + 1/D_s1)
    // Householder Qh q = alpha e1
    double hw[VM], nq = 0;
#pragma unroll
    for (int i = 0; i < VM; i++) nq += Jl[0][i] * Jl[0][i];                    // (Jl is zero beyond v)
    nq = sqrt(nq);
    double alpha = Jl[0][0] > 0 ? -nq : nq, nw = 0;
#pragma unroll
    for (int i = 0; i < VM; i++) { hw[i] = Jl[0][i] - (i == 0 ? alpha : 0.0); nw += hw[i] * hw[i]; }
    const double hb = nw > 0 ? 2.0 * rcp_nr<RS_>(nw) : 0.0;
    // Ht = Qh Kb Qh
#pragma unroll
    for (int j = 0; j < VM; j++) {
        double col[VM];
#pragma unroll
        for (int i = 0; i < VM; i++) col[i] = Kb[i * VM + j];
        hh_apply<VM>(v, hw, hb, col);
#pragma unroll
        for (int i = 0; i < VM; i++) Kb[i * VM + j] = col[i];
    }
#pragma unroll
    for (int i = 0; i < VM; i++) hh_apply<VM>(v, hw, hb, Kb + i * VM);
    double a00 = Kb[0], det = a00 * (-dc1) - alpha * alpha;
    if (!(det < 0)) bad = 1;
    const double idet = rcp_nr<RS_>(det), Mi0 = -dc1 * idet, Mi1 = -alpha * idet, Mi2 = a00 * idet;
    double hc[VM - 1], Hr[(VM - 1) * (VM - 1)];
#pragma unroll
    for (int i = 0; i < VM - 1; i++) hc[i] = (i + 1 < v) ? Kb[(i + 1) * VM] : 0.0;
#pragma unroll
    for (int i = 0; i < VM - 1; i++)
#pragma unroll
        for (int j = 0; j < VM - 1; j++) Hr[i * (VM - 1) + j] = Kb[(i + 1) * VM + (j + 1)] - Mi0 * hc[i] * hc[j];
    if (v > 1) bad |= ldl_fact<(VM > 1 ? VM - 1 : 1), RS_>(v - 1, Hr);
    // solve K^{-1} col  for col = [r_lam(v); r_y]
    auto ksolve = [&](double *col /* VM+1 */) {
        hh_apply<VM>(v, hw, hb, col);
        double g0 = col[0], gy = col[VM];
        double t0 = Mi0 * g0 + Mi1 * gy;
        double rr[VM - 1];
#pragma unroll
        for (int i = 0; i < VM - 1; i++) rr[i] = (i + 1 < v) ? col[i + 1] - hc[i] * t0 : 0.0;
        if (v > 1) ldl_solve<(VM > 1 ? VM - 1 : 1)>(v - 1, Hr, rr);
        double hl = 0;
#pragma unroll
        for (int i = 0; i < VM - 1; i++) if (i + 1 < v) hl += hc[i] * rr[i];
        g0 -= hl;
        col[0] = Mi0 * g0 + Mi1 * gy;
        col[VM] = Mi1 * g0 + Mi2 * gy;
#pragma unroll
        for (int i = 0; i < VM - 1; i++) col[i + 1] = (i + 1 < v) ? rr[i] : 0.0;
        hh_apply<VM>(v, hw, hb, col);
    };
    if (MODE == 0) {
        st->bad |= bad;
        // Z = K^{-1} [Cp | rk]
        double Z[VM + 1][4];
#pragma unroll
        for (int cI = 0; cI < 4; cI++) {
            double col[VM + 1];
#pragma unroll
            for (int i = 0; i < VM; i++) col[i] = cI < 3 ? Cp[i][cI] : rk[i];
            col[VM] = cI < 3 ? 0.0 : rk[VM];
            ksolve(col);
#pragma unroll
            for (int i = 0; i <= VM; i++) Z[i][cI] = col[i];
        }
        const double Hpp22 = LSQ ? 0.0 : y[1] * (-cs * p1 - sn * p2) + y[2] * (sn * p1 - cs * p2) + y[3] * off * (-cs * p1 - sn * p2);
        int q = 0;
#pragma unroll
        for (int a_ = 0; a_ < 3; a_++) {
#pragma unroll
            for (int b_ = 0; b_ < 3; b_++) if (b_ >= a_) {
                double s_ = (a_ == 2 && b_ == 2) ? Hpp22 : 0.0;
#pragma unroll
                for (int r = 0; r < 3; r++) s_ += Jp[r][a_] * W[r][VM + b_];
#pragma unroll
                for (int i = 0; i < VM; i++) s_ -= Cp[i][a_] * Z[i][b_];
                cond->Hpp[q++] = s_;
            }
            double s_ = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) s_ += Jp[r][a_] * W[r][VM + 3];
#pragma unroll
            for (int i = 0; i < VM; i++) s_ -= Cp[i][a_] * Z[i][3];
            cond->gcorr[a_] = s_;
            cond->gz[a_] = Jp[0][a_] * y[1] + Jp[1][a_] * y[2] + Jp[2][a_] * y[3];
        }
    } else {
        double col[VM + 1];
#pragma unroll
        for (int i = 0; i < VM; i++) col[i] = rk[i] - (Cp[i][0] * dp[0] + Cp[i][1] * dp[1] + Cp[i][2] * dp[2]);
        col[VM] = rk[VM];
        ksolve(col);
        double r3[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            // dy234 = T^{-1}(Jl dlam + Jp dp - r234) = W[:, :v] dlam + W[:, v:v+3] dp - W[:, v+3]
            double a_ = -W[r][VM + 3];
#pragma unroll
            for (int i = 0; i < VM; i++) a_ += W[r][i] * col[i];
#pragma unroll
            for (int i = 0; i < 3; i++) a_ += W[r][VM + i] * dp[i];
            r3[r] = a_;
        }
        step->dy[0] = col[VM]; step->dy[1] = r3[0]; step->dy[2] = r3[1]; step->dy[3] = r3[2];
#pragma unroll
        for (int i = 0; i < VM; i++) step->dlam[i] = col[i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double jy = (i == 0 ? r3[0] : (i == 1 ? r3[1] : (i == 2 ? -r3[0] : -r3[1]))) + Jmu[2][i] * r3[2];
            step->dmu[i] = (-r_mu[i] - jy) * iDmu[i];
        }
        step->dsl = (-r_sl - (c.dist ? col[VM] : r3[2])) * iDsl;
        step->dso = (r3[2] - r_so) * iDso;
    }
}

// ---------------------------------------------------------------- DualMultWS: one (pose, obstacle) convex problem
// max d = -g'mu + (A e - b)'lam  s.t. |A'lam|^2<=1, G'mu + R'A'lam = 0, lam,mu>=0   (DualMultWS.jl:52-73)
// feasible-start primal-dual path following (sigma = 0.1) down to an average complementarity of 1e-9.
template <int VM>
OBCA_FN void dualws_one(int v, const double *a1, const double *a2, const double *bj, const double g[4], double ex, double ey,
                        double cs, double sn, double *lam, double *mu, double *dout) {
    double Q0[VM], Q1[VM], cl[VM], amax = 0;
#pragma unroll
    for (int i = 0; i < VM; i++) {
        double x1 = i < v ? a1[i] : 0.0, x2 = i < v ? a2[i] : 0.0;
        Q0[i] = cs * x1 + sn * x2; Q1[i] = -sn * x1 + cs * x2;
        cl[i] = x1 * ex + x2 * ey - (i < v ? bj[i] : 0.0);
        double nr = sqrt(x1 * x1 + x2 * x2); amax = fmax(amax, nr);
    }
    double zl[VM], zm[4], zh = 1, eta0 = 0, eta1 = 0;
#pragma unroll
    for (int i = 0; i < VM; i++) { lam[i] = i < v ? 0.5 / (v * fmax(amax, 1e-12)) : 0.0; zl[i] = 1; }
    {
        double q0 = 0, q1 = 0;
#pragma unroll
        for (int i = 0; i < VM; i++) { q0 += Q0[i] * lam[i]; q1 += Q1[i] * lam[i]; }
        mu[0] = 1 + fmax(0.0, -q0); mu[2] = mu[0] + q0; mu[1] = 1 + fmax(0.0, -q1); mu[3] = mu[1] + q1;
#pragma unroll
        for (int i = 0; i < 4; i++) zm[i] = 1;
    }
    const double Em0[4] = {1, 0, -1, 0}, Em1[4] = {0, 1, 0, -1};
    const double iv5 = 1.0 / (v + 5);
    for (int it = 0; it < 60; it++) {
        double p1 = 0, p2 = 0;
#pragma unroll
        for (int i = 0; i < VM; i++) if (i < v) { p1 += a1[i] * lam[i]; p2 += a2[i] * lam[i]; }
        double h = 1 - p1 * p1 - p2 * p2;
        double gap = h * zh;
#pragma unroll
        for (int i = 0; i < VM; i++) if (i < v) gap += lam[i] * zl[i];
#pragma unroll
        for (int i = 0; i < 4; i++) gap += mu[i] * zm[i];
        double mbar = gap * iv5;
        double gh[VM], rl[VM], rm[4], rmax = 0;
#pragma unroll
        for (int i = 0; i < VM; i++) {
            if (i < v) {
                gh[i] = -2 * (p1 * a1[i] + p2 * a2[i]);
                rl[i] = -cl[i] + Q0[i] * eta0 + Q1[i] * eta1 - zl[i] - zh * gh[i];
                rmax = fmax(rmax, fabs(rl[i]));
            } else { gh[i] = 0; rl[i] = 0; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { rm[i] = g[i] + Em0[i] * eta0 + Em1[i] * eta1 - zm[i]; rmax = fmax(rmax, fabs(rm[i])); }
        if (mbar < 1e-9 && rmax < 1e-9) break;
        double mt = 0.1 * mbar;
        // reciprocals of the barrier variables, once per iteration (reciprocal + Newton, obca_model.h: rcp_nr); the subproblem is a chain of
        // dependent scalar operations per lane, and an IEEE division is 11 of them
        const double ih = rcp_nr(h);
        double il[VM], im[4];
#pragma unroll
        for (int i = 0; i < VM; i++) il[i] = i < v ? rcp_nr(lam[i]) : 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) im[i] = rcp_nr(mu[i]);
        double Hl[VM * VM], bl[VM], Dm[4], iDm[4], bm[4];
#pragma unroll
        for (int i = 0; i < VM; i++) {
#pragma unroll
            for (int j = 0; j < VM; j++)
                Hl[i * VM + j] = (i < v && j < v) ? zh * 2 * (a1[i] * a1[j] + a2[i] * a2[j]) + (zh * ih) * gh[i] * gh[j] : 0.0;
            if (i < v) { Hl[i * VM + i] += zl[i] * il[i]; bl[i] = -(rl[i] + zl[i] - mt * il[i] + (zh - mt * ih) * gh[i]); }
            else bl[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { Dm[i] = zm[i] * im[i]; iDm[i] = rcp_nr(Dm[i]); bm[i] = -(rm[i] + zm[i] - mt * im[i]); }
        if (ldl_fact<VM>(v, Hl)) break;
        double HiQ0[VM], HiQ1[VM], Hib[VM];
#pragma unroll
        for (int i = 0; i < VM; i++) { HiQ0[i] = Q0[i]; HiQ1[i] = Q1[i]; Hib[i] = bl[i]; }
        ldl_solve<VM>(v, Hl, HiQ0); ldl_solve<VM>(v, Hl, HiQ1); ldl_solve<VM>(v, Hl, Hib);
        double S00 = 0, S01 = 0, S11 = 0, rs0 = 0, rs1 = 0;
#pragma unroll
        for (int i = 0; i < VM; i++) if (i < v) {
            S00 += Q0[i] * HiQ0[i]; S01 += Q0[i] * HiQ1[i]; S11 += Q1[i] * HiQ1[i];
            rs0 += Q0[i] * Hib[i]; rs1 += Q1[i] * Hib[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            S00 += Em0[i] * Em0[i] * iDm[i]; S01 += Em0[i] * Em1[i] * iDm[i]; S11 += Em1[i] * Em1[i] * iDm[i];
            rs0 += Em0[i] * bm[i] * iDm[i]; rs1 += Em1[i] * bm[i] * iDm[i];
        }
        double Lc[3];
        if (!chol2(S00, S01, S11, Lc)) break;
        double de0 = rs0, de1 = rs1;
        chol2_solve(Lc, de0, de1);
        double dl[VM], dm[4], dzl[VM], dzm[4], ghd = 0;
#pragma unroll
        for (int i = 0; i < VM; i++) {
            dl[i] = i < v ? Hib[i] - HiQ0[i] * de0 - HiQ1[i] * de1 : 0.0;
            dzl[i] = i < v ? mt * il[i] - zl[i] - zl[i] * il[i] * dl[i] : 0.0;
            ghd += gh[i] * dl[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dm[i] = (bm[i] - Em0[i] * de0 - Em1[i] * de1) * iDm[i];
            dzm[i] = mt * im[i] - zm[i] - zm[i] * im[i] * dm[i];
        }
        double dzh = mt * ih - zh - zh * ih * ghd;
        double a = 1, tb = 0.995, cc;
#define OB_FTB(val, dv) { cc = (dv) < 0 ? -tb * (val) * rcp_nr(dv) : 1e300; if (cc < a) a = cc; }
#pragma unroll
        for (int i = 0; i < VM; i++) if (i < v) { OB_FTB(lam[i], dl[i]); OB_FTB(zl[i], dzl[i]); }
#pragma unroll
        for (int i = 0; i < 4; i++) { OB_FTB(mu[i], dm[i]); OB_FTB(zm[i], dzm[i]); }
        OB_FTB(zh, dzh);
#undef OB_FTB
        for (int bt = 0; bt < 60; bt++) {
            double q1 = 0, q2 = 0;
#pragma unroll
            for (int i = 0; i < VM; i++) if (i < v) { q1 += a1[i] * (lam[i] + a * dl[i]); q2 += a2[i] * (lam[i] + a * dl[i]); }
            if (1 - q1 * q1 - q2 * q2 >= (1 - tb) * h) break;
            a *= 0.7;
        }
#pragma unroll
        for (int i = 0; i < VM; i++) if (i < v) { lam[i] += a * dl[i]; zl[i] += a * dzl[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { mu[i] += a * dm[i]; zm[i] += a * dzm[i]; }
        zh += a * dzh; eta0 += a * de0; eta1 += a * de1;
    }
    double dv = 0;
#pragma unroll
    for (int i = 0; i < VM; i++) if (i < v) dv += cl[i] * lam[i];
#pragma unroll
    for (int i = 0; i < 4; i++) dv -= g[i] * mu[i];
    *dout = dv;
}

}  // namespace obca
