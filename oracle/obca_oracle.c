/*
 * obca_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, sequential, fp64 restatement of the reference's parking signed-distance hot path:
 *   /root/reference/AutonomousParking/ParkingSignedDist.jl:29-314  (NLP definition, warm start, exit flag)
 *   /root/reference/AutonomousParking/DualMultWS.jl:29-86          (dual-multiplier warm start)
 * The arithmetic of the reference lives in third-party code that is NOT in /root/reference and not
 * pinned (JuMP <=0.18 + Ipopt 3.12.x + MUMPS); what is restated here is the
 * reference's problem statement (file:line cited at each block) solved with IPOPT's published
 * algorithm (Waechter & Biegler 2006, "Algorithm A": monotone barrier, filter line search, inertia
 * correction) using the option values of the reference's call sites (ParkingSignedDist.jl:41-43,
 * DualMultWS.jl:36-37).
 *
 * PARITY UNPINNED: the reference ships no golden vectors for this path and Julia/IPOPT cannot run here.
 * The oracle is pinned instead by (i) known-answer geometry for DualMultWS, (ii) derivative checks against
 * autograd, (iii) dense independent solvers (oracle/ipm_dense.py at N = 8 / 24; oracle/ipm_ref80.py: the reference's
 * UN-reformulated NLP at N = 80, tests/golden/dense_N80.npz) and (iv) the reference's own feasibility checker
 * (ParkingConstraints.jl) restated in obca_amd/validate.py and in ref_constraints() below.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this file.
 *
 * Linear algebra: the Newton/KKT system is solved in structured form
 *   (a) every (stage, obstacle) block (lambda_j, mu_j, sl_j, slack, 4 multipliers) is condensed onto the
 *       pose (X,Y,psi) of its stage,
 *   (b) the steering-rate row is condensed onto (delta_{k-1}, delta_k, t),
 *   (c) the remaining optimal-control problem in (x_k, u_{k-1} copy, u_k) is solved by a Riccati recursion,
 *   (d) the global time scale t and the terminal multiplier nu are a 5x5 border.
 * Inertia is read off the signs of the pivots of (a)-(d); wrong signs trigger IPOPT's delta_w ladder.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VMAX 8
#define NOBMAX 16
#define NCOL 6 /* right-hand sides through the Riccati: main, t, nu1..nu4 */

typedef struct {
    int N, nOb, M, fixTime;
    int dist;   /* 1: ParkingDist.jl (no slack sl, |A'lam|^2 <= 1 with its own slack stored in the sl slot, weight 0.5 on a^2); 0: ParkingSignedDist.jl */
    int vOb[NOBMAX], roff[NOBMAX + 1];
    double Ts, L, g[4], off, XYb[4], x0[4], xF[4];
    const double *A, *b, *rx, *ry, *ryaw; /* A: M x 2 row major */
    double xl[4], xu[4];
    double An[2 * NOBMAX * VMAX], bn[NOBMAX * VMAX], rn[NOBMAX * VMAX];   /* unit-length rows a_i / |a_i|, b_i / |a_i| and the row lengths |a_i| (setup_prob) */
} prob_t;

typedef struct {
    double tol;
    int max_iter;
    double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
    double dw_min, dw0, dw_max, kw_inc0, kw_inc, kw_dec, dc_bar, kappa_c;
    double gamma_theta, gamma_phi, delta, s_theta, s_phi, eta_phi, gamma_alpha, s_max, kappa_sigma;
    double constr_viol_tol, dual_inf_tol, compl_inf_tol, rho_term;
    int lsq_init, verbose;
    int max_soc;      /* second-order correction trials per iteration (IPOPT max_soc = 4); 0 = off, the default: see DESIGN.md section 2 for the measured A/B */
    int recalc_y;     /* recalc_y = "yes" (ParkingSignedDist.jl:41): least-squares multipliers once the constraint violation is below 1e-6; 0 = off (default) */
    int obj_scaling;  /* IPOPT's gradient-based objective scaling: a no-op on this path (|grad f|_inf at the reference's start is the slack penalty, exactly 100: factor 1); the quadcopter oracle applies it */
    int restoration;  /* stand-in for IPOPT's restoration phase on degenerate obstacle blocks (restore_blocks below): 0 = off (default); 1 = at the start of an attempt and where IPOPT would
                         enter restoration (failed line search / inertia ladder exhausted); 2 = the latter only (test knob); obca_reference_opts of the HIP library sets 1 */
} opts_t;

void obca_oracle_default_opts(opts_t *o) {
    o->tol = 1e-5; o->max_iter = 200;               /* ParkingSignedDist.jl:42 */
    o->mu_init = 0.1; o->kappa_eps = 10; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99;
    o->bound_push = 1e-2; o->bound_frac = 1e-2;
    o->dw_min = 1e-12;                               /* min_hessian_perturbation, :43 */
    o->dw0 = 1e-4; o->dw_max = 1e40; o->kw_inc0 = 100; o->kw_inc = 8; o->kw_dec = 1.0 / 3;
    o->dc_bar = 1e-7;                                /* jacobian_regularization_value, :43 */
    o->kappa_c = 0.25;
    o->gamma_theta = 1e-5; o->gamma_phi = 1e-8; o->delta = 1; o->s_theta = 1.1; o->s_phi = 2.3;
    o->eta_phi = 1e-8; o->gamma_alpha = 0.05; o->s_max = 100; o->kappa_sigma = 1e10;
    o->constr_viol_tol = 1e-4; o->dual_inf_tol = 1; o->compl_inf_tol = 1e-4;
    o->rho_term = 1e3; o->lsq_init = 0; o->verbose = 0;
    o->max_soc = 0; o->recalc_y = 0; o->obj_scaling = 0; o->restoration = 0;                 /* the three IPOPT switches are off by default and set by the caller (never through the environment: a leaked variable would change what the parity tests compare) */
}

/* ------------------------------------------------------------------ iterate layout (one flat vector) */
typedef struct {
    int x, u, t, lam, mu, sl, so, ss;            /* primal */
    int pi, nu, yg, yo;                           /* equality multipliers */
    int zxL, zxU, zuL, zuU, ztL, ztU, zlam, zmu, zso, zssL, zssU, zs1; /* bound multipliers (zs1: of the norm-row slack, ParkingDist only) */
    int nprimal, len;
} lay_t;

static void make_layout(const prob_t *p, lay_t *l) {
    int N = p->N, N1 = N + 1, nOb = p->nOb, M = p->M, o = 0;
    l->x = o; o += 4 * N1; l->u = o; o += 2 * N; l->t = o; o += 1;
    l->lam = o; o += M * N1; l->mu = o; o += 4 * nOb * N1; l->sl = o; o += nOb * N1;
    l->so = o; o += nOb * N1; l->ss = o; o += N; l->nprimal = o;
    l->pi = o; o += 4 * N; l->nu = o; o += 4; l->yg = o; o += N; l->yo = o; o += 4 * nOb * N1;
    l->zxL = o; o += 4 * N1; l->zxU = o; o += 4 * N1; l->zuL = o; o += 2 * N; l->zuU = o; o += 2 * N;
    l->ztL = o; o += 1; l->ztU = o; o += 1; l->zlam = o; o += M * N1; l->zmu = o; o += 4 * nOb * N1;
    l->zso = o; o += nOb * N1; l->zssL = o; o += N; l->zssU = o; o += N; l->zs1 = o; o += nOb * N1; l->len = o;
}

static const double UL[2] = {-0.6, -0.4}, UU[2] = {0.6, 0.4}; /* ParkingSignedDist.jl:100-101 */
static const double TL = 0.8, TU = 1.2;                       /* :110 */
static const double SSB = 0.6;                                /* :167-173 */
static const double DMIN = 0.05;                              /* :33 */

/* ------------------------------------------------------------------ model pieces */
/* bicycle model with 2nd-order terms, ParkingSignedDist.jl:147-150.  vars: psi,v,delta,a,t */
static void dyn_eval(const prob_t *p, const double *x, const double *u, double t, double F[4],
                     double dF[4][5] /* d(F_i - x_i)/d(psi,v,delta,a,t) or NULL */,
                     const double *w /* multipliers or NULL */, double HL[5][5] /* sum_i w_i * Hess(F_i) */) {
    double Ts = p->Ts, L = p->L, psi = x[2], v = x[3], de = u[0], a = u[1];
    double tau = Ts * t, s = v + 0.5 * tau * a, T = tan(de), Tp = 1 + T * T;
    double phi = psi + tau * v * T / (2 * L), c = cos(phi), sn = sin(phi);
    F[0] = x[0] + tau * s * c; F[1] = x[1] + tau * s * sn; F[2] = psi + tau * s * T / L; F[3] = v + tau * a;
    if (!dF) return;
    double dtau[5] = {0, 0, 0, 0, Ts};
    double ds[5] = {0, 1, 0, 0.5 * tau, 0.5 * Ts * a};
    double dphi[5] = {1, tau * T / (2 * L), tau * v * Tp / (2 * L), 0, Ts * v * T / (2 * L)};
    double dT[5] = {0, 0, Tp, 0, 0};
    double g1[3] = {s * c, tau * c, -tau * s * sn};   /* d(tau s cos phi)/d(tau,s,phi) */
    double g2[3] = {s * sn, tau * sn, tau * s * c};
    double g3[3] = {s * T / L, tau * T / L, tau * s / L}; /* d(tau s T/L)/d(tau,s,T) */
    for (int i = 0; i < 5; i++) {
        dF[0][i] = g1[0] * dtau[i] + g1[1] * ds[i] + g1[2] * dphi[i];
        dF[1][i] = g2[0] * dtau[i] + g2[1] * ds[i] + g2[2] * dphi[i];
        dF[2][i] = g3[0] * dtau[i] + g3[1] * ds[i] + g3[2] * dT[i];
        dF[3][i] = 0;
    }
    dF[2][0] += 0; /* psi enters F_psi only through the leading psi (handled as identity) */
    dF[3][3] = tau; dF[3][4] = Ts * a;
    if (!w) return;
    /* second derivatives by the chain rule: H = sum_ab Gm_ab dm_a dm_b^T + sum_a gm_a Hm_a */
    double Hs[5][5] = {{0}}, Hphi[5][5] = {{0}}, HT[5][5] = {{0}};
    Hs[3][4] = Hs[4][3] = 0.5 * Ts;
    Hphi[1][2] = Hphi[2][1] = tau * Tp / (2 * L);
    Hphi[1][4] = Hphi[4][1] = Ts * T / (2 * L);
    Hphi[2][2] = tau * v * T * Tp / L;
    Hphi[2][4] = Hphi[4][2] = Ts * v * Tp / (2 * L);
    HT[2][2] = 2 * T * Tp;
    double G1[3][3] = {{0, c, -s * sn}, {c, 0, -tau * sn}, {-s * sn, -tau * sn, -tau * s * c}};
    double G2[3][3] = {{0, sn, s * c}, {sn, 0, tau * c}, {s * c, tau * c, -tau * s * sn}};
    double G3[3][3] = {{0, T / L, s / L}, {T / L, 0, tau / L}, {s / L, tau / L, 0}};
    const double *dm12[3] = {dtau, ds, dphi};
    const double *dm3[3] = {dtau, ds, dT};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            double h1 = 0, h2 = 0, h3 = 0;
            for (int a_ = 0; a_ < 3; a_++)
                for (int b_ = 0; b_ < 3; b_++) {
                    h1 += G1[a_][b_] * dm12[a_][i] * dm12[b_][j];
                    h2 += G2[a_][b_] * dm12[a_][i] * dm12[b_][j];
                    h3 += G3[a_][b_] * dm3[a_][i] * dm3[b_][j];
                }
            h1 += g1[1] * Hs[i][j] + g1[2] * Hphi[i][j];
            h2 += g2[1] * Hs[i][j] + g2[2] * Hphi[i][j];
            h3 += g3[1] * Hs[i][j] + g3[2] * HT[i][j];
            double h4 = ((i == 3 && j == 4) || (i == 4 && j == 3)) ? Ts : 0.0;
            HL[i][j] = w[0] * h1 + w[1] * h2 + w[2] * h3 + w[3] * h4;
        }
}

/* objective pieces, ParkingSignedDist.jl:78-92 */
static double wa_of(const prob_t *p) { return (p->fixTime || p->dist) ? 0.5 : 0.1; }   /* ParkingDist.jl:87 */
static double wpsi_of(const prob_t *p) { return p->fixTime ? 1e-2 : 1e-4; }

/* obstacle rows of one (stage, obstacle), ParkingSignedDist.jl:190-207 */
typedef struct { double p1, p2, beta, cs, sn; } obs_aux;
static void obs_rows(const prob_t *p, int j, const double *x, const double *lam, const double *mu, double sl,
                     double so, double c[4], obs_aux *ax) {
    const double *Aj = p->A + 2 * p->roff[j], *bj = p->b + p->roff[j];
    int v = p->vOb[j];
    double p1 = 0, p2 = 0, beta = 0;
    for (int i = 0; i < v; i++) { p1 += Aj[2 * i] * lam[i]; p2 += Aj[2 * i + 1] * lam[i]; beta += bj[i] * lam[i]; }
    double cs = cos(x[2]), sn = sin(x[2]);
    c[0] = p1 * p1 + p2 * p2 - 1 + (p->dist ? sl : 0.0);          /* ParkingDist.jl:200: <= 1, slack kept in the sl slot */
    c[1] = mu[0] - mu[2] + cs * p1 + sn * p2;
    c[2] = mu[1] - mu[3] - sn * p1 + cs * p2;
    c[3] = -(p->g[0] * mu[0] + p->g[1] * mu[1] + p->g[2] * mu[2] + p->g[3] * mu[3]) + (x[0] + cs * p->off) * p1 +
           (x[1] + sn * p->off) * p2 - beta + (p->dist ? 0.0 : sl) - DMIN - so;   /* ParkingDist.jl:207-208: no slack */
    if (ax) { ax->p1 = p1; ax->p2 = p2; ax->beta = beta; ax->cs = cs; ax->sn = sn; }
}

/* objective and constraint 1-norm / inf-norm at a primal point (line search) */
static void eval_f_theta(const prob_t *p, const lay_t *l, const double *z, double *f, double *th1, double *thinf) {
    int N = p->N, nOb = p->nOb, M = p->M;
    double t = z[l->t], q = t * p->Ts, wa = wa_of(p), wpsi = wpsi_of(p);
    double J = 0, th = 0, ti = 0;
    for (int k = 0; k < N; k++) {
        const double *u = z + l->u + 2 * k;
        double w0 = k ? u[-2] : 0, w1 = k ? u[-1] : 0;
        J += 0.01 * u[0] * u[0] + wa * u[1] * u[1];
        J += 0.1 * ((u[0] - w0) * (u[0] - w0) + (u[1] - w1) * (u[1] - w1)) / (q * q);
        double F[4];
        dyn_eval(p, z + l->x + 4 * k, u, t, F, NULL, NULL, NULL);
        for (int i = 0; i < 4; i++) { double r = fabs(z[l->x + 4 * (k + 1) + i] - F[i]); th += r; if (r > ti) ti = r; }
        double r = fabs((w0 - u[0]) / q - z[l->ss + k]); th += r; if (r > ti) ti = r;
    }
    if (!p->fixTime) J += (N + 1) * (0.5 * t + t * t);
    for (int k = 0; k <= N; k++) {
        const double *x = z + l->x + 4 * k;
        J += 1e-4 * x[3] * x[3] + 1e-3 * (x[0] - p->rx[k]) * (x[0] - p->rx[k]) + 1e-3 * (x[1] - p->ry[k]) * (x[1] - p->ry[k]) +
             wpsi * (x[2] - p->ryaw[k]) * (x[2] - p->ryaw[k]);
        for (int j = 0; j < nOb; j++) {
            double c[4], sl = z[l->sl + k * nOb + j];
            if (!p->dist) J += 1e2 * sl + 1e4 * sl * sl;
            obs_rows(p, j, x, z + l->lam + k * M + p->roff[j], z + l->mu + 4 * (k * nOb + j), sl, z[l->so + k * nOb + j], c, NULL);
            for (int i = 0; i < 4; i++) { double r = fabs(c[i]); th += r; if (r > ti) ti = r; }
        }
    }
    for (int i = 0; i < 4; i++) { double r = fabs(z[l->x + 4 * N + i] - p->xF[i]); th += r; if (r > ti) ti = r; }
    *f = J; *th1 = th; if (thinf) *thinf = ti;
}

static double barrier_terms(const prob_t *p, const lay_t *l, const double *z) {
    /* sum of log-distances to the bounds; returns sum mult*log(.) */
    int N = p->N, nOb = p->nOb, M = p->M;
    double s = 0;
    for (int k = 1; k <= N; k++)
        for (int i = 0; i < 4; i++) if (i != 2) s += log(z[l->x + 4 * k + i] - p->xl[i]) + log(p->xu[i] - z[l->x + 4 * k + i]);
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < 2; i++) s += log(z[l->u + 2 * k + i] - UL[i]) + log(UU[i] - z[l->u + 2 * k + i]);
        s += log(z[l->ss + k] + SSB) + log(SSB - z[l->ss + k]);
    }
    if (!p->fixTime) s += (N + 1) * (log(z[l->t] - TL) + log(TU - z[l->t]));
    for (int i = 0; i < M * (N + 1); i++) s += log(z[l->lam + i]);
    for (int i = 0; i < 4 * nOb * (N + 1); i++) s += log(z[l->mu + i]);
    for (int i = 0; i < nOb * (N + 1); i++) s += log(z[l->so + i]);
    if (p->dist) for (int i = 0; i < nOb * (N + 1); i++) s += log(z[l->sl + i]);
    return s;
}

/* ------------------------------------------------------------------ small dense helpers */
static int chol2(const double Q[2][2], double Lc[3]) { /* Q = L L^T, L = [l0 0; l1 l2] */
    if (!(Q[0][0] > 0)) return 0;
    Lc[0] = sqrt(Q[0][0]); Lc[1] = Q[1][0] / Lc[0];
    double d = Q[1][1] - Lc[1] * Lc[1];
    if (!(d > 0)) return 0;
    Lc[2] = sqrt(d);
    return 1;
}
static void chol2_solve(const double Lc[3], double b[2]) {
    b[0] /= Lc[0]; b[1] = (b[1] - Lc[1] * b[0]) / Lc[2];
    b[1] /= Lc[2]; b[0] = (b[0] - Lc[1] * b[1]) / Lc[0];
}
/* LDL^T without pivoting of an n x n symmetric matrix (n<=8); returns #negative pivots, -1 if a zero pivot */
static int ldl_n(int n, double *Am /* n x n row-major, lower used; overwritten: strict lower = L, diag = D */) {
    int neg = 0;
    for (int j = 0; j < n; j++) {
        double d = Am[j * n + j];
        for (int k = 0; k < j; k++) d -= Am[j * n + k] * Am[j * n + k] * Am[k * n + k];
        if (d == 0 || d != d) return -1;
        if (d < 0) neg++;
        Am[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = Am[i * n + j];
            for (int k = 0; k < j; k++) s -= Am[i * n + k] * Am[j * n + k] * Am[k * n + k];
            Am[i * n + j] = s / d;
        }
    }
    return neg;
}
static void ldl_solve(int n, const double *Am, double *b) {
    for (int i = 0; i < n; i++) for (int k = 0; k < i; k++) b[i] -= Am[i * n + k] * b[k];
    for (int i = 0; i < n; i++) b[i] /= Am[i * n + i];
    for (int i = n - 1; i >= 0; i--) for (int k = i + 1; k < n; k++) b[i] -= Am[k * n + i] * b[k];
}

/* ------------------------------------------------------------------ workspace of one Newton system */
typedef struct {
    int v;
    double Dso, Dsl, Dmu[4], Dlam[VMAX];
    double Tf[9];                 /* LDL of T (rows c2,c3,c4), positive definite */
    double Jl[4][VMAX];           /* d c_r / d lambda */
    double Jp[4][3];              /* d c_r / d (X,Y,psi) */
    double r234[3];               /* rhs of rows 2..4 after slack/mu elimination */
    /* (lambda, y1) block, factorised in the null space of q = d c1/d lambda: Householder Qh q = alpha e1,
       2x2 pivot on (lambda~_0, y1), then LDL of the remaining (v-1) x (v-1) reduced Hessian (must be PD) */
    double hw[VMAX], Minv[3], hc[VMAX], Hr[VMAX * VMAX];
    double Cp[VMAX + 1][3];       /* coupling of (lambda,y1) to the pose, after y234 elimination */
    double rk[VMAX + 1];          /* rhs of (lambda,y1) */
    double r_so, r_sl, r_mu[4];   /* stationarity residuals (barrier form) */
    double c[4];
} obs_fact;

typedef struct {
    const prob_t *p; const lay_t *l;
    int N;
    /* per stage */
    double (*Hs)[8][8];  /* Hessian over (x,w,u) */
    double (*hz)[8];     /* gradient, z-form (uses bound multipliers): for the optimality error */
    double (*hb)[8];     /* gradient, barrier form: right-hand side */
    double (*Ht)[8];     /* coupling to t */
    double Htt, gt_z, gt_b;
    double (*Ad)[4][4], (*Bd)[4][2], (*Ftd)[4], (*dd)[4]; /* dynamics linearisation, d = -(x+ - F) */
    double *sig_g, *rg;  /* steering-row condensation: sigma_k, modified residual */
    double (*gg)[3];     /* d g / d(w0, delta, t) */
    double *Dss, *r_ss;
    obs_fact *of;        /* (N+1)*nOb */
    /* Riccati */
    double (*P)[6][6], (*K)[2][6], (*Lq)[3];
    double (*pv)[NCOL][6], (*kf)[NCOL][2];
    double (*ds)[NCOL][6], (*du)[NCOL][2], (*pic)[NCOL][4];
    /* errors */
    double dinf, cinf_mu0, pinf, sumy, sumz; int nb, nm;
    const double *csoc;  /* second-order correction: constraint values that replace c(z) on the right-hand side (layout pi | nu | yg | yo), or NULL */
} kkt_t;

static void *xcalloc(size_t n, size_t s) { void *q = calloc(n ? n : 1, s); if (!q) { fprintf(stderr, "oom\n"); exit(1); } return q; }

static kkt_t *kkt_alloc(const prob_t *p, const lay_t *l) {
    kkt_t *k = xcalloc(1, sizeof *k);
    int N1 = p->N + 1;
    k->p = p; k->l = l; k->N = p->N;
    k->Hs = xcalloc(N1, sizeof *k->Hs); k->hz = xcalloc(N1, sizeof *k->hz); k->hb = xcalloc(N1, sizeof *k->hb);
    k->Ht = xcalloc(N1, sizeof *k->Ht);
    k->Ad = xcalloc(N1, sizeof *k->Ad); k->Bd = xcalloc(N1, sizeof *k->Bd); k->Ftd = xcalloc(N1, sizeof *k->Ftd);
    k->dd = xcalloc(N1, sizeof *k->dd);
    k->sig_g = xcalloc(N1, sizeof(double)); k->rg = xcalloc(N1, sizeof(double)); k->gg = xcalloc(N1, sizeof *k->gg);
    k->Dss = xcalloc(N1, sizeof(double)); k->r_ss = xcalloc(N1, sizeof(double));
    k->of = xcalloc((size_t)N1 * p->nOb, sizeof *k->of);
    k->P = xcalloc(N1, sizeof *k->P); k->K = xcalloc(N1, sizeof *k->K); k->Lq = xcalloc(N1, sizeof *k->Lq);
    k->pv = xcalloc(N1, sizeof *k->pv); k->kf = xcalloc(N1, sizeof *k->kf);
    k->ds = xcalloc(N1, sizeof *k->ds); k->du = xcalloc(N1, sizeof *k->du); k->pic = xcalloc(N1 + 1, sizeof *k->pic);
    return k;
}
static void kkt_free(kkt_t *k) {
    free(k->Hs); free(k->hz); free(k->hb); free(k->Ht); free(k->Ad); free(k->Bd); free(k->Ftd); free(k->dd);
    free(k->sig_g); free(k->rg); free(k->gg); free(k->Dss); free(k->r_ss); free(k->of);
    free(k->P); free(k->K); free(k->Lq); free(k->pv); free(k->kf); free(k->ds); free(k->du); free(k->pic); free(k);
}


static void hh_apply(int v, const double *w, double *x) { /* x <- (I - 2 w w^T) x, |w|=1 or w=0 */
    double s = 0; for (int i = 0; i < v; i++) s += w[i] * x[i];
    for (int i = 0; i < v; i++) x[i] -= 2 * s * w[i];
}
/* factor [[Hb, q],[q^T, -dc]]; returns 1 iff inertia is (v, 1, 0) */
static int lamblock_factor(obs_fact *F, int v, const double *Hb /* v x v full */, const double *q, double dc) {
    if (v < 1 || v > VMAX) return 0;      /* (callers pass the row count of an obstacle, checked at entry; stated here for the compiler's bounds analysis) */
    double nq = 0; for (int i = 0; i < v; i++) nq += q[i] * q[i];
    nq = sqrt(nq);
    double alpha = q[0] > 0 ? -nq : nq, w[VMAX], nw = 0;
    for (int i = 0; i < v; i++) { w[i] = q[i] - (i == 0 ? alpha : 0); nw += w[i] * w[i]; }
    nw = sqrt(nw);
    for (int i = 0; i < v; i++) F->hw[i] = nw > 0 ? w[i] / nw : 0;
    double Ht[VMAX * VMAX];
    for (int j = 0; j < v; j++) { double col[VMAX]; for (int i = 0; i < v; i++) col[i] = Hb[i * v + j]; hh_apply(v, F->hw, col); for (int i = 0; i < v; i++) Ht[i * v + j] = col[i]; }
    for (int i = 0; i < v; i++) hh_apply(v, F->hw, Ht + i * v); /* rows */
    double a = Ht[0], det = a * (-dc) - alpha * alpha;
    if (!(det < 0)) return 0;
    F->Minv[0] = -dc / det; F->Minv[1] = -alpha / det; F->Minv[2] = a / det;
    int m = v - 1;
    for (int i = 0; i < m; i++) F->hc[i] = Ht[(i + 1) * v];
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) F->Hr[i * m + j] = Ht[(i + 1) * v + (j + 1)] - F->Minv[0] * F->hc[i] * F->hc[j];
    if (m > 0 && ldl_n(m, F->Hr) != 0) return 0;
    return 1;
}
static void lamblock_solve(const obs_fact *F, int v, double *col /* in: [r_lam; r_y] out: [lam; y1] */) {
    int m = v - 1;
    hh_apply(v, F->hw, col);
    double g0 = col[0], gy = col[v];
    double t0 = F->Minv[0] * g0 + F->Minv[1] * gy;
    double rr[VMAX];
    for (int i = 0; i < m; i++) rr[i] = col[i + 1] - F->hc[i] * t0;
    if (m > 0) ldl_solve(m, F->Hr, rr);
    double hl = 0; for (int i = 0; i < m; i++) hl += F->hc[i] * rr[i];
    g0 -= hl;
    col[0] = F->Minv[0] * g0 + F->Minv[1] * gy;
    col[v] = F->Minv[1] * g0 + F->Minv[2] * gy;
    for (int i = 0; i < m; i++) col[i + 1] = rr[i];
    hh_apply(v, F->hw, col);
}

/* two-sided bound helper: returns Sigma, adds gradient contributions */
static inline void bound2(double v, double lo, double hi, double zL, double zU, double mu, double mult, double *Sig,
                          double *gz, double *gb, double *cmax, double *sumz, int lsq) {
    double dL = v - lo, dU = hi - v;
    *Sig = mult * (zL / dL + zU / dU);
    *gz += mult * (-zL + zU);
    *gb += lsq ? mult * (-zL + zU) : mult * (-mu / dL + mu / dU);   /* least-squares multiplier mode: the right-hand side is the gradient of the Lagrangian WITH the bound multipliers */
    double c1 = fabs(dL * zL), c2 = fabs(dU * zU); /* complementarity at mu=0 */
    if (c1 > *cmax) *cmax = c1; if (c2 > *cmax) *cmax = c2;
    *sumz += fabs(zL) + fabs(zU);
}

/*
 * Assemble the condensed Newton system at iterate z with barrier mu, regularisation (dw, dc).
 * lsq != 0: "least-squares multiplier" mode -- Hessian := identity, no second derivatives, dw=dc=0
 * (IPOPT's initial y, Waechter & Biegler eq. (36)).
 * Returns 1 if all block pivots have the expected signs.
 */
static int kkt_assemble(kkt_t *K, const double *z, double mu, double dw, double dc, int lsq) {
    const prob_t *p = K->p; const lay_t *l = K->l;
    int N = p->N, nOb = p->nOb, M = p->M, ok = 1;
    double t = z[l->t], q = t * p->Ts, wa = wa_of(p), wpsi = wpsi_of(p);
    double hsc = lsq ? 0.0 : 1.0; /* scale of true Hessian terms */
    double cmax = 0, sumz = 0, dmax = 0, pmax = 0, sumy = 0;
    int nb = 0, nm = 0;
    K->Htt = 0; K->gt_z = 0; K->gt_b = 0;
    if (!p->fixTime) {
        double Sig, gz = 0, gb = 0;
        bound2(t, TL, TU, z[l->ztL], z[l->ztU], mu, N + 1, &Sig, &gz, &gb, &cmax, &sumz, lsq);
        nb += 2 * (N + 1); sumz += N * (fabs(z[l->ztL]) + fabs(z[l->ztU]));
        double gf = (N + 1) * (0.5 + 2 * t);
        K->Htt = lsq ? (double)(N + 1) : (2.0 * (N + 1) + Sig + dw);      /* (least-squares system: t stands for the N + 1 timeScale variables of the reference's model: N + 1 unit diagonal entries) */
        K->gt_z = gf + gz; K->gt_b = gf + gb;
    } else K->Htt = 1.0;
    for (int k = 0; k <= N; k++) {
        double (*H)[8] = K->Hs[k];
        double *hz = K->hz[k], *hb = K->hb[k], *Ht = K->Ht[k];
        memset(H, 0, sizeof K->Hs[k]); memset(hz, 0, sizeof K->hz[k]); memset(hb, 0, sizeof K->hb[k]);
        memset(Ht, 0, sizeof K->Ht[k]);
        const double *x = z + l->x + 4 * k;
        /* ---- state cost + bounds (ParkingSignedDist.jl:82-83/90-91, :104-106) */
        double gx[4] = {2e-3 * (x[0] - p->rx[k]), 2e-3 * (x[1] - p->ry[k]), 2 * wpsi * (x[2] - p->ryaw[k]), 2e-4 * x[3]};
        double hx[4] = {2e-3, 2e-3, 2 * wpsi, 2e-4};
        for (int i = 0; i < 4; i++) {
            hz[i] = gx[i]; hb[i] = gx[i];
            double Sig = 0;
            if (i != 2 && k >= 1) {
                bound2(x[i], p->xl[i], p->xu[i], z[l->zxL + 4 * k + i], z[l->zxU + 4 * k + i], mu, 1, &Sig, &hz[i], &hb[i], &cmax, &sumz, lsq);
                nb += 2;
            }
            H[i][i] = lsq ? 1.0 : (hx[i] + Sig + dw);
        }
        /* ---- obstacle blocks: condense onto (X,Y,psi) */
        for (int j = 0; j < nOb; j++) {
            obs_fact *F = &K->of[k * nOb + j];
            int v = p->vOb[j]; F->v = v;
            const double *Aj = p->A + 2 * p->roff[j], *bj = p->b + p->roff[j];
            const double *lam = z + l->lam + k * M + p->roff[j], *mu_ = z + l->mu + 4 * (k * nOb + j);
            const double *zl = z + l->zlam + k * M + p->roff[j], *zm = z + l->zmu + 4 * (k * nOb + j);
            const double *y = z + l->yo + 4 * (k * nOb + j);
            double sl = z[l->sl + k * nOb + j], so = z[l->so + k * nOb + j], zso = z[l->zso + k * nOb + j];
            obs_aux ax;
            obs_rows(p, j, x, lam, mu_, sl, so, F->c, &ax);
            for (int r = 0; r < 4; r++) { double a_ = fabs(F->c[r]); if (a_ > pmax) pmax = a_; sumy += fabs(y[r]); }
            double cr[4];
            for (int r = 0; r < 4; r++) cr[r] = K->csoc ? K->csoc[(l->yo - l->pi) + 4 * (k * nOb + j) + r] : F->c[r];
            nm += 4;
            double cs = ax.cs, sn = ax.sn, p1 = ax.p1, p2 = ax.p2, off = p->off;
            /* Jacobians */
            for (int i = 0; i < v; i++) {
                double a1 = Aj[2 * i], a2 = Aj[2 * i + 1];
                F->Jl[0][i] = 2 * (p1 * a1 + p2 * a2);
                F->Jl[1][i] = cs * a1 + sn * a2;
                F->Jl[2][i] = -sn * a1 + cs * a2;
                F->Jl[3][i] = (x[0] + cs * off) * a1 + (x[1] + sn * off) * a2 - bj[i];
            }
            double Jp[4][3] = {{0, 0, 0}, {0, 0, -sn * p1 + cs * p2}, {0, 0, -cs * p1 - sn * p2}, {p1, p2, off * (-sn * p1 + cs * p2)}};
            memcpy(F->Jp, Jp, sizeof Jp);
            static const double Jm[4][4] = {{0, 0, 0, 0}, {1, 0, -1, 0}, {0, 1, 0, -1}, {0, 0, 0, 0}};
            double Jmu[4][4];
            memcpy(Jmu, Jm, sizeof Jm);
            for (int i = 0; i < 4; i++) Jmu[3][i] = -p->g[i];
            /* stationarity residuals of the local variables */
            double rso_z = -y[3] - zso, rso_b = -y[3] - mu / so;
            double rsl = 1e2 + 2e4 * sl + y[3];
            F->Dso = lsq ? 1.0 : (zso / so + dw); F->Dsl = lsq ? 1.0 : (2e4 + dw);
            F->r_so = lsq ? rso_z : rso_b; F->r_sl = rsl;
            if (p->dist) {   /* the sl slot holds the slack of the norm row: s1 >= 0, gradient y1, multiplier zs1 */
                double zs1 = z[l->zs1 + k * nOb + j];
                rsl = y[0] - zs1; F->r_sl = lsq ? rsl : y[0] - mu / sl; F->Dsl = lsq ? 1.0 : zs1 / sl + dw;
                double c_ = fabs(sl * zs1); if (c_ > cmax) cmax = c_; sumz += fabs(zs1); nb++;
            }
            if (fabs(rso_z) > dmax) dmax = fabs(rso_z); if (fabs(rsl) > dmax) dmax = fabs(rsl);
            { double c_ = fabs(so * zso); if (c_ > cmax) cmax = c_; sumz += fabs(zso); nb++; }
            for (int i = 0; i < 4; i++) {
                double jy = 0; for (int r = 1; r < 4; r++) jy += Jmu[r][i] * y[r];
                double rz = jy - zm[i]; F->r_mu[i] = lsq ? rz : jy - mu / mu_[i];
                F->Dmu[i] = lsq ? 1.0 : (zm[i] / mu_[i] + dw);
                if (fabs(rz) > dmax) dmax = fabs(rz);
                double c_ = fabs(mu_[i] * zm[i]); if (c_ > cmax) cmax = c_; sumz += fabs(zm[i]); nb++;
            }
            double rl_b[VMAX];
            for (int i = 0; i < v; i++) {
                double jy = 0; for (int r = 0; r < 4; r++) jy += F->Jl[r][i] * y[r];
                double rz = jy - zl[i]; rl_b[i] = lsq ? rz : jy - mu / lam[i];
                F->Dlam[i] = lsq ? 1.0 : (zl[i] / lam[i] + dw);
                if (fabs(rz) > dmax) dmax = fabs(rz);
                double c_ = fabs(lam[i] * zl[i]); if (c_ > cmax) cmax = c_; sumz += fabs(zl[i]); nb++;
            }
            /* pose gradient from these rows: Jp^T y */
            for (int r = 1; r < 4; r++) for (int i = 0; i < 3; i++) { hz[i] += Jp[r][i] * y[r]; hb[i] += Jp[r][i] * y[r]; }
            /* Lagrangian Hessian pieces */
            double Hll[VMAX][VMAX], Hlp[VMAX][3], Hpp[3][3] = {{0}};
            for (int i = 0; i < v; i++) {
                double a1 = Aj[2 * i], a2 = Aj[2 * i + 1];
                for (int m_ = 0; m_ < v; m_++) Hll[i][m_] = hsc * y[0] * 2 * (a1 * Aj[2 * m_] + a2 * Aj[2 * m_ + 1]);
                Hlp[i][0] = hsc * y[3] * a1; Hlp[i][1] = hsc * y[3] * a2;
                Hlp[i][2] = hsc * (y[1] * (-sn * a1 + cs * a2) + y[2] * (-cs * a1 - sn * a2) + y[3] * off * (-sn * a1 + cs * a2));
            }
            Hpp[2][2] = hsc * (y[1] * (-cs * p1 - sn * p2) + y[2] * (sn * p1 - cs * p2) + y[3] * off * (-cs * p1 - sn * p2));
            /* rows 2..4 after eliminating so, sl, mu:  Jl dlam + Jp dpose - T dy = r234 */
            double T[9] = {0};
            for (int r = 0; r < 3; r++) {
                for (int s_ = 0; s_ <= r; s_++) {
                    double a_ = 0; for (int i = 0; i < 4; i++) a_ += Jmu[r + 1][i] * Jmu[s_ + 1][i] / F->Dmu[i];
                    T[r * 3 + s_] = a_;
                }
                T[r * 3 + r] += lsq ? 0.0 : dc;
            }
            T[8] += 1.0 / F->Dso + (p->dist ? 0.0 : 1.0 / F->Dsl);
            for (int r = 0; r < 3; r++) {
                double a_ = lsq ? 0.0 : -cr[r + 1];
                for (int i = 0; i < 4; i++) a_ += Jmu[r + 1][i] * F->r_mu[i] / F->Dmu[i];
                F->r234[r] = a_;
            }
            F->r234[2] += -F->r_so / F->Dso + (p->dist ? 0.0 : F->r_sl / F->Dsl);
            memcpy(F->Tf, T, sizeof T);
            if (ldl_n(3, F->Tf) != 0) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "T fail k=%d j=%d\n", k, j); }
            /* (lambda,y1) block */
            int n = v + 1;
            double Kb[VMAX * VMAX] = {0}, W[3][VMAX + 3 + 1]; /* W = T^{-1} [Jl234 | Jp234 | r234] */
            for (int c_ = 0; c_ < v + 4; c_++) {
                double col[3];
                for (int r = 0; r < 3; r++) col[r] = c_ < v ? F->Jl[r + 1][c_] : (c_ < v + 3 ? Jp[r + 1][c_ - v] : F->r234[r]);
                ldl_solve(3, F->Tf, col);
                for (int r = 0; r < 3; r++) W[r][c_] = col[r];
            }
            for (int i = 0; i < v; i++) {
                for (int m_ = 0; m_ < v; m_++) {
                    double a_ = Hll[i][m_];
                    for (int r = 0; r < 3; r++) a_ += F->Jl[r + 1][i] * W[r][m_];
                    Kb[i * v + m_] = a_;
                }
                Kb[i * v + i] += F->Dlam[i];
                for (int c_ = 0; c_ < 3; c_++) {
                    double a_ = Hlp[i][c_];
                    for (int r = 0; r < 3; r++) a_ += F->Jl[r + 1][i] * W[r][v + c_];
                    F->Cp[i][c_] = a_;
                }
                double a_ = -rl_b[i];
                for (int r = 0; r < 3; r++) a_ += F->Jl[r + 1][i] * W[r][v + 3];
                F->rk[i] = a_;
            }
            F->Cp[v][0] = F->Cp[v][1] = F->Cp[v][2] = 0; F->rk[v] = (lsq ? 0.0 : -cr[0]) + (p->dist ? F->r_sl / F->Dsl : 0.0);   /* (the eliminated norm-row slack of ParkingDist stays in the least-squares system too) */
            if (!lamblock_factor(F, v, Kb, F->Jl[0], (lsq ? 0.0 : dc) + (p->dist ? 1.0 / F->Dsl : 0.0))) {
                ok = 0;
                if (getenv("OBCA_DBG")) fprintf(stderr, "lamblock fail k=%d j=%d y1=%g dw=%g\n", k, j, y[0], dw);
            }
            /* Schur complement onto the pose */
            double Z[VMAX + 1][4]; /* K^{-1} [Cp | rk] */
            for (int c_ = 0; c_ < 4; c_++) {
                double col[VMAX + 1];
                for (int i = 0; i < n; i++) col[i] = c_ < 3 ? F->Cp[i][c_] : F->rk[i];
                lamblock_solve(F, v, col);
                for (int i = 0; i < n; i++) Z[i][c_] = col[i];
            }
            for (int a_ = 0; a_ < 3; a_++) {
                for (int b_ = 0; b_ < 3; b_++) {
                    double s_ = Hpp[a_][b_];
                    for (int r = 0; r < 3; r++) s_ += Jp[r + 1][a_] * W[r][v + b_];
                    for (int i = 0; i < n; i++) s_ -= F->Cp[i][a_] * Z[i][b_];
                    H[a_][b_] += s_;
                }
                /* rhs:  pose row gets  -(... ) ; hb is a gradient (rhs = -hb), so subtract */
                double s_ = 0;
                for (int r = 0; r < 3; r++) s_ += Jp[r + 1][a_] * W[r][v + 3];
                for (int i = 0; i < n; i++) s_ -= F->Cp[i][a_] * Z[i][3];
                hb[a_] -= s_;
            }
        }
        if (k == N) {
            /* terminal equality x_N = xF (ParkingSignedDist.jl:128-131): augmented-Lagrangian shift rho on x_N */
            for (int i = 0; i < 4; i++) {
                double e = -(x[i] - p->xF[i]);
                if (fabs(e) > pmax) pmax = fabs(e);
            }
            continue;
        }
        /* ---- input cost, rate cost, bounds (ParkingSignedDist.jl:78-80/86-88, :100-101) */
        const double *u = z + l->u + 2 * k;
        double w[2] = {k ? u[-2] : 0, k ? u[-1] : 0};
        double cu[2] = {0.01, wa};
        double rr = 0.1 / (q * q), e1 = u[0] - w[0], e2 = u[1] - w[1], rv = rr * (e1 * e1 + e2 * e2);
        for (int i = 0; i < 2; i++) {
            double ei = i ? e2 : e1, Sig = 0;
            double gu = 2 * cu[i] * u[i] + 2 * rr * ei;
            hz[6 + i] += gu; hb[6 + i] += gu;
            hz[4 + i] += -2 * rr * ei; hb[4 + i] += -2 * rr * ei;
            bound2(u[i], UL[i], UU[i], z[l->zuL + 2 * k + i], z[l->zuU + 2 * k + i], mu, 1, &Sig, &hz[6 + i], &hb[6 + i], &cmax, &sumz, lsq);
            nb += 2;
            H[6 + i][6 + i] += lsq ? 1.0 : (2 * cu[i] + 2 * rr + Sig + dw);
            H[4 + i][4 + i] += hsc * 2 * rr;
            H[4 + i][6 + i] += -hsc * 2 * rr; H[6 + i][4 + i] += -hsc * 2 * rr;
            if (!p->fixTime) { Ht[6 + i] += -hsc * 4 * rr * ei / t; Ht[4 + i] += hsc * 4 * rr * ei / t; }
        }
        if (!p->fixTime) { K->gt_z += -2 * rv / t; K->gt_b += -2 * rv / t; K->Htt += hsc * 6 * rv / (t * t); }
        /* ---- steering-rate row  g=(w0-delta)/(t Ts) - ss = 0, |ss|<=0.6  (ParkingSignedDist.jl:157-174) */
        {
            double g = (w[0] - u[0]) / q, ss = z[l->ss + k], yg = z[l->yg + k];
            double gg[3] = {1 / q, -1 / q, p->fixTime ? 0 : -g / t};
            memcpy(K->gg[k], gg, sizeof gg);
            double Sig, gz = 0, gb = 0;
            bound2(ss, -SSB, SSB, z[l->zssL + k], z[l->zssU + k], mu, 1, &Sig, &gz, &gb, &cmax, &sumz, lsq);
            nb += 2; nm += 1; sumy += fabs(yg);
            double rz = -yg + gz, rb = -yg + gb;
            if (fabs(rz) > dmax) dmax = fabs(rz);
            double res = g - ss; if (fabs(res) > pmax) pmax = fabs(res);
            K->Dss[k] = lsq ? 1.0 : (Sig + dw); K->r_ss[k] = lsq ? rz : rb;
            double sig = 1.0 / (1.0 / K->Dss[k] + (lsq ? 0 : dc));
            K->sig_g[k] = sig; K->rg[k] = (lsq ? 0.0 : (K->csoc ? K->csoc[(l->yg - l->pi) + k] : res)) + K->r_ss[k] / K->Dss[k];
            /* indices of (w0, delta) inside the stage vector */
            int id[2] = {4, 6};
            for (int a_ = 0; a_ < 2; a_++) {
                hz[id[a_]] += gg[a_] * yg; hb[id[a_]] += gg[a_] * (yg + sig * K->rg[k]);
                for (int b_ = 0; b_ < 2; b_++) H[id[a_]][id[b_]] += sig * gg[a_] * gg[b_];
                if (!p->fixTime) Ht[id[a_]] += sig * gg[a_] * gg[2] + hsc * yg * (a_ == 0 ? -1 / (q * t) : 1 / (q * t));
            }
            if (!p->fixTime) {
                K->gt_z += gg[2] * yg; K->gt_b += gg[2] * (yg + sig * K->rg[k]);
                K->Htt += sig * gg[2] * gg[2] + hsc * yg * 2 * g / (t * t);
            }
        }
        /* ---- dynamics x_{k+1} - F(x_k,u_k,t) = 0 with multiplier pi_k (ParkingSignedDist.jl:139-155) */
        {
            double F[4], dF[4][5], HL[5][5];
            const double *pi = z + l->pi + 4 * k;
            dyn_eval(p, x, u, t, F, dF, pi, HL);
            double (*A)[4] = K->Ad[k]; double (*B)[2] = K->Bd[k];
            for (int i = 0; i < 4; i++) {
                for (int j = 0; j < 4; j++) A[i][j] = (i == j);
                A[i][2] += dF[i][0]; A[i][3] += dF[i][1]; B[i][0] = dF[i][2]; B[i][1] = dF[i][3];
                K->Ftd[k][i] = p->fixTime ? 0 : dF[i][4];
                double r = z[l->x + 4 * (k + 1) + i] - F[i];
                K->dd[k][i] = lsq ? 0.0 : -(K->csoc ? K->csoc[4 * k + i] : r); if (fabs(r) > pmax) pmax = fabs(r);
                sumy += fabs(pi[i]);
            }
            nm += 4;
            /* constraint is x+ - F: Lagrangian Hessian = -sum pi_i Hess F_i ; vars (psi,v,delta,a,t) -> stage idx (2,3,6,7,t) */
            static const int id[4] = {2, 3, 6, 7};
            for (int a_ = 0; a_ < 4; a_++) {
                for (int b_ = 0; b_ < 4; b_++) H[id[a_]][id[b_]] += -hsc * HL[a_][b_];
                if (!p->fixTime) Ht[id[a_]] += -hsc * HL[a_][4];
            }
            if (!p->fixTime) K->Htt += -hsc * HL[4][4];
        }
    }
    /* terminal residual e and rho shift */
    K->pinf = pmax; K->cinf_mu0 = cmax; K->sumz = sumz; K->sumy = sumy; K->nb = nb; K->nm = nm + 4;
    K->dinf = dmax;
    for (int i = 0; i < 4; i++) K->sumy += fabs(z[l->nu + i]);
    return ok;
}

/* Add the dynamics / terminal multiplier terms J^T y to the stage gradients (so that the Riccati solves for multiplier
 * INCREMENTS: errors of the recursion are then relative to the step, not to the multiplier) and return the dual
 * infeasibility max |grad L| over x_k, u_k, t. */
static double stage_dual_inf(kkt_t *K, const double *z) {
    const prob_t *p = K->p; const lay_t *l = K->l;
    int N = p->N;
    double dmax = 0;
    for (int k = 1; k <= N; k++) /* x_k */
        for (int i = 0; i < 4; i++) {
            double r = z[l->pi + 4 * (k - 1) + i];
            if (k < N) for (int j = 0; j < 4; j++) r -= K->Ad[k][j][i] * z[l->pi + 4 * k + j];
            else r += z[l->nu + i];
            K->hz[k][i] += r; K->hb[k][i] += r;
            if (fabs(K->hz[k][i]) > dmax) dmax = fabs(K->hz[k][i]);
        }
    for (int i = 0; i < 4; i++) { /* x_0 is a constant: its row is not part of the system */
        double r = 0; for (int j = 0; j < 4; j++) r -= K->Ad[0][j][i] * z[l->pi + j];
        K->hz[0][i] += r; K->hb[0][i] += r;
    }
    for (int k = 0; k < N; k++) { /* u_k : own part + copy part of stage k+1 */
        for (int i = 0; i < 2; i++) {
            double r = 0;
            for (int j = 0; j < 4; j++) r -= K->Bd[k][j][i] * z[l->pi + 4 * k + j];
            K->hz[k][6 + i] += r; K->hb[k][6 + i] += r;
            double tot = K->hz[k][6 + i] + (k + 1 < N ? K->hz[k + 1][4 + i] : 0);
            if (fabs(tot) > dmax) dmax = fabs(tot);
        }
        for (int j = 0; j < 4; j++) { double r = K->Ftd[k][j] * z[l->pi + 4 * k + j]; K->gt_z -= r; K->gt_b -= r; }
    }
    if (!p->fixTime && fabs(K->gt_z) > dmax) dmax = fabs(K->gt_z);
    return dmax;
}

/* Riccati factorisation + solves; fills the direction d (same layout as z).  Returns 1 if inertia is right. */
static int kkt_solve(kkt_t *K, const double *z, double mu, double dc, double rho, int lsq, double *d) {
    const prob_t *p = K->p; const lay_t *l = K->l;
    int N = p->N, nOb = p->nOb, M = p->M, ok = 1;
    /* columns: 0 main (h=hb, off=d), 1 t (h=Ht, off=Ft), 2..5 nu_i (h=e_i on x_N) */
    double Pn[6][6] = {{0}}, pn[NCOL][6] = {{0}};
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Pn[i][j] = K->Hs[N][i][j];
    double e[4];
    for (int i = 0; i < 4; i++) { e[i] = lsq ? 0.0 : -(K->csoc ? K->csoc[(l->nu - l->pi) + i] : z[l->x + 4 * N + i] - p->xF[i]); Pn[i][i] += rho; }
    for (int i = 0; i < 6; i++) { pn[0][i] = K->hb[N][i]; pn[1][i] = K->Ht[N][i]; }
    for (int i = 0; i < 4; i++) { pn[0][i] -= rho * e[i]; pn[2 + i][i] = 1.0; }
    memcpy(K->P[N], Pn, sizeof Pn); memcpy(K->pv[N], pn, sizeof pn);
    for (int k = N - 1; k >= 0; k--) {
        /* F = [A 0 B; 0 0 I] : (x,w,u) -> (x+, w+) */
        double Fm[6][8] = {{0}};
        for (int i = 0; i < 4; i++) { for (int j = 0; j < 4; j++) Fm[i][j] = K->Ad[k][i][j]; Fm[i][6] = K->Bd[k][i][0]; Fm[i][7] = K->Bd[k][i][1]; }
        Fm[4][6] = 1; Fm[5][7] = 1;
        double PF[6][8], Q[8][8];
        for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) { double s = 0; for (int a = 0; a < 6; a++) s += K->P[k + 1][i][a] * Fm[a][j]; PF[i][j] = s; }
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { double s = K->Hs[k][i][j]; for (int a = 0; a < 6; a++) s += Fm[a][i] * PF[a][j]; Q[i][j] = s; }
        double Quu[2][2] = {{Q[6][6], Q[6][7]}, {Q[7][6], Q[7][7]}};
        if (!chol2(Quu, K->Lq[k])) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "Quu fail k=%d %g %g %g\n", k, Quu[0][0], Quu[0][1], Quu[1][1]); return 0; }
        for (int j = 0; j < 6; j++) { double b[2] = {-Q[6][j], -Q[7][j]}; chol2_solve(K->Lq[k], b); K->K[k][0][j] = b[0]; K->K[k][1][j] = b[1]; }
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) K->P[k][i][j] = Q[i][j] + Q[i][6] * K->K[k][0][j] + Q[i][7] * K->K[k][1][j];
        for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) { double s = 0.5 * (K->P[k][i][j] + K->P[k][j][i]); K->P[k][i][j] = K->P[k][j][i] = s; }
        for (int c = 0; c < NCOL; c++) {
            double off[6] = {0}, hv[8] = {0};
            if (c == 0) { for (int i = 0; i < 4; i++) off[i] = K->dd[k][i]; for (int i = 0; i < 8; i++) hv[i] = K->hb[k][i]; }
            else if (c == 1) { for (int i = 0; i < 4; i++) off[i] = K->Ftd[k][i]; for (int i = 0; i < 8; i++) hv[i] = K->Ht[k][i]; }
            double tmp[6], qv[8];
            for (int i = 0; i < 6; i++) { double s = K->pv[k + 1][c][i]; for (int a = 0; a < 6; a++) s += K->P[k + 1][i][a] * off[a]; tmp[i] = s; }
            for (int i = 0; i < 8; i++) { double s = hv[i]; for (int a = 0; a < 6; a++) s += Fm[a][i] * tmp[a]; qv[i] = s; }
            double b[2] = {-qv[6], -qv[7]}; chol2_solve(K->Lq[k], b);
            K->kf[k][c][0] = b[0]; K->kf[k][c][1] = b[1];
            for (int i = 0; i < 6; i++) K->pv[k][c][i] = qv[i] + Q[i][6] * b[0] + Q[i][7] * b[1];
        }
    }
    /* forward sweeps */
    for (int c = 0; c < NCOL; c++) {
        double s[6] = {0};
        for (int k = 0; k < N; k++) {
            memcpy(K->ds[k][c], s, sizeof s);
            double uu[2];
            for (int i = 0; i < 2; i++) { double a = K->kf[k][c][i]; for (int j = 0; j < 6; j++) a += K->K[k][i][j] * s[j]; uu[i] = a; }
            K->du[k][c][0] = uu[0]; K->du[k][c][1] = uu[1];
            double sn[6];
            for (int i = 0; i < 4; i++) {
                double a = (c == 0) ? K->dd[k][i] : (c == 1 ? K->Ftd[k][i] : 0);
                for (int j = 0; j < 4; j++) a += K->Ad[k][i][j] * s[j];
                a += K->Bd[k][i][0] * uu[0] + K->Bd[k][i][1] * uu[1];
                sn[i] = a;
            }
            sn[4] = uu[0]; sn[5] = uu[1];
            memcpy(s, sn, sizeof s);
            for (int i = 0; i < 4; i++) { double a = K->pv[k + 1][c][i]; for (int j = 0; j < 6; j++) a += K->P[k + 1][i][j] * s[j]; K->pic[k + 1][c][i] = -a; }
        }
        memcpy(K->ds[N][c], s, sizeof s);
    }
    /* 5x5 border in (dt, nu):  unknown responses are additive: dz = z0 + zt*dt + sum znu_i*nu_i */
    double Mb[5][5] = {{0}}, rb[5] = {0};
    {
        double att = K->Htt, rt = -K->gt_b;
        double atn[4] = {0};
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < 6; i++) { att += K->Ht[k][i] * K->ds[k][1][i]; rt -= K->Ht[k][i] * K->ds[k][0][i]; for (int c = 0; c < 4; c++) atn[c] += K->Ht[k][i] * K->ds[k][2 + c][i]; }
            if (k < N) {
                for (int i = 0; i < 2; i++) { att += K->Ht[k][6 + i] * K->du[k][1][i]; rt -= K->Ht[k][6 + i] * K->du[k][0][i]; for (int c = 0; c < 4; c++) atn[c] += K->Ht[k][6 + i] * K->du[k][2 + c][i]; }
                for (int i = 0; i < 4; i++) { att -= K->Ftd[k][i] * K->pic[k + 1][1][i]; rt += K->Ftd[k][i] * K->pic[k + 1][0][i]; for (int c = 0; c < 4; c++) atn[c] -= K->Ftd[k][i] * K->pic[k + 1][2 + c][i]; }
            }
        }
        Mb[0][0] = att; rb[0] = rt;
        for (int c = 0; c < 4; c++) {
            /* use the symmetric value from the nu rows (response of x_N to dt) */
            Mb[0][1 + c] = atn[c]; Mb[1 + c][0] = K->ds[N][1][c];
            for (int c2 = 0; c2 < 4; c2++) Mb[1 + c][1 + c2] = K->ds[N][2 + c2][c];
            /* no delta_c on the terminal rows: with the rho shift their block is -(S^-1 + rho)^-1, always well scaled */
            rb[1 + c] = e[c] - K->ds[N][0][c];
        }
        if (p->fixTime) { for (int c = 0; c < 5; c++) Mb[0][c] = Mb[c][0] = 0; Mb[0][0] = 1; rb[0] = 0; }
        for (int a = 1; a < 5; a++) for (int b = 1; b < a; b++) { double s = 0.5 * (Mb[a][b] + Mb[b][a]); Mb[a][b] = Mb[b][a] = s; }
    }
    double dt, nu[4];
    {
        /* eliminate nu first (negative definite), then t (must be positive) */
        double S[16];
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) S[a * 4 + b] = -Mb[1 + a][1 + b];
        if (ldl_n(4, S) != 0) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "nu block fail\n"); }
        double col[4], colr[4];
        for (int a = 0; a < 4; a++) { col[a] = -Mb[1 + a][0]; colr[a] = -rb[1 + a]; }
        ldl_solve(4, S, col); ldl_solve(4, S, colr); /* col = Snn^{-1} a_nt ; colr = Snn^{-1} r_n (with Snn = Mb[1:,1:]) */
        double piv = Mb[0][0], rr = rb[0];
        for (int a = 0; a < 4; a++) { piv -= Mb[0][1 + a] * col[a]; rr -= Mb[0][1 + a] * colr[a]; }
        if (!(piv > 0)) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "t pivot fail %g\n", piv); }
        dt = rr / piv;
        for (int a = 0; a < 4; a++) nu[a] = colr[a] - col[a] * dt;
    }
    if (!ok) return 0;
    /* compose the stage direction */
    memset(d, 0, sizeof(double) * l->len);
    d[l->t] = p->fixTime ? 0 : dt;
    for (int k = 0; k <= N; k++) {
        double dx[6];
        for (int i = 0; i < 6; i++) { double a = K->ds[k][0][i] + K->ds[k][1][i] * dt; for (int c = 0; c < 4; c++) a += K->ds[k][2 + c][i] * nu[c]; dx[i] = a; }
        for (int i = 0; i < 4; i++) d[l->x + 4 * k + i] = dx[i];
        if (k < N) {
            for (int i = 0; i < 2; i++) { double a = K->du[k][0][i] + K->du[k][1][i] * dt; for (int c = 0; c < 4; c++) a += K->du[k][2 + c][i] * nu[c]; d[l->u + 2 * k + i] = a; }
            for (int i = 0; i < 4; i++) {
                double a = K->pic[k + 1][0][i] + K->pic[k + 1][1][i] * dt; for (int c = 0; c < 4; c++) a += K->pic[k + 1][2 + c][i] * nu[c];
                d[l->pi + 4 * k + i] = a; /* multiplier increment (the gradient already holds J^T pi) */
            }
        }
    }
    for (int i = 0; i < 4; i++) d[l->nu + i] = nu[i];
    /* back-substitute the condensed rows */
    for (int k = 0; k < N; k++) {
        double dw0 = k ? d[l->u + 2 * (k - 1)] : 0, dde = d[l->u + 2 * k];
        double lin = K->gg[k][0] * dw0 + K->gg[k][1] * dde + K->gg[k][2] * d[l->t];
        double dy = K->sig_g[k] * (lin + K->rg[k]);
        d[l->yg + k] = dy;
        double dss = (dy - K->r_ss[k]) / K->Dss[k];
        d[l->ss + k] = dss;
        double ss = z[l->ss + k], zL = z[l->zssL + k], zU = z[l->zssU + k];
        d[l->zssL + k] = mu / (ss + SSB) - zL - zL / (ss + SSB) * dss;
        d[l->zssU + k] = mu / (SSB - ss) - zU + zU / (SSB - ss) * dss;
    }
    for (int k = 0; k <= N; k++)
        for (int j = 0; j < nOb; j++) {
            obs_fact *F = &K->of[k * nOb + j];
            int v = F->v, n = v + 1;
            double dp[3] = {d[l->x + 4 * k], d[l->x + 4 * k + 1], d[l->x + 4 * k + 2]};
            double col[VMAX + 1];
            for (int i = 0; i < n; i++) col[i] = F->rk[i] - (F->Cp[i][0] * dp[0] + F->Cp[i][1] * dp[1] + F->Cp[i][2] * dp[2]);
            lamblock_solve(F, v, col);
            double dy[4]; dy[0] = col[v];
            double r3[3];
            for (int r = 0; r < 3; r++) {
                double a = -F->r234[r];
                for (int i = 0; i < v; i++) a += F->Jl[r + 1][i] * col[i];
                for (int i = 0; i < 3; i++) a += F->Jp[r + 1][i] * dp[i];
                r3[r] = a;
            }
            ldl_solve(3, F->Tf, r3);
            dy[1] = r3[0]; dy[2] = r3[1]; dy[3] = r3[2];
            int bo = k * nOb + j;
            for (int r = 0; r < 4; r++) d[l->yo + 4 * bo + r] = dy[r];
            for (int i = 0; i < v; i++) {
                int idx = k * M + p->roff[j] + i;
                d[l->lam + idx] = col[i];
                double la = z[l->lam + idx], zz = z[l->zlam + idx];
                d[l->zlam + idx] = mu / la - zz - zz / la * col[i];
            }
            for (int i = 0; i < 4; i++) {
                double jy = (i == 0 ? dy[1] : i == 1 ? dy[2] : i == 2 ? -dy[1] : -dy[2]) - p->g[i] * dy[3];
                double dm = (-F->r_mu[i] - jy) / F->Dmu[i];
                int idx = 4 * bo + i;
                d[l->mu + idx] = dm;
                double m_ = z[l->mu + idx], zz = z[l->zmu + idx];
                d[l->zmu + idx] = mu / m_ - zz - zz / m_ * dm;
            }
            d[l->sl + bo] = (-F->r_sl - (p->dist ? dy[0] : dy[3])) / F->Dsl;
            if (p->dist) { double s1 = z[l->sl + bo], zz1 = z[l->zs1 + bo]; d[l->zs1 + bo] = mu / s1 - zz1 - zz1 / s1 * d[l->sl + bo]; }
            double dso = (dy[3] - F->r_so) / F->Dso;
            d[l->so + bo] = dso;
            double so = z[l->so + bo], zz = z[l->zso + bo];
            d[l->zso + bo] = mu / so - zz - zz / so * dso;
        }
    /* bound multipliers of x,u,t */
    for (int k = 1; k <= N; k++)
        for (int i = 0; i < 4; i++) if (i != 2) {
            int idx = 4 * k + i; double xv = z[l->x + idx], dxv = d[l->x + idx], zL = z[l->zxL + idx], zU = z[l->zxU + idx];
            d[l->zxL + idx] = mu / (xv - p->xl[i]) - zL - zL / (xv - p->xl[i]) * dxv;
            d[l->zxU + idx] = mu / (p->xu[i] - xv) - zU + zU / (p->xu[i] - xv) * dxv;
        }
    for (int k = 0; k < N; k++)
        for (int i = 0; i < 2; i++) {
            int idx = 2 * k + i; double uv = z[l->u + idx], duv = d[l->u + idx], zL = z[l->zuL + idx], zU = z[l->zuU + idx];
            d[l->zuL + idx] = mu / (uv - UL[i]) - zL - zL / (uv - UL[i]) * duv;
            d[l->zuU + idx] = mu / (UU[i] - uv) - zU + zU / (UU[i] - uv) * duv;
        }
    if (!p->fixTime) {
        double t = z[l->t], zL = z[l->ztL], zU = z[l->ztU];
        d[l->ztL] = mu / (t - TL) - zL - zL / (t - TL) * dt;
        d[l->ztU] = mu / (TU - t) - zU + zU / (TU - t) * dt;
    }
    return 1;
}

/* ------------------------------------------------------------------ step-length rules */
static double ftb_one(double val, double dv, double tau) { return dv < 0 ? -tau * val / dv : 1e300; }

static void frac_to_boundary(const prob_t *p, const lay_t *l, const double *z, const double *d, double tau, double *ap, double *az) {
    int N = p->N, nOb = p->nOb, M = p->M;
    double a = 1.0, b = 1.0, c;
#define PR(val, dv) { c = ftb_one(val, dv, tau); if (c < a) a = c; }
#define DU(val, dv) { c = ftb_one(val, dv, tau); if (c < b) b = c; }
    for (int k = 1; k <= N; k++)
        for (int i = 0; i < 4; i++) if (i != 2) {
            int idx = 4 * k + i;
            PR(z[l->x + idx] - p->xl[i], d[l->x + idx]); PR(p->xu[i] - z[l->x + idx], -d[l->x + idx]);
            DU(z[l->zxL + idx], d[l->zxL + idx]); DU(z[l->zxU + idx], d[l->zxU + idx]);
        }
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < 2; i++) {
            int idx = 2 * k + i;
            PR(z[l->u + idx] - UL[i], d[l->u + idx]); PR(UU[i] - z[l->u + idx], -d[l->u + idx]);
            DU(z[l->zuL + idx], d[l->zuL + idx]); DU(z[l->zuU + idx], d[l->zuU + idx]);
        }
        PR(z[l->ss + k] + SSB, d[l->ss + k]); PR(SSB - z[l->ss + k], -d[l->ss + k]);
        DU(z[l->zssL + k], d[l->zssL + k]); DU(z[l->zssU + k], d[l->zssU + k]);
    }
    if (!p->fixTime) {
        PR(z[l->t] - TL, d[l->t]); PR(TU - z[l->t], -d[l->t]);
        DU(z[l->ztL], d[l->ztL]); DU(z[l->ztU], d[l->ztU]);
    }
    for (int i = 0; i < M * (N + 1); i++) { PR(z[l->lam + i], d[l->lam + i]); DU(z[l->zlam + i], d[l->zlam + i]); }
    for (int i = 0; i < 4 * nOb * (N + 1); i++) { PR(z[l->mu + i], d[l->mu + i]); DU(z[l->zmu + i], d[l->zmu + i]); }
    for (int i = 0; i < nOb * (N + 1); i++) { PR(z[l->so + i], d[l->so + i]); DU(z[l->zso + i], d[l->zso + i]); }
    if (p->dist) for (int i = 0; i < nOb * (N + 1); i++) { PR(z[l->sl + i], d[l->sl + i]); DU(z[l->zs1 + i], d[l->zs1 + i]); }
#undef PR
#undef DU
    *ap = a; *az = b;
}

/* directional derivative of the barrier function along the primal step */
static double barrier_dir(const prob_t *p, const lay_t *l, const double *z, const double *d, double mu) {
    int N = p->N, nOb = p->nOb, M = p->M;
    double t = z[l->t], q = t * p->Ts, wa = wa_of(p), wpsi = wpsi_of(p), g = 0;
    for (int k = 0; k <= N; k++) {
        const double *x = z + l->x + 4 * k, *dx = d + l->x + 4 * k;
        g += 2e-3 * (x[0] - p->rx[k]) * dx[0] + 2e-3 * (x[1] - p->ry[k]) * dx[1] + 2 * wpsi * (x[2] - p->ryaw[k]) * dx[2] + 2e-4 * x[3] * dx[3];
        if (k >= 1) for (int i = 0; i < 4; i++) if (i != 2) g += (-mu / (x[i] - p->xl[i]) + mu / (p->xu[i] - x[i])) * dx[i];
        for (int j = 0; j < nOb; j++) {
            int bo = k * nOb + j;
            g += (p->dist ? -mu / z[l->sl + bo] : 1e2 + 2e4 * z[l->sl + bo]) * d[l->sl + bo] - mu / z[l->so + bo] * d[l->so + bo];
            for (int i = 0; i < 4; i++) g -= mu / z[l->mu + 4 * bo + i] * d[l->mu + 4 * bo + i];
        }
        for (int i = 0; i < M; i++) g -= mu / z[l->lam + k * M + i] * d[l->lam + k * M + i];
    }
    double gt = 0;
    for (int k = 0; k < N; k++) {
        const double *u = z + l->u + 2 * k, *du = d + l->u + 2 * k;
        double w[2] = {k ? u[-2] : 0, k ? u[-1] : 0}, dwv[2] = {k ? du[-2] : 0, k ? du[-1] : 0};
        double cu[2] = {0.01, wa}, rr = 0.1 / (q * q), rv = 0;
        for (int i = 0; i < 2; i++) {
            double ei = u[i] - w[i];
            g += (2 * cu[i] * u[i] + 2 * rr * ei) * du[i] - 2 * rr * ei * dwv[i];
            g += (-mu / (u[i] - UL[i]) + mu / (UU[i] - u[i])) * du[i];
            rv += rr * ei * ei;
        }
        gt += -2 * rv / t;
        double ss = z[l->ss + k];
        g += (-mu / (ss + SSB) + mu / (SSB - ss)) * d[l->ss + k];
    }
    if (!p->fixTime) {
        gt += (N + 1) * (0.5 + 2 * t) + (N + 1) * (-mu / (t - TL) + mu / (TU - t));
        g += gt * d[l->t];
    }
    return g;
}

/* values of all equality rows at z (layout pi | nu | yg | yo, signs as the Newton system uses them) */
static void constraint_values(const prob_t *p, const lay_t *l, const double *z, double *c) {
    int N = p->N, nOb = p->nOb, M = p->M;
    double t = z[l->t], q = t * p->Ts;
    for (int k = 0; k < N; k++) {
        double F[4]; const double *u = z + l->u + 2 * k;
        dyn_eval(p, z + l->x + 4 * k, u, t, F, NULL, NULL, NULL);
        for (int i = 0; i < 4; i++) c[4 * k + i] = z[l->x + 4 * (k + 1) + i] - F[i];
        c[(l->yg - l->pi) + k] = ((k ? u[-2] : 0) - u[0]) / q - z[l->ss + k];
    }
    for (int i = 0; i < 4; i++) c[(l->nu - l->pi) + i] = z[l->x + 4 * N + i] - p->xF[i];
    for (int k = 0; k <= N; k++) for (int j = 0; j < nOb; j++) {
        int bo = k * nOb + j;
        obs_rows(p, j, z + l->x + 4 * k, z + l->lam + k * M + p->roff[j], z + l->mu + 4 * bo, z[l->sl + bo], z[l->so + bo], c + (l->yo - l->pi) + 4 * bo, NULL);
    }
}

/* ------------------------------------------------------------------ the interior-point driver */
typedef struct { int status; int iters; int nreg; double obj, pinf, dinf, cinf, mu, t; } result_t;
enum { ST_OPTIMAL = 0, ST_USERLIMIT = 1, ST_ERROR = 2, ST_INFEASIBLE = 3 };

static void push_bounds(const prob_t *p, const lay_t *l, const opts_t *o, double *z) {
    int N = p->N, nOb = p->nOb, M = p->M;
#define PUSH2(v, lo, hi) { double pl = fmin(o->bound_push * fmax(1, fabs(lo)), o->bound_frac * ((hi) - (lo))); \
                           double pu = fmin(o->bound_push * fmax(1, fabs(hi)), o->bound_frac * ((hi) - (lo))); \
                           if ((v) < (lo) + pl) (v) = (lo) + pl; if ((v) > (hi) - pu) (v) = (hi) - pu; }
    for (int k = 1; k <= N; k++) for (int i = 0; i < 4; i++) if (i != 2) PUSH2(z[l->x + 4 * k + i], p->xl[i], p->xu[i]);
    for (int k = 0; k < N; k++) { for (int i = 0; i < 2; i++) PUSH2(z[l->u + 2 * k + i], UL[i], UU[i]); PUSH2(z[l->ss + k], -SSB, SSB); }
    if (!p->fixTime) PUSH2(z[l->t], TL, TU);
    for (int i = 0; i < M * (N + 1); i++) if (z[l->lam + i] < o->bound_push) z[l->lam + i] = o->bound_push;
    for (int i = 0; i < 4 * nOb * (N + 1); i++) if (z[l->mu + i] < o->bound_push) z[l->mu + i] = o->bound_push;
    for (int i = 0; i < nOb * (N + 1); i++) if (z[l->so + i] < o->bound_push) z[l->so + i] = o->bound_push;
    if (p->dist) for (int i = 0; i < nOb * (N + 1); i++) if (z[l->sl + i] < o->bound_push) z[l->sl + i] = o->bound_push;
#undef PUSH2
}

static void reset_bound_mults(const prob_t *p, const lay_t *l, const opts_t *o, double *z, double mu) {
    int N = p->N, nOb = p->nOb, M = p->M; double ks = o->kappa_sigma;
#define CL(zz, dist) { double lo = mu / (ks * (dist)), hi = ks * mu / (dist); if ((zz) < lo) (zz) = lo; if ((zz) > hi) (zz) = hi; }
    for (int k = 1; k <= N; k++) for (int i = 0; i < 4; i++) if (i != 2) { int idx = 4 * k + i; CL(z[l->zxL + idx], z[l->x + idx] - p->xl[i]); CL(z[l->zxU + idx], p->xu[i] - z[l->x + idx]); }
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < 2; i++) { int idx = 2 * k + i; CL(z[l->zuL + idx], z[l->u + idx] - UL[i]); CL(z[l->zuU + idx], UU[i] - z[l->u + idx]); }
        CL(z[l->zssL + k], z[l->ss + k] + SSB); CL(z[l->zssU + k], SSB - z[l->ss + k]);
    }
    if (!p->fixTime) { CL(z[l->ztL], z[l->t] - TL); CL(z[l->ztU], TU - z[l->t]); }
    for (int i = 0; i < M * (N + 1); i++) CL(z[l->zlam + i], z[l->lam + i]);
    for (int i = 0; i < 4 * nOb * (N + 1); i++) CL(z[l->zmu + i], z[l->mu + i]);
    for (int i = 0; i < nOb * (N + 1); i++) CL(z[l->zso + i], z[l->so + i]);
    if (p->dist) for (int i = 0; i < nOb * (N + 1); i++) CL(z[l->zs1 + i], z[l->sl + i]);
#undef CL
}

/* Block feasibility restoration: a stand-in for what IPOPT's restoration phase does for this model's one structural degeneracy (opts.restoration; the quadcopter oracle's
 * restore_blocks is its sibling).  DualMultWS (DualMultWS.jl:52-73) returns lambda = mu = 0 for a pose whose car rectangle touches or penetrates the obstacle -- the distance
 * is 0 and the dual of |A'lam| <= 1 is free to vanish -- and the signed-distance NLP started there has |A'lam|^2 = 0 against the EQUALITY |A'lam|^2 == 1
 * (ParkingSignedDist.jl:196) with a vanishing row gradient 2 A A'lam: a rank-deficient start the interior point does not leave (the multipliers stay near 0, the separation
 * row cannot grip the pose).  A block (stage k, obstacle j) with |A'lam|^2 < 1/4 is DEGENERATE; it gets the feasible dual of the obstacle's best edge instead:
 *     lam = e_s,  s = argmax_i d_i,  d_i = a_i . c - b_i - (g_1 |a_i . e_psi| + g_2 |a_i . e_perp|)      (rows of unit length; c: centre of the car rectangle)
 *     mu = the non-negative split of -R'a_s: (max(-e1, 0), max(-e2, 0), max(e1, 0), max(e2, 0)),  e = R'a_s
 * i.e. |A'lam| = 1, G'mu + R'A'lam = 0 hold exactly and the separation row takes the value d_s (the signed distance along that edge normal: negative when the pose penetrates,
 * absorbed by the row's penalised slack sl as the reference intends).  mid = 1 (inside a solve): the other multipliers take the bound push, the row's slack so its value,
 * the block's bound multipliers 1 and its equality multipliers 0.  Returns the number of blocks repaired. */
static int restore_blocks(const prob_t *p, const lay_t *l, const opts_t *o, double *z, int mid) {
    int N = p->N, nOb = p->nOb, M = p->M, nrep = 0;
    for (int k = 0; k <= N; k++) for (int j = 0; j < nOb; j++) {
        const double *Aj = p->A + 2 * p->roff[j], *bj = p->b + p->roff[j]; const int v = p->vOb[j], bo = k * nOb + j;
        double *lam = z + l->lam + k * M + p->roff[j], *mu = z + l->mu + 4 * bo; const double *x = z + l->x + 4 * k;
        double p1 = 0, p2 = 0;
        for (int i = 0; i < v; i++) { p1 += Aj[2 * i] * lam[i]; p2 += Aj[2 * i + 1] * lam[i]; }
        if (p1 * p1 + p2 * p2 >= 0.25) continue;
        const double cs = cos(x[2]), sn = sin(x[2]), cx = x[0] + cs * p->off, cy = x[1] + sn * p->off;
        double best = -1e300, be1 = 0, be2 = 0; int s = 0;
        for (int i = 0; i < v; i++) {
            const double a1 = Aj[2 * i], a2 = Aj[2 * i + 1], e1 = cs * a1 + sn * a2, e2 = -sn * a1 + cs * a2;
            const double d = a1 * cx + a2 * cy - bj[i] - (p->g[0] * fabs(e1) + p->g[1] * fabs(e2));
            if (d > best) { best = d; s = i; be1 = e1; be2 = e2; }
        }
        const double lo = mid ? o->bound_push : 0.0;
        for (int i = 0; i < v; i++) lam[i] = i == s ? 1.0 : lo;
        mu[0] = fmax(-be1, lo); mu[1] = fmax(-be2, lo); mu[2] = fmax(be1, lo); mu[3] = fmax(be2, lo);
        if (mid) {
            double c[4];
            obs_rows(p, j, x, lam, mu, z[l->sl + bo], 0.0, c, NULL);
            if (p->dist) { z[l->sl + bo] = fmax(-(c[0] - z[l->sl + bo]), o->bound_push); z[l->zs1 + bo] = 1.0; }
            z[l->so + bo] = fmax(c[3], o->bound_push); z[l->zso + bo] = 1.0;
            for (int i = 0; i < v; i++) z[l->zlam + k * M + p->roff[j] + i] = 1.0;
            for (int i = 0; i < 4; i++) { z[l->zmu + 4 * bo + i] = 1.0; z[l->yo + 4 * bo + i] = 0.0; }
        }
        nrep++;
    }
    return nrep;
}
#define MAX_RESTORE 3      /* restorations per attempt */

#define FILT_MAX 512

static void ipm_solve(const prob_t *p, const lay_t *l, const opts_t *o, double *z, result_t *res) {
    int N = p->N, nOb = p->nOb, M = p->M;
    kkt_t *K = kkt_alloc(p, l);
    double *d = xcalloc(l->len, sizeof(double)), *zt = xcalloc(l->len, sizeof(double)), *dsoc = xcalloc(l->len, sizeof(double));
    double *csoc = xcalloc(l->zxL - l->pi, sizeof(double)), *ctr = xcalloc(l->zxL - l->pi, sizeof(double));
    int max_soc = o->max_soc, nsoc = 0, nsoc_acc = 0, recalc_y = o->recalc_y, nrecalc = 0;
    double mu = o->mu_init, tau = fmax(o->tau_min, 1 - mu), dw_last = 0;
    double filt[FILT_MAX][2]; int nf = 0;
    /* initial point: x_0 = x0, x_N stays at the warm start (pulled to xF by the Newton step), bounds pushed */
    for (int i = 0; i < 4; i++) z[l->x + i] = p->x0[i];
    z[l->t] = p->fixTime ? 1.0 : z[l->t];
    int nrest = 0;
    if (o->restoration == 1) restore_blocks(p, l, o, z, 0);      /* degenerate blocks of the warm start (DualMultWS at a touching / penetrating pose): before the slacks take their values */
    /* slacks take the row values (ParkingSignedDist.jl gives them no start; IPOPT uses s = d(x)) */
    {
        double q = z[l->t] * p->Ts;
        for (int k = 0; k < N; k++) z[l->ss + k] = ((k ? z[l->u + 2 * (k - 1)] : 0) - z[l->u + 2 * k]) / q;
        for (int k = 0; k <= N; k++) for (int j = 0; j < nOb; j++) {
            double c[4]; int bo = k * nOb + j;
            if (p->dist) z[l->sl + bo] = 0.0;
            obs_rows(p, j, z + l->x + 4 * k, z + l->lam + k * M + p->roff[j], z + l->mu + 4 * bo, z[l->sl + bo], 0.0, c, NULL);
            z[l->so + bo] = c[3];
            if (p->dist) z[l->sl + bo] = -c[0];         /* slack of |A'lam|^2 <= 1 takes the row value */
        }
    }
    push_bounds(p, l, o, z);
    for (int i = l->nprimal; i < l->len; i++) z[i] = 0;
    for (int i = l->zxL; i < l->len; i++) z[i] = 1.0;
    if (o->lsq_init) {
        /* least-squares multipliers: the same structured solve with H := I */
        kkt_assemble(K, z, 0.0, 0, 0, 1);      /* (lsq mode: the right-hand side is the z-form gradient, bound2) */
        stage_dual_inf(K, z);
        if (kkt_solve(K, z, 0.0, 0, 0.0, 1, d)) {
            double ymax = 0;
            for (int i = l->pi; i < l->zxL; i++) { double a = fabs(z[i] + d[i]); if (a > ymax || a != a) ymax = a; }
            if (ymax <= 1e3 && ymax == ymax) for (int i = l->pi; i < l->zxL; i++) z[i] += d[i];
        }
    }
    double f, th, thinf;
    eval_f_theta(p, l, z, &f, &th, &thinf);
    double th_min = 1e-4 * fmax(1, th), th_max = 1e4 * fmax(1, th);
    int it = 0, status = ST_USERLIMIT, nreg = 0;
    double Emu_dinf = 0, Emu_pinf = 0;
    for (;;) {
        /* optimality error at mu=0 and at mu */
        kkt_assemble(K, z, mu, 0, 0, 0);
        double dinf = fmax(K->dinf, stage_dual_inf(K, z)), pinf = K->pinf, cinf0 = K->cinf_mu0;
        double sd = fmax(o->s_max, (K->sumy + K->sumz) / (K->nm + K->nb)) / o->s_max;
        double sc = fmax(o->s_max, K->sumz / K->nb) / o->s_max;
        double E0 = fmax(dinf / sd, fmax(pinf, cinf0 / sc));
        eval_f_theta(p, l, z, &f, &th, &thinf);
        if (o->verbose) printf("it %3d f=% .8e pinf=%.2e dinf=%.2e cinf=%.2e mu=%.1e dw=%.1e t=%.4f\n", it, f, pinf, dinf, cinf0, mu, dw_last, z[l->t]);
        Emu_dinf = dinf; Emu_pinf = pinf;
        if (E0 <= o->tol && pinf <= o->constr_viol_tol && dinf <= o->dual_inf_tol && cinf0 <= o->compl_inf_tol) { status = ST_OPTIMAL; break; }
        if (it >= o->max_iter) { status = ST_USERLIMIT; break; }
        if (!(f == f) || !(pinf == pinf)) { status = ST_ERROR; break; }
        /* barrier update (complementarity error at mu: |v z - mu|) */
        for (;;) {
            /* cinf(mu): recompute cheaply from products */
            double cm = 0;
#define CM(val, zz) { double c_ = fabs((val) * (zz) - mu); if (c_ > cm) cm = c_; }
            for (int k = 1; k <= N; k++) for (int i = 0; i < 4; i++) if (i != 2) { int idx = 4 * k + i; CM(z[l->x + idx] - p->xl[i], z[l->zxL + idx]); CM(p->xu[i] - z[l->x + idx], z[l->zxU + idx]); }
            for (int k = 0; k < N; k++) { for (int i = 0; i < 2; i++) { int idx = 2 * k + i; CM(z[l->u + idx] - UL[i], z[l->zuL + idx]); CM(UU[i] - z[l->u + idx], z[l->zuU + idx]); }
                CM(z[l->ss + k] + SSB, z[l->zssL + k]); CM(SSB - z[l->ss + k], z[l->zssU + k]); }
            if (!p->fixTime) { CM(z[l->t] - TL, z[l->ztL]); CM(TU - z[l->t], z[l->ztU]); }
            for (int i = 0; i < M * (N + 1); i++) CM(z[l->lam + i], z[l->zlam + i]);
            for (int i = 0; i < 4 * nOb * (N + 1); i++) CM(z[l->mu + i], z[l->zmu + i]);
            for (int i = 0; i < nOb * (N + 1); i++) CM(z[l->so + i], z[l->zso + i]);
            if (p->dist) for (int i = 0; i < nOb * (N + 1); i++) CM(z[l->sl + i], z[l->zs1 + i]);
#undef CM
            double Emu = fmax(dinf / sd, fmax(pinf, cm / sc));
            if (Emu <= o->kappa_eps * mu && mu > o->tol / 10) {
                mu = fmax(o->tol / 10, fmin(o->kappa_mu * mu, pow(mu, o->theta_mu)));
                tau = fmax(o->tau_min, 1 - mu); nf = 0;
            } else break;
        }
        /* search direction with inertia correction (IPOPT Algorithm IC) */
        double dw = 0, dc = o->dc_bar * pow(mu, o->kappa_c);
        int ok = 0;
        for (int tr = 0; tr < 60; tr++) {
            int a = kkt_assemble(K, z, mu, dw, dc, 0);
            stage_dual_inf(K, z);
            if (a) a = kkt_solve(K, z, mu, dc, o->rho_term, 0, d);
            if (a) { ok = 1; break; }
            nreg++;
            if (dw == 0) dw = dw_last == 0 ? o->dw0 : fmax(o->dw_min, o->kw_dec * dw_last);
            else dw *= (dw_last == 0 ? o->kw_inc0 : o->kw_inc);
            if (dw > o->dw_max) break;
        }
        /* where IPOPT would enter its restoration phase: repair the degenerate obstacle blocks (if there are any), restart the barrier, empty the filter */
#define RESTORE_AND_CONTINUE { nrest++; mu = o->mu_init; tau = fmax(o->tau_min, 1 - mu); nf = 0; dw_last = 0; eval_f_theta(p, l, z, &f, &th, &thinf); \
                               th_min = 1e-4 * fmax(1, th); th_max = 1e4 * fmax(1, th); continue; }
        if (!ok) { if (o->restoration && nrest < MAX_RESTORE && restore_blocks(p, l, o, z, 1) > 0) RESTORE_AND_CONTINUE; status = ST_ERROR; break; }
        if (dw > 0) dw_last = dw;
        double ap, az;
        frac_to_boundary(p, l, z, d, tau, &ap, &az);
        double phi = f - mu * barrier_terms(p, l, z);
        double gd = barrier_dir(p, l, z, d, mu);
        double amin;
        if (gd < 0) {
            amin = fmin(o->gamma_theta, o->gamma_phi * th / (-gd));
            if (th <= th_min) amin = fmin(amin, o->delta * pow(th, o->s_theta) / pow(-gd, o->s_phi));
        } else amin = o->gamma_theta;
        amin *= o->gamma_alpha;
        double alpha = ap; int acc = 0, ntrial = 0;
        while (alpha >= amin) {
            ntrial++;
            for (int i = 0; i < l->nprimal; i++) zt[i] = z[i] + alpha * d[i];
            double ft, tht, thi;
            eval_f_theta(p, l, zt, &ft, &tht, &thi);
            if (ft == ft && tht == tht && tht < th_max) {
                double pht = ft - mu * barrier_terms(p, l, zt);
                int okf = (pht == pht);
                for (int i = 0; i < nf && okf; i++) if (!(tht < filt[i][0] || pht < filt[i][1])) okf = 0;
                if (okf) {
                    int sw = gd < 0 && alpha * pow(-gd, o->s_phi) > o->delta * pow(th, o->s_theta);
                    int armijo = pht <= phi + o->eta_phi * alpha * gd;
                    if (th <= th_min && sw) { if (armijo) { acc = 1; break; } }
                    else if (tht <= (1 - o->gamma_theta) * th || pht <= phi - o->gamma_phi * th) {
                        acc = 1;
                        if (!(sw && armijo) && nf < FILT_MAX) { filt[nf][0] = (1 - o->gamma_theta) * th; filt[nf][1] = phi - o->gamma_phi * th; nf++; }
                        break;
                    }
                }
            }
            /* second-order correction (IPOPT A-5.5..A-5.9, max_soc = 4, kappa_soc = 0.99): only for the first trial step and only if it did not reduce theta */
            if (max_soc > 0 && ntrial == 1 && ft == ft && tht == tht && tht >= th) {
                int nc = l->zxL - l->pi; double th_old = 0, th_tr = tht, asoc = alpha;
                constraint_values(p, l, z, csoc);
                for (int ps = 0; ps < max_soc && !acc && (ps == 0 || th_tr <= 0.99 * th_old); ps++) {
                    th_old = th_tr;
                    constraint_values(p, l, zt, ctr);
                    for (int i = 0; i < nc; i++) csoc[i] = asoc * csoc[i] + ctr[i];
                    K->csoc = csoc;
                    int a = kkt_assemble(K, z, mu, dw, dc, 0);
                    stage_dual_inf(K, z);
                    if (a) a = kkt_solve(K, z, mu, dc, o->rho_term, 0, dsoc);
                    K->csoc = NULL;
                    if (!a) break;
                    double azs; frac_to_boundary(p, l, z, dsoc, tau, &asoc, &azs);
                    for (int i = 0; i < l->nprimal; i++) zt[i] = z[i] + asoc * dsoc[i];
                    eval_f_theta(p, l, zt, &ft, &tht, &thi);
                    nsoc++;
                    if (!(ft == ft && tht == tht)) break;
                    th_tr = tht;
                    if (tht < th_max) {
                        double pht = ft - mu * barrier_terms(p, l, zt);
                        int okf = (pht == pht);
                        for (int i = 0; i < nf && okf; i++) if (!(tht < filt[i][0] || pht < filt[i][1])) okf = 0;
                        if (okf) {
                            int sw = gd < 0 && alpha * pow(-gd, o->s_phi) > o->delta * pow(th, o->s_theta);
                            int armijo = pht <= phi + o->eta_phi * alpha * gd;
                            if (th <= th_min && sw) { if (armijo) acc = 1; }
                            else if (tht <= (1 - o->gamma_theta) * th || pht <= phi - o->gamma_phi * th) {
                                acc = 1;
                                if (!(sw && armijo) && nf < FILT_MAX) { filt[nf][0] = (1 - o->gamma_theta) * th; filt[nf][1] = phi - o->gamma_phi * th; nf++; }
                            }
                        }
                    }
                    if (acc) { memcpy(d, dsoc, sizeof(double) * l->len); alpha = asoc; az = azs; nsoc_acc++; }
                }
                if (acc) break;
            }
            alpha *= 0.5;
        }
        if (o->verbose > 1) printf("   ls: alpha_max %.3e accepted %.3e trials %d\n", ap, alpha, ntrial);
        if (!acc) { if (o->restoration && nrest < MAX_RESTORE && restore_blocks(p, l, o, z, 1) > 0) RESTORE_AND_CONTINUE; status = ST_ERROR; break; } /* IPOPT would enter restoration here */
#undef RESTORE_AND_CONTINUE
        for (int i = 0; i < l->nprimal; i++) z[i] += alpha * d[i];
        double ay = fmin(alpha, az); /* alpha_for_y = "min" (ParkingSignedDist.jl:41) */
        for (int i = l->pi; i < l->zxL; i++) z[i] += ay * d[i];
        for (int i = l->zxL; i < l->len; i++) z[i] += az * d[i];
        reset_bound_mults(p, l, o, z, mu);
        if (recalc_y) {   /* recalc_y = "yes" (ParkingSignedDist.jl:41), recalc_y_feas_tol = 1e-6: least-squares equality multipliers once the point is (nearly) feasible */
            double f2, th2, thi2;
            eval_f_theta(p, l, z, &f2, &th2, &thi2);
            if (thi2 < 1e-6) {
                kkt_assemble(K, z, 0.0, 0, 0, 1);      /* (lsq mode: every gradient in its z-form, the condensed blocks' corrections included) */
                stage_dual_inf(K, z);
                if (kkt_solve(K, z, 0.0, 0, 0.0, 1, dsoc)) {
                    int fin = 1;
                    for (int i = l->pi; i < l->zxL; i++) if (!(dsoc[i] == dsoc[i]) || fabs(dsoc[i]) > 1e300) fin = 0;
                    if (fin) { for (int i = l->pi; i < l->zxL; i++) z[i] += dsoc[i]; nrecalc++; }
                }
            }
        }
        it++;
    }
    eval_f_theta(p, l, z, &f, &th, &thinf);
    res->status = status; res->iters = it; res->nreg = nreg; res->obj = f; res->pinf = Emu_pinf; res->dinf = Emu_dinf;
    res->mu = mu; res->t = z[l->t]; res->cinf = K->cinf_mu0;
    if (getenv("OBCA_SOC_STAT")) fprintf(stderr, "soc %d accepted %d recalc_y %d iters %d status %d\n", nsoc, nsoc_acc, nrecalc, it, status);
    free(d); free(zt); free(dsoc); free(csoc); free(ctr); kkt_free(K);
}

/* ------------------------------------------------------------------ DualMultWS (DualMultWS.jl:29-86) */
/*
 * For a fixed pose the reference's model separates into (N+1)*nOb independent convex problems
 *     max  d = -g'mu + (A e - b)'lam     s.t. |A'lam|^2 <= 1,  G'mu + R'A'lam = 0,  lam,mu >= 0
 * (DualMultWS.jl:52-73; e = centre of the car rectangle).  Each is solved by a feasible-start primal-dual
 * path-following method (long step, sigma = 0.1) down to an average complementarity of 1e-9.
 */
static void dualws_one(int v, const double *Aj, const double *bj, const double g[4], double ex, double ey, double cs,
                       double sn, double *lam, double *mu, double *dout) {
    double Q[2][VMAX], cl[VMAX], amax = 0;
    for (int i = 0; i < v; i++) {
        double a1 = Aj[2 * i], a2 = Aj[2 * i + 1];
        Q[0][i] = cs * a1 + sn * a2; Q[1][i] = -sn * a1 + cs * a2;
        cl[i] = a1 * ex + a2 * ey - bj[i];
        double nr = sqrt(a1 * a1 + a2 * a2); if (nr > amax) amax = nr;
    }
    double zl[VMAX], zm[4], zh = 1, eta[2] = {0, 0};
    for (int i = 0; i < v; i++) { lam[i] = 0.5 / (v * fmax(amax, 1e-12)); zl[i] = 1; }
    {
        double q0 = 0, q1 = 0;
        for (int i = 0; i < v; i++) { q0 += Q[0][i] * lam[i]; q1 += Q[1][i] * lam[i]; }
        mu[0] = 1 + fmax(0, -q0); mu[2] = mu[0] + q0; mu[1] = 1 + fmax(0, -q1); mu[3] = mu[1] + q1;
        for (int i = 0; i < 4; i++) zm[i] = 1;
    }
    static const double Em[2][4] = {{1, 0, -1, 0}, {0, 1, 0, -1}};
    for (int it = 0; it < 60; it++) {
        double p1 = 0, p2 = 0;
        for (int i = 0; i < v; i++) { p1 += Aj[2 * i] * lam[i]; p2 += Aj[2 * i + 1] * lam[i]; }
        double h = 1 - p1 * p1 - p2 * p2;
        double gap = h * zh; for (int i = 0; i < v; i++) gap += lam[i] * zl[i]; for (int i = 0; i < 4; i++) gap += mu[i] * zm[i];
        double mbar = gap / (v + 5);
        /* dual residuals (gradient of  -d + eta'E - zl'lam - zm'mu - zh*h) */
        double gh[VMAX], rl[VMAX], rm[4], rmax = 0;
        for (int i = 0; i < v; i++) {
            gh[i] = -2 * (p1 * Aj[2 * i] + p2 * Aj[2 * i + 1]); /* dh/dlam */
            rl[i] = -cl[i] + Q[0][i] * eta[0] + Q[1][i] * eta[1] - zl[i] - zh * gh[i];
            if (fabs(rl[i]) > rmax) rmax = fabs(rl[i]);
        }
        for (int i = 0; i < 4; i++) { rm[i] = g[i] + Em[0][i] * eta[0] + Em[1][i] * eta[1] - zm[i]; if (fabs(rm[i]) > rmax) rmax = fabs(rm[i]); }
        if (mbar < 1e-9 && rmax < 1e-9) break;
        double sig = 0.1, mt = sig * mbar;
        /* reduced system in (dlam, dmu, deta) */
        double Hl[VMAX * VMAX], bl[VMAX], Dm[4], bm[4];
        for (int i = 0; i < v; i++) {
            for (int j = 0; j < v; j++)
                Hl[i * v + j] = zh * 2 * (Aj[2 * i] * Aj[2 * j] + Aj[2 * i + 1] * Aj[2 * j + 1]) + (zh / h) * gh[i] * gh[j];
            Hl[i * v + i] += zl[i] / lam[i];
            bl[i] = -(rl[i] + zl[i] - mt / lam[i] + (zh - mt / h) * gh[i]);
        }
        for (int i = 0; i < 4; i++) { Dm[i] = zm[i] / mu[i]; bm[i] = -(rm[i] + zm[i] - mt / mu[i]); }
        if (ldl_n(v, Hl) != 0) break;
        double HiQ[2][VMAX], Hib[VMAX];
        for (int r = 0; r < 2; r++) { for (int i = 0; i < v; i++) HiQ[r][i] = Q[r][i]; ldl_solve(v, Hl, HiQ[r]); }
        for (int i = 0; i < v; i++) Hib[i] = bl[i];
        ldl_solve(v, Hl, Hib);
        double S[2][2] = {{0, 0}, {0, 0}}, rs[2] = {0, 0};
        for (int r = 0; r < 2; r++) {
            for (int s_ = 0; s_ < 2; s_++) {
                double a = 0; for (int i = 0; i < v; i++) a += Q[r][i] * HiQ[s_][i];
                for (int i = 0; i < 4; i++) a += Em[r][i] * Em[s_][i] / Dm[i];
                S[r][s_] = a;
            }
            double a = 0; for (int i = 0; i < v; i++) a += Q[r][i] * Hib[i];
            for (int i = 0; i < 4; i++) a += Em[r][i] * bm[i] / Dm[i];
            rs[r] = a; /* E H^{-1} b  (equality residual is zero: feasible start) */
        }
        double Lc[3], de[2] = {rs[0], rs[1]};
        if (!chol2(S, Lc)) break;
        chol2_solve(Lc, de);
        double dl[VMAX], dm[4];
        for (int i = 0; i < v; i++) dl[i] = Hib[i] - HiQ[0][i] * de[0] - HiQ[1][i] * de[1];
        for (int i = 0; i < 4; i++) dm[i] = (bm[i] - Em[0][i] * de[0] - Em[1][i] * de[1]) / Dm[i];
        double dzl[VMAX], dzm[4], ghd = 0;
        for (int i = 0; i < v; i++) { dzl[i] = mt / lam[i] - zl[i] - zl[i] / lam[i] * dl[i]; ghd += gh[i] * dl[i]; }
        for (int i = 0; i < 4; i++) dzm[i] = mt / mu[i] - zm[i] - zm[i] / mu[i] * dm[i];
        double dzh = mt / h - zh - zh / h * ghd;
        double a = 1, tb = 0.995, c;
        for (int i = 0; i < v; i++) { c = ftb_one(lam[i], dl[i], tb); if (c < a) a = c; c = ftb_one(zl[i], dzl[i], tb); if (c < a) a = c; }
        for (int i = 0; i < 4; i++) { c = ftb_one(mu[i], dm[i], tb); if (c < a) a = c; c = ftb_one(zm[i], dzm[i], tb); if (c < a) a = c; }
        c = ftb_one(zh, dzh, tb); if (c < a) a = c;
        /* keep h > (1-tb) h along the quadratic */
        for (int bt = 0; bt < 60; bt++) {
            double q1 = 0, q2 = 0;
            for (int i = 0; i < v; i++) { q1 += Aj[2 * i] * (lam[i] + a * dl[i]); q2 += Aj[2 * i + 1] * (lam[i] + a * dl[i]); }
            if (1 - q1 * q1 - q2 * q2 >= (1 - tb) * h) break;
            a *= 0.7;
        }
        for (int i = 0; i < v; i++) { lam[i] += a * dl[i]; zl[i] += a * dzl[i]; }
        for (int i = 0; i < 4; i++) { mu[i] += a * dm[i]; zm[i] += a * dzm[i]; }
        zh += a * dzh; eta[0] += a * de[0]; eta[1] += a * de[1];
    }
    double dv = 0;
    for (int i = 0; i < v; i++) dv += cl[i] * lam[i];
    for (int i = 0; i < 4; i++) dv -= g[i] * mu[i];
    *dout = dv;
}

/* ------------------------------------------------------------------ C entry points (called through ctypes by tests) */
static void setup_prob(prob_t *p, int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                       const double *x0, const double *xF, int nOb, const int *vOb, const double *A, const double *b,
                       const double *rx, const double *ry, const double *ryaw, int unit_rows) {
    memset(p, 0, sizeof *p);
    p->N = N; p->Ts = Ts; p->L = L; p->fixTime = fixTime; p->nOb = nOb;
    double W_ev = ego[1] + ego[3], L_ev = ego[0] + ego[2];          /* ParkingSignedDist.jl:182-188 */
    p->g[0] = L_ev / 2; p->g[1] = W_ev / 2; p->g[2] = L_ev / 2; p->g[3] = W_ev / 2;
    p->off = (ego[0] + ego[2]) / 2 - ego[2];
    memcpy(p->XYb, XYb, 4 * sizeof(double));
    if (x0) memcpy(p->x0, x0, 4 * sizeof(double));
    if (xF) memcpy(p->xF, xF, 4 * sizeof(double));
    p->roff[0] = 0;
    for (int j = 0; j < nOb; j++) { p->vOb[j] = vOb[j]; p->roff[j + 1] = p->roff[j] + vOb[j]; }
    p->M = p->roff[nOb];
    /* The solve runs on unit-length half-space rows: a_i / |a_i|, b_i / |a_i| describe the same obstacle, lambda_i scales with |a_i| (A'lam and b'lam do not
     * change) and is handed back in the caller's scaling.  obstHrep.jl:57-86 leaves the rows of a sloped edge unnormalised ([-s 1], |a| up to 1e3 for a steep one);
     * IPOPT's default gradient-based scaling stands between such rows and the reference's solves, nothing did here (config-5 instances with |a| > 100 failed). */
    for (int r = 0; r < p->M; r++) {
        double n = unit_rows ? hypot(A[2 * r], A[2 * r + 1]) : 1.0;
        if (!(n > 0)) n = 1.0;
        p->rn[r] = n; p->An[2 * r] = A[2 * r] / n; p->An[2 * r + 1] = A[2 * r + 1] / n; p->bn[r] = b[r] / n;
    }
    p->A = p->An; p->b = p->bn; p->rx = rx; p->ry = ry; p->ryaw = ryaw;
    p->xl[0] = XYb[0]; p->xu[0] = XYb[1]; p->xl[1] = XYb[2]; p->xu[1] = XYb[3]; p->xl[2] = -1e300; p->xu[2] = 1e300;
    p->xl[3] = -1; p->xu[3] = 2;                                      /* :104-106 */
}

/* lWS: (N+1) x M, nWS: (N+1) x 4nOb (row-major = the transposed shapes DualMultWS.jl:81-84 returns), d: (N+1) x nOb */
int obca_oracle_dualmult_ws(int N, int nOb, const int *vOb, const double *A, const double *b, const double *rx,
                            const double *ry, const double *ryaw, const double ego[4], double *lWS, double *nWS, double *dd) {
    prob_t p; double XYb[4] = {0, 0, 0, 0};
    if (nOb > NOBMAX) return -1;
    for (int j = 0; j < nOb; j++) if (vOb[j] > VMAX || vOb[j] < 1) return -1;
    setup_prob(&p, N, 1, 1, ego, XYb, 0, NULL, NULL, nOb, vOb, A, b, rx, ry, ryaw, 1);
    for (int k = 0; k <= N; k++) {
        double cs = cos(ryaw[k]), sn = sin(ryaw[k]);
        for (int j = 0; j < nOb; j++)
            dualws_one(p.vOb[j], p.A + 2 * p.roff[j], p.b + p.roff[j], p.g, rx[k] + cs * p.off, ry[k] + sn * p.off, cs, sn,
                       lWS + k * p.M + p.roff[j], nWS + 4 * (k * nOb + j), dd + k * nOb + j);
        for (int r = 0; r < p.M; r++) lWS[k * p.M + r] /= p.rn[r];          /* back to the caller's row scaling */
    }
    return 0;
}

/*
 * The reference's own acceptance test, restated with its quirks (ParkingConstraints.jl:29-149): in variable-time mode only the
 * speed row of the dynamics survives in c3 (:76-79), the obstacle rows c6 are overwritten per obstacle so only the LAST obstacle counts
 * (:108-130), the c6[4] row ignores the slack, the steering-rate row divides by timeScale[1] only (:88).  Returns 1 if every class is
 * <= 5e-5 (:133-139).  sd = 1: signed-distance variant (|A'lam|^2 == 1), sd = 0: distance variant (<= 1).
 */
static int ref_constraints(const prob_t *p, const lay_t *l, const double *z, int sd) {
    int N = p->N, nOb = p->nOb, M = p->M; const double tol = 5e-5;
    double t = p->fixTime ? 1.0 : z[l->t];
    double c0 = -1e300, c2 = 0, c3 = 0, c5 = 0, c6 = -1e300, m0 = 0, m1 = 0, lmin = 1e300, nmin = 1e300;
    for (int k = 0; k < N; k++) { m0 = fmax(m0, fabs(z[l->u + 2 * k])); m1 = fmax(m1, fabs(z[l->u + 2 * k + 1])); }
    for (int i = 0; i < M * (N + 1); i++) lmin = fmin(lmin, z[l->lam + i]);
    for (int i = 0; i < 4 * nOb * (N + 1); i++) nmin = fmin(nmin, z[l->mu + i]);
    c0 = fmax(fmax(m0 - 0.6, m1 - 0.4), fmax(fabs(t - 1) - 0.2, fmax(-lmin, -nmin)));
    for (int i = 0; i < 4; i++) c2 = fmax(c2, fabs(z[l->x + 4 * N + i] - p->xF[i]));
    for (int k = 0; k < N; k++) {
        double F[4]; dyn_eval(p, z + l->x + 4 * k, z + l->u + 2 * k, t, F, NULL, NULL, NULL);
        if (p->fixTime) { for (int i = 0; i < 4; i++) c3 = fmax(c3, fabs(z[l->x + 4 * (k + 1) + i] - F[i])); }
        else c3 = fmax(c3, fabs(z[l->x + 4 * (k + 1) + 3] - F[3]));
        double prev = k ? z[l->u + 2 * (k - 1)] : 0.0;
        c5 = fmax(c5, fabs(z[l->u + 2 * k] - prev) / (t * p->Ts));
    }
    c5 -= 0.6;
    int j = nOb - 1;                                                   /* only the last obstacle survives the overwrite */
    for (int k = 0; k <= N && j >= 0; k++) {
        double c[4]; const double *x = z + l->x + 4 * k;
        int dist_keep = p->dist; ((prob_t *)p)->dist = 0;              /* evaluate the plain rows: no slack in either variant (:127-128) */
        obs_rows(p, j, x, z + l->lam + k * M + p->roff[j], z + l->mu + 4 * (k * nOb + j), 0.0, 0.0, c, NULL);
        ((prob_t *)p)->dist = dist_keep;
        c6 = fmax(c6, sd ? fabs(c[0] + 1) - 1 : c[0]);                /* abs(|p|^2) - 1 (sd) or |p|^2 - 1 */
        c6 = fmax(c6, fmax(fabs(c[1]), fabs(c[2])));
        c6 = fmax(c6, -c[3]);                                          /* -(row) + dmin <= 0 ; obs_rows already subtracts DMIN */
    }
    return c0 <= tol && c2 <= tol && c3 <= tol && c5 <= tol && (nOb == 0 || c6 <= tol);
}

/*
 * One parking solve.  Array conventions (all fp64, "stage-contiguous" = the reference's column-major xp etc.):
 *   xWS 4 x (N+1) stage-contiguous, uWS 2 x N, lWS M x (N+1) stage-contiguous, nWS 4nOb x (N+1)
 *   outputs xp 4(N+1), up 2N, timeScale (N+1), lp M(N+1), np 4nOb(N+1), slp nOb(N+1)
 *   info[8] = {status, iterations, objective, pinf, dinf, mu, #regularisations, t}
 * dist = 0: ParkingSignedDist.  exitflag follows ParkingSignedDist.jl:256-290: Optimal -> 1; Error/UserLimit -> one retry from the last
 *           iterate; if that fails too the reference's feasibility check decides (feasible -> 1).
 * dist = 1: ParkingDist.  exitflag follows ParkingDist.jl:245-289: Optimal -> 1; otherwise the feasibility check runs first (feasible -> 1),
 *           else one retry; after a failed retry the check is INVERTED in the reference (infeasible -> 1) -- reproduced.
 */
static double *g_zfull = NULL;      /* test hook: when set, the full primal-dual iterate (layout of make_layout) is copied out after the solve */
static int parking_solve(int dist, int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                         const double *x0, const double *xF, int nOb, const int *vOb, const double *A,
                         const double *b, const double *rx, const double *ry, const double *ryaw,
                         const double *xWS, const double *uWS, const double *lWS, const double *nWS,
                         const opts_t *opt, double *xp, double *up, double *tsp, double *lp, double *np,
                         double *slp, int *exitflag, double *info) {
    prob_t p; lay_t l; opts_t o;
    if (nOb > NOBMAX) return -1;
    for (int j = 0; j < nOb; j++) if (vOb[j] > VMAX || vOb[j] < 1) return -1;
    if (opt) o = *opt; else obca_oracle_default_opts(&o);
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, 1);
    p.dist = dist;
    make_layout(&p, &l);
    double *z = xcalloc(l.len, sizeof(double));
    memcpy(z + l.x, xWS, sizeof(double) * 4 * (N + 1));
    memcpy(z + l.u, uWS, sizeof(double) * 2 * N);
    z[l.t] = 1.0;                                                      /* ParkingSignedDist.jl:214 */
    for (int k = 0; k <= N; k++) for (int q = 0; q < p.M; q++) z[l.lam + k * p.M + q] = lWS[k * p.M + q] * p.rn[q];      /* dual warm start: caller's row scaling -> unit rows */
    memcpy(z + l.mu, nWS, sizeof(double) * 4 * nOb * (N + 1));
    result_t r;
    ipm_solve(&p, &l, &o, z, &r);
    int ef = (r.status == ST_OPTIMAL);
    int iters = r.iters;
    int retry = !ef && (r.status == ST_ERROR || r.status == ST_USERLIMIT);
    if (retry && dist && ref_constraints(&p, &l, z, 0)) { ef = 1; retry = 0; }       /* ParkingDist.jl:259-260, :284-285 */
    if (retry) {
        /* second attempt from the last iterate (ParkingSignedDist.jl:259-263) */
        double *z2 = xcalloc(l.len, sizeof(double));
        memcpy(z2, z, sizeof(double) * l.nprimal);
        result_t r2;
        ipm_solve(&p, &l, &o, z2, &r2);
        iters += r2.iters;
        if (r2.status == ST_OPTIMAL) { ef = 1; memcpy(z, z2, sizeof(double) * l.len); r = r2; }
        else {
            if (r2.obj == r2.obj) { memcpy(z, z2, sizeof(double) * l.len); r = r2; }
            int feas = ref_constraints(&p, &l, z, dist ? 0 : 1);
            ef = dist ? !feas : feas;                                   /* ParkingSignedDist.jl:278-283 / ParkingDist.jl:277-282 (inverted, Q6) */
        }
        free(z2);
    }
    memcpy(xp, z + l.x, sizeof(double) * 4 * (N + 1));
    memcpy(up, z + l.u, sizeof(double) * 2 * N);
    for (int k = 0; k <= N; k++) tsp[k] = fixTime ? 1.0 : z[l.t];
    for (int k = 0; k <= N; k++) for (int q = 0; q < p.M; q++) lp[k * p.M + q] = z[l.lam + k * p.M + q] / p.rn[q];      /* back to the caller's row scaling */
    memcpy(np, z + l.mu, sizeof(double) * 4 * nOb * (N + 1));
    if (slp) memcpy(slp, z + l.sl, sizeof(double) * nOb * (N + 1));
    *exitflag = ef;
    if (info) { info[0] = r.status; info[1] = iters; info[2] = r.obj; info[3] = r.pinf; info[4] = r.dinf; info[5] = r.mu; info[6] = r.nreg; info[7] = r.t; }
    if (g_zfull) {
        memcpy(g_zfull, z, sizeof(double) * l.len);
        for (int k = 0; k <= N; k++) for (int q = 0; q < p.M; q++) { g_zfull[l.lam + k * p.M + q] /= p.rn[q]; g_zfull[l.zlam + k * p.M + q] *= p.rn[q]; }
    }
    free(z);
    return 0;
}
/* the same solve, additionally returning the full primal-dual iterate x,u,t,lam,mu,sl,so,ss | pi,nu,yg,yo | bound multipliers (for the independent
 * optimality certificate of tests/golden/make_kkt_pin.py: multipliers of the oracle, derivatives of oracle/nlp_ref.py) */
int obca_oracle_parking_signed_dist_full(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                                         const double *x0, const double *xF, int nOb, const int *vOb, const double *A,
                                         const double *b, const double *rx, const double *ry, const double *ryaw,
                                         const double *xWS, const double *uWS, const double *lWS, const double *nWS,
                                         const opts_t *opt, double *xp, double *up, double *tsp, double *lp, double *np,
                                         double *slp, int *exitflag, double *info, double *zfull) {
    g_zfull = zfull;
    int rc = parking_solve(0, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, opt, xp, up, tsp, lp, np, slp, exitflag, info);
    g_zfull = NULL;
    return rc;
}
int obca_oracle_parking_signed_dist(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                                    const double *x0, const double *xF, int nOb, const int *vOb, const double *A,
                                    const double *b, const double *rx, const double *ry, const double *ryaw,
                                    const double *xWS, const double *uWS, const double *lWS, const double *nWS,
                                    const opts_t *opt, double *xp, double *up, double *tsp, double *lp, double *np,
                                    double *slp, int *exitflag, double *info) {
    return parking_solve(0, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, opt, xp, up, tsp, lp, np, slp, exitflag, info);
}
/* ParkingDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)  (ParkingDist.jl:29); slp returns the norm-row slack */
int obca_oracle_parking_dist(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime,
                             const double *x0, const double *xF, int nOb, const int *vOb, const double *A,
                             const double *b, const double *rx, const double *ry, const double *ryaw,
                             const double *xWS, const double *uWS, const double *lWS, const double *nWS,
                             const opts_t *opt, double *xp, double *up, double *tsp, double *lp, double *np,
                             double *slp, int *exitflag, double *info) {
    return parking_solve(1, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, xWS, uWS, lWS, nWS, opt, xp, up, tsp, lp, np, slp, exitflag, info);
}
/* test hook: the acceptance test on a returned solution (same argument conventions) */
int obca_oracle_ref_constraints(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime, const double *x0,
                                const double *xF, int nOb, const int *vOb, const double *A, const double *b, const double *xp,
                                const double *up, double t, const double *lp, const double *np, int sd) {
    prob_t p; lay_t l;
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, NULL, NULL, NULL, 0);
    make_layout(&p, &l);
    double *z = xcalloc(l.len, sizeof(double));
    memcpy(z + l.x, xp, sizeof(double) * 4 * (N + 1)); memcpy(z + l.u, up, sizeof(double) * 2 * N); z[l.t] = t;
    memcpy(z + l.lam, lp, sizeof(double) * p.M * (N + 1)); memcpy(z + l.mu, np, sizeof(double) * 4 * nOb * (N + 1));
    int r = ref_constraints(&p, &l, z, sd);
    free(z);
    return r;
}

int obca_oracle_eval(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime, const double *x0,
                     const double *xF, int nOb, const int *vOb, const double *A, const double *b, const double *rx,
                     const double *ry, const double *ryaw, const double *zin /* packed primal in oracle layout */,
                     double *f, double *theta1, double *thetainf) {
    prob_t p; lay_t l;
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, 0);
    make_layout(&p, &l);
    eval_f_theta(&p, &l, zin, f, theta1, thetainf);
    return l.len;
}

int obca_oracle_layout(int N, int nOb, const int *vOb, int *out /* 26 ints */) {
    prob_t p; lay_t l; memset(&p, 0, sizeof p);
    p.N = N; p.nOb = nOb; p.M = 0; for (int j = 0; j < nOb; j++) p.M += vOb[j];
    make_layout(&p, &l);
    memcpy(out, &l, sizeof l);
    return (int)(sizeof l / sizeof(int));
}

/* one regularised Newton direction at a full primal-dual point z (oracle layout); returns inertia-ok flag */
/* the Newton system of z with GIVEN constraint values on the right-hand side (what a second-order correction solves: csoc in the layout pi | nu | yg | yo);
 * tests/test_oracle_cpu.py pins it against a dense solve with autograd derivatives */
int obca_oracle_newton_soc(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime, const double *x0,
                           const double *xF, int nOb, const int *vOb, const double *A, const double *b, const double *rx,
                           const double *ry, const double *ryaw, const double *z, double mu, double dw, double dc, double rho,
                           const double *csoc, double *d, int dist) {
    prob_t p; lay_t l;
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, 0);
    p.dist = dist;
    make_layout(&p, &l);
    kkt_t *K = kkt_alloc(&p, &l);
    K->csoc = csoc;
    int ok = kkt_assemble(K, z, mu, dw, dc, 0);
    stage_dual_inf(K, z);
    if (ok) ok = kkt_solve(K, z, mu, dc, rho, 0, d);
    K->csoc = NULL;
    kkt_free(K);
    return ok;
}

/* the least-squares multiplier step at z (what recalc_y and lsq_init take): d[pi .. zxL) = the increment of the equality multipliers; tests/test_oracle_cpu.py pins it against a
 * dense solve of [I J'; J 0] with autograd derivatives */
int obca_oracle_lsq_multipliers(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime, const double *x0,
                                const double *xF, int nOb, const int *vOb, const double *A, const double *b, const double *rx,
                                const double *ry, const double *ryaw, const double *z, double *d, int dist) {
    prob_t p; lay_t l;
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, 0);
    p.dist = dist;
    make_layout(&p, &l);
    kkt_t *K = kkt_alloc(&p, &l);
    kkt_assemble(K, z, 0.0, 0, 0, 1);
    stage_dual_inf(K, z);
    int ok = kkt_solve(K, z, 0.0, 0, 0.0, 1, d);
    kkt_free(K);
    return ok;
}

int obca_oracle_newton(int N, double Ts, double L, const double ego[4], const double XYb[4], int fixTime, const double *x0,
                       const double *xF, int nOb, const int *vOb, const double *A, const double *b, const double *rx,
                       const double *ry, const double *ryaw, const double *z, double mu, double dw, double dc, double rho,
                       double *d, double *errs /* dinf,pinf,cinf */, int dist) {
    prob_t p; lay_t l;
    setup_prob(&p, N, Ts, L, ego, XYb, fixTime, x0, xF, nOb, vOb, A, b, rx, ry, ryaw, 0);
    p.dist = dist;
    make_layout(&p, &l);
    kkt_t *K = kkt_alloc(&p, &l);
    int ok = kkt_assemble(K, z, mu, dw, dc, 0);
    double sdi = stage_dual_inf(K, z);
    if (errs) { errs[0] = fmax(K->dinf, sdi); errs[1] = K->pinf; errs[2] = K->cinf_mu0; }
    if (ok) ok = kkt_solve(K, z, mu, dc, rho, 0, d);
    kkt_free(K);
    return ok;
}
