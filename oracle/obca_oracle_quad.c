/*
 * obca_oracle_quad.c -- CPU ORACLE for the quadcopter signed-distance path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of /root/reference/QuadcopterNavigation/QuadcopterSignedDist.jl:25-300
 *   variables :34-49, constants :51-62, objective :65-71, bounds :74-105, boundary :108-134, Euler dynamics :136-158,
 *   box-obstacle rows :162-197, initial point :199-210, exit flag :227-288, returned arrays :290-298
 * solved with the same interior-point algorithm as oracle/obca_oracle.c (IPOPT's Algorithm A with the option values of
 * QuadcopterSignedDist.jl:28-31: tol=1e-5, min_hessian_perturbation=1e-10, jacobian_regularization_value=1e-7, no max_iter
 * -> IPOPT's default 3000, alpha_for_y=min, recalc_y=no).  PARITY UNPINNED, exactly as for the parking oracle (see its header).
 *
 * Reformulations that leave feasible set and optimum unchanged: the timeScale chain (:157) is one scalar t with multiplicity
 * N+1; x[:,1]==x0 is eliminated; the gyroscopic terms use x[10..12] of STAGE 1 (single-index Julia quirk), i.e. the
 * constants x0[10..12]; the `>= R` rows get a slack.
 *
 * Structured Newton solve: every (stage, box) block (lambda in R^6, slack, row slack, 2 multipliers) is condensed onto the
 * position (x1,x2,x3); the optimal-control problem in (x_k in R^12, copy of u_{k-1}; u_k in R^4) is solved by a Riccati recursion
 * with 14 right-hand sides (main, t, nu_1..12); (t, nu) form a 13x13 border.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NXS 12
#define NU 4
#define NS (NXS + NU)      /* Riccati state: x and the copy w = u_{k-1} */
#define NZ (NS + NU)       /* stage vector (x, w, u) */
#define NC (2 + NXS)       /* right-hand sides */
#define NOB 5
#define NL 6

typedef struct {
    double tol; int max_iter;
    double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
    double dw_min, dw0, dw_max, kw_inc0, kw_inc, kw_dec, dc_bar, kappa_c;
    double gamma_theta, gamma_phi, delta, s_theta, s_phi, eta_phi, gamma_alpha, s_max, kappa_sigma;
    double constr_viol_tol, dual_inf_tol, compl_inf_tol, rho_term;
    int lsq_init, verbose;
    int max_soc;      /* second-order correction trials per iteration (IPOPT's default: 4); 0 = off (default) */
    int recalc_y_;    /* (slot of the parking oracle's recalc_y: the quadcopter call sets recalc_y = "no") */
    int obj_scaling;  /* 1: IPOPT's gradient-based scaling of the objective (nlp_scaling_max_gradient = 100): the algorithm runs on sf f with sf = 100 / max(100, |grad f(start)|_inf);
                         the unscaled tolerances (dual_inf_tol, compl_inf_tol) are tested on the unscaled quantities, the reported objective is unscaled.  0 = off (default) */
} opts_t;

void obca_oracle_quad_default_opts(opts_t *o) {
    o->tol = 1e-5; o->max_iter = 3000;             /* QuadcopterSignedDist.jl:28-31 (no max_iter given: IPOPT default) */
    o->mu_init = 0.1; o->kappa_eps = 10; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99;
    o->bound_push = 1e-2; o->bound_frac = 1e-2;
    o->dw_min = 1e-10;                              /* min_hessian_perturbation */
    o->dw0 = 1e-4; o->dw_max = 1e40; o->kw_inc0 = 100; o->kw_inc = 8; o->kw_dec = 1.0 / 3;
    o->dc_bar = 1e-7; o->kappa_c = 0.25;            /* jacobian_regularization_value */
    o->gamma_theta = 1e-5; o->gamma_phi = 1e-8; o->delta = 1; o->s_theta = 1.1; o->s_phi = 2.3;
    o->eta_phi = 1e-8; o->gamma_alpha = 0.05; o->s_max = 100; o->kappa_sigma = 1e10;
    o->constr_viol_tol = 1e-4; o->dual_inf_tol = 1; o->compl_inf_tol = 1e-4; o->rho_term = 1e3; o->lsq_init = 0; o->verbose = 0; o->recalc_y_ = 0; o->obj_scaling = 0;
    o->max_soc = 0;                                 /* second-order correction: an option the caller sets (opts.max_soc), never the environment */
}

/* model constants, QuadcopterSignedDist.jl:51-62 */
static const double MASS = 0.5, GRAV = 9.81, KF = 0.0611, KM = 0.0015, ARM = 0.225;
static const double INERT[3] = {3.9e-3, 4.4e-3, 4.9e-3};

typedef struct {
    int N; double Ts, R, x0[NXS], xF[NXS], ob[NOB][NL], wH, gyro[3];
    int dist;    /* 1: QuadcopterDist.jl (no slack variable, x[10] in [-1.5, 3]); 0: QuadcopterSignedDist.jl */
} prob_t;

typedef struct { int x, u, t, lam, s, so, n, pi, nu, yo, m; } lay_t;
static void make_layout(int N, lay_t *l) {
    int o = 0, N1 = N + 1;
    l->x = o; o += NXS * N1; l->u = o; o += NU * N; l->t = o; o += 1; l->lam = o; o += NL * NOB * N1; l->s = o; o += NOB * N1;
    l->so = o; o += NOB * N1; l->n = o;
    o = 0; l->pi = o; o += NXS * N; l->nu = o; o += NXS; l->yo = o; o += 2 * NOB * N1; l->m = o;
}

static const double XLB[NXS] = {0, 0, 0, -3, -0.2, -0.2, -1, -1, -1, -1, -1, -1};  /* :78-94 */
static const double XUB[NXS] = {10, 10, 5, 3, 0.2, 0.2, 1, 1, 1, 1, 1, 1};

/* iterate: primal v[n], equality multipliers y[m], bound multipliers zL[n], zU[n]; lb/ub/mult are per-variable tables */
typedef struct { const prob_t *p; lay_t l; double *lb, *ub, *mult; double sf; /* objective scaling factor (1: none) */ } model_t;

static void *xcalloc(size_t n, size_t s) { void *q = calloc(n ? n : 1, s); if (!q) { fprintf(stderr, "oom\n"); exit(1); } return q; }

static void model_init(model_t *M, const prob_t *p) {
    M->p = p; make_layout(p->N, &M->l); M->sf = 1.0;
    const lay_t *l = &M->l; int N = p->N, n = l->n;
    M->lb = xcalloc(n, 8); M->ub = xcalloc(n, 8); M->mult = xcalloc(n, 8);
    for (int i = 0; i < n; i++) { M->lb[i] = -INFINITY; M->ub[i] = INFINITY; M->mult[i] = 1; }
    for (int k = 1; k <= N; k++) for (int i = 0; i < NXS; i++) { M->lb[l->x + NXS * k + i] = XLB[i]; M->ub[l->x + NXS * k + i] = XUB[i]; }
    if (p->dist) for (int k = 1; k <= N; k++) { M->lb[l->x + NXS * k + 9] = -1.5; M->ub[l->x + NXS * k + 9] = 3.0; }   /* QuadcopterDist.jl:88 */
    for (int i = 0; i < NU * N; i++) { M->lb[l->u + i] = 1.2; M->ub[l->u + i] = 7.8; }      /* :74-77 */
    M->lb[l->t] = 0.5; M->ub[l->t] = 2.0; M->mult[l->t] = N + 1;                               /* :97 */
    for (int i = 0; i < NL * NOB * (N + 1); i++) M->lb[l->lam + i] = 0;                         /* :99-103 */
    for (int i = 0; i < NOB * (N + 1); i++) { M->lb[l->s + i] = p->dist ? -INFINITY : 0; M->lb[l->so + i] = 0; }     /* :105; no slack variable in QuadcopterDist: it stays frozen at 0 here */
}
static void model_free(model_t *M) { free(M->lb); free(M->ub); free(M->mult); }
/* |grad f|_inf at v over the variables of the reference's model (QuadcopterSignedDist.jl:66-73; every one of the N + 1 timeScale variables carries 0.25 + 10 t): what IPOPT's
 * gradient-based scaling looks at.  At the reference's start it is the slack penalty, 1e2 + 2e3 * 1 = 2 100 (scaling factor 100 / 2 100); 10.25 for QuadcopterDist (factor 1). */
static double objective_gradient_max(const model_t *M, const double *v) {
    const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N; double g = fabs(0.25 + 10 * v[l->t]);
    for (int k = 0; k <= N; k++) for (int i = 9; i < 12; i++) g = fmax(g, fabs(2e-4 * v[l->x + NXS * k + i]));
    for (int k = 0; k < N; k++) for (int j = 0; j < NU; j++) {
        double gu = -2e-3 * (p->wH - v[l->u + NU * k + j]);
        if (k >= 1) gu += -2e-2 * (v[l->u + NU * (k - 1) + j] - v[l->u + NU * k + j]);
        if (k + 1 < N) gu += 2e-2 * (v[l->u + NU * k + j] - v[l->u + NU * (k + 1) + j]);
        g = fmax(g, fabs(gu)); }
    for (int i = 0; i < NL * NOB * (N + 1); i++) g = fmax(g, fabs(2e-4 * v[l->lam + i]));
    if (!p->dist) for (int i = 0; i < NOB * (N + 1); i++) g = fmax(g, fabs(1e2 + 2e3 * v[l->s + i]));
    return g;
}

/* -------------------------------------------------------------- dynamics g(x,u) with x+ = x + t Ts g  (:136-156) */
/* local variable order for derivatives: a4,a5,a6 (angles x[3..5]), r10,r11,r12 (rates x[9..11]), u1..u4  -> 10 vars */
#define NV 10
static void dyn_g(const prob_t *p, const double *x, const double *u, double g[NXS], double dg[NXS][NV] /* or NULL */,
                  const double *w /* 12 weights or NULL */, double HG[NV][NV] /* sum_i w_i Hess g_i */) {
    double s4 = sin(x[3]), c4 = cos(x[3]), s5 = sin(x[4]), c5 = cos(x[4]), s6 = sin(x[5]), c6 = cos(x[5]);
    double T4 = s4 / c4, S4 = 1 / c4, r10 = x[9], r11 = x[10], r12 = x[11];
    double U = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3], kap = KF / MASS;
    double g4 = c5 * r10 + s5 * r12, h = s5 * r10 - c5 * r12;
    double E7 = s4 * c5 * s6 + s5 * c6, E8 = -s4 * c5 * c6 + s5 * s6, E9 = c4 * c5;
    g[0] = x[6]; g[1] = x[7]; g[2] = x[8];
    g[3] = g4; g[4] = T4 * h + r11; g[5] = -S4 * h;
    g[6] = kap * U * E7; g[7] = kap * U * E8; g[8] = kap * U * E9 - GRAV;
    g[9] = (ARM * KF * (u[1] * u[1] - u[3] * u[3]) - (INERT[2] - INERT[1]) * p->gyro[1] * p->gyro[2]) / INERT[0];
    g[10] = (ARM * KF * (u[2] * u[2] - u[0] * u[0]) - (INERT[0] - INERT[2]) * p->gyro[0] * p->gyro[2]) / INERT[1];
    g[11] = (KM * (u[0] * u[0] - u[1] * u[1] + u[2] * u[2] - u[3] * u[3]) - (INERT[1] - INERT[0]) * p->gyro[0] * p->gyro[1]) / INERT[2];
    if (!dg) return;
    memset(dg, 0, sizeof(double) * NXS * NV);
    double T4p = S4 * S4, S4p = S4 * T4;
    /* g4 */ dg[3][1] = -h; dg[3][3] = c5; dg[3][5] = s5;
    /* g5 = T4 h + r11 */ dg[4][0] = T4p * h; dg[4][1] = T4 * g4; dg[4][3] = T4 * s5; dg[4][4] = 1; dg[4][5] = -T4 * c5;
    /* g6 = -S4 h */ dg[5][0] = -S4p * h; dg[5][1] = -S4 * g4; dg[5][3] = -S4 * s5; dg[5][5] = S4 * c5;
    double E7d[3] = {c4 * c5 * s6, -s4 * s5 * s6 + c5 * c6, s4 * c5 * c6 - s5 * s6};
    double E8d[3] = {-c4 * c5 * c6, s4 * s5 * c6 + c5 * s6, s4 * c5 * s6 + s5 * c6};
    double E9d[3] = {-s4 * c5, -c4 * s5, 0};
    for (int a = 0; a < 3; a++) { dg[6][a] = kap * U * E7d[a]; dg[7][a] = kap * U * E8d[a]; dg[8][a] = kap * U * E9d[a]; }
    for (int j = 0; j < 4; j++) { dg[6][6 + j] = 2 * kap * u[j] * E7; dg[7][6 + j] = 2 * kap * u[j] * E8; dg[8][6 + j] = 2 * kap * u[j] * E9; }
    dg[9][6 + 1] = 2 * ARM * KF * u[1] / INERT[0]; dg[9][6 + 3] = -2 * ARM * KF * u[3] / INERT[0];
    dg[10][6 + 2] = 2 * ARM * KF * u[2] / INERT[1]; dg[10][6 + 0] = -2 * ARM * KF * u[0] / INERT[1];
    dg[11][6 + 0] = 2 * KM * u[0] / INERT[2]; dg[11][6 + 1] = -2 * KM * u[1] / INERT[2];
    dg[11][6 + 2] = 2 * KM * u[2] / INERT[2]; dg[11][6 + 3] = -2 * KM * u[3] / INERT[2];
    if (!w) return;
    memset(HG, 0, sizeof(double) * NV * NV);
#define SYM(i, j, v) { double v_ = (v); HG[i][j] += v_; if ((i) != (j)) HG[j][i] += v_; }
    /* g4: (5,5) -g4 ; (5,r10) -s5 ; (5,r12) c5 */
    SYM(1, 1, w[3] * (-g4)); SYM(1, 3, w[3] * (-s5)); SYM(1, 5, w[3] * c5);
    /* g5 = T4 h + r11 */
    SYM(0, 0, w[4] * 2 * T4 * T4p * h); SYM(0, 1, w[4] * T4p * g4); SYM(0, 3, w[4] * T4p * s5); SYM(0, 5, w[4] * (-T4p * c5));
    SYM(1, 1, w[4] * (-T4 * h)); SYM(1, 3, w[4] * T4 * c5); SYM(1, 5, w[4] * T4 * s5);
    /* g6 = -S4 h ; S4'' = S4 (T4^2 + S4^2) */
    SYM(0, 0, w[5] * (-S4 * (T4 * T4 + S4 * S4) * h)); SYM(0, 1, w[5] * (-S4p * g4)); SYM(0, 3, w[5] * (-S4p * s5)); SYM(0, 5, w[5] * S4p * c5);
    SYM(1, 1, w[5] * S4 * h); SYM(1, 3, w[5] * (-S4 * c5)); SYM(1, 5, w[5] * (-S4 * s5));
    /* g7..g9 = kap U E */
    double E7h[3][3] = {{-s4 * c5 * s6, -c4 * s5 * s6, c4 * c5 * c6}, {0, -E7, -s4 * s5 * c6 - c5 * s6}, {0, 0, -E7}};
    double E8h[3][3] = {{s4 * c5 * c6, c4 * s5 * c6, c4 * c5 * s6}, {0, -E8, -s4 * s5 * s6 + c5 * c6}, {0, 0, -E8}};
    double E9h[3][3] = {{-c4 * c5, s4 * s5, 0}, {0, -c4 * c5, 0}, {0, 0, 0}};
    for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) SYM(a, b, kap * U * (w[6] * E7h[a][b] + w[7] * E8h[a][b] + w[8] * E9h[a][b]));
    for (int j = 0; j < 4; j++) {
        SYM(6 + j, 6 + j, 2 * kap * (w[6] * E7 + w[7] * E8 + w[8] * E9));
        for (int a = 0; a < 3; a++) SYM(a, 6 + j, 2 * kap * u[j] * (w[6] * E7d[a] + w[7] * E8d[a] + w[8] * E9d[a]));
    }
    /* g10..g12: quadratic in u */
    SYM(6 + 1, 6 + 1, w[9] * 2 * ARM * KF / INERT[0]); SYM(6 + 3, 6 + 3, -w[9] * 2 * ARM * KF / INERT[0]);
    SYM(6 + 2, 6 + 2, w[10] * 2 * ARM * KF / INERT[1]); SYM(6 + 0, 6 + 0, -w[10] * 2 * ARM * KF / INERT[1]);
    SYM(6 + 0, 6 + 0, w[11] * 2 * KM / INERT[2]); SYM(6 + 1, 6 + 1, -w[11] * 2 * KM / INERT[2]);
    SYM(6 + 2, 6 + 2, w[11] * 2 * KM / INERT[2]); SYM(6 + 3, 6 + 3, -w[11] * 2 * KM / INERT[2]);
#undef SYM
}
/* stage-vector index of the local derivative variables: a4..a6 -> x[3..5], r -> x[9..11], u -> NS + j */
static const int VIDX[NV] = {3, 4, 5, 9, 10, 11, NS + 0, NS + 1, NS + 2, NS + 3};

/* obstacle rows of one (stage, box): c1 = |q|^2 - 1, c2 = -b'lam + p'q + 0.01 s - R - so  (:169-171) */
static void obs_rows(const prob_t *p, int j, const double *x, const double *lam, double s, double so, double c[2], double q[3]) {
    double bl = 0;
    for (int i = 0; i < 3; i++) q[i] = lam[i] - lam[3 + i];
    for (int i = 0; i < NL; i++) bl += p->ob[j][i] * lam[i];
    c[0] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] - 1;
    c[1] = -bl + x[0] * q[0] + x[1] * q[1] + x[2] * q[2] + (p->dist ? 0.0 : 0.01 * s) - p->R - so;
}

static void eval_f_theta(const model_t *M, const double *v, double *f, double *th1, double *thinf) {
    const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N;
    double t = v[l->t], tau = t * p->Ts, J = 0, th = 0, ti = 0;
    for (int k = 0; k < N; k++) {
        const double *u = v + l->u + NU * k, *x = v + l->x + NXS * k;
        for (int j = 0; j < NU; j++) { J += 1e-3 * (p->wH - u[j]) * (p->wH - u[j]); if (k >= 1) J += 1e-2 * (u[j - NU] - u[j]) * (u[j - NU] - u[j]); }
        double g[NXS]; dyn_g(p, x, u, g, NULL, NULL, NULL);
        for (int i = 0; i < NXS; i++) { double r = fabs(v[l->x + NXS * (k + 1) + i] - x[i] - tau * g[i]); th += r; if (r > ti) ti = r; }
    }
    J += (N + 1) * (0.25 * t + 5 * t * t);
    /* (J is scaled by M->sf where it is handed out, below) */
    for (int k = 0; k <= N; k++) {
        const double *x = v + l->x + NXS * k;
        J += 1e-4 * (x[9] * x[9] + x[10] * x[10] + x[11] * x[11]);
        for (int j = 0; j < NOB; j++) {
            int bo = k * NOB + j; const double *lam = v + l->lam + NL * bo; double s = v[l->s + bo], c[2], q[3];
            if (!p->dist) J += 1e2 * s + 1e3 * s * s;
            for (int i = 0; i < NL; i++) J += 1e-4 * lam[i] * lam[i];
            obs_rows(p, j, x, lam, s, v[l->so + bo], c, q);
            for (int i = 0; i < 2; i++) { double r = fabs(c[i]); th += r; if (r > ti) ti = r; }
        }
    }
    for (int i = 0; i < NXS; i++) { double r = fabs(v[l->x + NXS * N + i] - p->xF[i]); th += r; if (r > ti) ti = r; }
    *f = M->sf * J; *th1 = th; if (thinf) *thinf = ti;
}
static double barrier_sum(const model_t *M, const double *v) {
    double s = 0; int n = M->l.n;
    for (int i = 0; i < n; i++) { if (isfinite(M->lb[i])) s += M->mult[i] * log(v[i] - M->lb[i]); if (isfinite(M->ub[i])) s += M->mult[i] * log(M->ub[i] - v[i]); }
    return s;
}

/* -------------------------------------------------------------- small dense helpers */
static int ldl_n(int n, double *A) { /* lower in; L strict lower, D diag; returns #neg pivots, -1 on zero/NaN */
    int neg = 0;
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k] * A[k * n + k];
        if (d == 0 || d != d) return -1;
        if (d < 0) neg++;
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double s = A[i * n + j]; for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k] * A[k * n + k]; A[i * n + j] = s / d; }
    }
    return neg;
}
static void ldl_solve(int n, const double *A, double *b) {
    for (int i = 0; i < n; i++) for (int k = 0; k < i; k++) b[i] -= A[i * n + k] * b[k];
    for (int i = 0; i < n; i++) b[i] /= A[i * n + i];
    for (int i = n - 1; i >= 0; i--) for (int k = i + 1; k < n; k++) b[i] -= A[k * n + i] * b[k];
}

/* -------------------------------------------------------------- per-block factorisation record */
typedef struct {
    double Dl[NL], Ds, Dso, T2, iT2;     /* diagonals; T2 = 1e-4/Ds + 1/Dso + dc */
    double g1[NL], g2[NL], q[3];
    double hw[NL], Minv[3], hc[NL], Hr[(NL - 1) * (NL - 1)];   /* (lambda, y1) block in the null space of g1 */
    double Cp[NL][3], rk[NL + 1], r2;    /* coupling to p, rhs of (lambda,y1), modified rhs of row 2 */
    double r_s, r_so, c[2];
} obs_fact;

static void hh_apply(int v, const double *w, double *x) { double s = 0; for (int i = 0; i < v; i++) s += w[i] * x[i]; for (int i = 0; i < v; i++) x[i] -= 2 * s * w[i]; }
static int lamblock_factor(obs_fact *F, const double *Hb, const double *qv, double dc) {
    const int v = NL; double nq = 0; for (int i = 0; i < v; i++) nq += qv[i] * qv[i]; nq = sqrt(nq);
    double alpha = qv[0] > 0 ? -nq : nq, w[NL], nw = 0;
    for (int i = 0; i < v; i++) { w[i] = qv[i] - (i == 0 ? alpha : 0); nw += w[i] * w[i]; } nw = sqrt(nw);
    for (int i = 0; i < v; i++) F->hw[i] = nw > 0 ? w[i] / nw : 0;
    double Ht[NL * NL];
    for (int j = 0; j < v; j++) { double col[NL]; for (int i = 0; i < v; i++) col[i] = Hb[i * v + j]; hh_apply(v, F->hw, col); for (int i = 0; i < v; i++) Ht[i * v + j] = col[i]; }
    for (int i = 0; i < v; i++) hh_apply(v, F->hw, Ht + i * v);
    double a = Ht[0], det = a * (-dc) - alpha * alpha;
    if (!(det < 0)) return 0;
    F->Minv[0] = -dc / det; F->Minv[1] = -alpha / det; F->Minv[2] = a / det;
    int m = v - 1;
    for (int i = 0; i < m; i++) F->hc[i] = Ht[(i + 1) * v];
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) F->Hr[i * m + j] = Ht[(i + 1) * v + (j + 1)] - F->Minv[0] * F->hc[i] * F->hc[j];
    return ldl_n(m, F->Hr) == 0;
}
static void lamblock_solve(const obs_fact *F, double *col /* [r_lam(6); r_y1] -> [lam; y1] */) {
    const int v = NL, m = v - 1;
    hh_apply(v, F->hw, col);
    double g0 = col[0], gy = col[v], t0 = F->Minv[0] * g0 + F->Minv[1] * gy, rr[NL];
    for (int i = 0; i < m; i++) rr[i] = col[i + 1] - F->hc[i] * t0;
    ldl_solve(m, F->Hr, rr);
    double hl = 0; for (int i = 0; i < m; i++) hl += F->hc[i] * rr[i];
    g0 -= hl;
    col[0] = F->Minv[0] * g0 + F->Minv[1] * gy; col[v] = F->Minv[1] * g0 + F->Minv[2] * gy;
    for (int i = 0; i < m; i++) col[i + 1] = rr[i];
    hh_apply(v, F->hw, col);
}

/* -------------------------------------------------------------- Newton system workspace */
typedef struct {
    const model_t *M; int N;
    double (*H)[NZ][NZ], (*hz)[NZ], (*hb)[NZ], (*Ht)[NZ];
    double Htt, gt_z, gt_b;
    double (*A)[NXS][NXS], (*B)[NXS][NU], (*Ft)[NXS], (*dd)[NXS];
    obs_fact *of;
    double (*P)[NS][NS], (*K)[NU][NS], (*Lq)[NU * NU];
    double (*pv)[NC][NS], (*kf)[NC][NU], (*ds)[NC][NS], (*du)[NC][NU], (*pic)[NC][NXS];
    double dinf, pinf, cinf0, cinfmu, sumy, sumz; int nb, nm;
    const double *csoc;      /* second-order correction: constraint values that replace c(v) on the right-hand side (layout pi | nu | yo), or NULL */
    int lsq;                 /* least-squares multiplier mode (IPOPT's initial multipliers): Hessian := identity on every primal variable, right-hand side := the gradient of
                                the Lagrangian with the bound multipliers (z-form) and zero for the constraint rows; call with y = 0, dw = dc = rho = 0 */
} kkt_t;
static kkt_t *kkt_alloc(const model_t *M) {
    kkt_t *K = xcalloc(1, sizeof *K); int N1 = M->p->N + 1; K->M = M; K->N = M->p->N;
    K->H = xcalloc(N1, sizeof *K->H); K->hz = xcalloc(N1, sizeof *K->hz); K->hb = xcalloc(N1, sizeof *K->hb); K->Ht = xcalloc(N1, sizeof *K->Ht);
    K->A = xcalloc(N1, sizeof *K->A); K->B = xcalloc(N1, sizeof *K->B); K->Ft = xcalloc(N1, sizeof *K->Ft); K->dd = xcalloc(N1, sizeof *K->dd);
    K->of = xcalloc((size_t)N1 * NOB, sizeof *K->of);
    K->P = xcalloc(N1, sizeof *K->P); K->K = xcalloc(N1, sizeof *K->K); K->Lq = xcalloc(N1, sizeof *K->Lq);
    K->pv = xcalloc(N1, sizeof *K->pv); K->kf = xcalloc(N1, sizeof *K->kf); K->ds = xcalloc(N1, sizeof *K->ds); K->du = xcalloc(N1, sizeof *K->du);
    K->pic = xcalloc(N1 + 1, sizeof *K->pic);
    return K;
}
static void kkt_free(kkt_t *K) {
    free(K->H); free(K->hz); free(K->hb); free(K->Ht); free(K->A); free(K->B); free(K->Ft); free(K->dd); free(K->of);
    free(K->P); free(K->K); free(K->Lq); free(K->pv); free(K->kf); free(K->ds); free(K->du); free(K->pic); free(K);
}

/* bound bookkeeping of one primal variable: Sigma, gradient parts, complementarity */
typedef struct { double Sig, gz, gb; } bnd_t;
static bnd_t bound_terms(const model_t *M, int i, const double *v, const double *zL, const double *zU, double mu, double *c0, double *cmu, double *sumz, int *nb) {
    bnd_t r = {0, 0, 0}; double m = M->mult[i];
    if (isfinite(M->lb[i])) { double d = v[i] - M->lb[i], c = d * zL[i]; r.Sig += m * zL[i] / d; r.gz -= m * zL[i]; r.gb -= m * mu / d;
        if (fabs(c) > *c0) *c0 = fabs(c); if (fabs(c - mu) > *cmu) *cmu = fabs(c - mu); *sumz += m * fabs(zL[i]); *nb += (int)m; }
    if (isfinite(M->ub[i])) { double d = M->ub[i] - v[i], c = d * zU[i]; r.Sig += m * zU[i] / d; r.gz += m * zU[i]; r.gb += m * mu / d;
        if (fabs(c) > *c0) *c0 = fabs(c); if (fabs(c - mu) > *cmu) *cmu = fabs(c - mu); *sumz += m * fabs(zU[i]); *nb += (int)m; }
    return r;
}

static int kkt_assemble(kkt_t *K, const double *v, const double *y, const double *zL, const double *zU, double mu, double dw, double dc) {
    const model_t *M = K->M; const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N, ok = 1;
    double t = v[l->t], tau = t * p->Ts, c0 = 0, cmu = 0, sumz = 0, sumy = 0, dmax = 0, pmax = 0; int nb = 0, nm = 0;
    const int lsq = K->lsq; const double sf = M->sf;
#define LSQ_B(b) do { if (lsq) (b).gb = (b).gz; } while (0)      /* least-squares mode: the gradient with the bound multipliers themselves */
    {
        bnd_t b = bound_terms(M, l->t, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b);
        double gf = sf * (N + 1) * (0.25 + 10 * t);
        K->Htt = lsq ? (double)(N + 1) : sf * 10.0 * (N + 1) + b.Sig + dw;      /* (t stands for the N + 1 timeScale variables of the reference's model: N + 1 unit diagonal entries) */ K->gt_z = gf + b.gz; K->gt_b = gf + b.gb;
    }
    for (int k = 0; k <= N; k++) {
        double (*H)[NZ] = K->H[k]; double *hz = K->hz[k], *hb = K->hb[k], *Ht = K->Ht[k];
        memset(H, 0, sizeof K->H[k]); memset(hz, 0, sizeof K->hz[k]); memset(hb, 0, sizeof K->hb[k]); memset(Ht, 0, sizeof K->Ht[k]);
        const double *x = v + l->x + NXS * k;
        for (int i = 0; i < NXS; i++) {
            double gx = (i >= 9) ? sf * 2e-4 * x[i] : 0, hx = (i >= 9) ? sf * 2e-4 : 0, Sig = 0;
            hz[i] = gx; hb[i] = gx;
            if (k >= 1) { bnd_t b = bound_terms(M, l->x + NXS * k + i, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b); Sig = b.Sig; hz[i] += b.gz; hb[i] += b.gb; }
            H[i][i] = lsq ? 1.0 : hx + Sig + dw;
        }
        for (int j = 0; j < NOB; j++) {   /* condense the box block onto the position */
            int bo = k * NOB + j; obs_fact *F = &K->of[bo];
            const double *lam = v + l->lam + NL * bo, *yv = y + l->yo + 2 * bo; double s = v[l->s + bo], so = v[l->so + bo];
            obs_rows(p, j, x, lam, s, so, F->c, F->q);
            for (int r = 0; r < 2; r++) { if (fabs(F->c[r]) > pmax) pmax = fabs(F->c[r]); sumy += fabs(yv[r]); } nm += 2;
            double rl_b[NL];
            for (int i = 0; i < NL; i++) {
                double sg = i < 3 ? 1.0 : -1.0; int a = i % 3;
                F->g1[i] = 2 * sg * F->q[a]; F->g2[i] = -p->ob[j][i] + sg * x[a];
                bnd_t b = bound_terms(M, l->lam + NL * bo + i, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b);
                double gl = sf * 2e-4 * lam[i] + F->g1[i] * yv[0] + F->g2[i] * yv[1];
                if (fabs(gl + b.gz) > dmax) dmax = fabs(gl + b.gz);
                rl_b[i] = gl + b.gb; F->Dl[i] = lsq ? 1.0 : sf * 2e-4 + b.Sig + dw;
            }
            { bnd_t b = bound_terms(M, l->s + bo, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b); double gs = sf * (1e2 + 2e3 * s) + 0.01 * yv[1];
              if (!p->dist && fabs(gs + b.gz) > dmax) dmax = fabs(gs + b.gz); F->r_s = gs + b.gb; F->Ds = lsq ? 1.0 : sf * 2e3 + b.Sig + dw; }
            if (p->dist) { F->r_s = 0; F->Ds = INFINITY; }       /* frozen: every 1/Ds term below vanishes, ds = 0 */
            { bnd_t b = bound_terms(M, l->so + bo, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b); double gs = -yv[1];
              if (fabs(gs + b.gz) > dmax) dmax = fabs(gs + b.gz); F->r_so = gs + b.gb; F->Dso = lsq ? 1.0 : b.Sig + dw; }
            /* row 2 after eliminating s and so:  g2'dlam + q'dp - T2 dy2 = r2 */
            F->T2 = 1e-4 / F->Ds + 1.0 / F->Dso + dc; F->iT2 = 1.0 / F->T2;
            F->r2 = -(lsq ? 0.0 : K->csoc ? K->csoc[l->yo + 2 * bo + 1] : F->c[1]) + 0.01 * F->r_s / F->Ds - F->r_so / F->Dso;
            for (int i = 0; i < 3; i++) { hz[i] += F->q[i] * yv[1]; hb[i] += F->q[i] * yv[1]; }
            double Hb[NL * NL];
            for (int i = 0; i < NL; i++) {
                int a = i % 3; double sg = i < 3 ? 1.0 : -1.0;
                for (int m_ = 0; m_ < NL; m_++) {
                    double sm = m_ < 3 ? 1.0 : -1.0;
                    Hb[i * NL + m_] = ((m_ % 3) == a ? 2 * yv[0] * sg * sm : 0.0) + F->g2[i] * F->g2[m_] * F->iT2;
                }
                Hb[i * NL + i] += F->Dl[i];
                for (int c_ = 0; c_ < 3; c_++) F->Cp[i][c_] = (c_ == a ? yv[1] * sg : 0.0) + F->g2[i] * F->q[c_] * F->iT2;
                F->rk[i] = -rl_b[i] + F->g2[i] * F->r2 * F->iT2;
            }
            F->rk[NL] = -(lsq ? 0.0 : K->csoc ? K->csoc[l->yo + 2 * bo] : F->c[0]);
            if (!lamblock_factor(F, Hb, F->g1, dc)) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "lamblock fail k=%d j=%d\n", k, j); }
            double Z[NL + 1][4];
            for (int c_ = 0; c_ < 4; c_++) { double col[NL + 1]; for (int i = 0; i < NL; i++) col[i] = c_ < 3 ? F->Cp[i][c_] : F->rk[i]; col[NL] = c_ < 3 ? 0 : F->rk[NL];
                lamblock_solve(F, col); for (int i = 0; i <= NL; i++) Z[i][c_] = col[i]; }
            for (int a = 0; a < 3; a++) {
                for (int b_ = 0; b_ < 3; b_++) { double s_ = F->q[a] * F->q[b_] * F->iT2; for (int i = 0; i < NL; i++) s_ -= F->Cp[i][a] * Z[i][b_]; H[a][b_] += s_; }
                double s_ = F->q[a] * F->r2 * F->iT2; for (int i = 0; i < NL; i++) s_ -= F->Cp[i][a] * Z[i][3];
                hb[a] -= s_;
            }
        }
        if (k == N) { for (int i = 0; i < NXS; i++) { double e = fabs(x[i] - p->xF[i]); if (e > pmax) pmax = e; } continue; }
        const double *u = v + l->u + NU * k;
        for (int j = 0; j < NU; j++) {
            bnd_t b = bound_terms(M, l->u + NU * k + j, v, zL, zU, mu, &c0, &cmu, &sumz, &nb); LSQ_B(b);
            double gu = -sf * 2e-3 * (p->wH - u[j]), hu = sf * 2e-3; const double w2 = sf * 2e-2;
            if (k >= 1) { double e = u[j - NU] - u[j]; gu += -w2 * e; hu += w2; hz[NXS + j] += w2 * e; hb[NXS + j] += w2 * e;
                if (!lsq) { H[NXS + j][NXS + j] += w2; H[NXS + j][NS + j] += -w2; H[NS + j][NXS + j] += -w2; } }
            hz[NS + j] += gu + b.gz; hb[NS + j] += gu + b.gb; H[NS + j][NS + j] += lsq ? 1.0 : hu + b.Sig + dw;
        }
        {   /* dynamics x+ - x - t Ts g(x,u) = 0 with multiplier pi_k */
            double g[NXS], dg[NXS][NV], HG[NV][NV]; const double *pi = y + l->pi + NXS * k;
            dyn_g(p, x, u, g, dg, pi, HG);
            for (int i = 0; i < NXS; i++) {
                for (int j = 0; j < NXS; j++) K->A[k][i][j] = (i == j);
                for (int j = 0; j < NU; j++) K->B[k][i][j] = 0;
                if (i < 3) K->A[k][i][6 + i] += tau;
                for (int c_ = 0; c_ < NV; c_++) { int id = VIDX[c_]; if (id < NXS) K->A[k][i][id] += tau * dg[i][c_]; else K->B[k][i][id - NS] += tau * dg[i][c_]; }
                K->Ft[k][i] = p->Ts * g[i];
                double r = v[l->x + NXS * (k + 1) + i] - x[i] - tau * g[i];
                K->dd[k][i] = -(lsq ? 0.0 : K->csoc ? K->csoc[l->pi + NXS * k + i] : r); if (fabs(r) > pmax) pmax = fabs(r); sumy += fabs(pi[i]);
            }
            nm += NXS;
            /* Lagrangian Hessian of -pi'(t Ts g): -tau sum pi_i Hess g_i on the local variables; cross terms with t: -Ts pi'dg */
            for (int a = 0; a < NV; a++) {
                for (int b_ = 0; b_ < NV; b_++) H[VIDX[a]][VIDX[b_]] += -tau * HG[a][b_];
                double s_ = 0; for (int i = 0; i < NXS; i++) s_ += pi[i] * dg[i][a];
                Ht[VIDX[a]] += -p->Ts * s_;
            }
            for (int i = 0; i < 3; i++) Ht[6 + i] += -p->Ts * pi[i];
        }
    }
    for (int i = 0; i < NXS; i++) sumy += fabs(y[l->nu + i]);
#undef LSQ_B
    K->pinf = pmax; K->cinf0 = c0; K->cinfmu = cmu; K->sumz = sumz; K->sumy = sumy; K->nb = nb; K->nm = nm + NXS; K->dinf = dmax;
    return ok;
}

/* add J^T(pi,nu) to the stage gradients; return max |grad L| over x,u,t */
static double stage_dual_inf(kkt_t *K, const double *y) {
    const lay_t *l = &K->M->l; int N = K->N; double dmax = 0;
    for (int k = 0; k <= N; k++)
        for (int i = 0; i < NXS; i++) {
            double r = k >= 1 ? y[l->pi + NXS * (k - 1) + i] : 0;
            if (k < N) for (int j = 0; j < NXS; j++) r -= K->A[k][j][i] * y[l->pi + NXS * k + j]; else r += y[l->nu + i];
            K->hz[k][i] += r; K->hb[k][i] += r;
            if (k >= 1 && fabs(K->hz[k][i]) > dmax) dmax = fabs(K->hz[k][i]);
        }
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < NU; i++) {
            double r = 0; for (int j = 0; j < NXS; j++) r -= K->B[k][j][i] * y[l->pi + NXS * k + j];
            K->hz[k][NS + i] += r; K->hb[k][NS + i] += r;
            double tot = K->hz[k][NS + i] + (k + 1 < N ? K->hz[k + 1][NXS + i] : 0);
            if (fabs(tot) > dmax) dmax = fabs(tot);
        }
        for (int j = 0; j < NXS; j++) { double r = K->Ft[k][j] * y[l->pi + NXS * k + j]; K->gt_z -= r; K->gt_b -= r; }
    }
    if (fabs(K->gt_z) > dmax) dmax = fabs(K->gt_z);
    return dmax;
}

/* Riccati + border + back-substitution; fills dv (primal), dy (multiplier increments).  Returns 1 if inertia is right. */
static int kkt_solve(kkt_t *K, const double *v, double dc, double rho, double *dv, double *dy) {
    const model_t *M = K->M; const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N, ok = 1;
    double e[NXS];
    memset(K->P[N], 0, sizeof K->P[N]); memset(K->pv[N], 0, sizeof K->pv[N]);
    for (int i = 0; i < NS; i++) for (int j = 0; j < NS; j++) K->P[N][i][j] = K->H[N][i][j];
    for (int i = 0; i < NXS; i++) { e[i] = -(K->lsq ? 0.0 : K->csoc ? K->csoc[l->nu + i] : v[l->x + NXS * N + i] - p->xF[i]); K->P[N][i][i] += rho; }
    for (int i = 0; i < NS; i++) { K->pv[N][0][i] = K->hb[N][i]; K->pv[N][1][i] = K->Ht[N][i]; }
    for (int i = 0; i < NXS; i++) { K->pv[N][0][i] -= rho * e[i]; K->pv[N][2 + i][i] = 1.0; }
    for (int k = N - 1; k >= 0; k--) {
        static double Fm[NS][NZ], PF[NS][NZ], Q[NZ][NZ];
        memset(Fm, 0, sizeof Fm);
        for (int i = 0; i < NXS; i++) { for (int j = 0; j < NXS; j++) Fm[i][j] = K->A[k][i][j]; for (int j = 0; j < NU; j++) Fm[i][NS + j] = K->B[k][i][j]; }
        for (int j = 0; j < NU; j++) Fm[NXS + j][NS + j] = 1;
        for (int i = 0; i < NS; i++) for (int j = 0; j < NZ; j++) { double s = 0; for (int a = 0; a < NS; a++) s += K->P[k + 1][i][a] * Fm[a][j]; PF[i][j] = s; }
        for (int i = 0; i < NZ; i++) for (int j = 0; j < NZ; j++) { double s = K->H[k][i][j]; for (int a = 0; a < NS; a++) s += Fm[a][i] * PF[a][j]; Q[i][j] = s; }
        double *Lq = K->Lq[k];
        for (int i = 0; i < NU; i++) for (int j = 0; j < NU; j++) Lq[i * NU + j] = Q[NS + i][NS + j];
        if (ldl_n(NU, Lq) != 0) { if (getenv("OBCA_DBG")) fprintf(stderr, "Quu fail k=%d\n", k); return 0; }   /* Quu must be positive definite */
        for (int j = 0; j < NS; j++) { double b[NU]; for (int i = 0; i < NU; i++) b[i] = -Q[NS + i][j]; ldl_solve(NU, Lq, b); for (int i = 0; i < NU; i++) K->K[k][i][j] = b[i]; }
        for (int i = 0; i < NS; i++) for (int j = 0; j < NS; j++) { double s = Q[i][j]; for (int a = 0; a < NU; a++) s += Q[i][NS + a] * K->K[k][a][j]; K->P[k][i][j] = s; }
        for (int i = 0; i < NS; i++) for (int j = 0; j < i; j++) { double s = 0.5 * (K->P[k][i][j] + K->P[k][j][i]); K->P[k][i][j] = K->P[k][j][i] = s; }
        for (int c = 0; c < NC; c++) {
            double off[NS] = {0}, hv[NZ] = {0}, tmp[NS], qv[NZ], b[NU];
            if (c == 0) { for (int i = 0; i < NXS; i++) off[i] = K->dd[k][i]; for (int i = 0; i < NZ; i++) hv[i] = K->hb[k][i]; }
            else if (c == 1) { for (int i = 0; i < NXS; i++) off[i] = K->Ft[k][i]; for (int i = 0; i < NZ; i++) hv[i] = K->Ht[k][i]; }
            for (int i = 0; i < NS; i++) { double s = K->pv[k + 1][c][i]; for (int a = 0; a < NS; a++) s += K->P[k + 1][i][a] * off[a]; tmp[i] = s; }
            for (int i = 0; i < NZ; i++) { double s = hv[i]; for (int a = 0; a < NS; a++) s += Fm[a][i] * tmp[a]; qv[i] = s; }
            for (int i = 0; i < NU; i++) b[i] = -qv[NS + i];
            ldl_solve(NU, Lq, b);
            for (int i = 0; i < NU; i++) K->kf[k][c][i] = b[i];
            for (int i = 0; i < NS; i++) { double s = qv[i]; for (int a = 0; a < NU; a++) s += Q[i][NS + a] * b[a]; K->pv[k][c][i] = s; }
        }
    }
    for (int c = 0; c < NC; c++) {
        double s[NS] = {0};
        for (int k = 0; k < N; k++) {
            memcpy(K->ds[k][c], s, sizeof s);
            double uu[NU], sn[NS];
            for (int i = 0; i < NU; i++) { double a = K->kf[k][c][i]; for (int j = 0; j < NS; j++) a += K->K[k][i][j] * s[j]; uu[i] = a; K->du[k][c][i] = a; }
            for (int i = 0; i < NXS; i++) {
                double a = (c == 0) ? K->dd[k][i] : (c == 1 ? K->Ft[k][i] : 0);
                for (int j = 0; j < NXS; j++) a += K->A[k][i][j] * s[j];
                for (int j = 0; j < NU; j++) a += K->B[k][i][j] * uu[j];
                sn[i] = a;
            }
            for (int j = 0; j < NU; j++) sn[NXS + j] = uu[j];
            memcpy(s, sn, sizeof s);
            for (int i = 0; i < NXS; i++) { double a = K->pv[k + 1][c][i]; for (int j = 0; j < NS; j++) a += K->P[k + 1][i][j] * s[j]; K->pic[k + 1][c][i] = -a; }
        }
        memcpy(K->ds[N][c], s, sizeof s);
    }
    /* (1 + NXS) border in (dt, nu) */
    enum { NBD = 1 + NXS };
    double Mb[NBD][NBD], rb[NBD];
    memset(Mb, 0, sizeof Mb);
    {
        double att = K->Htt, rt = -K->gt_b, atn[NXS] = {0};
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < NS; i++) { att += K->Ht[k][i] * K->ds[k][1][i]; rt -= K->Ht[k][i] * K->ds[k][0][i]; for (int c = 0; c < NXS; c++) atn[c] += K->Ht[k][i] * K->ds[k][2 + c][i]; }
            if (k < N) {
                for (int i = 0; i < NU; i++) { att += K->Ht[k][NS + i] * K->du[k][1][i]; rt -= K->Ht[k][NS + i] * K->du[k][0][i]; for (int c = 0; c < NXS; c++) atn[c] += K->Ht[k][NS + i] * K->du[k][2 + c][i]; }
                for (int i = 0; i < NXS; i++) { att -= K->Ft[k][i] * K->pic[k + 1][1][i]; rt += K->Ft[k][i] * K->pic[k + 1][0][i]; for (int c = 0; c < NXS; c++) atn[c] -= K->Ft[k][i] * K->pic[k + 1][2 + c][i]; }
            }
        }
        Mb[0][0] = att; rb[0] = rt;
        for (int c = 0; c < NXS; c++) {
            Mb[0][1 + c] = atn[c]; Mb[1 + c][0] = K->ds[N][1][c];
            for (int c2 = 0; c2 < NXS; c2++) Mb[1 + c][1 + c2] = K->ds[N][2 + c2][c];
            rb[1 + c] = e[c] - K->ds[N][0][c];
        }
        for (int a = 1; a < NBD; a++) for (int b = 1; b < a; b++) { double s = 0.5 * (Mb[a][b] + Mb[b][a]); Mb[a][b] = Mb[b][a] = s; }
    }
    double dt, nu[NXS];
    {
        double S[NXS * NXS], col[NXS], colr[NXS];
        for (int a = 0; a < NXS; a++) for (int b = 0; b < NXS; b++) S[a * NXS + b] = -Mb[1 + a][1 + b];
        if (ldl_n(NXS, S) != 0) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "nu block fail\n"); }
        for (int a = 0; a < NXS; a++) { col[a] = -Mb[1 + a][0]; colr[a] = -rb[1 + a]; }
        ldl_solve(NXS, S, col); ldl_solve(NXS, S, colr);
        double piv = Mb[0][0], rr = rb[0];
        for (int a = 0; a < NXS; a++) { piv -= Mb[0][1 + a] * col[a]; rr -= Mb[0][1 + a] * colr[a]; }
        if (!(piv > 0)) { ok = 0; if (getenv("OBCA_DBG")) fprintf(stderr, "t pivot fail %g\n", piv); }
        dt = rr / piv;
        for (int a = 0; a < NXS; a++) nu[a] = colr[a] - col[a] * dt;
    }
    if (!ok) return 0;
    memset(dv, 0, sizeof(double) * l->n); memset(dy, 0, sizeof(double) * l->m);
    dv[l->t] = dt;
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < NXS; i++) { double a = K->ds[k][0][i] + K->ds[k][1][i] * dt; for (int c = 0; c < NXS; c++) a += K->ds[k][2 + c][i] * nu[c]; dv[l->x + NXS * k + i] = a; }
        if (k < N) {
            for (int i = 0; i < NU; i++) { double a = K->du[k][0][i] + K->du[k][1][i] * dt; for (int c = 0; c < NXS; c++) a += K->du[k][2 + c][i] * nu[c]; dv[l->u + NU * k + i] = a; }
            for (int i = 0; i < NXS; i++) { double a = K->pic[k + 1][0][i] + K->pic[k + 1][1][i] * dt; for (int c = 0; c < NXS; c++) a += K->pic[k + 1][2 + c][i] * nu[c]; dy[l->pi + NXS * k + i] = a; }
        }
    }
    for (int i = 0; i < NXS; i++) dy[l->nu + i] = nu[i];
    for (int k = 0; k <= N; k++)
        for (int j = 0; j < NOB; j++) {
            int bo = k * NOB + j; obs_fact *F = &K->of[bo];
            double dp[3] = {dv[l->x + NXS * k], dv[l->x + NXS * k + 1], dv[l->x + NXS * k + 2]}, col[NL + 1];
            for (int i = 0; i < NL; i++) col[i] = F->rk[i] - (F->Cp[i][0] * dp[0] + F->Cp[i][1] * dp[1] + F->Cp[i][2] * dp[2]);
            col[NL] = F->rk[NL];
            lamblock_solve(F, col);
            double a = -F->r2; for (int i = 0; i < NL; i++) a += F->g2[i] * col[i]; for (int i = 0; i < 3; i++) a += F->q[i] * dp[i];
            double dy2 = a * F->iT2;
            dy[l->yo + 2 * bo] = col[NL]; dy[l->yo + 2 * bo + 1] = dy2;
            for (int i = 0; i < NL; i++) dv[l->lam + NL * bo + i] = col[i];
            dv[l->s + bo] = (-F->r_s - 0.01 * dy2) / F->Ds;
            dv[l->so + bo] = (dy2 - F->r_so) / F->Dso;
        }
    return 1;
}


/*
 * Block feasibility restoration (stands in for the part of IPOPT's restoration phase this path needs).  The reference starts every lambda at
 * 0.05 (QuadcopterSignedDist.jl:204-208) where A'lambda = 0: the gradient of |A'lambda|^2 == 1 vanishes, the constraint Jacobian is rank
 * deficient and IPOPT leaves the point through its restoration phase (minimise the constraint violation; the authors mention its messages,
 * mainQuadcopter.jl:140).  For FIXED positions the violation of the two rows of a (stage, box) block is minimised in closed form: lambda =
 * the dual solution of the point-to-box distance (|A'lambda| = 1 exactly), row slack = row value.  The restoration step therefore resets the
 * multipliers lambda / row slacks / row multipliers of every block at the current positions, keeps x, u, timeScale and their multipliers,
 * restarts the barrier parameter and clears the filter.  It runs (a) at the start if some block has A'lambda = 0 and (b) where IPOPT would
 * enter restoration (line search or inertia correction failed), at most MAX_RESTORE times.  The HIP solver does exactly the same.
 */
#define MAX_RESTORE 3
static void dual_ws_block(const prob_t *p, int j, const double *x, double *lam) {
    double d[3], n2 = 0, q[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) { double hi = p->ob[j][i], lo = -p->ob[j][3 + i], c = x[i] < lo ? lo : (x[i] > hi ? hi : x[i]); d[i] = x[i] - c; n2 += d[i] * d[i]; }
    if (n2 > 1e-16) { double in = 1 / sqrt(n2); for (int i = 0; i < 3; i++) q[i] = d[i] * in; }
    else {   /* inside the box: normal of the nearest face */
        int best = 0; double bd = 1e300, sg = 1;
        for (int i = 0; i < 3; i++) { double hi = p->ob[j][i] - x[i], lo = x[i] + p->ob[j][3 + i]; if (hi < bd) { bd = hi; best = i; sg = 1; } if (lo < bd) { bd = lo; best = i; sg = -1; } }
        q[best] = sg;
    }
    for (int i = 0; i < 3; i++) { lam[i] = q[i] > 0 ? q[i] : 0; lam[3 + i] = q[i] < 0 ? -q[i] : 0; }
}
static double min_norm2(const model_t *M, const double *v) {
    const prob_t *p = M->p; const lay_t *l = &M->l; double mn = 1e300;
    for (int k = 0; k <= p->N; k++) for (int j = 0; j < NOB; j++) { const double *lam = v + l->lam + NL * (k * NOB + j); double n2 = 0;
        for (int i = 0; i < 3; i++) n2 += (lam[i] - lam[3 + i]) * (lam[i] - lam[3 + i]); if (n2 < mn) mn = n2; }
    return mn;
}
static void restore_blocks(const model_t *M, const opts_t *o, double *v, double *y, double *zL) {
    const prob_t *p = M->p; const lay_t *l = &M->l; const double pl = o->bound_push;
    for (int k = 0; k <= p->N; k++) for (int j = 0; j < NOB; j++) {
        const int bo = k * NOB + j; double *lam = v + l->lam + NL * bo, c[2], q[3];
        dual_ws_block(p, j, v + l->x + NXS * k, lam);
        obs_rows(p, j, v + l->x + NXS * k, lam, v[l->s + bo], 0.0, c, q);
        v[l->so + bo] = c[1] < pl ? pl : c[1];
        for (int i = 0; i < NL; i++) { if (lam[i] < pl) lam[i] = pl; zL[l->lam + NL * bo + i] = 1.0; }
        if (!p->dist) { if (v[l->s + bo] < pl) v[l->s + bo] = pl; zL[l->s + bo] = 1.0; }
        zL[l->so + bo] = 1.0;
        y[l->yo + 2 * bo] = 0.0; y[l->yo + 2 * bo + 1] = 0.0;
    }
}

/* -------------------------------------------------------------- interior-point driver */
/* equality rows at v in the layout of the multipliers (pi | nu | yo), as the assembly forms them: what a second-order correction accumulates */
static void constraint_values(const model_t *M, const double *v, double *c) {
    const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N; double tau = v[l->t] * p->Ts;
    for (int k = 0; k < N; k++) {
        double g[NXS]; const double *x = v + l->x + NXS * k;
        dyn_g(p, x, v + l->u + NU * k, g, NULL, NULL, NULL);
        for (int i = 0; i < NXS; i++) c[l->pi + NXS * k + i] = v[l->x + NXS * (k + 1) + i] - x[i] - tau * g[i];
    }
    for (int i = 0; i < NXS; i++) c[l->nu + i] = v[l->x + NXS * N + i] - p->xF[i];
    for (int k = 0; k <= N; k++) for (int j = 0; j < NOB; j++) { int bo = k * NOB + j; double q[3];
        obs_rows(p, j, v + l->x + NXS * k, v + l->lam + NL * bo, v[l->s + bo], v[l->so + bo], c + l->yo + 2 * bo, q); }
}
typedef struct { int status, iters, nreg; double obj, pinf, dinf, mu, t; } result_t;
enum { ST_OPTIMAL = 0, ST_USERLIMIT = 1, ST_ERROR = 2 };
#define FILT_MAX 4096

#define RESTORE_AND_CONTINUE do { restore_blocks(M, o, v, y, zL); nrest++; mu = o->mu_init; tau = fmax(o->tau_min, 1 - mu); nf = 0; dw_last = 0; \
        eval_f_theta(M, v, &f, &th, &thinf); th_min = 1e-4 * fmax(1, th); th_max = 1e4 * fmax(1, th); goto next_iter; } while (0)
static __thread int g_nsoc = 0, g_nsoc_acc = 0;      /* diagnostic: corrections tried / accepted by the calling thread */
int obca_oracle_quad_soc_counts(int *acc) { if (acc) *acc = g_nsoc_acc; return g_nsoc; }
static void ipm_solve(const model_t *M, const opts_t *o, double *v, double *y, double *zL, double *zU, result_t *res) {
    const prob_t *p = M->p; const lay_t *l = &M->l; int N = p->N, n = l->n, m = l->m; const double sf = M->sf;
    kkt_t *K = kkt_alloc(M);
    double *dv = xcalloc(n, 8), *dy = xcalloc(m, 8), *dzL = xcalloc(n, 8), *dzU = xcalloc(n, 8), *vt = xcalloc(n, 8);
    double mu = o->mu_init, tau = fmax(o->tau_min, 1 - mu), dw_last = 0;
    static double filt[FILT_MAX][2]; int nf = 0;
    for (int i = 0; i < NXS; i++) v[l->x + i] = p->x0[i];
    /* row slack takes the row value, then everything is pushed inside its bounds (IPOPT sec. 3.6) */
    for (int k = 0; k <= N; k++) for (int j = 0; j < NOB; j++) { int bo = k * NOB + j; double c[2], q[3];
        obs_rows(p, j, v + l->x + NXS * k, v + l->lam + NL * bo, v[l->s + bo], 0.0, c, q); v[l->so + bo] = c[1]; }
    for (int i = 0; i < n; i++) {
        double lo = M->lb[i], hi = M->ub[i];
        if (i >= l->x && i < l->x + NXS) continue;    /* x_0 is a constant */
        if (isfinite(lo) && isfinite(hi)) {
            double pl = fmin(o->bound_push * fmax(1, fabs(lo)), o->bound_frac * (hi - lo)), pu = fmin(o->bound_push * fmax(1, fabs(hi)), o->bound_frac * (hi - lo));
            if (v[i] < lo + pl) v[i] = lo + pl; if (v[i] > hi - pu) v[i] = hi - pu;
        } else if (isfinite(lo)) { double pl = o->bound_push * fmax(1, fabs(lo)); if (v[i] < lo + pl) v[i] = lo + pl; }
    }
    memset(y, 0, sizeof(double) * m);
    for (int i = 0; i < n; i++) { zL[i] = isfinite(M->lb[i]) ? 1.0 : 0.0; zU[i] = isfinite(M->ub[i]) ? 1.0 : 0.0; }
    for (int i = 0; i < NXS; i++) { zL[l->x + i] = zU[l->x + i] = 0; }
    double f, th, thinf;
    eval_f_theta(M, v, &f, &th, &thinf);
    double th_min = 1e-4 * fmax(1, th), th_max = 1e4 * fmax(1, th);
    int it = 0, status = ST_USERLIMIT, nreg = 0, nrest = 0; double last_pinf = 0, last_dinf = 0;
    if (o->lsq_init) {      /* IPOPT's initial multipliers: least-squares estimate at the starting point, kept if its largest entry is <= constr_mult_init_max = 1e3 (else y = 0);
                               the same structured solve with the Hessian replaced by the identity.  (At the reference's own start, lambda = 0.05, the system is singular: y stays 0.) */
        K->lsq = 1;
        int a = kkt_assemble(K, v, y, zL, zU, 0.0, 0, 0); stage_dual_inf(K, y);
        if (a) a = kkt_solve(K, v, 0.0, 0.0, dv, dy);
        K->lsq = 0;
        if (a) { double ymax = 0; for (int i = 0; i < m; i++) { double q = fabs(dy[i]); if (q > ymax || q != q) ymax = q; }
                 if (ymax <= 1e3 && ymax == ymax) memcpy(y, dy, sizeof(double) * m); }
    }
    if (min_norm2(M, v) < 1e-12) {      /* rank-deficient start (the reference's lambda = 0.05): restoration before the first iteration */
        restore_blocks(M, o, v, y, zL); nrest++;
        eval_f_theta(M, v, &f, &th, &thinf); th_min = 1e-4 * fmax(1, th); th_max = 1e4 * fmax(1, th);
    }
    for (;;) {
        kkt_assemble(K, v, y, zL, zU, mu, 0, 0);
        double dinf = fmax(K->dinf, stage_dual_inf(K, y)), pinf = K->pinf, cinf0 = K->cinf0;
        double sd = fmax(o->s_max, (K->sumy + K->sumz) / (K->nm + K->nb)) / o->s_max, sc = fmax(o->s_max, K->sumz / K->nb) / o->s_max;
        double E0 = fmax(dinf / sd, fmax(pinf, cinf0 / sc));
        eval_f_theta(M, v, &f, &th, &thinf);
        last_pinf = pinf; last_dinf = dinf;
        if (o->verbose) printf("it %3d f=% .8e pinf=%.2e dinf=%.2e cinf=%.2e mu=%.1e dw=%.1e t=%.4f\n", it, f, pinf, dinf, cinf0, mu, dw_last, v[l->t]);
        if (E0 <= o->tol && pinf <= o->constr_viol_tol && dinf / M->sf <= o->dual_inf_tol && cinf0 / M->sf <= o->compl_inf_tol) { status = ST_OPTIMAL; break; }      /* (the three *_tol are IPOPT's tolerances on the UNSCALED problem) */
        if (it >= o->max_iter) { status = ST_USERLIMIT; break; }
        if (!(f == f) || !(pinf == pinf) || !(dinf == dinf)) { status = ST_ERROR; break; }
        for (;;) {
            double cm = 0;
            for (int i = 0; i < n; i++) {
                if (i >= l->x && i < l->x + NXS) continue;
                if (isfinite(M->lb[i])) { double c = fabs((v[i] - M->lb[i]) * zL[i] - mu); if (c > cm) cm = c; }
                if (isfinite(M->ub[i])) { double c = fabs((M->ub[i] - v[i]) * zU[i] - mu); if (c > cm) cm = c; }
            }
            double Emu = fmax(dinf / sd, fmax(pinf, cm / sc));
            if (Emu <= o->kappa_eps * mu && mu > o->tol / 10) { mu = fmax(o->tol / 10, fmin(o->kappa_mu * mu, pow(mu, o->theta_mu))); tau = fmax(o->tau_min, 1 - mu); nf = 0; }
            else break;
        }
        double dw = 0, dc = o->dc_bar * pow(mu, o->kappa_c); int ok = 0;
        for (int tr = 0; tr < 60; tr++) {
            int a = kkt_assemble(K, v, y, zL, zU, mu, dw, dc);
            stage_dual_inf(K, y);
            if (a) a = kkt_solve(K, v, dc, o->rho_term, dv, dy);
            if (a) { ok = 1; break; }
            nreg++;
            if (dw == 0) dw = dw_last == 0 ? o->dw0 : fmax(o->dw_min, o->kw_dec * dw_last); else dw *= (dw_last == 0 ? o->kw_inc0 : o->kw_inc);
            if (dw > o->dw_max) break;
        }
        if (!ok) {
            if (nrest < MAX_RESTORE) { RESTORE_AND_CONTINUE; }
            status = ST_ERROR; break;
        }
        if (dw > 0) dw_last = dw;
        double ap = 1, az = 1, gd = 0;
        {   /* bound multiplier steps, fraction to the boundary, directional derivative of the barrier function */
            double t = v[l->t], wsum = 0;
            for (int i = 0; i < n; i++) {
                dzL[i] = dzU[i] = 0;
                if (i >= l->x && i < l->x + NXS) continue;
                if (isfinite(M->lb[i])) { double d = v[i] - M->lb[i]; dzL[i] = mu / d - zL[i] - zL[i] / d * dv[i]; gd -= M->mult[i] * mu / d * dv[i];
                    if (dv[i] < 0) ap = fmin(ap, -tau * d / dv[i]); if (dzL[i] < 0) az = fmin(az, -tau * zL[i] / dzL[i]); }
                if (isfinite(M->ub[i])) { double d = M->ub[i] - v[i]; dzU[i] = mu / d - zU[i] + zU[i] / d * dv[i]; gd += M->mult[i] * mu / d * dv[i];
                    if (dv[i] > 0) ap = fmin(ap, tau * d / dv[i]); if (dzU[i] < 0) az = fmin(az, -tau * zU[i] / dzU[i]); }
            }
            (void)wsum;
            for (int k = 0; k < N; k++) for (int j = 0; j < NU; j++) { const double *u = v + l->u + NU * k; double gu = -sf * 2e-3 * (p->wH - u[j]);
                if (k >= 1) { double e = u[j - NU] - u[j]; gu -= sf * 2e-2 * e; gd += sf * 2e-2 * e * dv[l->u + NU * (k - 1) + j]; }
                gd += gu * dv[l->u + NU * k + j]; }
            gd += sf * (N + 1) * (0.25 + 10 * t) * dv[l->t];
            for (int k = 0; k <= N; k++) { for (int i = 9; i < 12; i++) gd += sf * 2e-4 * v[l->x + NXS * k + i] * dv[l->x + NXS * k + i];
                for (int j = 0; j < NOB; j++) { int bo = k * NOB + j; if (!p->dist) gd += sf * (1e2 + 2e3 * v[l->s + bo]) * dv[l->s + bo];
                    for (int i = 0; i < NL; i++) gd += sf * 2e-4 * v[l->lam + NL * bo + i] * dv[l->lam + NL * bo + i]; } }
        }
        double phi = f - mu * barrier_sum(M, v), amin;
        if (gd < 0) { amin = fmin(o->gamma_theta, o->gamma_phi * th / (-gd)); if (th <= th_min) amin = fmin(amin, o->delta * pow(th, o->s_theta) / pow(-gd, o->s_phi)); }
        else amin = o->gamma_theta;
        amin *= o->gamma_alpha;
        double alpha = ap; int acc = 0;
        while (alpha >= amin) {
            for (int i = 0; i < n; i++) vt[i] = v[i] + alpha * dv[i];
            double ft, tht, thi; eval_f_theta(M, vt, &ft, &tht, &thi);
            if (ft == ft && tht == tht && tht < th_max) {
                double pht = ft - mu * barrier_sum(M, vt); int okf = (pht == pht);
                for (int i = 0; i < nf && okf; i++) if (!(tht < filt[i][0] || pht < filt[i][1])) okf = 0;
                if (okf) {
                    int sw = gd < 0 && alpha * pow(-gd, o->s_phi) > o->delta * pow(th, o->s_theta), armijo = pht <= phi + o->eta_phi * alpha * gd;
                    if (th <= th_min && sw) { if (armijo) { acc = 1; break; } }
                    else if (tht <= (1 - o->gamma_theta) * th || pht <= phi - o->gamma_phi * th) { acc = 1;
                        if (!(sw && armijo) && nf < FILT_MAX) { filt[nf][0] = (1 - o->gamma_theta) * th; filt[nf][1] = phi - o->gamma_phi * th; nf++; } break; }
                }
            }
            /* second-order correction (IPOPT A-5.5..A-5.9, kappa_soc = 0.99): after a rejected FIRST trial step that did not reduce theta; option max_soc */
            if (o->max_soc > 0 && alpha == ap && ft == ft && tht == tht && tht >= th) {
                double *cs = xcalloc(m, 8), *ct = xcalloc(m, 8), *dvs = xcalloc(n, 8), *dys = xcalloc(m, 8);
                double th_old = 0, th_tr = tht, asoc = alpha, azs = az;
                constraint_values(M, v, cs);
                for (int ps = 0; ps < o->max_soc && !acc && (ps == 0 || th_tr <= 0.99 * th_old); ps++) {
                    th_old = th_tr;
                    constraint_values(M, vt, ct);
                    for (int i = 0; i < m; i++) cs[i] = asoc * cs[i] + ct[i];
                    K->csoc = cs;
                    int a = kkt_assemble(K, v, y, zL, zU, mu, dw, dc);
                    stage_dual_inf(K, y);
                    if (a) a = kkt_solve(K, v, dc, o->rho_term, dvs, dys);
                    K->csoc = NULL;
                    if (!a) break;
                    asoc = 1; azs = 1;
                    for (int i = 0; i < n; i++) {
                        if (i >= l->x && i < l->x + NXS) continue;
                        if (isfinite(M->lb[i])) { double d = v[i] - M->lb[i], dz = mu / d - zL[i] - zL[i] / d * dvs[i]; if (dvs[i] < 0) asoc = fmin(asoc, -tau * d / dvs[i]); if (dz < 0) azs = fmin(azs, -tau * zL[i] / dz); }
                        if (isfinite(M->ub[i])) { double d = M->ub[i] - v[i], dz = mu / d - zU[i] + zU[i] / d * dvs[i]; if (dvs[i] > 0) asoc = fmin(asoc, tau * d / dvs[i]); if (dz < 0) azs = fmin(azs, -tau * zU[i] / dz); }
                    }
                    for (int i = 0; i < n; i++) vt[i] = v[i] + asoc * dvs[i];
                    eval_f_theta(M, vt, &ft, &tht, &thi); g_nsoc++;
                    if (!(ft == ft && tht == tht)) break;
                    th_tr = tht;
                    if (tht < th_max) {
                        double pht = ft - mu * barrier_sum(M, vt); int okf = (pht == pht);
                        for (int i = 0; i < nf && okf; i++) if (!(tht < filt[i][0] || pht < filt[i][1])) okf = 0;
                        if (okf) {
                            int sw = gd < 0 && alpha * pow(-gd, o->s_phi) > o->delta * pow(th, o->s_theta), armijo = pht <= phi + o->eta_phi * alpha * gd;
                            if (th <= th_min && sw) { if (armijo) acc = 1; }
                            else if (tht <= (1 - o->gamma_theta) * th || pht <= phi - o->gamma_phi * th) { acc = 1;
                                if (!(sw && armijo) && nf < FILT_MAX) { filt[nf][0] = (1 - o->gamma_theta) * th; filt[nf][1] = phi - o->gamma_phi * th; nf++; } }
                        }
                    }
                    if (acc) {      /* the correction is the step: its multiplier steps with it */
                        memcpy(dv, dvs, sizeof(double) * n); memcpy(dy, dys, sizeof(double) * m); alpha = asoc; az = azs; g_nsoc_acc++;
                        for (int i = 0; i < n; i++) { dzL[i] = dzU[i] = 0; if (i >= l->x && i < l->x + NXS) continue;
                            if (isfinite(M->lb[i])) { double d = v[i] - M->lb[i]; dzL[i] = mu / d - zL[i] - zL[i] / d * dv[i]; }
                            if (isfinite(M->ub[i])) { double d = M->ub[i] - v[i]; dzU[i] = mu / d - zU[i] + zU[i] / d * dv[i]; } }
                    }
                }
                free(cs); free(ct); free(dvs); free(dys);
                if (acc) break;
                kkt_assemble(K, v, y, zL, zU, mu, dw, dc); stage_dual_inf(K, y);      /* (the workspace holds the correction system: back to the iteration's own, as the block records are used again) */
            }
            alpha *= 0.5;
        }
        if (!acc) {
            if (nrest < MAX_RESTORE) { RESTORE_AND_CONTINUE; }
            status = ST_ERROR; break;
        }
        double ay = fmin(alpha, az);
        for (int i = 0; i < n; i++) v[i] += alpha * dv[i];
        for (int i = 0; i < m; i++) y[i] += ay * dy[i];
        for (int i = 0; i < n; i++) {
            if (i >= l->x && i < l->x + NXS) continue;
            if (isfinite(M->lb[i])) { double d = v[i] - M->lb[i], z = zL[i] + az * dzL[i], lo = mu / (o->kappa_sigma * d), hi = o->kappa_sigma * mu / d; zL[i] = z < lo ? lo : (z > hi ? hi : z); }
            if (isfinite(M->ub[i])) { double d = M->ub[i] - v[i], z = zU[i] + az * dzU[i], lo = mu / (o->kappa_sigma * d), hi = o->kappa_sigma * mu / d; zU[i] = z < lo ? lo : (z > hi ? hi : z); }
        }
        it++;
        next_iter:;
    }
    eval_f_theta(M, v, &f, &th, &thinf);
    res->status = status; res->iters = it; res->nreg = nreg; res->obj = f / M->sf; res->pinf = last_pinf; res->dinf = last_dinf / M->sf; res->mu = mu; res->t = v[l->t];
    free(dv); free(dy); free(dzL); free(dzU); free(vt); kkt_free(K);
}

/* -------------------------------------------------------------- C entry points (ctypes) */
static void setup_prob(prob_t *p, int N, double Ts, double R, const double *x0, const double *xF, const double *ob /* 5 x 6 */, int dist) {
    memset(p, 0, sizeof *p); p->dist = dist;
    p->N = N; p->Ts = Ts; p->R = R;
    memcpy(p->x0, x0, sizeof p->x0); memcpy(p->xF, xF, sizeof p->xF); memcpy(p->ob, ob, sizeof p->ob);
    p->wH = sqrt((MASS * GRAV) / (KF * 4));                    /* :62 */
    for (int i = 0; i < 3; i++) p->gyro[i] = x0[9 + i];        /* x[10], x[11], x[12] with a single index = stage 1 */
}

int obca_oracle_quad_layout(int N, int *out /* 11 ints */) { lay_t l; make_layout(N, &l); memcpy(out, &l, sizeof l); return (int)(sizeof l / sizeof(int)); }

/*
 * Dual warm start (the quadcopter analogue of DualMultWS.jl): the reference starts every lambda at 0.05 (:204-208), where
 * A'lambda = 0, so the gradient of the row |A'lambda|^2 == 1 vanishes and the constraint Jacobian is rank deficient; IPOPT gets
 * away from that point through its restoration phase (the authors mention its messages, mainQuadcopter.jl:140); here the block
 * restoration above plays that part.  With dual_ws != 0 the lambdas start at the closed-form dual solution of the point-to-box distance at
 * the warm-start position: q = unit vector from the box to the point (or the least-penetration face normal inside the box),
 * lambda = [max(q,0); max(-q,0)], so that |A'lambda| = 1 and the separation row equals the (signed) distance.
 */
static void quad_dual_ws(const prob_t *p, const lay_t *l, double *v) {
    for (int k = 0; k <= p->N; k++)
        for (int j = 0; j < NOB; j++) dual_ws_block(p, j, v + l->x + NXS * k, v + l->lam + NL * (k * NOB + j));
}

/*
 * QuadcopterSignedDist(x0,xF,N,Ts,R,ob1..ob5,xWS,uWS,timeWS)  (QuadcopterSignedDist.jl:25).
 * xWS 12 x (N+1) stage-contiguous; uWS is ignored as in the reference (:202); ob: 5 x 6 [xmax,ymax,zmax,-xmin,-ymin,-zmin].
 * outputs xp 12(N+1), up 4N, tsp (N+1), lp 30 x (N+1) stage-contiguous (rows l1..l5 stacked, :295), slack 5(N+1);
 * exitflag 0/1/2 (:229-234, :285-288); info[8] = {status, iterations, objective, pinf, dinf, mu, nreg, t}.
 */
static double *g_qfull = NULL;      /* test hook: v[n] | y[m] | zL[n] | zU[n] of the final iterate */
int obca_oracle_quadcopter_signed_dist(int N, double Ts, double R, const double *x0, const double *xF, const double *ob,
                                       const double *xWS, double timeWS, int flags /* bit 0: dual warm start, bit 1: QuadcopterDist */, const opts_t *opt, double *xp, double *up, double *tsp,
                                       double *lp, double *slp, int *exitflag, double *info) {
    prob_t p; model_t M; opts_t o;
    if (opt) o = *opt; else obca_oracle_quad_default_opts(&o);
    const int dual_ws = flags & 1, dist = (flags >> 1) & 1;
    setup_prob(&p, N, Ts, R, x0, xF, ob, dist); model_init(&M, &p);
    const lay_t *l = &M.l; int n = l->n, m = l->m;
    double *v = xcalloc(n, 8), *y = xcalloc(m, 8), *zL = xcalloc(n, 8), *zU = xcalloc(n, 8);
    memcpy(v + l->x, xWS, sizeof(double) * NXS * (N + 1));                  /* :201 */
    for (int i = 0; i < NU * N; i++) v[l->u + i] = p.wH;                    /* :202 */
    v[l->t] = timeWS;                                                       /* :199 */
    for (int i = 0; i < NL * NOB * (N + 1); i++) v[l->lam + i] = 0.05;      /* :204-208 */
    if (dual_ws) quad_dual_ws(&p, l, v);
    for (int i = 0; i < NOB * (N + 1); i++) v[l->s + i] = dist ? 0.0 : 1.0; /* :210 */
    if (o.obj_scaling) { double gm = objective_gradient_max(&M, v); M.sf = gm > 100.0 ? 100.0 / gm : 1.0; }      /* nlp_scaling_method = gradient-based, nlp_scaling_max_gradient = 100 */
    result_t r; ipm_solve(&M, &o, v, y, zL, zU, &r);
    int ef = r.status == ST_OPTIMAL ? 1 : 0;                                /* flag = 1 branch, :229-234 */
    double ssum = 0; for (int i = 0; i < NOB * (N + 1); i++) ssum += v[l->s + i];
    if (!dist && ef == 1 && ssum > 1e-3) ef = 2;                            /* :285-288 (QuadcopterDist has no such check) */
    memcpy(xp, v + l->x, sizeof(double) * NXS * (N + 1)); memcpy(up, v + l->u, sizeof(double) * NU * N);
    for (int k = 0; k <= N; k++) tsp[k] = v[l->t];
    memcpy(lp, v + l->lam, sizeof(double) * NL * NOB * (N + 1));
    if (slp) memcpy(slp, v + l->s, sizeof(double) * NOB * (N + 1));
    *exitflag = ef;
    if (info) { info[0] = r.status; info[1] = r.iters; info[2] = r.obj; info[3] = r.pinf; info[4] = r.dinf; info[5] = r.mu; info[6] = r.nreg; info[7] = r.t; }
    if (g_qfull) { memcpy(g_qfull, v, 8 * n); memcpy(g_qfull + n, y, 8 * m); memcpy(g_qfull + n + m, zL, 8 * n); memcpy(g_qfull + 2 * n + m, zU, 8 * n); }
    free(v); free(y); free(zL); free(zU); model_free(&M);
    return 0;
}
/* the same solve, additionally returning the full primal-dual iterate (independent optimality certificate, tests/golden/make_kkt_pin.py) */
int obca_oracle_quadcopter_signed_dist_full(int N, double Ts, double R, const double *x0, const double *xF, const double *ob, const double *xWS, double timeWS, int flags,
                                            const opts_t *opt, double *xp, double *up, double *tsp, double *lp, double *slp, int *exitflag, double *info, double *full) {
    g_qfull = full;
    int rc = obca_oracle_quadcopter_signed_dist(N, Ts, R, x0, xF, ob, xWS, timeWS, flags, opt, xp, up, tsp, lp, slp, exitflag, info);
    g_qfull = NULL;
    return rc;
}

/* test hook: one regularised Newton direction at a full primal-dual point */
/* the least-squares multiplier estimate at (v, zL, zU) (what lsq_init takes): tests pin it against a dense least-squares solve on the autograd Jacobian */
int obca_oracle_quad_lsq_multipliers(int N, double Ts, double R, const double *x0, const double *xF, const double *ob, const double *v,
                                     const double *zL, const double *zU, double *yls, int dist) {
    prob_t p; model_t M; setup_prob(&p, N, Ts, R, x0, xF, ob, dist); model_init(&M, &p);
    kkt_t *K = kkt_alloc(&M); const lay_t *l = &M.l;
    double *y0 = xcalloc(l->m, 8), *dv = xcalloc(l->n, 8);
    K->lsq = 1;
    int ok = kkt_assemble(K, v, y0, zL, zU, 0.0, 0, 0); stage_dual_inf(K, y0);
    if (ok) ok = kkt_solve(K, v, 0.0, 0.0, dv, yls);
    free(y0); free(dv); kkt_free(K); model_free(&M);
    return ok;
}
int obca_oracle_quad_newton(int N, double Ts, double R, const double *x0, const double *xF, const double *ob, const double *v,
                            const double *y, const double *zL, const double *zU, double mu, double dw, double dc, double rho,
                            double *dv, double *dy, double *errs, int dist) {
    prob_t p; model_t M; setup_prob(&p, N, Ts, R, x0, xF, ob, dist); model_init(&M, &p);
    kkt_t *K = kkt_alloc(&M);
    int ok = kkt_assemble(K, v, y, zL, zU, mu, dw, dc);
    double sdi = stage_dual_inf(K, y);
    if (errs) { errs[0] = fmax(K->dinf, sdi); errs[1] = K->pinf; errs[2] = K->cinf0; }
    if (ok) ok = kkt_solve(K, v, dc, rho, dv, dy);
    kkt_free(K); model_free(&M);
    return ok;
}
