// obca_quad_solver.h -- one quadcopter signed-distance NLP instance (QuadcopterSignedDist.jl)
// solved by ONE wavefront (64-thread workgroup), four instances per CU
// (round 1 / early round 2: two wavefronts per instance, two instances per CU --
// but the stage-parallel phases have 61 items and both sweeps run on one wavefront,
// so the second wavefront idled three quarters of the time; with the MFMA sweep an instance needs neither its lanes nor the LDS they came with).
//
// Same programming model, interior-point algorithm and phase structure as obca_solver.h (parking); what differs is the model
// (12 states, 4 rotor speeds, 5 box obstacles with 6 multipliers each, slack >= 0), hence the sizes: Riccati state 16 (x and the
// copy of u_{k-1}), 4 inputs, 14 right-hand sides (main, t, nu_1..12), a 13x13 (t, nu) border, and a dense pre-zeroed stage record.
#pragma once
#include "obca_solver.h"
#include "obca_quad_model.h"

namespace obca {
namespace quad {

#define QNT 64                       // threads per instance: one wavefront
#define QNMAX 128                    // longest horizon (the forward-sweep trajectory lives in LDS: (QNMAX + 2) x 16 doubles)
#define QSR 736                      // doubles per stage record: H 20x20 | Fh 16x18 | hc 20x2
#define QSR_H 0
#define QSR_F 400                    // Fh[a][c]: c<12 x columns (A), 12..15 u columns (B / identity), 16 = d, 17 = Ft
#define QSR_HC 688                   // hc[i][0] = gradient (barrier form), hc[i][1] = d/dt column
#define QRR 768                      // doubles per Riccati record
#define QRR_K 0                      // K 4x16
#define QRR_KF 64                    // feed-forward 4x14
#define QRR_PX 120                   // rows 0..11 of P (12x16)
#define QRR_PV 312                   // rows 0..11 of p (12x14)
#define QRR_CL 480                   // closed loop: Acl 16x16 then bcl 16
#define QFILT 224
#define QFC 18                       // columns of Fh
#define QQC (QZ + QC)                // columns of the extended matrix (34)

// ---------------------------------------------------------------- physical layout of the stage record
// The logical stage record is dense (QSR = 736 doubles: H 20x20 | Fh 16x18 | hc
// 20x2) but only 291 of its entries are ever non-zero.  In HBM it is stored PACKED
// (QSP = 288 doubles): QR(o) maps a logical offset o to its slot (structural zeros
// share one slot that holds 0, the constant 1 of the identity entries another).
// The kernels are limited by HBM traffic: the dense record cost 5.9 KB per stage and pass to read and -- written 8 bytes at a time into a
// sparse pattern -- 32 bytes per non-zero to write.  The -DOBCA_QUAD_RICCATI_LDS variant keeps the dense record (its sweep copies records into LDS wholesale).
#define QP_LOC 0        // H[v][v'] over the 10 local variables (angles, rates, inputs)
#define QP_POS 100      // H[0..2][0..2]
#define QP_VEL 109      // H[6..8] diagonal
#define QP_W 112        // H[w_j][w_j], then H[w_j][u_j], then H[u_j][w_j]
#define QP_FD 124       // d (12), Ft (12)
#define QP_TAU 148      // F[i][6 + i], i < 3
#define QP_ONE 151      // the 1 of every identity entry of F
#define QP_FLOC 152     // F[3..11][local columns]
#define QP_HC 242       // hc 20 x 2
#define QR_ZERO 282
#define QSP 288
OBCA_FN int q_vpos(int i) { return (i >= 3 && i < 6) ? i - 3 : ((i >= 9 && i < 12) ? i - 6 : ((i >= QS && i < QZ) ? i - 10 : -1)); }   // inverse of q_vidx
OBCA_FN int QR(int o) {
    if (o >= QSR_HC) return o < QSR_HC + 2 * QZ ? QP_HC + (o - QSR_HC) : QR_ZERO;
    if (o >= QSR_F) {
        const int i = (o - QSR_F) / QFC, cI = (o - QSR_F) % QFC;
        if (cI >= 16) return i < QX ? QP_FD + (cI - 16) * QX + i : QR_ZERO;
        const int id = cI < QX ? cI : cI + QU;                       // stage-vector index of the column
        if (i < 3) return cI == i ? QP_ONE : (cI == 6 + i ? QP_TAU + i : QR_ZERO);
        if (i < QX) { const int vp = q_vpos(id); return vp >= 0 ? QP_FLOC + (i - 3) * QV + vp : ((cI == i && i >= 6 && i < 9) ? QP_ONE : QR_ZERO); }
        return cI == i ? QP_ONE : QR_ZERO;
    }
    const int i = o / QZ, j = o % QZ, vi = q_vpos(i), vj = q_vpos(j);
    if (vi >= 0 && vj >= 0) return QP_LOC + vi * QV + vj;
    if (i < 3 && j < 3) return QP_POS + i * 3 + j;
    if (i == j && i >= 6 && i < 9) return QP_VEL + (i - 6);
    if (i >= QX && i < QS) return j == i ? QP_W + (i - QX) : (j == i + QU ? QP_W + 4 + (i - QX) : QR_ZERO);
    if (i >= QS && j == i - QU) return QP_W + 8 + (i - QS);
    return QR_ZERO;
}

#if defined(OBCA_EMU) && defined(OBCA_EMU_RACE)
#define QPAR(lane) for (int lane = 0; (race::lane = lane) < QNT; ++lane)      // (see PAR in obca_solver.h)
#define QNLT QNT
#elif defined(OBCA_EMU)
#define QPAR(lane) for (int lane = 0; lane < QNT; ++lane)
#define QNLT QNT
#else
#define QPAR(lane) PAR(lane)
#define QNLT 1
#endif
OBCA_FN double red_sum(const double *r) { return red_sum_t<QNT>(r); }      // (hide the one-wavefront reductions of the parking solver)
OBCA_FN double red_max(const double *r) { return red_max_t<QNT>(r); }
OBCA_FN double red_min(const double *r) { return red_min_t<QNT>(r); }

struct QLay { int x, u, t, lam, s, so, n, pi, nu, yo, m, zL, zU, len; };   // iterate buffer: v[n] | y[m] | zL[n] | zU[n]
OBCA_HD void q_make_layout(int N, QLay &l) {
    int o = 0, N1 = N + 1;
    l.x = o; o += QX * N1; l.u = o; o += QU * N; l.t = o; o += 1; l.lam = o; o += QL * QOB * N1; l.s = o; o += QOB * N1; l.so = o; o += QOB * N1; l.n = o;
    l.pi = o; o += QX * N; l.nu = o; o += QX; l.yo = o; o += 2 * QOB * N1; l.m = o - l.n;
    l.zL = o; o += l.n; l.zU = o; o += l.n; l.len = o;
}
// direction buffer: dv[n] then dy[m] (same offsets as v / y)

// d: the direction buffer in use (one of the two behind d0, see QCS)   // prob: Ts, R, x0[12], xF[12], ob[30], xWS..., see host packing
struct QInst { const gdbl *prob; gdbl *z, *d, *as, *rs, *oc, *d0; };
#define QPH_TS 0
#define QPH_R 1
#define QPH_X0 2
#define QPH_XF 14
#define QPH_OB 26
#define QPH_TWS 56
#define QPH_DWS 57
#define QPH_DIST 58
#define QPH_SIZE 64
// direction memory of an instance, behind d0: two direction buffers (dv | dy, n +
// m doubles each: a second-order correction is solved into the one the iteration's
// own direction is not in, so a correction that is rejected costs nothing but its own
// solve) and the rows of the correction (IPOPT A-5.7: c_soc = alpha c_soc + c(trial);
// m doubles, numbered like the multipliers)
#define QDIR(sh, which) ((sh).inst.d0 + (size_t)(which) * ((sh).l.n + (sh).l.m))
#define QCS(sh) QDIR(sh, 2)
#define QDIR_DOUBLES(l) (2 * ((l).n + (l).m) + (l).m)
#define QCS_PI(sh) ((sh).l.pi - (sh).l.n)
#define QCS_NU(sh) ((sh).l.nu - (sh).l.n)
#define QCS_YO(sh) ((sh).l.yo - (sh).l.n)

struct QShared {
    QConsts c; QLay l; QInst inst; AsmOut A, A2, Ap; StepOut S; double trial[4];
    double ob[QOB * QL];
    alignas(16) double red[16][QNT];   // reductions; during the sweeps the same memory holds That|Qhat or -- together with Pn, pn, sg behind it, which are dead
                                       // by then -- the forward-sweep ring (2 x QFW_CH x QFW_SZ = 2 016 doubles <= 16 QNT + 256 + 224 + 736)
    double Pn[QS * QS], pn[QS * QC], sg[QSR], Khat[QU * 30], Bm[QC * QC], sB[4 * QC], Lq[QU * QU];
    double bord[13 * 13 + 3 * 13], coef[QC];
#ifdef OBCA_EMU
    alignas(16) double traj[(QNMAX + 2) * (QS + QU)];
#endif
    double filt[QFILT][2];
    int ric_ok, bord_ok;
    // 1: the system being solved is a second-order correction's, the terminal right-hand side comes from QCS; 2: the least-squares multiplier system,
    int soc_on;
                                       // it is zero (the assembly / block phases have variants of their own)
    int hintl[QNT];                    // per lane: a box block whose inertia was wrong in an earlier assembly (-1: none), see q_block_bad
    double prof[16]; long long tlast;      // diagnostic per-phase cycle counters (-DOBCA_PROFILE)
};
// The forward-sweep trajectory ((N + 2) x 16 doubles) and, behind it, kf_k(coef)
// (N x 4) live in dynamic LDS behind the static block: the launch sizes it for the
// batch's horizon (N = 60: 9.9 KB, four instances per CU).  It is addressed through
// the array itself, never through a pointer kept in memory: a loaded pointer is
// "generic", its accesses become flat_load / flat_store, and a flat access waits for vmcnt(0) -- for every HBM gather in flight -- before it returns.
#ifdef OBCA_EMU
static QShared gq_sh;
#define QTRAJ(sh) ((sh).traj)
#else
__shared__ QShared gq_sh;
extern __shared__ __attribute__((aligned(16))) double gq_traj[];
#define QTRAJ(sh) gq_traj
#endif
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)
#define QPROF(id) do { long long now_ = clock64(); if (LANE0) { gq_sh.prof[id] += (double)(now_ - gq_sh.tlast); gq_sh.tlast = now_; } } while (0)
#else
#define QPROF(id) ((void)0)
#endif
enum { QPF_INIT = 0, QPF_ASM_OBS, QPF_ASM_STAGE, QPF_RIC, QPF_BORDER, QPF_CL, QPF_FWD, QPF_BS_STAGE, QPF_BS_OBS, QPF_TRIAL, QPF_APPLY, QPF_OTHER };

// bounds of primal variable i
struct QBnd { double lo, hi; int hasL, hasU; double mult; };
OBCA_FN QBnd q_bounds(const QLay &l, int N, int i, int dist) {
    QBnd b; b.lo = 0; b.hi = 0; b.hasL = 0; b.hasU = 0; b.mult = 1;
    if (i < l.u) { int k = i / QX, cI = i - k * QX; if (k >= 1) { b.lo = q_xlb(cI, dist); b.hi = q_xub(cI, dist); b.hasL = b.hasU = 1; } }
    else if (i < l.t) { b.lo = Q_ULO; b.hi = Q_UHI; b.hasL = b.hasU = 1; }
    else if (i == l.t) { b.lo = Q_TLO; b.hi = Q_THI; b.hasL = b.hasU = 1; b.mult = N + 1; }
    else if (!(dist && i >= l.s && i < l.so)) { b.hasL = 1; }      // lam, s, so >= 0 (QuadcopterDist: no slack variable, frozen at 0)
    return b;
}

OBCA_FN void q_load_obs(const QShared &sh, const gdbl *z, int k, int j, QObsIn &in) {
    const QLay &l = sh.l; const int bo = k * QOB + j;
#pragma unroll
    for (int i = 0; i < QL; i++) { in.b[i] = sh.ob[j * QL + i]; in.lam[i] = z[l.lam + QL * bo + i]; in.zl[i] = z[l.zL + l.lam + QL * bo + i]; }
    in.s = z[l.s + bo]; in.zs = z[l.zL + l.s + bo]; in.so = z[l.so + bo]; in.zso = z[l.zL + l.so + bo];
    in.y[0] = z[l.yo + 2 * bo]; in.y[1] = z[l.yo + 2 * bo + 1];
    in.p[0] = z[l.x + QX * k]; in.p[1] = z[l.x + QX * k + 1]; in.p[2] = z[l.x + QX * k + 2];
}

// ---------------------------------------------------------------- assemble, part (a): box blocks
// RHS = 1: the system of a second-order correction (constraint right-hand sides from QCS); 2: the least-squares multiplier system (q_obs_block<.., LSQ>).
template <int RHS>
                        // Variants of their own, so that the options cost the iterations nothing
OBCA_FN void q_assemble_obs(QShared &sh, double mu, double dw, double dc) {
    const QConsts &c = sh.c; const int N = c.N; const gdbl *z = sh.inst.z, *cs = QCS(sh);
    constexpr int soc = RHS == 1, LSQ = RHS == 2;
    QPAR(lane) {
        QObsStats st; st.dmax = st.pmax = st.cmax0 = st.sumz = st.sumy = 0; st.cmin = 1e300; st.cmax = -1e300; st.bad = 0;
        double fsl = 0, th = 0, bar = 0; int badit = -1;
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in);
            ObsCond cd;
            const int bad0 = st.bad;
            double crs[2] = {0, 0};
            if (soc) { crs[0] = cs[QCS_YO(sh) + 2 * it]; crs[1] = cs[QCS_YO(sh) + 2 * it + 1]; }
            q_obs_block<0, LSQ>(c, in, mu, dw, dc, &cd, &st, nullptr, nullptr, soc ? crs : nullptr);
            if (st.bad && !bad0) badit = it;
            gdbl *o = sh.inst.oc + (size_t)it * OB_OC;
#pragma unroll
            for (int i = 0; i < 6; i++) o[i] = cd.Hpp[i];
#pragma unroll
            for (int i = 0; i < 3; i++) { o[6 + i] = cd.gz[i]; o[9 + i] = cd.gcorr[i]; }
            if (!c.dist) fsl += 1e2 * in.s + 1e3 * in.s * in.s;
            double r[2], q[3]; q_obs_rows(c, in, r, q);
            th += fabs(r[0]) + fabs(r[1]);
#pragma unroll
            for (int i = 0; i < QL; i++) fsl += 1e-4 * in.lam[i] * in.lam[i];
            {   // barrier of the item as one log of the product of its eight distances (log_prod, obca_solver.h)
                double dd[QL + 2];
#pragma unroll
                for (int i = 0; i < QL; i++) dd[i] = in.lam[i];
                dd[QL] = c.dist ? 1.0 : in.s; dd[QL + 1] = in.so;
                bar += log_prod(dd);
            }
        }
        sh.red[0][lane] = st.dmax; sh.red[1][lane] = st.pmax; sh.red[2][lane] = st.cmax0; sh.red[3][lane] = st.cmin; sh.red[12][lane] = st.cmax;
        sh.red[4][lane] = st.sumz; sh.red[5][lane] = st.sumy; sh.red[6][lane] = fsl; sh.red[7][lane] = th;
        sh.red[8][lane] = bar; sh.red[9][lane] = st.bad ? 1.0 : 0.0;
        if (st.bad) sh.hintl[lane] = badit;
    }
    SYNC();
    AsmOut &P = sh.Ap;
    P.dinf = red_max(sh.red[0]); P.pinf = red_max(sh.red[1]); P.cinf0 = red_max(sh.red[2]); P.cmin = red_min(sh.red[3]); P.cmax = red_max(sh.red[12]);
    P.sumz = red_sum(sh.red[4]); P.sumy = red_sum(sh.red[5]); P.f = red_sum(sh.red[6]); P.th1 = red_sum(sh.red[7]);
    P.bar = red_sum(sh.red[8]);
    P.ok = !(red_max(sh.red[9]) > 0.5);
    SYNC();
}

// Inertia of remembered box blocks at a given delta_w.  IPOPT tries delta_w = 0 in
// every iteration and climbs a ladder of regularisations until the inertia is right;
// on this problem most iterations fail the first rung(s) at the block level (the
// norm row |A'lam|^2 == 1 makes a lambda block indefinite whenever its multiplier is
// negative), and mostly in blocks that failed an iteration earlier.  Every lane
// remembers the block it last saw fail (hintl) and re-tests just that block: if any of
// them still fails, the rung is known to fail without being assembled, and the solve moves up the ladder exactly as it would have.
OBCA_FN int q_block_bad(QShared &sh, double mu, double dw, double dc) {
    const QConsts &c = sh.c; const gdbl *z = sh.inst.z;
    QPAR(lane) {
        const int it = sh.hintl[lane];
        QObsStats st; st.dmax = st.pmax = st.cmax0 = st.sumz = st.sumy = 0; st.cmin = 1e300; st.cmax = -1e300; st.bad = 0;
        if (it >= 0) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in);
            q_obs_block<2>(c, in, mu, dw, dc, nullptr, &st, nullptr, nullptr);
        }
        sh.red[9][lane] = st.bad ? 1.0 : 0.0;
    }
    SYNC();
    const int bad = red_max(sh.red[9]) > 0.5;
    SYNC();
    return bad;
}

// ---------------------------------------------------------------- assemble, part (b): stages (writes the non-zeros of the dense record)
template <int RHS>
OBCA_FN void q_assemble_stage(QShared &sh, double mu, double dw, double dc, AsmOut &out) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; const gdbl *z = sh.inst.z;
    const double t = z[l.t], tau = t * c.Ts;
    const gdbl *cs = QCS(sh); constexpr int soc = RHS == 1, LSQ = RHS == 2;
    double dinf = sh.Ap.dinf, pinf = sh.Ap.pinf, c0 = sh.Ap.cinf0, cmn = sh.Ap.cmin, cmx = sh.Ap.cmax, sumz = sh.Ap.sumz, sumy = sh.Ap.sumy, f = sh.Ap.f,
           th1 = sh.Ap.th1, bar = sh.Ap.bar;
    const int ok = sh.Ap.ok;
    QPAR(lane) {
        double dmax = 0, pmax = 0, lc0 = 0, lcmn = 1e300, lcmx = -1e300, lsz = 0, lsy = 0, lf = 0, lth = 0, lbar = 0, lgtb = 0, lgtz = 0;
        for (int k = lane; k <= N; k += QNT) {
            gdbl *rec = sh.inst.as + (size_t)k * QSP;
            double x[QX], hz[QX], hb[QX], xd[QX], Hpos[6] = {0, 0, 0, 0, 0, 0};
            BarAcc ba, bb, bu; bar_init(ba); bar_init(bb); bar_init(bu);      // barrier distances: states 0..5, states 6..11, inputs
#pragma unroll
            for (int i = 0; i < QX; i++) {
                x[i] = z[l.x + QX * k + i];
                const double gx = i >= 9 ? c.sf * 2e-4 * x[i] : 0.0;
                hz[i] = gx; hb[i] = gx; xd[i] = LSQ ? 1.0 : (i >= 9 ? c.sf * 2e-4 : 0.0) + dw;
                if (i >= 9) lf += 1e-4 * x[i] * x[i];
                if (k >= 1) {
                    B2 b = bound2(x[i], q_xlb(i, c.dist), q_xub(i, c.dist), z[l.zL + l.x + QX * k + i], z[l.zU + l.x + QX * k + i], mu, 1, lc0, lcmn, lcmx, lsz);
                    if (LSQ) { b.Sig = 0; b.gb = b.gz; }
                    xd[i] += b.Sig; hz[i] += b.gz; hb[i] += b.gb; bar_mul(i < 6 ? ba : bb, x[i] - q_xlb(i, c.dist), q_xub(i, c.dist) - x[i]);
                }
            }
            for (int j = 0; j < QOB; j++) {
                const gdbl *o = sh.inst.oc + (size_t)(k * QOB + j) * OB_OC;
#pragma unroll
                for (int i = 0; i < 6; i++) Hpos[i] += o[i];
#pragma unroll
                for (int i = 0; i < 3; i++) { hz[i] += o[6 + i]; hb[i] += o[6 + i] - o[9 + i]; }
            }
            if (k == N) {
#pragma unroll
                for (int i = 0; i < QX; i++) {
                    const double e = fabs(x[i] - c.xF[i]); pmax = fmax(pmax, e); lth += e;
                    const double r = z[l.pi + QX * (N - 1) + i] + z[l.nu + i];
                    hz[i] += r; hb[i] += r; dmax = fmax(dmax, fabs(hz[i]));
                    lsy += fabs(z[l.nu + i]);
                    rec[QR(QSR_H + i * QZ + i)] = xd[i] + (i < 3 ? Hpos[i == 0 ? 0 : (i == 1 ? 3 : 5)] : 0.0);
                    rec[QR(QSR_HC + 2 * i)] = hb[i]; rec[QR(QSR_HC + 2 * i + 1)] = 0.0;
                }
                rec[QR(QSR_H + 0 * QZ + 1)] = Hpos[1]; rec[QR(QSR_H + 1 * QZ + 0)] = Hpos[1]; rec[QR(QSR_H + 0 * QZ + 2)] = Hpos[2];
                rec[QR(QSR_H + 2 * QZ + 0)] = Hpos[2];
                rec[QR(QSR_H + 1 * QZ + 2)] = Hpos[4]; rec[QR(QSR_H + 2 * QZ + 1)] = Hpos[4];
                lbar += bar_log(ba) + bar_log(bb);
                continue;
            }
            double u[QU], pi[QX], g[QX], dg[9][QV], HG[55];
#pragma unroll
            for (int j = 0; j < QU; j++) u[j] = z[l.u + QU * k + j];
#pragma unroll
            for (int i = 0; i < QX; i++) { pi[i] = z[l.pi + QX * k + i]; lsy += fabs(pi[i]); }
            // everything else the stage reads from the iterate, before the first store into
            // the record (a load behind a store that may alias waits for its own round trip)
            double xn[QX], pim[QX], um[QU], un[QU], zLu[QU], zUu[QU];
            const int km = k >= 1 ? k - 1 : 0, kn = k + 1 < N ? k + 1 : k;
#pragma unroll
            for (int i = 0; i < QX; i++) { xn[i] = z[l.x + QX * (k + 1) + i]; pim[i] = z[l.pi + QX * km + i]; }
#pragma unroll
            for (int j = 0; j < QU; j++) { um[j] = z[l.u + QU * km + j]; un[j] = z[l.u + QU * kn + j]; zLu[j] = z[l.zL + l.u + QU * k + j]; zUu[j] = z[l.zU + l.u + QU * k + j]; }
            double csr[QX];
#pragma unroll
            for (int i = 0; i < QX; i++) csr[i] = soc ? cs[QCS_PI(sh) + QX * k + i] : 0.0;
            dyn_g_derivs(c, x, u, pi, g, dg, HG);
            // residual, F columns d / Ft, A, B
#pragma unroll
            for (int i = 0; i < QX; i++) {
                const double r = xn[i] - x[i] - tau * g[i];
                pmax = fmax(pmax, fabs(r)); lth += fabs(r);
                rec[QR(QSR_F + i * QFC + 16)] = LSQ ? 0.0 : (soc ? -csr[i] : -r); rec[QR(QSR_F + i * QFC + 17)] = c.Ts * g[i];
            }
#pragma unroll
            for (int i = 0; i < 3; i++) rec[QR(QSR_F + i * QFC + 6 + i)] = tau;
#pragma unroll
            for (int i = 3; i < QX; i++)
#pragma unroll
                for (int a = 0; a < QV; a++) {
                    const int id = q_vidx(a);
                    if (id < QX) rec[QR(QSR_F + i * QFC + id)] = (id == i ? 1.0 : 0.0) + tau * dg[i - 3][a];
                    else rec[QR(QSR_F + i * QFC + 12 + (id - QS))] = tau * dg[i - 3][a];
                }
            // J^T pi for x_k, u_k, t ; Hessian cross terms with t
            double Ht[QZ];
#pragma unroll
            for (int i = 0; i < QZ; i++) Ht[i] = 0;
            double ATpi[QX];
#pragma unroll
            for (int i = 0; i < QX; i++) ATpi[i] = pi[i];
#pragma unroll
            for (int i = 0; i < 3; i++) { ATpi[6 + i] += tau * pi[i]; Ht[6 + i] += -c.Ts * pi[i]; }
            double BTpi[QU] = {0, 0, 0, 0}, gtl = 0;
#pragma unroll
            for (int a = 0; a < QV; a++) {
                double s_ = 0;
#pragma unroll
                for (int i = 3; i < QX; i++) s_ += dg[i - 3][a] * pi[i];
                const int id = q_vidx(a);
                if (id < QX) ATpi[id] += tau * s_; else BTpi[id - QS] += tau * s_;
                Ht[id] += -c.Ts * s_;
            }
#pragma unroll
            for (int i = 0; i < QX; i++) gtl += c.Ts * g[i] * pi[i];
            lgtz -= gtl; lgtb -= gtl;
#pragma unroll
            for (int i = 0; i < QX; i++) {
                const double r = (k >= 1 ? pim[i] : 0.0) - ATpi[i];
                hz[i] += r; hb[i] += r; if (k >= 1 && fabs(hz[i]) > dmax) dmax = fabs(hz[i]);
            }
            // inputs: costs, bounds, copy terms
            double hzu[QU], hbu[QU], hzw[QU], ud[QU], wn[QU];
#pragma unroll
            for (int j = 0; j < QU; j++) {
                B2 b = bound2(u[j], Q_ULO, Q_UHI, zLu[j], zUu[j], mu, 1, lc0, lcmn, lcmx, lsz);
                if (LSQ) { b.Sig = 0; b.gb = b.gz; }
                bar_mul(bu, u[j] - Q_ULO, Q_UHI - u[j]);
                const double w2 = c.sf * 2e-2;
                double gu = -c.sf * 2e-3 * (c.wH - u[j]), hu = c.sf * 2e-3; hzw[j] = 0;
                lf += 1e-3 * (c.wH - u[j]) * (c.wH - u[j]);
                if (k >= 1) { const double e = um[j] - u[j]; gu += -w2 * e; hu += w2; hzw[j] = w2 * e; lf += 1e-2 * e * e; }
                hzu[j] = gu + b.gz - BTpi[j]; hbu[j] = gu + b.gb - BTpi[j]; ud[j] = LSQ ? 1.0 : hu + b.Sig + dw;
                wn[j] = (k + 1 < N) ? w2 * (u[j] - un[j]) : 0.0;     // copy part living in stage k+1
                const double tot = hzu[j] + wn[j]; dmax = fmax(dmax, fabs(tot));
            }
            // write H: x diagonal + position block, local 10x10 block (-tau HG), w/u coupling
#pragma unroll
            for (int i = 0; i < QX; i++) rec[QR(QSR_H + i * QZ + i)] = xd[i] + (i < 3 ? Hpos[i == 0 ? 0 : (i == 1 ? 3 : 5)] : 0.0);
            rec[QR(QSR_H + 0 * QZ + 1)] = Hpos[1]; rec[QR(QSR_H + 1 * QZ + 0)] = Hpos[1]; rec[QR(QSR_H + 0 * QZ + 2)] = Hpos[2];
            rec[QR(QSR_H + 2 * QZ + 0)] = Hpos[2];
            rec[QR(QSR_H + 1 * QZ + 2)] = Hpos[4]; rec[QR(QSR_H + 2 * QZ + 1)] = Hpos[4];
#pragma unroll
            for (int a = 0; a < QV; a++)
#pragma unroll
                for (int b_ = 0; b_ < QV; b_++) {
                    const int ia = q_vidx(a), ib = q_vidx(b_);
                    double v = -tau * HG[q_pidx(a, b_)];
                    if (a == b_) v += (ia < QX) ? xd[ia] : ud[ia - QS];
                    rec[QR(QSR_H + ia * QZ + ib)] = v;
                }
#pragma unroll
            for (int j = 0; j < QU; j++) {
                const double ww = (k >= 1 && !LSQ) ? c.sf * 2e-2 : 0.0;
                rec[QR(QSR_H + (QX + j) * QZ + (QX + j))] = ww; rec[QR(QSR_H + (QX + j) * QZ + (QS + j))] = -ww;
                rec[QR(QSR_H + (QS + j) * QZ + (QX + j))] = -ww;
            }
            // gradients / t-columns
#pragma unroll
            for (int i = 0; i < QX; i++) { rec[QR(QSR_HC + 2 * i)] = hb[i]; rec[QR(QSR_HC + 2 * i + 1)] = Ht[i]; }
#pragma unroll
            for (int j = 0; j < QU; j++) {
                rec[QR(QSR_HC + 2 * (QX + j))] = hzw[j]; rec[QR(QSR_HC + 2 * (QX + j) + 1)] = 0.0;
                rec[QR(QSR_HC + 2 * (QS + j))] = hbu[j]; rec[QR(QSR_HC + 2 * (QS + j) + 1)] = Ht[QS + j];
            }
            lbar += bar_log(ba) + bar_log(bb) + bar_log(bu);
        }
        sh.red[0][lane] = dmax; sh.red[1][lane] = pmax; sh.red[2][lane] = lc0; sh.red[3][lane] = lcmn; sh.red[12][lane] = lcmx;
        sh.red[4][lane] = lsz; sh.red[5][lane] = lsy; sh.red[6][lane] = lf; sh.red[7][lane] = lth;
        sh.red[8][lane] = lbar; sh.red[10][lane] = lgtb; sh.red[11][lane] = lgtz;
    }
    SYNC();
    dinf = fmax(dinf, red_max(sh.red[0])); pinf = fmax(pinf, red_max(sh.red[1])); c0 = fmax(c0, red_max(sh.red[2])); cmn = fmin(cmn, red_min(sh.red[3]));
    cmx = fmax(cmx, red_max(sh.red[12]));
    sumz += red_sum(sh.red[4]); sumy += red_sum(sh.red[5]); f += red_sum(sh.red[6]); th1 += red_sum(sh.red[7]); bar += red_sum(sh.red[8]);
    double gtb = red_sum(sh.red[10]), gtz = red_sum(sh.red[11]);
    SYNC();
    double d0 = 0, d2 = 0;
    B2 b = bound2(t, Q_TLO, Q_THI, z[l.zL + l.t], z[l.zU + l.t], mu, N + 1, d0, cmn, cmx, d2);
    if (LSQ) b.gb = b.gz;
    c0 = fmax(c0, d0); sumz += (N + 1) * (fabs(z[l.zL + l.t]) + fabs(z[l.zU + l.t]));
    const double gf = c.sf * (N + 1) * (0.25 + 10 * t);
    gtb += gf + b.gb; gtz += gf + b.gz;
    f += (N + 1) * (0.25 * t + 5 * t * t); bar += (N + 1) * log((t - Q_TLO) * (Q_THI - t));
    dinf = fmax(dinf, fabs(gtz));
    out.ok = ok; out.dinf = dinf; out.pinf = pinf; out.cinf0 = c0; out.cmin = cmn; out.cmax = cmx; out.sumy = sumy; out.sumz = sumz;
    // (least-squares system: t stands for the N + 1 timeScale variables of the reference's model)
    out.f = c.sf * f; out.th1 = th1; out.bar = bar; out.Htt = LSQ ? (double)(N + 1) : c.sf * 10.0 * (N + 1) + b.Sig + dw; out.gtb = gtb;
    out.nb = 2 * QX * N + 2 * QU * N + 2 * (N + 1) + (QL + 2 - (c.dist ? 1 : 0)) * QOB * (N + 1);
    out.nm = QX * N + QX + 2 * QOB * (N + 1);
}

#define QRR_PAD 767                                  // unused slot of the Riccati record: target of dummy stores
// ---------------------------------------------------------------- Riccati backward sweep on the matrix cores
// (Round 1 ran the sweep as four LDS / VALU phases: bound by LDS bandwidth, every fp64
// FMA of its products read two operands from LDS, 10 k clocks per stage.)  Here the
// whole recursion of an instance runs on its wavefront (QNT = 64: the instance IS one wavefront; the WAVE0 sections below are
// written for any QNT) with 16 x 16 fp64 tiles in registers and v_mfma_f64_16x16x4_f64 (23 per stage); the stage record is gathered
// from HBM straight into operand layout (software-pipelined QMD stages ahead), nothing but the symmetrisation of P goes through LDS.
//   lane = 16 g + j.  wv_mfma(C, a, b): C[i][n] += sum_{k<4} a(lane (k, i)) * b(lane
//   (k, n));  accumulator register r of lane (g, j) = C[g + 4 r][j] ("D layout").
//   Register kb of a tile in D layout is the B operand of K-block kb (rows 4 kb .. 4 kb + 3), and the A operand of the TRANSPOSED tile.
// Tiles (rows x columns; x = 12 states, w = copy of u_{k-1} (4), u = 4 inputs, rhs = main, t, nu_1..12):
//   PD   (x,w) x (x,w)  value function, symmetric         pnD  (x,w) x rhs
//   FXD0 FX[:, x|u columns]   FXD1 FX[:, d|Ft] (columns 0, 1)         FX = [A B d Ft; 0 I 0 0] is the 16 x 18 block of the stage record
//   Th0 = PD FXD0                         (x,w) x (x|u)          Th1 = pnD + PD FXD1                        (x,w) x rhs
//   Q0  = H + FXD0' Th0                   (x,u) x (x|u)          Q1  = hc + FXD0' Th1                       (x,u) x rhs
//   (the rows of a tile follow the columns of FXD0: x in registers 0..2, u in register
//   3.  The rows / columns of the input copy w carry no product -- F has zero
//   columns there -- and are known in closed form: H[w_j][w_j] = ww, H[w_j][u_j] = -ww,
//   gradient hc[w]; they are written over register 3 once the u rows are used up.)
//   Quu = Q0[u rows][u columns]: LDL' (uniform);  every lane solves the gains of its own column (x | w columns and the rhs columns)
//   P' = Q0(x rows | closed-form w rows) + Q[., u] K    p' = Q1 + Q[., u] Kf    (Q[.,
//   u] = transpose of the u rows / the closed-form w rows), P' symmetrised through LDS
//   border constants  Bm += FXD1' (Th1 + pnD) + Q1[u rows]' Kf   (the static parts off_a.(P
//   off_b + p_b) + off_b.p_a -- the last one as its transpose, the tile is symmetrised
//   at the end -- and the gain part, all into one accumulator tile)
//   23 MFMAs per stage (a v_mfma_f64_16x16x4_f64 occupies the matrix pipe for 64 clocks on gfx950: fp64 matrix rate = fp64 vector rate).
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)
#define QSEG(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); seg[i] += (double)(t_ - segt); segt = t_; __builtin_amdgcn_sched_barrier(0); } while (0)      // diagnostic: where a stage of the sweep spends its clocks (prof[12..15])
#else
#define QSEG(i) ((void)0)
#endif
#define QMD 2                        // stages the gathers run ahead
#define QMG 18                       // gathers per lane and stage
struct QMPlan { int off[QMG]; };
OBCA_FN void qm_plan(int lane, QMPlan &p) {
    const int g = lane >> 4, j = lane & 15, Z = QR_ZERO;          // Z: the slot of the structural zeros (lanes without an element)
    const int col = j < QX ? j : j + QU;                    // column of the stage vector (x | u) that tile column j stands for
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        p.off[kb] = QR(QSR_F + (4 * kb + g) * QFC + j);                             // [A B; 0 I]: rows 4 kb + g, columns x (12) and u (4)
        p.off[4 + kb] = j < 2 ? QR(QSR_F + (4 * kb + g) * QFC + 16 + j) : Z;       // d, Ft
        if (kb < 3) {
            p.off[8 + kb] = QR(QSR_H + (g + 4 * kb) * QZ + col);                    // H rows x of register kb
            p.off[12 + kb] = j < 2 ? QR(QSR_HC + 2 * (g + 4 * kb) + j) : Z;
        }
    }
    p.off[11] = QR(QSR_H + (QS + g) * QZ + col);                                    // H rows u: register 3 of the tile (its w rows are known in closed form)
    p.off[15] = j < 2 ? QR(QSR_HC + 2 * (QS + g) + j) : Z;                          // hc rows u
    p.off[16] = j < 2 ? QR(QSR_HC + 2 * (QX + g) + j) : Z;                          // hc rows w
    p.off[17] = QR(QSR_H + QX * QZ + QX);                                           // ww = H[w_0][w_0] (uniform)
}
OBCA_FN void qm_gather(const gdbl *rec, const QMPlan &p, double (&v)[QMG]) {
#pragma unroll
    for (int e = 0; e < QMG; e++) v[e] = rec[p.off[e]];
}
template <int PIPE>
OBCA_FN int q_riccati_stage_mfma(QShared &sh, const int k, const QMPlan (&plan)[OBCA_NLT], double (&PD)[4][OBCA_NLT], double (&pnD)[4][OBCA_NLT], double (&BmD)[4][OBCA_NLT],
                                 double (&nv)[OBCA_NLT][QMD][QMG], const int slot, const double (*raw)[QMG], double (&seg)[4], long long &segt) {
    // Tiles (row = lane group + 4 register, column = lane & 15): Q0 = [H | .] + [A B;
    // 0 I]' Th0 and Q1 = hc + [A B; 0 I]' Th1 have the x rows in registers 0..2 and
    // the u rows in register 3; the rows of the input copy w carry no product (F has zero columns there) and are filled in closed form below.
    double FXD0[4][OBCA_NLT], FXD1[4][OBCA_NLT], Th0[4][OBCA_NLT], Th1[4][OBCA_NLT];
    double Q0[4][OBCA_NLT], Q1[4][OBCA_NLT], ww[OBCA_NLT], hw[OBCA_NLT];
    PAR64(lane) {
        const int L_ = LI(lane); const QMPlan &p = plan[L_];
        double v[QMG];
        // The prefetched operands are MOVED out of the slot's registers (a real v_mov:
        // a plain assignment is only a rename, the old values would stay live in the
        // slot's registers for the whole stage, the re-issued gathers would land elsewhere and the copy back at the loop edge would wait for them -- vmcnt(3)
        // after every pair of stages, measured) so that the gathers of the stage QMD ahead return straight into the registers they are consumed from.
#pragma unroll
        for (int e = 0; e < QMG; e++) {
#ifndef OBCA_EMU
            if (PIPE) asm volatile("v_mov_b64 %0, %1" : "=v"(v[e]) : "v"(nv[L_][slot][e]));
            else v[e] = raw[L_][e];
#else
            v[e] = PIPE ? nv[L_][slot][e] : raw[L_][e];
#endif
        }
        // re-issue the slot (clamped, unconditional)
        if (PIPE) { const int kl = k - QMD > 0 ? k - QMD : 0; qm_gather(sh.inst.as + (size_t)kl * QSP, p, nv[L_][slot]); }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            FXD0[r][L_] = v[r]; FXD1[r][L_] = v[4 + r];
            Q0[r][L_] = v[8 + r]; Q1[r][L_] = v[12 + r]; Th0[r][L_] = 0.0; Th1[r][L_] = pnD[r][L_];
        }
        hw[L_] = v[16]; ww[L_] = v[17];
    }
#pragma unroll
    for (int kb = 0; kb < 4; kb++) { wv_mfma(Th0, PD[kb], FXD0[kb]); wv_mfma(Th1, PD[kb], FXD1[kb]); }
    // static parts of the border constants, FXD1' Th1 + pnD' FXD1 (pnD: still the next
    // stage's p): the second product is the transpose of FXD1' pnD and the accumulator
    // tile is symmetrised when the sweep ends (sh.Bm = (B + B') / 2), so FXD1' (Th1 + pnD) carries both -- four products per stage instead of eight
    double Tb[4][OBCA_NLT];
    PAR64(lane) {
        const int L_ = LI(lane);
#pragma unroll
        for (int r = 0; r < 4; r++) Tb[r][L_] = Th1[r][L_] + pnD[r][L_];
    }
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        wv_mfma(Q0, FXD0[kb], Th0[kb]); wv_mfma(Q1, FXD0[kb], Th1[kb]);
        wv_mfma(BmD, FXD1[kb], Tb[kb]);
    }
    QSEG(0);
    // Quu = Q0[u rows][u columns]: lane (a, 12 + b) of register 3
    double Lq[QU * QU];
#pragma unroll
    for (int a = 0; a < QU; a++)
#pragma unroll
        for (int b_ = 0; b_ < QU; b_++) Lq[a * QU + b_] = WV_READLANE(Q0[3], 16 * a + QX + b_);
    const int ok = UNIFORM(ldl_fact<QU>(QU, Lq) ? 0 : 1);        // (no early exit; after a failed pivot the rest of the group runs on garbage)
    const double wwu = WV_READLANE(ww, 0);
    // the four u-row entries of every column: rows u_0..u_3 sit in lane groups 0..3 of register 3
    double c0[QU][OBCA_NLT], c1[QU][OBCA_NLT];
#pragma unroll
    for (int a = 0; a < QU; a++) { wv_shfl_group(c0[a], Q0[3], a); wv_shfl_group(c1[a], Q1[3], a); }
    QSEG(1);
    double Aq[OBCA_NLT], Bk0[OBCA_NLT], Bk1[OBCA_NLT], Sn[4][OBCA_NLT], Qu1[OBCA_NLT];
    gdbl *ro = sh.inst.rs + (size_t)k * QRR;
    PAR64(lane) {
        const int L_ = LI(lane), g = lane >> 4, j = lane & 15;
        double b0[QU], b1[QU];
#pragma unroll
        for (int a = 0; a < QU; a++) { b0[a] = j < QX ? -c0[a][L_] : (a == j - QX ? wwu : 0.0); b1[a] = -c1[a][L_]; }      // w columns: -Q[u][w_c] = ww e_c
        ldl_solve<QU>(QU, Lq, b0); ldl_solve<QU>(QU, Lq, b1);
        const double k0 = g == 0 ? b0[0] : (g == 1 ? b0[1] : (g == 2 ? b0[2] : b0[3])), k1 = g == 0 ? b1[0] : (g == 1 ? b1[1] : (g == 2 ? b1[2] : b1[3]));
        Bk0[L_] = k0; Bk1[L_] = j < QC ? k1 : 0.0;
        // Q[row][u_g]: transpose of the u rows for the x rows, closed form for the w rows
        Aq[L_] = j < QX ? Q0[3][L_] : (j - QX == g ? -wwu : 0.0);
#pragma unroll
        // w rows / columns of [H | .]: ww on the (w, w) diagonal, nothing else
        for (int r = 0; r < 4; r++) Sn[r][L_] = j < QX ? (r < 3 ? Q0[r][L_] : 0.0) : ((g + 4 * r) == j ? wwu : 0.0);
        // u rows of the right-hand sides go to the border constants, register 3 becomes the w rows (hc only)
        Qu1[L_] = Q1[3][L_]; Q1[3][L_] = hw[L_];
        ro[QRR_K + g * QS + j] = k0;                                              // gains: row g, column j of the 4 x 16 / 4 x 14 blocks
        ro[j < QC ? QRR_KF + g * QC + j : QRR_PAD] = k1;
    }
    QSEG(2);
    wv_mfma(Sn, Aq, Bk0); wv_mfma(Q1, Aq, Bk1); wv_mfma(BmD, Qu1, Bk1);
    // symmetrise the value function through LDS (without it round-off flipped the Quu > 0 inertia test on this badly scaled problem)
    double *tr = &sh.red[0][0];
    PAR64(lane) {
        const int L_ = LI(lane), g = lane >> 4, j = lane & 15;
#pragma unroll
        // row stride 17: the transposed read below would hit one LDS bank 16 times with stride 16
        for (int r = 0; r < 4; r++) tr[(g + 4 * r) * 17 + j] = Sn[r][L_];
    }
    LDS_SYNC();
    PAR64(lane) {
        const int L_ = LI(lane), g = lane >> 4, j = lane & 15;
        // the four transposed entries are read together (the compiler otherwise reads two, waits, works, reads two, waits: a second LDS round trip per stage)
        double t_[4];
#pragma unroll
        for (int r = 0; r < 4; r++) t_[r] = tr[j * 17 + (g + 4 * r)];
#ifndef OBCA_EMU
        asm volatile("" : "+v"(t_[0]), "+v"(t_[1]), "+v"(t_[2]), "+v"(t_[3]));
#endif
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double v = 0.5 * (Sn[r][L_] + t_[r]);
            PD[r][L_] = v; pnD[r][L_] = j < QC ? Q1[r][L_] : 0.0;
            if (r < 3) { ro[QRR_PX + (g + 4 * r) * QS + j] = v; ro[j < QC ? QRR_PV + (g + 4 * r) * QC + j : QRR_PAD] = pnD[r][L_]; }      // rows 0..11 of P / p
        }
    }
    LDS_SYNC();
    QSEG(3);
    return ok;
}

OBCA_FN int q_riccati_body_mfma(QShared &sh, double rho) {      // wavefront 0
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = UNIFORM(c.N); const gdbl *z = sh.inst.z;
    int ok = 1;
    WAVE0_BEGIN
        double nv[OBCA_NLT][QMD][QMG], raw[OBCA_NLT][QMG], PD[4][OBCA_NLT], pnD[4][OBCA_NLT], BmD[4][OBCA_NLT];
        QMPlan plan[OBCA_NLT];
        double seg[4] = {0, 0, 0, 0}; long long segt = 0;
        PAR64(lane) {   // terminal cost-to-go: P_N = H_N[x, x] + rho I, p_N = (hb_N - rho e, 0, e_i)
            const int L_ = LI(lane), g = lane >> 4, j = lane & 15;
            qm_plan(lane, plan[L_]);
            const gdbl *rec = sh.inst.as + (size_t)N * QSP;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = g + 4 * r; double v = 0.0, w = 0.0;
                if (i < QX && j < QX) { v = rec[QR(QSR_H + i * QZ + j)]; if (i == j) v += rho; }
                if (i < QX) { if (j == 0) w = rec[QR(QSR_HC + 2 * i)] - rho * (sh.soc_on == 2 ? 0.0 : (sh.soc_on ? -QCS(sh)[QCS_NU(sh) + i] : -(z[l.x + QX * N + i] - c.xF[i]))); else if (j >= 2 && j < QC) w = (j - 2 == i) ? 1.0 : 0.0; }
                PD[r][L_] = v; pnD[r][L_] = w; BmD[r][L_] = 0.0;
            }
        }
        int k = N - 1, fin = 0;
        for (; k >= 0 && (k + 1) % QMD != 0; k--) {      // head: synchronous gathers until the remaining stage count is a multiple of QMD
            PAR64(lane) { qm_gather(sh.inst.as + (size_t)k * QSP, plan[LI(lane)], raw[LI(lane)]); }
            if (!q_riccati_stage_mfma<0>(sh, k, plan, PD, pnD, BmD, nv, 0, raw, seg, segt)) { ok = 0; fin = 1; break; }
        }
        if (!fin && k >= 0) {
            PAR64(lane) {
#pragma unroll
                for (int ju = 0; ju < QMD; ju++) { const int st = k - ju > 0 ? k - ju : 0; qm_gather(sh.inst.as + (size_t)st * QSP, plan[LI(lane)], nv[LI(lane)][ju]); }
#ifndef OBCA_EMU
#pragma unroll
                for (int ju = 0; ju < QMD; ju++)
#pragma unroll
                    for (int e = 0; e < QMG; e++) asm volatile("" : "+v"(nv[0][ju][e]));
#endif
            }
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)
            segt = clock64();
#endif
            for (int kb = k; kb >= QMD - 1 && ok; kb -= QMD) {
#pragma unroll
                for (int ju = 0; ju < QMD; ju++) ok &= q_riccati_stage_mfma<1>(sh, kb - ju, plan, PD, pnD, BmD, nv, ju, nullptr, seg, segt);
            }
        }
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)
        if (LANE0) { for (int i = 0; i < 4; i++) sh.prof[12 + i] += seg[i]; }
#endif
        // border constants to LDS, exactly symmetric (the accumulator tile is symmetric up to round-off)
        double *tr = &sh.red[0][0];
        PAR64(lane) {
            const int L_ = LI(lane), g = lane >> 4, j = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; r++) tr[(g + 4 * r) * 16 + j] = BmD[r][L_];
        }
        LDS_SYNC();
        PAR64(lane) {
            const int g = lane >> 4, j = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; r++) { const int a = g + 4 * r; if (a < QC && j < QC) sh.Bm[a * QC + j] = 0.5 * (tr[a * 16 + j] + tr[j * 16 + a]); }
        }
    WAVE0_END
    return ok;
}

OBCA_FN int q_riccati_backward(QShared &sh, double rho) {
    const int ok = q_riccati_body_mfma(sh, rho);      // (meaningful on wavefront 0 only: it publishes the flag)
    WAVE0_BEGIN
        PAR64(lane) { if (lane == 0) sh.ric_ok = ok; }
    WAVE0_END
    SYNC();
    return sh.ric_ok;
}

// ---------------------------------------------------------------- border, forward sweep, stage-parallel back-substitution
OBCA_FN void q_direction_main(QShared &sh, const AsmOut &A, double mu, double dw, double dc, double rho, double tau, StepOut &so) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; const gdbl *z = sh.inst.z; gdbl *d = sh.inst.d;
    // ---- 13x13 border in (dt, nu) from the bilinear constants: eliminate nu (S = -B(nu,nu) must be PD) then t.  Wavefront 0.
    WAVE0_BEGIN
        double *S = sh.bord, *col = sh.bord + 169, *colr = col + 13;   // S 12x12 (stride 12)
        PAR64(lane) {
            for (int it = lane; it < 144; it += 64) { int a = it / 12, b_ = it % 12; S[it] = -sh.Bm[(2 + a) * QC + (2 + b_)]; }
            if (lane < 12) { col[lane] = -sh.Bm[(2 + lane) * QC + 1]; colr[lane] = -((sh.soc_on == 2 ? 0.0 : (sh.soc_on ? -QCS(sh)[QCS_NU(sh) + lane] : -(z[l.x + QX * N + lane] - c.xF[lane]))) - sh.Bm[(2 + lane) * QC + 0]); }
            if (lane == 0) sh.bord_ok = 1;
        }
        LDS_SYNC();
        for (int j = 0; j < 12; j++) {   // cooperative LDL^T (L in the strict lower triangle, D on the diagonal)
            PAR64(lane) {
                if (lane == 0) { double dj = S[j * 12 + j]; for (int kk = 0; kk < j; kk++) dj -= S[j * 12 + kk] * S[j * 12 + kk] * S[kk * 12 + kk]; if (!(dj > 0)) sh.bord_ok = 0; S[j * 12 + j] = dj; }
            }
            LDS_SYNC();
            PAR64(lane) {
                if (lane > j && lane < 12) { double s_ = S[lane * 12 + j]; for (int kk = 0; kk < j; kk++) s_ -= S[lane * 12 + kk] * S[j * 12 + kk] * S[kk * 12 + kk]; S[lane * 12 + j] = s_ / S[j * 12 + j]; }
            }
            LDS_SYNC();
        }
        PAR64(lane) {
            if (lane < 2) {   // two right-hand sides
                double *b_ = lane ? colr : col;
                for (int i = 0; i < 12; i++) for (int kk = 0; kk < i; kk++) b_[i] -= S[i * 12 + kk] * b_[kk];
                for (int i = 0; i < 12; i++) b_[i] /= S[i * 12 + i];
                for (int i = 11; i >= 0; i--) for (int kk = i + 1; kk < 12; kk++) b_[i] -= S[kk * 12 + i] * b_[kk];
            }
        }
        LDS_SYNC();
        PAR64(lane) {
            if (lane == 0) {
                double piv = A.Htt + sh.Bm[1 * QC + 1], rr = -A.gtb - sh.Bm[1 * QC + 0];
                for (int a = 0; a < 12; a++) { piv -= sh.Bm[1 * QC + 2 + a] * col[a]; rr -= sh.Bm[1 * QC + 2 + a] * colr[a]; }
                if (!(piv > 0)) sh.bord_ok = 0;
                const double dt = rr / piv;
                sh.coef[0] = 1.0; sh.coef[1] = dt;
                for (int a = 0; a < 12; a++) sh.coef[2 + a] = colr[a] - col[a] * dt;
            }
        }
    WAVE0_END
    SYNC();
    QPROF(QPF_BORDER);
    {   // stored ONCE: every lane writes the shared slot, it must never hold an intermediate value
        const int ok = sh.bord_ok; so.ok = ok;
        if (!ok) return;
    }
    const double dt = sh.coef[1];
    // ---- forward recursion:  u_k = K_k s_k + kf_k(coef),  x_{k+1} = A_k x_k + B_k u_k
    // + d_k + dt Ft_k,  w_{k+1} = u_k  (s = (x, w); no closed-loop matrices are formed).
    // A chain of N dependent steps, so what counts is the latency of one step.  Lane 4
    // i + c owns chunk c (four terms) of row i: the 16 terms of a gain row / the 12 + 4
    // terms of a state row are summed inside a quad of lanes (two DPP exchanges), the
    // four inputs reach every lane through v_readlane, and the new state goes to the LDS
    // trajectory -- one LDS round trip per stage, every LDS operand a 16-byte read.  What
    // does not depend on the state is taken out of the chain: kf_k(coef) for all stages
    // is formed stage-parallel beforehand (LDS, behind the trajectory), and the stage
    // data (the first 12 rows of the dense FX block, 216 doubles of the stage record, and
    // the gains K, 64 doubles of the Riccati record) is gathered from HBM QFWD stages
    // ahead, five values per lane through offsets tabulated once per sweep, into a
    // double-buffered LDS slot.  (Round 2: two LDS phases per stage with 16- and 18-term
    // sums on 8 + 12 lanes, 8-byte LDS reads, record offsets recomputed for every
    // gathered value, chunks of three stages gathered one chunk ahead: 1 800 clocks per stage alone, 2 700 with four instances per CU.)
#ifndef QFWD
#define QFWD 6                       // stages the gathers run ahead
#endif
#define QFW_F 216
#define QFW_SZ (QFW_F + QU * QS)     // 280 values per stage
#define QFW_SLOT 288                 // slot stride (16-byte aligned; [280, 288) is the pad the lanes without a fifth value write to)
#define QFW_PER 5
    {
        double *ring = &sh.red[0][0];
        double *kfc = QTRAJ(sh) + (size_t)(N + 2) * QS;          // kf_k(coef): N x 4, behind the trajectory (dynamic LDS)
        // per lane: where its values of a stage live (>= 0: stage record, < 0: -1 - offset in the Riccati record)
        int fo[OBCA_NL][QFW_PER];
        double nvf[OBCA_NL][QFWD][QFW_PER];
        QPAR(lane) {
            const int L_ = LI(lane);
#pragma unroll
            for (int r = 0; r < QFW_PER; r++) { const int e = lane + 64 * r; fo[L_][r] = e < QFW_F ? QR(QSR_F + e) : (e < QFW_SZ ? -1 - (QRR_K + (e - QFW_F)) : QR(QSR_F)); }
            for (int it = lane; it < QU * N; it += QNT) {
                const gdbl *kf = sh.inst.rs + (size_t)(it >> 2) * QRR + QRR_KF + (it & 3) * QC;
                double a0 = 0, a1 = 0;
#pragma unroll
                for (int cc = 0; cc < QC; cc += 2) { a0 = fma(kf[cc], sh.coef[cc], a0); a1 = fma(kf[cc + 1], sh.coef[cc + 1], a1); }
                kfc[it] = a0 + a1;
            }
            if (lane < QS) QTRAJ(sh)[lane] = 0.0;
#define QFW_LOAD(st_, dst_) { const int sc_ = (st_) < N ? (st_) : N - 1; const gdbl *as_ = sh.inst.as + (size_t)sc_ * QSP, *rs_ = sh.inst.rs + (size_t)sc_ * QRR; \
                              _Pragma("unroll") for (int r = 0; r < QFW_PER; r++) { const int o_ = fo[L_][r]; (dst_)[r] = o_ >= 0 ? as_[o_] : rs_[-1 - o_]; } }
#define QFW_STORE(st_, src_) { double *sl_ = ring + (size_t)((st_) & 1) * QFW_SLOT; \
                               _Pragma("unroll") for (int r = 0; r < QFW_PER; r++) { const int e_ = lane + 64 * r; sl_[e_ < QFW_SZ ? e_ : QFW_SZ + (lane & 7)] = (src_)[r]; } }
            { double v0[QFW_PER]; QFW_LOAD(0, v0); QFW_STORE(0, v0); }
#pragma unroll
            for (int j = 0; j < QFWD; j++) QFW_LOAD(1 + j, nvf[L_][(1 + j) % QFWD]);
        }
        LDS_SYNC();
        QPROF(QPF_CL);
        for (int kb = 0; kb < N; kb += QFWD) {
#pragma unroll
            for (int ju = 0; ju < QFWD; ju++) {
                const int k = kb + ju;
                if (k < N) {
                    const double *rec = ring + (size_t)(k & 1) * QFW_SLOT, *s_ = QTRAJ(sh) + (size_t)k * QS;
                    double fa[OBCA_NL][4], pK[OBCA_NL], t1[OBCA_NL], cst[OBCA_NL], uq[OBCA_NL], p2[OBCA_NL], xs[OBCA_NL];
                    QPAR(lane) {
                        const int L_ = LI(lane), i = lane >> 2, cI = lane & 3, ir = i < QX ? i : 0;
                        double sv[4], ka[4];
                        ldv<4>(s_ + 4 * cI, sv); ldv<4>(rec + ir * QFC + 4 * cI, fa[L_]); ldv<4>(rec + QFW_F + (i & 3) * QS + 4 * cI, ka);
                        cst[L_] = rec[ir * QFC + 16] + dt * rec[ir * QFC + 17];
                        pK[L_] = dot4_tree(cI == 0 ? kfc[QU * k + (i & 3)] : 0.0, ka, sv);
                        t1[L_] = dot4_tree(0.0, fa[L_], sv);
                    }
                    wquad_sum(pK, uq);
                    const double uu[4] = {WV_READLANE(uq, 0), WV_READLANE(uq, 4), WV_READLANE(uq, 8), WV_READLANE(uq, 12)};
                    QPAR(lane) { const int L_ = LI(lane); p2[L_] = (lane & 3) == 3 ? dot4_tree(cst[L_], fa[L_], uu) : t1[L_]; }
                    wquad_sum(p2, xs);
                    QPAR(lane) {
                        const int L_ = LI(lane), i = lane >> 2;
                        if ((lane & 3) == 0) QTRAJ(sh)[(size_t)(k + 1) * QS + i] = i < QX ? xs[L_] : (i == QX ? uu[0] : (i == QX + 1 ? uu[1] : (i == QX + 2 ? uu[2] : uu[3])));
                        // the gather of stage k+1 (issued QFWD stages ago) goes to the other slot; its registers take stage k+1+QFWD
#ifndef OBCA_QFW_NOGATHER      /* (diagnostic: the chain without its HBM gathers) */
                        QFW_STORE(k + 1, nvf[L_][(ju + 1) % QFWD]);
                        QFW_LOAD(k + 1 + QFWD, nvf[L_][(ju + 1) % QFWD]);
#endif
                    }
                    LDS_SYNC();
                }
            }
        }
#undef QFW_LOAD
#undef QFW_STORE
    }
    SYNC();
    QPROF(QPF_FWD);
    // ---- costate increments of the stages with a Riccati record behind them, d pi_k
    // = -(P_{k+1} s_{k+1} + p_{k+1} coef): one (stage, row) item per lane, so that
    // consecutive lanes read consecutive rows of the records (with one STAGE per lane every load of the 360 values touched 64 different lines); three items per
    // lane in flight, all loads before the first store (d may alias the records as far as the compiler knows)
    QPAR(lane) {
        const int nit = QX * (N - 1);
#define QCS_R 3
        for (int base = 0; base < nit; base += QCS_R * QNT) {
            double pv[QCS_R][QC], px[QCS_R][QS], a_[QCS_R];
#pragma unroll
            for (int r = 0; r < QCS_R; r++) {
                const int it = base + lane + QNT * r, ic = it < nit ? it : 0, k = ic / QX, i = ic - k * QX;
                const gdbl *r1 = sh.inst.rs + (size_t)(k + 1) * QRR;
#pragma unroll
                for (int cc = 0; cc < QC; cc++) pv[r][cc] = r1[QRR_PV + i * QC + cc];
#pragma unroll
                for (int j = 0; j < QS; j++) px[r][j] = r1[QRR_PX + i * QS + j];
            }
#pragma unroll
            for (int r = 0; r < QCS_R; r++) {
                const int it = base + lane + QNT * r, ic = it < nit ? it : 0, k = ic / QX;
                const double *sn = QTRAJ(sh) + (size_t)(k + 1) * QS;
                double a = 0;
#pragma unroll
                for (int cc = 0; cc < QC; cc++) a += pv[r][cc] * sh.coef[cc];
#pragma unroll
                for (int j = 0; j < QS; j++) a += px[r][j] * sn[j];
                a_[r] = -a;
            }
#pragma unroll
            for (int r = 0; r < QCS_R; r++) { const int it = base + lane + QNT * r; if (it < nit) d[l.pi + it] = a_[r]; }
        }
#undef QCS_R
    }
    // ---- stage-parallel: steps of x, u; the last costate increment; step-length / descent partials of x, u
    QPAR(lane) {
        double ap = 1.0, az = 1.0, gd = 0, cc_;
#define FTBP(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < ap) ap = cc_; }
#define FTBZ(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < az) az = cc_; }
        for (int k = lane; k <= N; k += QNT) {
            // All loads of the stage come first and all stores last: d and z may alias as far as the compiler knows, so a load after a store would wait for its
            // own round trip (the stage used to take one round trip per state and per costate row).
            const double *s = QTRAJ(sh) + (size_t)k * QS, *sn = QTRAJ(sh) + (size_t)(k + 1 <= N ? k + 1 : N) * QS;
            const int ku = k < N ? k : N - 1, km = ku >= 1 ? ku - 1 : 0;
            double xv[QX], zLx[QX], zUx[QX], uv[QU], um[QU], zLu[QU], zUu[QU], dpi[QX];
#pragma unroll
            for (int i = 0; i < QX; i++) { xv[i] = z[l.x + QX * k + i]; zLx[i] = z[l.zL + l.x + QX * k + i]; zUx[i] = z[l.zU + l.x + QX * k + i]; }
#pragma unroll
            for (int j = 0; j < QU; j++) { uv[j] = z[l.u + QU * ku + j]; um[j] = z[l.u + QU * km + j]; zLu[j] = z[l.zL + l.u + QU * ku + j]; zUu[j] = z[l.zU + l.u + QU * ku + j]; }
            if (k + 1 == N) {
                const gdbl *rN = sh.inst.as + (size_t)N * QSP;
#pragma unroll
                for (int i = 0; i < QX; i++) {
                    // (2: the least-squares multiplier system has a zero row there; QCS holds nothing yet)
                    const double e = sh.soc_on == 2 ? 0.0 : (sh.soc_on ? -QCS(sh)[QCS_NU(sh) + i] : -(z[l.x + QX * N + i] - c.xF[i]));
                    double a_ = (rN[QR(QSR_HC + 2 * i)] - rho * e) + sh.coef[2 + i];
                    for (int j = 0; j < QX; j++) a_ += (rN[QR(QSR_H + i * QZ + j)] + (i == j ? rho : 0.0)) * sn[j];
                    dpi[i] = -a_;
                }
            }
#pragma unroll
            for (int i = 0; i < QX; i++) {
                const double dx = s[i];
                d[l.x + QX * k + i] = dx;
                if (i >= 9) gd += c.sf * 2e-4 * xv[i] * dx;
                if (k >= 1) {
                    const double dL = xv[i] - q_xlb(i, c.dist), dU = q_xub(i, c.dist) - xv[i], zL = zLx[i], zU = zUx[i];
                    gd += (-rdiv(mu, dL) + rdiv(mu, dU)) * dx;
                    FTBP(dL, dx); FTBP(dU, -dx);
                    FTBZ(zL, rdiv(mu, dL) - zL - rdiv(zL, dL) * dx); FTBZ(zU, rdiv(mu, dU) - zU + rdiv(zU, dU) * dx);
                }
            }
            if (k < N) {
#pragma unroll
                for (int j = 0; j < QU; j++) {
                    const double du = sn[QX + j];        // w_{k+1} = u_k
                    d[l.u + QU * k + j] = du;
                    double gu = -c.sf * 2e-3 * (c.wH - uv[j]);
                    if (k >= 1) { const double e = um[j] - uv[j]; gu -= c.sf * 2e-2 * e; gd += c.sf * 2e-2 * e * s[QX + j]; }
                    const double dL = uv[j] - Q_ULO, dU = Q_UHI - uv[j], zL = zLu[j], zU = zUu[j];
                    gd += (gu - rdiv(mu, dL) + rdiv(mu, dU)) * du;
                    FTBP(dL, du); FTBP(dU, -du);
                    FTBZ(zL, rdiv(mu, dL) - zL - rdiv(zL, dL) * du); FTBZ(zU, rdiv(mu, dU) - zU + rdiv(zU, dU) * du);
                }
                if (k + 1 == N) {
#pragma unroll
                    for (int i = 0; i < QX; i++) d[l.pi + QX * k + i] = dpi[i];
                }
            }
        }
        sh.red[0][lane] = ap; sh.red[1][lane] = az; sh.red[2][lane] = gd;
        if (lane < QX) d[l.nu + lane] = sh.coef[2 + lane];
        if (lane == QX) d[l.t] = dt;
#undef FTBP
#undef FTBZ
    }
    SYNC();
    so.ap = red_min(sh.red[0]); so.az = red_min(sh.red[1]); so.gd = red_sum(sh.red[2]);
    SYNC();
    QPROF(QPF_BS_STAGE);
}

template <int RHS>
OBCA_FN void q_direction_obs(QShared &sh, double mu, double dw, double dc, double tau, StepOut &so) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; const gdbl *z = sh.inst.z; gdbl *d = sh.inst.d;
    double ap = so.ap, az = so.az, gd = so.gd;
    const double dt = sh.coef[1];
    const gdbl *cs = QCS(sh); constexpr int soc = RHS == 1, LSQ = RHS == 2;
    QPAR(lane) {
        double lap = 1.0, laz = 1.0, lgd = 0, cc_;
#define FTBP(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < lap) lap = cc_; }
#define FTBZ(val, dv) { cc_ = (dv) < 0 ? -tau * (val) * rcp_nr(dv) : 1e300; if (cc_ < laz) laz = cc_; }
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in);
            const double dp[3] = {d[l.x + QX * k], d[l.x + QX * k + 1], d[l.x + QX * k + 2]};
            QObsStep st;
            double crs[2] = {0, 0};
            if (soc) { crs[0] = cs[QCS_YO(sh) + 2 * it]; crs[1] = cs[QCS_YO(sh) + 2 * it + 1]; }
            q_obs_block<1, LSQ>(c, in, mu, dw, dc, nullptr, nullptr, dp, &st, soc ? crs : nullptr);
#pragma unroll
            for (int i = 0; i < QL; i++) {
                d[l.lam + QL * it + i] = st.dlam[i];
                lgd += (c.sf * 2e-4 * in.lam[i] - rdiv(mu, in.lam[i])) * st.dlam[i];
                FTBP(in.lam[i], st.dlam[i]); FTBZ(in.zl[i], rdiv(mu, in.lam[i]) - in.zl[i] - rdiv(in.zl[i], in.lam[i]) * st.dlam[i]);
            }
            d[l.s + it] = st.ds; d[l.so + it] = st.dso; d[l.yo + 2 * it] = st.dy[0]; d[l.yo + 2 * it + 1] = st.dy[1];
            lgd += -rdiv(mu, in.so) * st.dso;
            if (!c.dist) { lgd += (c.sf * (1e2 + 2e3 * in.s) - rdiv(mu, in.s)) * st.ds; FTBP(in.s, st.ds); FTBZ(in.zs, rdiv(mu, in.s) - in.zs - rdiv(in.zs, in.s) * st.ds); }
            FTBP(in.so, st.dso); FTBZ(in.zso, rdiv(mu, in.so) - in.zso - rdiv(in.zso, in.so) * st.dso);
        }
        sh.red[0][lane] = lap; sh.red[1][lane] = laz; sh.red[2][lane] = lgd;
#undef FTBP
#undef FTBZ
    }
    SYNC();
    ap = fmin(ap, red_min(sh.red[0])); az = fmin(az, red_min(sh.red[1])); gd += red_sum(sh.red[2]);
    SYNC();
    {
        const double t = z[l.t], dL = t - Q_TLO, dU = Q_THI - t, zL = z[l.zL + l.t], zU = z[l.zU + l.t];
        double cc_;
        cc_ = dt < 0 ? -tau * dL * rcp_nr(dt) : 1e300; if (cc_ < ap) ap = cc_;
        cc_ = -dt < 0 ? tau * dU * rcp_nr(dt) : 1e300; if (cc_ < ap) ap = cc_;
        const double dzL = rdiv(mu, dL) - zL - rdiv(zL, dL) * dt, dzU = rdiv(mu, dU) - zU + rdiv(zU, dU) * dt;
        cc_ = dzL < 0 ? -tau * zL * rcp_nr(dzL) : 1e300; if (cc_ < az) az = cc_;
        cc_ = dzU < 0 ? -tau * zU * rcp_nr(dzU) : 1e300; if (cc_ < az) az = cc_;
        gd += (c.sf * (N + 1) * (0.25 + 10 * t) + (N + 1) * (-rdiv(mu, dL) + rdiv(mu, dU))) * dt;
    }
    so.ap = ap; so.az = az; so.gd = gd;
}

// ---------------------------------------------------------------- objective / constraint 1-norm / barrier at v + alpha dv
OBCA_FN void q_eval_trial(QShared &sh, double alpha, double &f, double &th1, double &bar) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; const gdbl *z = sh.inst.z, *d = sh.inst.d;
    const double t = z[l.t] + alpha * d[l.t], tau = t * c.Ts;
    QPAR(lane) {
        double lf = 0, lth = 0, lbar = 0;
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in);
#pragma unroll
            for (int i = 0; i < QL; i++) { in.lam[i] += alpha * d[l.lam + QL * it + i]; lf += 1e-4 * in.lam[i] * in.lam[i]; }
            in.s += alpha * d[l.s + it]; in.so += alpha * d[l.so + it];
            {
                double dd[QL + 2];
#pragma unroll
                for (int i = 0; i < QL; i++) dd[i] = in.lam[i];
                dd[QL] = c.dist ? 1.0 : in.s; dd[QL + 1] = in.so;
                lbar += log_prod(dd);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) in.p[i] += alpha * d[l.x + QX * k + i];
            double r[2], q[3]; q_obs_rows(c, in, r, q);
            lth += fabs(r[0]) + fabs(r[1]);
            if (!c.dist) lf += 1e2 * in.s + 1e3 * in.s * in.s;
        }
        for (int k = lane; k <= N; k += QNT) {
            double x[QX];
            BarAcc ba, bb, bu; bar_init(ba); bar_init(bb); bar_init(bu);
#pragma unroll
            for (int i = 0; i < QX; i++) { x[i] = z[l.x + QX * k + i] + alpha * d[l.x + QX * k + i]; if (k >= 1) bar_mul(i < 6 ? ba : bb, x[i] - q_xlb(i, c.dist), q_xub(i, c.dist) - x[i]); }
            lf += 1e-4 * (x[9] * x[9] + x[10] * x[10] + x[11] * x[11]);
            if (k == N) {
#pragma unroll
                for (int i = 0; i < QX; i++) lth += fabs(x[i] - c.xF[i]);
            } else {
                double u[QU], g[QX];
#pragma unroll
                for (int j = 0; j < QU; j++) {
                    u[j] = z[l.u + QU * k + j] + alpha * d[l.u + QU * k + j];
                    lf += 1e-3 * (c.wH - u[j]) * (c.wH - u[j]); bar_mul(bu, u[j] - Q_ULO, Q_UHI - u[j]);
                    if (k >= 1) { const double e = z[l.u + QU * (k - 1) + j] + alpha * d[l.u + QU * (k - 1) + j] - u[j]; lf += 1e-2 * e * e; }
                }
                dyn_g_value(c, x, u, g);
#pragma unroll
                for (int i = 0; i < QX; i++) lth += fabs(z[l.x + QX * (k + 1) + i] + alpha * d[l.x + QX * (k + 1) + i] - x[i] - tau * g[i]);
            }
            lbar += bar_log(ba) + bar_log(bb) + bar_log(bu);
        }
        sh.red[0][lane] = lf; sh.red[1][lane] = lth; sh.red[2][lane] = lbar;
    }
    SYNC();
    // finish the values in registers and store each shared slot exactly once (every lane writes them; see eval_trial in obca_solver.h)
    double fr = red_sum(sh.red[0]), br = red_sum(sh.red[2]); const double tr = red_sum(sh.red[1]);
    SYNC();
    fr += (N + 1) * (0.25 * t + 5 * t * t); br += (N + 1) * log((t - Q_TLO) * (Q_THI - t));
    f = c.sf * fr; th1 = tr; bar = br;
}

// ---------------------------------------------------------------- rows of a second-order correction (IPOPT A-5.6 / A-5.7, option max_soc)
// cs <- a (first ? c(v) : cs) + c(v + a dv):  a = step length of the trial that was just rejected, dv = its direction (still in the direction buffer)
OBCA_FN void q_soc_rows(QShared &sh, double a, int first) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; const gdbl *z = sh.inst.z, *d = sh.inst.d; gdbl *cs = QCS(sh);
    QPAR(lane) {
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in);
            double r0[2], rt[2], q[3];
            if (first) q_obs_rows(c, in, r0, q); else { r0[0] = cs[QCS_YO(sh) + 2 * it]; r0[1] = cs[QCS_YO(sh) + 2 * it + 1]; }
#pragma unroll
            for (int i = 0; i < QL; i++) in.lam[i] += a * d[l.lam + QL * it + i];
            in.s += a * d[l.s + it]; in.so += a * d[l.so + it];
#pragma unroll
            for (int i = 0; i < 3; i++) in.p[i] += a * d[l.x + QX * k + i];
            q_obs_rows(c, in, rt, q);
            cs[QCS_YO(sh) + 2 * it] = a * r0[0] + rt[0]; cs[QCS_YO(sh) + 2 * it + 1] = a * r0[1] + rt[1];
        }
        for (int k = lane; k <= N; k += QNT) {
            double x[QX], xt[QX];
#pragma unroll
            for (int i = 0; i < QX; i++) { x[i] = z[l.x + QX * k + i]; xt[i] = x[i] + a * d[l.x + QX * k + i]; }
            if (k == N) {
#pragma unroll
                for (int i = 0; i < QX; i++) { const double r0 = first ? x[i] - c.xF[i] : cs[QCS_NU(sh) + i]; cs[QCS_NU(sh) + i] = a * r0 + (xt[i] - c.xF[i]); }
            } else {
                double u[QU], ut[QU], g[QX], gt[QX], r0[QX];
#pragma unroll
                for (int j = 0; j < QU; j++) { u[j] = z[l.u + QU * k + j]; ut[j] = u[j] + a * d[l.u + QU * k + j]; }
                const double t0 = z[l.t], tt = t0 + a * d[l.t];
                if (first) {
                    dyn_g_value(c, x, u, g);
#pragma unroll
                    for (int i = 0; i < QX; i++) r0[i] = z[l.x + QX * (k + 1) + i] - x[i] - t0 * c.Ts * g[i];
                } else {
#pragma unroll
                    for (int i = 0; i < QX; i++) r0[i] = cs[QCS_PI(sh) + QX * k + i];
                }
                dyn_g_value(c, xt, ut, gt);
#pragma unroll
                for (int i = 0; i < QX; i++)
                    cs[QCS_PI(sh) + QX * k + i] = a * r0[i] + (z[l.x + QX * (k + 1) + i] + a * d[l.x + QX * (k + 1) + i] - xt[i] - tt * c.Ts * gt[i]);
            }
        }
    }
    SYNC();
}

// ---------------------------------------------------------------- accept the step (generic over the primal vector)
OBCA_FN void q_apply_step(QShared &sh, double alpha, double ay, double az, double mu, double ks) {
    const QLay &l = sh.l; const int N = sh.c.N; gdbl *z = sh.inst.z; const gdbl *d = sh.inst.d;
    // The iterate is updated in place, so the compiler cannot hoist a load over an earlier store (may alias): QAP_R items per lane are processed at a time,
    // all their loads first (indices clamped, the multiplier arrays cover every primal variable), then the arithmetic and the stores -- one memory round trip
    // per chunk instead of one per item (a lone wavefront per SIMD has nothing else to hide the latency with).
#ifndef QAP_R
#define QAP_R 6
#endif
    QPAR(lane) {
        for (int base = 0; base < l.n; base += QAP_R * QNT) {
            double v[QAP_R], dv[QAP_R], zl[QAP_R], zu[QAP_R];
#pragma unroll
            // (only x, u, t have upper bounds)
            for (int r = 0; r < QAP_R; r++) { const int i = base + lane + QNT * r, ic = i < l.n ? i : 0; v[r] = z[ic]; dv[r] = d[ic]; zl[r] = z[l.zL + ic]; zu[r] = ic < l.lam ? z[l.zU + ic] : 0.0; }
#pragma unroll
            for (int r = 0; r < QAP_R; r++) {
                const int i = base + lane + QNT * r;
                if (i >= QX && i < l.n) {                  // x_0 is a constant
                    const QBnd b = q_bounds(l, N, i, sh.c.dist);
                    const double v1 = v[r] + alpha * dv[r];
                    if (b.hasL) z[l.zL + i] = clampz(zstep(zl[r], v[r] - b.lo, dv[r], mu, az), v1 - b.lo, mu, ks);
                    if (b.hasU) z[l.zU + i] = clampz(zstep(zu[r], b.hi - v[r], -dv[r], mu, az), b.hi - v1, mu, ks);
                    z[i] = v1;
                }
            }
        }
        for (int base = 0; base < l.m; base += QAP_R * QNT) {
            double yv[QAP_R], dy[QAP_R];
#pragma unroll
            for (int r = 0; r < QAP_R; r++) { const int i = base + lane + QNT * r, ic = i < l.m ? i : 0; yv[r] = z[l.n + ic]; dy[r] = d[l.n + ic]; }
#pragma unroll
            for (int r = 0; r < QAP_R; r++) { const int i = base + lane + QNT * r; if (i < l.m) z[l.n + i] = yv[r] + ay * dy[r]; }
        }
    }
#undef QAP_R
    SYNC();
}

// ---------------------------------------------------------------- starting point
OBCA_FN void q_init_point(QShared &sh, double bound_push, double bound_frac, double timeWS, int dual_ws, int obj_scaling) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; gdbl *z = sh.inst.z;
    QPAR(lane) {
        for (int i = lane; i < QX * (N + 1); i += QNT) z[l.x + i] = i < QX ? c.x0[i] : sh.inst.prob[QPH_SIZE + i];   // xWS, :201
        for (int i = lane; i < QU * N; i += QNT) z[l.u + i] = c.wH;                       // QuadcopterSignedDist.jl:202
        if (lane == 0) z[l.t] = timeWS;                                                   // :199
        for (int i = lane; i < QOB * (N + 1); i += QNT) z[l.s + i] = c.dist ? 0.0 : 1.0; // :210
        for (int i = lane; i < l.m; i += QNT) z[l.n + i] = 0.0;
        for (int i = lane; i < l.n; i += QNT) { z[l.zL + i] = 1.0; z[l.zU + i] = 1.0; }
        // stage / Riccati records: zero once, constants of the dense layout
        for (int i = lane; i < (N + 1) * QSP; i += QNT) sh.inst.as[i] = 0.0;
    }
    SYNC();
    QPAR(lane) {
        for (int k = lane; k <= N; k += QNT) {
            gdbl *rec = sh.inst.as + (size_t)k * QSP;
            for (int i = 0; i < 3; i++) { rec[QR(QSR_F + i * QFC + i)] = 1.0; rec[QR(QSR_F + (6 + i) * QFC + (6 + i))] = 1.0; }
            // the copy rows w+ = u of FX   // the other diagonal entries are rewritten every pass
            for (int j = 0; j < QU; j++) rec[QR(QSR_F + (QX + j) * QFC + QX + j)] = 1.0;
        }
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            double lam[QL], p[3] = {z[l.x + QX * k], z[l.x + QX * k + 1], z[l.x + QX * k + 2]};
            if (dual_ws) q_dual_ws(&sh.ob[j * QL], p, lam);
            else { for (int i = 0; i < QL; i++) lam[i] = 0.05; }                           // :204-208
            for (int i = 0; i < QL; i++) z[l.lam + QL * it + i] = lam[i];
        }
    }
    SYNC();
    QPAR(lane) {   // row slack = row value at the start, then everything is pushed inside its bounds
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            QObsIn in; q_load_obs(sh, z, k, j, in); in.so = 0;
            double r[2], q[3]; q_obs_rows(c, in, r, q);
            z[l.so + it] = r[1];
        }
    }
    SYNC();
    // IPOPT's gradient-based scaling of the objective (nlp_scaling_max_gradient = 100): |grad f|_inf at the starting point as given, over the variables of the
    if (obj_scaling) {
                            // reference's model (each of the N + 1 timeScale variables carries 0.25 + 10
                            // t).  At the reference's start it is the slack penalty: 1e2 + 2e3 = 2 100
        QPAR(lane) {
            double g = lane == 0 ? fabs(0.25 + 10 * z[l.t]) : 0.0;
            for (int i = lane; i < QX * (N + 1); i += QNT) if (i % QX >= 9) g = fmax(g, fabs(2e-4 * z[l.x + i]));
            for (int i = lane; i < QU * N; i += QNT) {
                const int k = i / QU;
                double gu = -2e-3 * (c.wH - z[l.u + i]);
                if (k >= 1) gu += -2e-2 * (z[l.u + i - QU] - z[l.u + i]);
                if (k + 1 < N) gu += 2e-2 * (z[l.u + i] - z[l.u + i + QU]);
                g = fmax(g, fabs(gu));
            }
            for (int i = lane; i < QL * QOB * (N + 1); i += QNT) g = fmax(g, fabs(2e-4 * z[l.lam + i]));
            if (!c.dist) for (int i = lane; i < QOB * (N + 1); i += QNT) g = fmax(g, fabs(1e2 + 2e3 * z[l.s + i]));
            sh.red[0][lane] = g;
        }
        SYNC();
        const double gm = red_max(sh.red[0]);
        SYNC();
        QPAR(lane) { if (lane == 0) sh.c.sf = gm > 100.0 ? 100.0 / gm : 1.0; }
        SYNC();
    }
    QPAR(lane) {
        for (int i = lane; i < l.n; i += QNT) {
            if (i < QX) continue;
            const QBnd b = q_bounds(l, N, i, sh.c.dist);
            if (b.hasL && b.hasU) z[i] = push2(z[i], b.lo, b.hi, bound_push, bound_frac);
            else if (b.hasL) z[i] = fmax(z[i], b.lo + bound_push * fmax(1.0, fabs(b.lo)));
        }
    }
    SYNC();
}

// ---------------------------------------------------------------- block feasibility restoration
// The reference starts every lambda at 0.05 (QuadcopterSignedDist.jl:204-208) where A'lambda = 0: the gradient of |A'lambda|^2 == 1 vanishes, the
// Jacobian is rank deficient and IPOPT leaves the point through its restoration phase (the authors mention its messages, mainQuadcopter.jl:140).
// For fixed positions the violation of the two rows of a (stage, box) block is minimised in closed form (point-to-box dual, q_dual_ws): the
// restoration resets lambda / row slack / row multipliers / their bound multipliers of every block at the current positions and keeps
// everything else; the driver then restarts the barrier parameter and clears the filter.  Same steps, same order as the CPU checker of the tests.
#define Q_MAX_RESTORE 3
OBCA_FN double q_min_norm2(QShared &sh) {
    const QLay &l = sh.l; const int N = sh.c.N; const gdbl *z = sh.inst.z;
    QPAR(lane) {
        double mn = 1e300;
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            double n2 = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) { const double q = z[l.lam + QL * it + i] - z[l.lam + QL * it + 3 + i]; n2 += q * q; }
            mn = fmin(mn, n2);
        }
        sh.red[0][lane] = mn;
    }
    SYNC();
    const double r = red_min(sh.red[0]);
    SYNC();
    return r;
}
OBCA_FN void q_restore_blocks(QShared &sh, double bound_push) {
    const QConsts &c = sh.c; const QLay &l = sh.l; const int N = c.N; gdbl *z = sh.inst.z;
    QPAR(lane) {
        for (int it = lane; it < (N + 1) * QOB; it += QNT) {
            const int k = it / QOB, j = it - k * QOB;
            double lam[QL]; const double p[3] = {z[l.x + QX * k], z[l.x + QX * k + 1], z[l.x + QX * k + 2]};
            q_dual_ws(&sh.ob[j * QL], p, lam);
            QObsIn in;
#pragma unroll
            for (int i = 0; i < QL; i++) { in.b[i] = sh.ob[j * QL + i]; in.lam[i] = lam[i]; }
            in.s = z[l.s + it]; in.so = 0; in.p[0] = p[0]; in.p[1] = p[1]; in.p[2] = p[2];
            double r[2], q[3]; q_obs_rows(c, in, r, q);
            z[l.so + it] = r[1] < bound_push ? bound_push : r[1];
#pragma unroll
            for (int i = 0; i < QL; i++) { z[l.lam + QL * it + i] = lam[i] < bound_push ? bound_push : lam[i]; z[l.zL + l.lam + QL * it + i] = 1.0; }
            if (!c.dist) { if (in.s < bound_push) z[l.s + it] = bound_push; z[l.zL + l.s + it] = 1.0; }
            z[l.zL + l.so + it] = 1.0;
            z[l.yo + 2 * it] = 0.0; z[l.yo + 2 * it + 1] = 0.0;
        }
    }
    SYNC();
}

// ---------------------------------------------------------------- phase entry points and driver
OBCA_PHASE void qph_init(double bp, double bf, double tws, int dws, int osc) { q_init_point(gq_sh, bp, bf, tws, dws, osc); QPROF(QPF_INIT); }
OBCA_PHASE double qph_min_norm2() { return q_min_norm2(gq_sh); }
OBCA_PHASE void qph_restore(double bp) { q_restore_blocks(gq_sh, bp); }
OBCA_PHASE int qph_block_bad(double mu, double dw, double dc) { const int b = q_block_bad(gq_sh, mu, dw, dc); QPROF(QPF_ASM_OBS); return b; }
OBCA_PHASE void qph_assemble_obs(double mu, double dw, double dc) { QPROF(QPF_OTHER); q_assemble_obs<0>(gq_sh, mu, dw, dc); QPROF(QPF_ASM_OBS); }
OBCA_PHASE void qph_assemble_stage(double mu, double dw, double dc, int second) { QShared &sh = gq_sh; q_assemble_stage<0>(sh, mu, dw, dc, second ? sh.A2 : sh.A); QPROF(QPF_ASM_STAGE); }
OBCA_FN void qph_assemble(double mu, double dw, double dc, int second) { qph_assemble_obs(mu, dw, dc); qph_assemble_stage(mu, dw, dc, second); }
OBCA_PHASE void qph_lsq_assemble() { QShared &sh = gq_sh; q_assemble_obs<2>(sh, 0.0, 0.0, 0.0); q_assemble_stage<2>(sh, 0.0, 0.0, 0.0, sh.A); QPROF(QPF_INIT); }
OBCA_PHASE void qph_lsq_direction_obs(double tau) { QShared &sh = gq_sh; q_direction_obs<2>(sh, 0.0, 0.0, 0.0, tau, sh.S); QPROF(QPF_INIT); }
// y <- the least-squares estimate in the direction buffer if its largest entry is <= constr_mult_init_max = 1e3 (IPOPT's default), else y stays 0
OBCA_PHASE void qph_lsq_take() {
    QShared &sh = gq_sh; const QLay &l = sh.l; gdbl *z = sh.inst.z; const gdbl *d = sh.inst.d;
    QPAR(lane) { double m_ = 0; for (int i = lane; i < l.m; i += QNT) { const double a = fabs(d[l.n + i]); if (a > m_ || a != a) m_ = a; } sh.red[0][lane] = m_; }
    SYNC();
    double ymax = 0;
    for (int i = 0; i < QNT; i++) { const double a = sh.red[0][i]; if (a > ymax || a != a) ymax = a; }
    SYNC();
    if (ymax <= 1e3 && ymax == ymax) { QPAR(lane) { for (int i = lane; i < l.m; i += QNT) z[l.n + i] = d[l.n + i]; } }
    SYNC();
    QPROF(QPF_INIT);
}
OBCA_PHASE void qph_soc_assemble(double mu, double dw, double dc) { QShared &sh = gq_sh; QPROF(QPF_OTHER); q_assemble_obs<1>(sh, mu, dw, dc); QPROF(QPF_ASM_OBS); q_assemble_stage<1>(sh, mu, dw, dc, sh.A2); QPROF(QPF_ASM_STAGE); }
OBCA_PHASE int qph_riccati(double rho) { const int ok = q_riccati_backward(gq_sh, rho); QPROF(QPF_RIC); return ok; }
OBCA_PHASE void qph_direction_main(double mu, double dw, double dc, double rho, double tau) { QShared &sh = gq_sh; q_direction_main(sh, sh.A, mu, dw, dc, rho, tau, sh.S); }
OBCA_PHASE void qph_soc_direction_obs(double mu, double dw, double dc, double tau) { QShared &sh = gq_sh; q_direction_obs<1>(sh, mu, dw, dc, tau, sh.S); QPROF(QPF_BS_OBS); }
OBCA_PHASE void qph_direction_obs(double mu, double dw, double dc, double tau) { QShared &sh = gq_sh; q_direction_obs<0>(sh, mu, dw, dc, tau, sh.S); QPROF(QPF_BS_OBS); }
OBCA_PHASE void qph_trial(double alpha) { QShared &sh = gq_sh; QPROF(QPF_OTHER); q_eval_trial(sh, alpha, sh.trial[0], sh.trial[1], sh.trial[2]); QPROF(QPF_TRIAL); }
OBCA_PHASE void qph_soc_rows(double a, int first) { q_soc_rows(gq_sh, a, first); QPROF(QPF_OTHER); }
OBCA_PHASE void qph_apply(double alpha, double ay, double az, double mu, double ks) { QPROF(QPF_OTHER); q_apply_step(gq_sh, alpha, ay, az, mu, ks); QPROF(QPF_APPLY); }

// info[8] = {status, iterations, objective, pinf, dinf, mu, #regularisations, exitflag}; exit flag per QuadcopterSignedDist.jl:229-234,285-288
OBCA_FN void q_solve_instance(int N, const Opts &o, double *info, int max_soc = 0, int lsq_init = 0, int obj_scaling = 0) {
    QShared &sh = gq_sh;
    QPAR(lane) {
        if (lane == 0) {
            QConsts &c = sh.c; const gdbl *p = sh.inst.prob;
            c.N = N; c.dist = (int)p[QPH_DIST]; c.Ts = p[QPH_TS]; c.R = p[QPH_R]; c.wH = sqrt((Q_MASS * Q_GRAV) / (Q_KF * 4)); c.sf = 1.0;
            for (int i = 0; i < QX; i++) { c.x0[i] = p[QPH_X0 + i]; c.xF[i] = p[QPH_XF + i]; }
            for (int i = 0; i < 3; i++) c.gyro[i] = c.x0[9 + i];                          // single-index x[10..12] = stage 1
            for (int i = 0; i < QOB * QL; i++) sh.ob[i] = p[QPH_OB + i];
            q_make_layout(N, sh.l);
            sh.soc_on = 0; sh.inst.d0 = sh.inst.d;
        }
    }
    SYNC();
    qph_init(o.bound_push, o.bound_frac, sh.inst.prob[QPH_TWS], (int)sh.inst.prob[QPH_DWS], obj_scaling);
    const double sf = sh.c.sf;
    double mu = o.mu_init, tau = fmax(o.tau_min, 1 - mu), dw_last = 0;
    int nf = 0, it = 0, status = ST_USERLIMIT, nreg = 0, nrest = 0, reset_th = 1, have_hint = 0, dcur = 0;      // dcur: the direction buffer in use
    const AsmOut &A = sh.A;
    double th_min = 0, th_max = 0, f = 0, pinf = 0, dinf = 0;
    double dc_mu = -1.0, dc_val = 0;
    QPAR(lane) { sh.hintl[lane] = -1; }
    SYNC();
    // IPOPT's initial multipliers: the least-squares estimate at the starting point through the same structured solve with the Hessian replaced by the identity
    if (lsq_init) {
                         // (at the reference's own start, lambda = 0.05, the system is singular: y stays 0)
        QPAR(lane) { if (lane == 0) sh.soc_on = 2; }
        SYNC();
        qph_lsq_assemble();
        int a_ = sh.A.ok;
        if (a_) a_ = qph_riccati(0.0);
        if (a_) { qph_direction_main(0.0, 0.0, 0.0, 0.0, tau); a_ = sh.S.ok; }
        if (a_) { qph_lsq_direction_obs(tau); qph_lsq_take(); }
        QPAR(lane) { if (lane == 0) sh.soc_on = 0; }
        SYNC();
    }
    if (qph_min_norm2() < 1e-12) { qph_restore(o.bound_push); nrest++; }      // rank-deficient start (the reference's lambda = 0.05): restoration first
    // where IPOPT would enter its restoration phase (line search or inertia correction failed): block restoration, barrier restart, empty filter
#define Q_USE_HINTS 1
#define Q_NEXT_RUNG(dw_) ((dw_) == 0 ? (dw_last == 0 ? o.dw0 : fmax(o.dw_min, o.kw_dec * dw_last)) : (dw_) * (dw_last == 0 ? o.kw_inc0 : o.kw_inc))
#define Q_RESTORE_AND_CONTINUE { qph_restore(o.bound_push); nrest++; mu = o.mu_init; tau = fmax(o.tau_min, 1 - mu); nf = 0; dw_last = 0; reset_th = 1; continue; }
    for (;;) {
        if (mu != dc_mu) { dc_val = o.dc_bar * pow(mu, o.kappa_c); dc_mu = mu; }   // a pow is a ~3k-clock dependent chain: keep it while mu stays
        double dc = dc_val;
        // rungs of the inertia ladder (IPOPT Algorithm IC) that a remembered block is known to fail are counted and skipped, see q_block_bad
        int nskip = 0; double dw_first = 0;
        if (have_hint) { while (dw_first <= o.dw_max && qph_block_bad(mu, dw_first, dc)) { nskip++; dw_first = Q_NEXT_RUNG(dw_first); } }
        if (dw_first > o.dw_max) { nskip = 0; dw_first = 0; }
        double dw_have = dw_first;                      // the regularisation of the system the records hold
        qph_assemble(mu, dw_have, dc, 0);
        if (!A.ok) have_hint = Q_USE_HINTS;
        if (reset_th) { th_min = 1e-4 * fmax(1.0, A.th1); th_max = 1e4 * fmax(1.0, A.th1); reset_th = 0; }
        f = A.f; pinf = A.pinf; dinf = A.dinf;
        const double sd = fmax(o.s_max, (A.sumy + A.sumz) / (A.nm + A.nb)) / o.s_max, sc = fmax(o.s_max, A.sumz / A.nb) / o.s_max;
        const double E0 = fmax(A.dinf / sd, fmax(A.pinf, A.cinf0 / sc));
        // (the three *_tol: IPOPT's tolerances on the UNSCALED problem)
        if (E0 <= o.tol && A.pinf <= o.constr_viol_tol && A.dinf / sf <= o.dual_inf_tol && A.cinf0 / sf <= o.compl_inf_tol) { status = ST_OPTIMAL; break; }
        if (it >= o.max_iter) { status = ST_USERLIMIT; break; }
        if (!(A.f == A.f) || !(A.pinf == A.pinf) || !(A.dinf == A.dinf)) { status = ST_ERROR; break; }
        int mu_changed = 0;
        {
            double cm = cinf_mu(A, mu);
            for (;;) {
                const double Emu = fmax(dinf / sd, fmax(pinf, cm / sc));
                if (Emu <= o.kappa_eps * mu && mu > o.tol / 10) {
                    mu = fmax(o.tol / 10, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
                    tau = fmax(o.tau_min, 1 - mu); nf = 0; mu_changed = 1;
                    dc_val = o.dc_bar * pow(mu, o.kappa_c); dc_mu = mu;
                    cm = cinf_mu(A, mu);      // complementarity error w.r.t. the new mu, from the extreme products of the assembly at hand (no re-assembly)
                } else break;
            }
        }
        dc = dc_val;
        if (mu_changed) {      // delta_c moved with mu, and the records hold the system of the error test
            nskip = 0; dw_first = 0; dw_have = -1.0;
            if (have_hint) { while (dw_first <= o.dw_max && qph_block_bad(mu, dw_first, dc)) { nskip++; dw_first = Q_NEXT_RUNG(dw_first); } }
            if (dw_first > o.dw_max) { nskip = 0; dw_first = 0; }
        }
        double dw = dw_first; int ok = 0;
        nreg += nskip;
        for (int tr = 0; tr < 60; tr++) {
            if (dw != dw_have) { qph_assemble(mu, dw, dc, 0); dw_have = dw; if (!A.ok) have_hint = Q_USE_HINTS; }
            int a_ = A.ok;
            if (a_) a_ = qph_riccati(o.rho_term);
            if (a_) { qph_direction_main(mu, dw, dc, o.rho_term, tau); a_ = sh.S.ok; }
            if (a_) { qph_direction_obs(mu, dw, dc, tau); ok = 1; break; }
            nreg++;
            dw = Q_NEXT_RUNG(dw);
            if (dw > o.dw_max) break;
        }
        if (!ok) { if (nrest < Q_MAX_RESTORE) Q_RESTORE_AND_CONTINUE; status = ST_ERROR; break; }
        if (dw > 0) dw_last = dw;
        const double th = A.th1, phi = A.f - mu * A.bar, gd = sh.S.gd, ap0 = sh.S.ap; double az = sh.S.az;
        double amin, pw_th = 0, pw_gd = 0;
        if (gd < 0) { amin = fmin(o.gamma_theta, o.gamma_phi * th / (-gd)); pw_th = pow(th, o.s_theta); pw_gd = pow(-gd, o.s_phi); if (th <= th_min) amin = fmin(amin, o.delta * pw_th / pw_gd); }
        else amin = o.gamma_theta;
        amin *= o.gamma_alpha;
        double alpha = sh.S.ap; int acc = 0;
        while (alpha >= amin) {
            qph_trial(alpha);
            const double ft = sh.trial[0], tht = sh.trial[1], pht = ft - mu * sh.trial[2];
            if (ft == ft && tht == tht && pht == pht && tht < th_max) {
                int okf = 1;
                for (int i = 0; i < nf && okf; i++) if (!(tht < sh.filt[i][0] || pht < sh.filt[i][1])) okf = 0;
                if (okf) {
                    const int sw = gd < 0 && alpha * pw_gd > o.delta * pw_th, armijo = pht <= phi + o.eta_phi * alpha * gd;
                    if (th <= th_min && sw) { if (armijo) { acc = 1; break; } }
                    else if (tht <= (1 - o.gamma_theta) * th || pht <= phi - o.gamma_phi * th) {
                        acc = 1;
                        if (!(sw && armijo) && nf < QFILT) {
                            QPAR(lane) { if (lane == 0) { sh.filt[nf][0] = (1 - o.gamma_theta) * th; sh.filt[nf][1] = phi - o.gamma_phi * th; } }
                            SYNC();
                            nf++;
                        }
                        break;
                    }
                }
            }
            // second-order correction (IPOPT A-5.5 .. A-5.9, kappa_soc = 0.99) after a
            // rejected FIRST trial step that did not reduce theta.  The correction solves the
            // iteration's own system with the constraint right-hand sides replaced (same
            // matrix: same regularisation, inertia already right) into the other direction
            // buffer: a correction that is rejected leaves the iteration's own direction where the backtracking goes on with it.
            if (max_soc > 0 && alpha == ap0 && ft == ft && tht == tht && tht >= th) {
                double th_old = 0, th_tr = tht, asoc = alpha, azs = az;
                for (int ps = 0; ps < max_soc && !acc && (ps == 0 || th_tr <= 0.99 * th_old); ps++) {
                    th_old = th_tr;
                    qph_soc_rows(asoc, ps == 0);      // (reads the direction of the trial just rejected: the iteration's own, then the last correction's)
                    QPAR(lane) { if (lane == 0) { sh.soc_on = 1; sh.inst.d = QDIR(sh, 1 - dcur); } }
                    SYNC();
                    qph_soc_assemble(mu, dw, dc);
                    int a_ = sh.A2.ok;
                    if (a_) a_ = qph_riccati(o.rho_term);
                    if (a_) { qph_direction_main(mu, dw, dc, o.rho_term, tau); a_ = sh.S.ok; }
                    if (a_) qph_soc_direction_obs(mu, dw, dc, tau);
                    QPAR(lane) { if (lane == 0) sh.soc_on = 0; }
                    SYNC();
                    if (!a_) break;
                    asoc = sh.S.ap; azs = sh.S.az;
                    qph_trial(asoc);
                    const double fs = sh.trial[0], ths = sh.trial[1], phs = fs - mu * sh.trial[2];
                    if (!(fs == fs && ths == ths)) break;
                    th_tr = ths;
                    if (ths < th_max && phs == phs) {
                        int okf = 1;
                        for (int i = 0; i < nf && okf; i++) if (!(ths < sh.filt[i][0] || phs < sh.filt[i][1])) okf = 0;
                        if (okf) {
                            const int sw = gd < 0 && alpha * pw_gd > o.delta * pw_th, armijo = phs <= phi + o.eta_phi * alpha * gd;
                            if (th <= th_min && sw) { if (armijo) acc = 1; }
                            else if (ths <= (1 - o.gamma_theta) * th || phs <= phi - o.gamma_phi * th) {
                                acc = 1;
                                if (!(sw && armijo) && nf < QFILT) {
                                    QPAR(lane) { if (lane == 0) { sh.filt[nf][0] = (1 - o.gamma_theta) * th; sh.filt[nf][1] = phi - o.gamma_phi * th; } }
                                    SYNC();
                                    nf++;
                                }
                            }
                        }
                    }
                    if (acc) { alpha = asoc; az = azs; dcur = 1 - dcur; }      // the correction is the step: its multiplier steps go with it
                }
                if (acc) break;
                QPAR(lane) { if (lane == 0) sh.inst.d = QDIR(sh, dcur); }      // back to the iteration's own direction
                SYNC();
            }
            alpha *= 0.5;
        }
        if (!acc) { if (nrest < Q_MAX_RESTORE) Q_RESTORE_AND_CONTINUE; status = ST_ERROR; break; }
        qph_apply(alpha, fmin(alpha, az), az, mu, o.kappa_sigma);
        it++;
    }
#undef Q_RESTORE_AND_CONTINUE
#undef Q_NEXT_RUNG
    // exit flag: 1 = Optimal, 2 = Optimal but sum(slack) > 1e-3, 0 otherwise
    QPAR(lane) { double s_ = 0; for (int i = lane; i < QOB * (N + 1); i += QNT) s_ += sh.inst.z[sh.l.s + i]; sh.red[0][lane] = s_; }
    SYNC();
    const double ssum = red_sum(sh.red[0]);
    SYNC();
    int ef = status == ST_OPTIMAL ? 1 : 0;
    if (!sh.c.dist && ef == 1 && ssum > 1e-3) ef = 2;
    QPAR(lane) { if (lane == 0) { info[0] = status; info[1] = it; info[2] = f / sf; info[3] = pinf; info[4] = dinf / sf; info[5] = mu; info[6] = nreg; info[7] = ef; } }
    SYNC();
}

}  // namespace quad
}  // namespace obca
