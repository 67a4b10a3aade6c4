// obca_solver_direction.h -- part of obca_solver.h (included from there, inside namespace obca; not a stand-alone header):
// border solve, closed-loop forward sweep, stage and block back-substitution, step lengths.

// ---------------------------------------------------------------- border solve + forward sweep + back-substitution

// part 1: border, closed loop, forward sweep, stage-parallel back-substitution; leaves partial (ap, az, gd) and (dt, nu) in LDS
// SOC = 1: the terminal row enters with c_soc;  LSQ = 1: with zero (least-squares multiplier system; call with mu = dw = dc = rho = 0)
template <int SOC = 0, int LSQ = 0>
OBCA_FN void direction_main(const Inst &I, Shared &sh, const AsmOut &A, double mu, double dw, double dc, double rho, double tau, StepOut &so) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N;
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
    // ---- forward recursion s_{k+1} = Acl_k s_k + bcl_k with the closed-loop maps Acl
    // = [A + B K ; K] (6x6), bcl = [B kf + off ; kf].  The recursion is a chain of N
    // dependent steps (~280 clocks each: a 4-deep fp64 dependency plus the broadcast),
    // so it runs TWO stages per step.  One lane per stage pair j builds the maps
    // of stages 2j and 2j+1 from the Riccati gains and the stage records, composes
    // them (Pm_j = Acl_{2j+1} Acl_{2j}, pb_j = Acl_{2j+1} bcl_{2j} + bcl_{2j+1}) into
    // LDS and KEEPS the plain map of stage 2j in registers; the sequential loop then
    // produces the even states s_{2j+2} from the composed maps (rows read from LDS
    // one step ahead, the state itself in scalar registers via v_readlane), and afterwards
    // every pair lane fills in its odd state s_{2j+1} = Acl_{2j} s_{2j} + bcl_{2j}.
    // (Round 2 wrote the 42-double closed-loop map of every stage to the Riccati record and the composed maps to a second HBM buffer, and the sequential loop
    // gathered both back through a ring of registers: 0.1 MB of traffic per pass and a loop whose step time followed the memory latency under load.)
    const int NP = UNIFORM(N / 2), NH = UNIFORM((N + 1) / 2);     // pairs; pair lanes incl. the single last stage of an odd horizon
    // composed maps: NP x 42 doubles behind the trajectory (the backward sweep's stage buffers are dead by now)
    double *pm = stg_base(sh);
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
                                              a3[i] = as_df(i, 1) >= 0 ? rec[AS_DF + as_df(i, 1)] : 0.0; dd[i] = rec[AS_DD + i];
                                              ft[i] = rec[AS_DF + as_df(i, 4)]; }
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
                    // the next step's row: its LDS reads are in flight during this step's arithmetic
                    const double *pn_ = pm + (size_t)(j + 1 < NP ? j + 1 : j) * 42;
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
                    for (int j = 0; j < 6; j++) a_ += r1[RS_PX + (j < i ? j * 6 + i : i * 6 + j)] * sn[j];      // (rows 0..3 of the symmetric P: the sweep stores entry (i, j) of the 4 x 4 block once, at (min, max))
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
                // du_k = K_k s_k + kf_k is the input copy dw_{k+1} of the NEXT state: the forward
                // sweep formed it already (rows 4, 5 of the closed-loop map), so it is read from the
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
// DBG = 1 (host emulation tests, least-squares multipliers): the obstacle part of
// the direction is also written to d; SOC = 1: block right-hand sides with c_soc
template <int VM, int DBG, int SOC = 0, int LSQ = 0, int KEEP = 0>
// KEEP = 1: the block steps stay in the caller's registers (keep[r] = step of item
// lane + 64 r) for the first trial of the line search, see assemble_obs<KEEP = 1>
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

