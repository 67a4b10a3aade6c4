// obca_solver_assemble.h -- part of obca_solver.h (included from there, inside namespace obca; not a stand-alone header):
// assembly of the condensed Newton system: (stage, obstacle) blocks and stage items, with the line search fused in.

// ---------------------------------------------------------------- assemble the condensed Newton system
// The line search is fused into the assembly (FUSED = 1): the trial point z + alpha d is formed on the fly, written to the second iterate buffer
// (Inst::zn) and assembled right there -- objective, constraint norm and barrier of the trial point ARE the f / th1 / bar of its assembly, and when the
// trial is accepted (the first one, as a rule) the two buffers swap and the next iteration starts with its Newton system already assembled.  Against
// separate trial / accept / assemble phases (round 2) the iterate is read once instead of three times per iteration and the obstacle part of the search
// direction is never stored: a (stage, obstacle) block recomputes its step from the pose step (obs_block<1>, the same code direction_obs ran).
// The obstacle part of the search direction (d lambda, d mu, d sl, d so, d y per (stage,
// obstacle) block) is needed twice: for the step lengths (direction_obs) and for
// the trial point (fused assembly).  OBCA_STORE_DOBS = 0 (default): the fused assembly
// recomputes it (obs_block<1>: ~25 % of that phase's arithmetic, no traffic);
// 1: direction_obs writes it to `d` and the fused assembly loads it with the block's
// iterate.  Measured on MI355X (config 2, profiles/r03_ab_obstacle_steps.txt): the same
// pipelined rate (212.8 k / 214.6 k solves/s), storing is 5 % quicker per pass for
// a lone instance and moves 10 % more HBM bytes (15.9 against 14.4 GB per launch).
#ifndef OBCA_STORE_DOBS
#define OBCA_STORE_DOBS 0
#endif
// step lengths (primal, equality multipliers, bound multipliers), kappa_sigma, delta_w of the factorisation that gave d
struct FuseArgs { double alpha, ay, az, ks, dw_dir; };
// part (a): one lane per (stage, obstacle) block; partial results go to sh.Ap
// SOC = 1: the system of a second-order correction step -- FUSED = 0: condensation with
// c_soc on the right-hand side; FUSED = 1: the block steps of the trial are those of
// the correction direction (recomputed with c_soc), the assembly at the trial point is the ordinary one.
// KEEP = 1 (with FUSED = 1): the block steps were kept in registers by direction_obs<KEEP
// = 1> of the same phase call (ph_direction2_trial: the first trial of a line search);
// item lane + 64 r finds its step in keep[r] and is not factorised a second time at the old point.
#define OB_KEEP 4          // rounds of (stage, obstacle) items whose steps a lane keeps: (N + 1) nOb <= 64 OB_KEEP (N = 80, 3 obstacles: 243 items)
// LSQ = 1 (with FUSED = 0): the blocks of the least-squares multiplier system (obs_block)
template <int VM, int FUSED, int SOC = 0, int LSQ = 0, int KEEP = 0>
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
        // KEEP: the rounds are unrolled so that keep[rr] is a fixed set of registers; otherwise one pass of the plain item loop
        for (int rr = 0; rr < (KEEP ? OB_KEEP : 1); rr++)
        // (the round loop is UNIFORM -- its bounds sit in scalar registers -- and a lane
        // without an item in the last round skips the item code: the ordered sum of the
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
                // pose step of the stage (x_0 is fixed: s_0 = 0)
                const double dp[3] = {g_traj[(size_t)k * 6], g_traj[(size_t)k * 6 + 1], g_traj[(size_t)k * 6 + 2]};
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
                // (explicit fma: the stage part forms the same values from the same operands, bit for bit)
                in.X = fma(fa.alpha, dp[0], in.X); in.Y = fma(fa.alpha, dp[1], in.Y); in.psi = fma(fa.alpha, dp[2], in.psi);
#pragma unroll
                for (int i = 0; i < VM; i++) { SEAM(in.lam[i]); SEAM(in.zl[i]); }
#pragma unroll
                for (int i = 0; i < 4; i++) { SEAM(in.mu[i]); SEAM(in.zm[i]); SEAM(in.y[i]); }
                SEAM(in.so); SEAM(in.zso); SEAM(in.sl); SEAM(in.zs1); SEAM(in.X); SEAM(in.Y); SEAM(in.psi);
            }
            obs_block<0, VM, (SOC && !FUSED) ? 1 : 0, LSQ>(c, in, mu, dw, dc, &cd, &st, nullptr, nullptr, crs);
            }
            // the condensed contribution goes into the stage's 12 sums in LDS (rounds 1-3 wrote
            // a record per (stage, obstacle) to HBM and the stage part read nOb of them back:
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
// The stage item is written in SECTIONS -- state x_k | condensed obstacle sums | inputs,
// rate cost, steering row | dynamics | finish -- each of which loads what it needs,
// folds it into the few accumulators of the stage record and stores what is final,
// with a scheduling barrier in between: the live set stays below the 256 registers a
// wavefront has when TWO of them share a SIMD (rounds 1-3 issued every load of the stage
// up front and kept ~430 registers alive, which fixed the kernel at one wavefront per
// SIMD).  The loads of a section are issued one section ahead, so a section's arithmetic runs in the shadow of the next one's memory round trip.
#ifdef OBCA_EMU
#define SECTION() ((void)0)
#else
#define SECTION() __builtin_amdgcn_sched_barrier(0)
#endif
// SOC = 1 (with FUSED = 0): steering and dynamics rows enter the right-hand side with c_soc;  LSQ = 1 (with FUSED = 0, mu = dw = dc = 0):
template <int FUSED, int SOC = 0, int LSQ = 0>
// the least-squares multiplier system -- unit Hessian, no second derivatives, zero constraint right-hand side, gradients in their z-form
OBCA_FN void assemble_stage(const Inst &I, Shared &sh, double mu_, double dw_, double dc_, const FuseArgs &fa_, AsmOut &out) {
    const Consts &c = sh.c; const Lay &l = sh.l;
    const int N = UNIFORM(c.N), nOb = c.nOb, M = c.M;
    const gdbl *z = I.z, *d = I.d; gdbl *zn = I.zn;
    // what is the same for every lane lives in scalar registers (as function arguments
    // and LDS reads these values would each hold two of the 256 vector registers)
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
            // gradient of the Lagrangian w.r.t. (X, Y, psi, v, w0, w1, delta, a): z-form (dual infeasibility) and barrier form (right-hand side)
            double hz[8], hb[8];
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
                // the trial point of this stage (steps: x in LDS, the rest in d).  Explicit fma
                // wherever a trial value is formed: neighbouring stages (and the obstacle blocks) form
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
            // ================================================================ section 2: condensed
            // obstacle contributions of this stage (summed over the obstacles by part (a))
            double H00, H01, H02, H11, H12, H22;
            {
                double oc_[12]; const double *os = ocs + (size_t)k * OB_OC;
#pragma unroll
                for (int i = 0; i < 12; i++) oc_[i] = os[i];
                H00 = Hd[0] + oc_[0]; H01 = oc_[1]; H02 = oc_[2]; H11 = Hd[1] + oc_[3]; H12 = oc_[4]; H22 = Hd[2] + oc_[5];
                // final: stored now, not carried through the dynamics
                rec[AS_H + 0] = H00; rec[AS_H + 1] = H01; rec[AS_H + 2] = H02; rec[AS_H + 3] = H11; rec[AS_H + 4] = H12;
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
            // ================================================================ section 3:
            // inputs u_k, their copy w_k = u_{k-1}, rate cost, bounds, steering-rate row
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
            // final
            rec[AS_H + 12] = H44; rec[AS_H + 13] = H46; rec[AS_H + 14] = H55; rec[AS_H + 15] = H57; rec[AS_HT + 2] = Ht4; rec[AS_HT + 3] = Ht5;
            rec[AS_HB + 4] = hb[4]; rec[AS_HB + 5] = hb[5];
            // (issued one section ahead) section 5: the next stage's inputs and steering
            // multiplier, for the part of u_k's dual infeasibility that lives in stage k + 1
            double un[2] = {z[l.u + 2 * kn], z[l.u + 2 * kn + 1]}, ygn = z[l.yg + kn], dun[2] = {0, 0}, dygn = 0;
            if (FUSED) { dun[0] = d[l.u + 2 * kn]; dun[1] = d[l.u + 2 * kn + 1]; dygn = d[l.yg + kn]; }
            SECTION();
            // ================================================================ section 4: dynamics
            // x_{k+1} - F(x_k,u_k,t) = 0, multiplier pi_k   (ParkingSignedDist.jl:139-155)
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
            // ================================================================ section 5: dual
            // infeasibility of u_k (own part + copy part living in stage k+1), barrier, stores
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
        // (least-squares system: t stands for the N + 1 timeScale variables of the reference's model, N + 1 unit diagonal entries)
        Htt += LSQ ? (double)(N + 1) : 2.0 * (N + 1) + b.Sig + dw;
        gtb += gf + (LSQ ? b.gz : b.gb); gtz += gf + b.gz;
        f += (N + 1) * (0.5 * t + t * t);
        bar += (N + 1) * log((t - OB_TL) * (OB_TU - t));
        dinf = fmax(dinf, fabs(gtz));
    } else { Htt = 1.0; gtb = 0; }
    // (cinf0: the largest |s z| of all complementarity pairs = the larger of the two extreme products in magnitude)
    out.ok = ok; out.dinf = dinf; out.pinf = pinf; out.cinf0 = fmax(fabs(cmn), fabs(cmx)); out.cmin = cmn; out.cmax = cmx;
    // the multiplier sums behind IPOPT's scaling factors s_d, s_c of the termination test (ipm_attempt).  From round 4 until the end of round 5 these two stores sat at the end of
    // a comment: the driver read whatever the LDS words held.  Zeros, NaNs and the previous workgroup's own leftovers give s_d = s_c = 1, which is also what the sums give
    // for nearly every instance -- so parity held; a large positive pattern left by ANOTHER kernel (another process on the GPU, or a workgroup of another horizon whose dynamic block
    // lay there) ends the solve early.  DESIGN.md section 11.
    out.sumy = sumy; out.sumz = sumz;
    out.f = f; out.th1 = th1; out.bar = bar; out.Htt = Htt; out.gtb = gtb; out.nb = nb; out.nm = nm;
    PROF(I, FUSED ? PF_APPLY : PF_ASM_STAGE);
}

