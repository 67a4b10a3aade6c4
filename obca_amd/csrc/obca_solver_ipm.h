// obca_solver_ipm.h -- part of obca_solver.h (included from there, inside namespace obca; not a stand-alone header):
// starting point, phase entry points, second-order correction, recalc_y / least-squares multipliers, the interior-point driver and solve_instance.

// ---------------------------------------------------------------- starting point (IPOPT sec. 3.6: push into the bounds, z=1, y=0)
OBCA_FN double push2(double v, double lo, double hi, double k1, double k2) {
    double pl = fmin(k1 * fmax(1.0, fabs(lo)), k2 * (hi - lo)), pu = fmin(k1 * fmax(1.0, fabs(hi)), k2 * (hi - lo));
    if (v < lo + pl) v = lo + pl;
    if (v > hi - pu) v = hi - pu;
    return v;
}
// Block feasibility restoration: a stand-in for what IPOPT's restoration phase does for this model's one structural degeneracy (obca_opts.restoration; the quadcopter kernel's
// q_restore_blocks is its sibling).  DualMultWS (DualMultWS.jl:52-73) returns lambda = mu = 0 for a pose whose car rectangle touches or penetrates the obstacle -- the distance is 0
// and the dual of |A'lam| <= 1 is free to vanish -- and the signed-distance NLP started there has |A'lam|^2 = 0 against the EQUALITY |A'lam|^2 == 1 (ParkingSignedDist.jl:196) with
// a vanishing row gradient 2 A A'lam: a rank-deficient start the interior point does not leave.  A block (stage k, obstacle j) with |A'lam|^2 < 1/4 is DEGENERATE; it gets the
// feasible dual of the obstacle's best edge:   lam = e_s,  s = argmax_i d_i,  d_i = a_i . c - b_i - (g_1 |a_i . e_psi| + g_2 |a_i . e_perp|)   (unit rows; c: centre of the car),
// mu = the non-negative split of -R'a_s.  Then |A'lam| = 1 and G'mu + R'A'lam = 0 hold exactly and the separation row takes the value d_s (the signed distance along that edge
// normal: negative when the pose penetrates, absorbed by the row's penalised slack as the reference intends).  mid = 1 (inside a solve): the other multipliers take the bound
// push, the row's slack its value, the block's bound multipliers 1 and its equality multipliers 0.  Returns the number of blocks repaired (wave-uniform).
template <int VM>
OBCA_FN int restore_blocks(const Inst &I, Shared &sh, int mid, double bound_push) {
    const Lay &l = sh.l; Consts c; obs_consts(sh.c, c);
    const int N = c.N, nOb = c.nOb, M = c.M; gdbl *z = I.z;
    double red[1][OBCA_NL];
    PAR(lane) {
        double cnt = 0;
        for (int it = lane; it < (N + 1) * nOb; it += OB_NT) {
            const int k = it / nOb, j = it - k * nOb, r0 = sh.roff[j];
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, j, in);
            double p1 = 0, p2 = 0;
#pragma unroll
            for (int i = 0; i < VM; i++) { p1 += in.a1[i] * in.lam[i]; p2 += in.a2[i] * in.lam[i]; }      // (rows beyond in.v: a = 0)
            if (p1 * p1 + p2 * p2 < 0.25) {
                double sn, cs; sincos_bounded(in.psi, &sn, &cs);
                const double cx = in.X + cs * c.off, cy = in.Y + sn * c.off;
                double best = -1e300, be1 = 0, be2 = 0; int s_ = 0;
#pragma unroll
                for (int i = 0; i < VM; i++) if (i < in.v) {
                    const double e1 = cs * in.a1[i] + sn * in.a2[i], e2 = -sn * in.a1[i] + cs * in.a2[i];
                    const double d_ = in.a1[i] * cx + in.a2[i] * cy - in.b[i] - (c.g[0] * fabs(e1) + c.g[1] * fabs(e2));
                    if (d_ > best) { best = d_; s_ = i; be1 = e1; be2 = e2; }
                }
                const double lo = mid ? bound_push : 0.0;
#pragma unroll
                for (int i = 0; i < VM; i++) if (i < in.v) { in.lam[i] = i == s_ ? 1.0 : lo; z[l.lam + k * M + r0 + i] = in.lam[i]; }
                in.mu[0] = fmax(-be1, lo); in.mu[1] = fmax(-be2, lo); in.mu[2] = fmax(be1, lo); in.mu[3] = fmax(be2, lo);
#pragma unroll
                for (int i = 0; i < 4; i++) z[l.mu + 4 * it + i] = in.mu[i];
                if (mid) {
                    in.so = 0; double r[4]; obs_rows<VM>(c, in, r);
                    if (c.dist) { z[l.sl + it] = fmax(-(r[0] - in.sl), bound_push); z[l.zs1 + it] = 1.0; }
                    z[l.so + it] = fmax(r[3], bound_push); z[l.zso + it] = 1.0;
#pragma unroll
                    for (int i = 0; i < VM; i++) if (i < in.v) z[l.zlam + k * M + r0 + i] = 1.0;
#pragma unroll
                    for (int i = 0; i < 4; i++) { z[l.zmu + 4 * it + i] = 1.0; z[l.yo + 4 * it + i] = 0.0; }
                }
                cnt += 1.0;
            }
        }
        red[0][LI(lane)] = cnt;
    }
    const double n_ = wred_sum(red[0]);
    SYNC();
    return (int)n_;
}
#define OB_MAX_RESTORE 3      // restorations per attempt

struct PushOpts { double bound_push, bound_frac; int restore; };
template <int VM>
OBCA_FN void init_point(const Inst &I, Shared &sh, const PushOpts &o) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb, M = c.M;
    gdbl *z = I.z;
    PAR(lane) {
        if (lane < 4) z[l.x + lane] = c.x0[lane];
        if (lane == 4 && c.fixTime) z[l.t] = 1.0;
        for (int i = l.pi + lane; i < l.zxL; i += OB_NT) z[i] = 0.0;
        for (int i = l.zxL + lane; i < l.len; i += OB_NT) z[i] = 1.0;
    }
    SYNC();
    if (o.restore) restore_blocks<VM>(I, sh, 0, o.bound_push);      // degenerate blocks of the warm start (DualMultWS at a touching / penetrating pose): before the slacks take their values
    const double q = z[l.t] * c.Ts;
    PAR(lane) {   // slacks take the row values at the (un-pushed) warm start
        for (int k = lane; k < N; k += OB_NT) z[l.ss + k] = ((k ? z[l.u + 2 * k - 2] : 0.0) - z[l.u + 2 * k]) / q;
        for (int it = lane; it < (N + 1) * nOb; it += OB_NT) {
            int k = it / nOb, j = it - k * nOb;
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, j, in);
            in.so = 0; if (c.dist) in.sl = 0;
            double r[4]; obs_rows<VM>(c, in, r);
            z[l.so + it] = r[3];
            if (c.dist) z[l.sl + it] = -r[0];          // slack of |A'lam|^2 <= 1 takes the row value
        }
    }
    SYNC();
    PAR(lane) {   // push into the interior
        for (int k = lane; k <= N; k += OB_NT) {
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) if (i != 2) z[l.x + 4 * k + i] = push2(z[l.x + 4 * k + i], c.xl[i], c.xu[i], o.bound_push, o.bound_frac);
            }
            if (k < N) {
                z[l.u + 2 * k] = push2(z[l.u + 2 * k], OB_UL0, OB_UU0, o.bound_push, o.bound_frac);
                z[l.u + 2 * k + 1] = push2(z[l.u + 2 * k + 1], OB_UL1, OB_UU1, o.bound_push, o.bound_frac);
                z[l.ss + k] = push2(z[l.ss + k], -OB_SSB, OB_SSB, o.bound_push, o.bound_frac);
            }
        }
        if (lane == 4 && !c.fixTime) z[l.t] = push2(z[l.t], OB_TL, OB_TU, o.bound_push, o.bound_frac);
        for (int i = lane; i < M * (N + 1); i += OB_NT) z[l.lam + i] = fmax(z[l.lam + i], o.bound_push);
        for (int i = lane; i < 4 * nOb * (N + 1); i += OB_NT) z[l.mu + i] = fmax(z[l.mu + i], o.bound_push);
        for (int i = lane; i < nOb * (N + 1); i += OB_NT) { z[l.so + i] = fmax(z[l.so + i], o.bound_push); if (c.dist) z[l.sl + i] = fmax(z[l.sl + i], o.bound_push); }
    }
    SYNC();
}

// ---------------------------------------------------------------- phase entry points (non-inlined; state lives in g_sh)
// the per-lane (stage, obstacle) code exists in three sizes (VM = 2, OB_VMID, OB_VMAX rows); an instance uses the smallest that holds its widest obstacle
#define VM_CALL(F, ...) do { if (g_sh.vmc == 0) F<2>(__VA_ARGS__); else if (g_sh.vmc == 1) F<OB_VMID>(__VA_ARGS__); else F<OB_VMAX>(__VA_ARGS__); } while (0)
OBCA_PHASE int ph_restore(double bound_push) { Shared &sh = g_sh; int n_ = 0; if (sh.vmc == 0) n_ = restore_blocks<2>(sh.inst, sh, 1, bound_push); else if (sh.vmc == 1) n_ = restore_blocks<OB_VMID>(sh.inst, sh, 1, bound_push); else n_ = restore_blocks<OB_VMAX>(sh.inst, sh, 1, bound_push); return n_; }
OBCA_PHASE void ph_init(double bound_push, double bound_frac) {
    Shared &sh = g_sh; PushOpts po = {bound_push, bound_frac, sh.soc.restoration == 1}; PROF(sh.inst, PF_OTHER);
    VM_CALL(init_point, sh.inst, sh, po);
    PROF(sh.inst, PF_INIT);
}
// Assembly of the Newton system at the current iterate (`which` = 0 -> sh.A, 1 -> sh.A2), and the fused line-search step (ph_fused: trial point -> Inst::zn,
// assembled -> sh.An).  The (stage, obstacle) part and the stage part of the common (<= 2 rows per obstacle) case share ONE non-inlined function.
#define OB_NOFUSE FuseArgs{0.0, 0.0, 0.0, 0.0, 0.0}
OBCA_PHASE void ph_assemble_obs2(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<2, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_PHASE void ph_assemble_obs4(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<OB_VMID, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_PHASE void ph_assemble_obs8(double mu, double dw, double dc) { Shared &sh = g_sh; assemble_obs<OB_VMAX, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); }
OBCA_FN void ph_assemble_obs(double mu, double dw, double dc) { if (g_sh.vmc == 0) ph_assemble_obs2(mu, dw, dc); else if (g_sh.vmc == 1) ph_assemble_obs4(mu, dw, dc); else ph_assemble_obs8(mu, dw, dc); }
OBCA_PHASE void ph_assemble_stage(double mu, double dw, double dc, int second) { Shared &sh = g_sh; assemble_stage<0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, second ? sh.A2 : sh.A); }
OBCA_PHASE void ph_assemble2(double mu, double dw, double dc, int second) { Shared &sh = g_sh; assemble_obs<2, 0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); assemble_stage<0>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, second ? sh.A2 : sh.A); }
OBCA_FN void ph_assemble(double mu, double dw, double dc, int second) { if (g_sh.vm2) ph_assemble2(mu, dw, dc, second); else { ph_assemble_obs(mu, dw, dc); ph_assemble_stage(mu, dw, dc, second); } }
OBCA_PHASE void ph_fused_obs4(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_obs<OB_VMID, 1>(sh.inst, sh, mu, 0.0, dc, fa); }
OBCA_PHASE void ph_fused_obs8(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_obs<OB_VMAX, 1>(sh.inst, sh, mu, 0.0, dc, fa); }
OBCA_PHASE void ph_fused_stage(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) { Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd}; assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An); }
OBCA_PHASE void ph_fused2(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {
    Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd};
    assemble_obs<2, 1>(sh.inst, sh, mu, 0.0, dc, fa); assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An);
}
OBCA_FN void ph_fused(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {
    if (g_sh.vm2) { ph_fused2(mu, dc, alpha, ay, az, ks, dwd); return; }
    if (g_sh.vmc == 1) ph_fused_obs4(mu, dc, alpha, ay, az, ks, dwd); else ph_fused_obs8(mu, dc, alpha, ay, az, ks, dwd);
    ph_fused_stage(mu, dc, alpha, ay, az, ks, dwd);
}
OBCA_PHASE int ph_riccati(double rho) { Shared &sh = g_sh; return riccati_backward(sh.inst, sh, rho); }
OBCA_PHASE void ph_direction_main(double mu, double dw, double dc, double rho, double tau) { Shared &sh = g_sh; direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S); }
OBCA_PHASE void ph_direction_obs2(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<2, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_PHASE void ph_direction_obs4(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<OB_VMID, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_PHASE void ph_direction_obs8(double mu, double dw, double dc, double tau) { Shared &sh = g_sh; direction_obs<OB_VMAX, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S); }
OBCA_FN void ph_direction_obs(double mu, double dw, double dc, double tau) { if (g_sh.vmc == 0) ph_direction_obs2(mu, dw, dc, tau); else if (g_sh.vmc == 1) ph_direction_obs4(mu, dw, dc, tau); else ph_direction_obs8(mu, dw, dc, tau); }
OBCA_PHASE void ph_direction2(double mu, double dw, double dc, double rho, double tau) {   // both parts in one call, see ph_assemble2
    Shared &sh = g_sh;
    direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (sh.S.ok) direction_obs<2, 0>(sh.inst, sh, mu, dw, dc, tau, sh.S);
}
// Search direction AND the block part of the first trial of the line search in one call
// (round 4).  The first trial always takes the fraction-to-the-boundary step lengths, which
// are known the moment the block back-substitution has been reduced over the wavefront
// -- so the blocks' steps stay in the lanes' registers (48 doubles for the 4 rounds of a
// 3-obstacle instance) and the trial point's block part is formed and condensed right away.
// Until round 4 the fused line search factorised every block a second time at the old
// point just to get that step back (a fifth of a pass).  The stage part of the trial follows
// as ph_fused_stage once the driver has set up the line search; later (backtracking)
// trials and everything on the cold paths recompute as before.  Bit for bit the numbers of the two-call sequence.
OBCA_PHASE void ph_direction2_trial(double mu, double dw, double dc, double rho, double tau, double ks) {
    Shared &sh = g_sh;
    direction_main(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (!sh.S.ok) return;
    ObsStep<2> keep[OB_KEEP][OBCA_NL];
    direction_obs<2, 0, 0, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S, keep);
    const FuseArgs fa = {sh.S.ap, fmin(sh.S.ap, sh.S.az), sh.S.az, ks, dw};
    assemble_obs<2, 1, 0, 0, 1>(sh.inst, sh, mu, 0.0, dc, fa, keep);
    PAR(lane) { if (lane == 0) sh.ft_done = 1; }
    LDS_SYNC();
}
// ks_first_trial > 0: also the block part of the first trial (main path only)
OBCA_FN void ph_direction(double mu, double dw, double dc, double rho, double tau, double ks_first_trial = 0.0) {
    if (g_sh.vm2 && g_sh.ft_ok && ks_first_trial > 0) { ph_direction2_trial(mu, dw, dc, rho, tau, ks_first_trial); return; }
    if (g_sh.vm2) { ph_direction2(mu, dw, dc, rho, tau); return; }
    ph_direction_main(mu, dw, dc, rho, tau);
    if (g_sh.S.ok) ph_direction_obs(mu, dw, dc, tau);
}
// ---- second-order correction (IPOPT A-5.5..A-5.9; Opts::max_soc > 0; cold path: one non-inlined function per step, every obstacle width inside)
// c_soc <- asoc * (first ? c(z) : c_soc) + c(zn)   (zn: the rejected trial point; rows as
// the assembly forms them: dynamics x_{k+1} - F, terminal x_N - xF, steering, obstacle rows)
template <int VM>
OBCA_FN void soc_accumulate(const Inst &I, Shared &sh, double asoc, int first) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb; gdbl *cs = sh.soc.csoc;
    PAR(lane) {
        for (int k = lane; k < N; k += OB_NT) {
            double v[2][5];
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const gdbl *z = w ? I.zn : I.z; const double t = z[l.t];
                double x[4], u[2], F[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = z[l.x + 4 * k + i];
                u[0] = z[l.u + 2 * k]; u[1] = z[l.u + 2 * k + 1];
                dyn_value(c, x, u, t, F);
#pragma unroll
                for (int i = 0; i < 4; i++) v[w][i] = z[l.x + 4 * (k + 1) + i] - F[i];
                v[w][4] = ((k ? z[l.u + 2 * k - 2] : 0.0) - u[0]) / (t * c.Ts) - z[l.ss + k];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) cs[4 * k + i] = asoc * (first ? v[0][i] : (double)cs[4 * k + i]) + v[1][i];
            cs[(l.yg - l.pi) + k] = asoc * (first ? v[0][4] : (double)cs[(l.yg - l.pi) + k]) + v[1][4];
        }
        if (lane < 4) { const int o_ = (l.nu - l.pi) + lane; cs[o_] = asoc * (first ? I.z[l.x + 4 * N + lane] - c.xF[lane] : (double)cs[o_]) + (I.zn[l.x + 4 * N + lane] - c.xF[lane]); }
        for (int it = lane; it < (N + 1) * nOb; it += OB_NT) {
            const int k = it / nOb, j = it - k * nOb; double r0[4], r1[4];
            { ObsIn<VM> in; load_obs<VM>(I, sh, I.z, k, j, in); obs_rows<VM>(c, in, r0); }
            { ObsIn<VM> in; load_obs<VM>(I, sh, I.zn, k, j, in); obs_rows<VM>(c, in, r1); }
#pragma unroll
            for (int r = 0; r < 4; r++) { const int o_ = (l.yo - l.pi) + 4 * it + r; cs[o_] = asoc * (first ? r0[r] : (double)cs[o_]) + r1[r]; }
        }
    }
    SYNC();
}
OBCA_PHASE void ph_soc_accumulate(double asoc, int first) { Shared &sh = g_sh; VM_CALL(soc_accumulate, sh.inst, sh, asoc, first); }
// the system at z with c_soc on the right-hand side (sh.A keeps the values of the iterate: f, theta, errors use the true rows)
OBCA_PHASE void ph_soc_assemble(double mu, double dw, double dc) {
    Shared &sh = g_sh;
    if (sh.vmc == 0) assemble_obs<2, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE);
    else if (sh.vmc == 1) assemble_obs<OB_VMID, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE); else assemble_obs<OB_VMAX, 0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE);
    assemble_stage<0, 1>(sh.inst, sh, mu, dw, dc, OB_NOFUSE, sh.A);
}
OBCA_PHASE int ph_soc_riccati(double rho) { Shared &sh = g_sh; return riccati_backward<1>(sh.inst, sh, rho); }
OBCA_PHASE void ph_soc_direction(double mu, double dw, double dc, double rho, double tau) {
    Shared &sh = g_sh;
    direction_main<1>(sh.inst, sh, sh.A, mu, dw, dc, rho, tau, sh.S);
    if (!sh.S.ok) return;
    if (sh.vmc == 0) direction_obs<2, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S);
    else if (sh.vmc == 1) direction_obs<OB_VMID, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S);
    else direction_obs<OB_VMAX, 0, 1>(sh.inst, sh, mu, dw, dc, tau, sh.S);
}
// trial point along the correction direction, assembled there as usual
OBCA_PHASE void ph_soc_fused(double mu, double dc, double alpha, double ay, double az, double ks, double dwd) {
    Shared &sh = g_sh; const FuseArgs fa = {alpha, ay, az, ks, dwd};
    if (sh.vmc == 0) assemble_obs<2, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa); else if (sh.vmc == 1) assemble_obs<OB_VMID, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa);
    else assemble_obs<OB_VMAX, 1, 1>(sh.inst, sh, mu, 0.0, dc, fa);
    assemble_stage<1>(sh.inst, sh, mu, 0.0, dc, fa, sh.An);
}

// ---- recalc_y = "yes" (ParkingSignedDist.jl:41; IPOPT recalc_y_feas_tol = 1e-6):
// once the iterate is (nearly) feasible its equality multipliers are replaced by the
// least-squares estimate -- the same structured solve with H := I, zero constraint right-hand
// side, gradients in their z-form; only the multiplier part of the solution is used.
// Cold path: one non-inlined function, every obstacle width inside.  1 = the multipliers were replaced (the assembly at hand is then stale).
#ifdef OBCA_EMU
static int g_emu_recalc_fail = 0;      // host test hook: every estimate is attempted (at every accepted iterate) and thrown away
#define OB_RECALC_FEAS_TOL (g_emu_recalc_fail ? 1e300 : 1e-6)
#else
#define OB_RECALC_FEAS_TOL 1e-6        // IPOPT recalc_y_feas_tol
#endif
// init = 1: IPOPT's initial multipliers (least-squares estimate at the starting point, kept only if its max-norm is <= constr_mult_init_max = 1e3)
OBCA_PHASE int ph_recalc_y(int init) {
    Shared &sh = g_sh; const Inst &I = sh.inst; const Lay &l = sh.l;
    if (sh.vmc == 0) assemble_obs<2, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE);
    else if (sh.vmc == 1) assemble_obs<OB_VMID, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE); else assemble_obs<OB_VMAX, 0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE);
    assemble_stage<0, 0, 1>(I, sh, 0.0, 0.0, 0.0, OB_NOFUSE, sh.A2);
    if (!riccati_backward(I, sh, 0.0)) return 0;
    direction_main<0, 1>(I, sh, sh.A2, 0.0, 0.0, 0.0, 0.0, 0.99, sh.S);
    if (!sh.S.ok) return 0;
    if (sh.vmc == 0) direction_obs<2, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S);
    else if (sh.vmc == 1) direction_obs<OB_VMID, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S);
    else direction_obs<OB_VMAX, 1, 0, 1>(I, sh, 0.0, 0.0, 0.0, 0.99, sh.S);
    double red[1][OBCA_NL];
    PAR(lane) { double w = 0; for (int i = l.pi + lane; i < l.zxL; i += OB_NT) { const double v = I.d[i], y1 = fabs(I.z[i] + v); w = (v == v && fabs(v) <= 1e300 && w <= 1e300) ? fmax(w, y1) : 1e301; } red[0][LI(lane)] = w; }
    const double ymax = wred_max(red[0]);
#ifdef OBCA_EMU
    if (g_emu_recalc_fail && !init) return 0;                                // (host test hook: the estimate is discarded AFTER the records were overwritten)
#endif
    // a non-finite entry (or, at the start, an estimate beyond constr_mult_init_max): keep the multipliers
    if (ymax > 1e300 || (init && ymax > 1e3)) return 0;
    PAR(lane) { for (int i = l.pi + lane; i < l.zxL; i += OB_NT) I.z[i] += I.d[i]; }
    SYNC();
    if (!init) sh.soc.nrecalc++;
    return 1;
}

// the iterate the solve ends with (or is parked at) must sit in the instance's own
// buffer `home`: copy it over if the last accepted trial left it in the other one
OBCA_PHASE void ph_bring_home() {
    Shared &sh = g_sh; Inst &I = sh.inst;
    PAR(lane) { for (int i = lane; i < sh.l.len; i += OB_NT) I.zn[i] = I.z[i]; }
    SYNC();
    PAR(lane) { if (lane == 0) { gdbl *t_ = I.z; I.z = I.zn; I.zn = t_; } }
    SYNC();
}

// ---------------------------------------------------------------- the interior-point driver
enum { ST_OPTIMAL = 0, ST_USERLIMIT = 1, ST_ERROR = 2, ST_SUSPENDED = 3 };

// Time slicing ("two-launch schedule").  A solve may be cut at the top of an interior-point iteration and continued by
// a later launch: everything the iteration loop carries across iterations besides the iterate itself (which lives in HBM anyway) is a
// handful of scalars and the filter, saved in the instance's slice record.  A resumed solve recomputes the assembly at the same point, so
// the sequence of iterates is bit-identical to an uninterrupted solve.  Record layout (doubles):
OBCA_FN double filt_get(const Shared &sh, const gdbl *st, int i, int c) { return i < OB_FILT_LDS ? sh.filt[i][c] : st[SL_FILT + 2 * i + c]; }

// The reference's acceptance test on the current iterate, with its quirks (ParkingConstraints.jl:29-149): in variable-time
// mode only the speed row of the dynamics is kept (:76-79), only the LAST obstacle's rows survive (:108-130), the separation row is
// evaluated without any slack, the steering rate divides by timeScale[1].  1 = every class <= 5e-5.  Cold path (failed attempts only).
template <int VM>
OBCA_FN int ref_constraints(const Inst &I, Shared &sh, int sd) {
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = c.N, nOb = c.nOb, M = c.M; const gdbl *z = I.z;
    const double t = c.fixTime ? 1.0 : z[l.t];
    double red[1][OBCA_NL];
    PAR(lane) {
        double w = -1e300;                                     // running max of every "should be <= 0" quantity
        for (int i = lane; i < M * (N + 1); i += OB_NT) w = fmax(w, -z[l.lam + i]);
        for (int i = lane; i < 4 * nOb * (N + 1); i += OB_NT) w = fmax(w, -z[l.mu + i]);
        for (int k = lane; k < N; k += OB_NT) {
            double x[4], u[2], F[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = z[l.x + 4 * k + i];
            u[0] = z[l.u + 2 * k]; u[1] = z[l.u + 2 * k + 1];
            w = fmax(w, fmax(fabs(u[0]) - 0.6, fabs(u[1]) - 0.4));
            dyn_value(c, x, u, t, F);
            if (c.fixTime) {
#pragma unroll
                for (int i = 0; i < 4; i++) w = fmax(w, fabs(z[l.x + 4 * (k + 1) + i] - F[i]));
            } else w = fmax(w, fabs(z[l.x + 4 * (k + 1) + 3] - F[3]));
            w = fmax(w, fabs(u[0] - (k ? z[l.u + 2 * k - 2] : 0.0)) / (t * c.Ts) - 0.6);
        }
        if (lane < 4) w = fmax(w, fabs(z[l.x + 4 * N + lane] - c.xF[lane]));
        if (lane == 4) w = fmax(w, fabs(t - 1) - 0.2);
        if (nOb > 0) for (int k = lane; k <= N; k += OB_NT) {
            ObsIn<VM> in; load_obs<VM>(I, sh, z, k, nOb - 1, in);
            in.sl = 0; in.so = 0;
            double p1 = 0, p2 = 0, beta = 0;
#pragma unroll
            for (int i = 0; i < VM; i++) if (i < in.v) { p1 += in.a1[i] * in.lam[i]; p2 += in.a2[i] * in.lam[i]; beta += in.b[i] * in.lam[i]; }
            double sn, cs; sincos_bounded(in.psi, &sn, &cs);
            const double r0 = p1 * p1 + p2 * p2 - 1;
            const double r1 = in.mu[0] - in.mu[2] + cs * p1 + sn * p2, r2 = in.mu[1] - in.mu[3] - sn * p1 + cs * p2;
            const double r3 = -(c.g[0] * in.mu[0] + c.g[1] * in.mu[1] + c.g[2] * in.mu[2] + c.g[3] * in.mu[3]) + (in.X + cs * c.off) * p1 +
                              (in.Y + sn * c.off) * p2 - beta - OB_DMIN;
            w = fmax(w, fmax(sd ? fabs(r0 + 1) - 1 : r0, fmax(fmax(fabs(r1), fabs(r2)), -r3)));
        }
        red[0][LI(lane)] = w;
    }
    const double worst = wred_max(red[0]);
    return worst <= 5e-5;
}
OBCA_PHASE int ph_ref_constraints(int sd) { Shared &sh = g_sh; return sh.vmc == 0 ? ref_constraints<2>(sh.inst, sh, sd) : (sh.vmc == 1 ? ref_constraints<OB_VMID>(sh.inst, sh, sd) : ref_constraints<OB_VMAX>(sh.inst, sh, sd)); }

// What the iteration loop carries lives in LDS (Shared::drv), not in registers: the phases are non-inlined calls that use the whole register file, so every
// value the driver kept in a register was spilled to scratch -- i.e. to HBM -- before each call and fetched back after it (~500 spill instructions in round 2's
// kernel body, a memory round trip behind every phase).  An LDS slot costs a
// ~100-clock read where the value is needed and nothing at a call.  (Measured and not
// kept: the state in registers between the calls and copied to / from LDS around each call -- the register allocator then spills MORE, 277 scratch stores / 571
// loads in the kernel body against 96 / 279.)
// pow of the driver as a call: inlined, the ~40 polynomial coefficients of the library routine were hoisted out of the iteration loop as loop invariants, did not survive the
// phase calls in registers, and were kept in SCRATCH -- 18 spill stores before the loop and a scratch load per coefficient at every evaluation (ISA of round 6).
OBCA_PHASE double ob_pow(double x, double y) { return pow(x, y); }
// PH: after a phase call nothing loaded before it is assumed still valid (a compiler-level memory clobber, no code): the optimiser knows which LDS words a phase
// writes, hoisted the loads of the option record and of other loop-invariant words out of the iteration loop -- and, every phase using the whole register file, had
// to keep them in SCRATCH: a spill store at the hoist, a scratch load (a memory round trip) at every use.  Re-reading LDS where the value is needed is the cheaper miss.
#ifdef OBCA_EMU
#define PH(call) call
#define OB_NOHOIST() ((void)0)
#else
#define PH(call) do { call; asm volatile("" ::: "memory"); } while (0)
#define OB_NOHOIST() asm volatile("" ::: "memory")
#endif
#ifdef OBCA_PROFILE_DRV      // (with -DOBCA_PROFILE: the driver's clocks split three ways -- slots ric_p1 / ric_p2 of the phase profile: line-search set-up, acceptance; the rest stays in `other`)
#define PROF_DRV(I, id) PROF(I, id)
#else
#define PROF_DRV(I, id) ((void)0)
#endif
// Second-order correction after the FIRST trial step of an iteration was rejected without
// reducing theta (IPOPT A-5.5..A-5.9, kappa_soc = 0.99): up to max_soc steps that
// solve the system of the iterate again with c_soc = alpha c(z) + c(trial) on the right-hand
// side, each tested like a trial step (with the ORIGINAL alpha in the switching and
// Armijo conditions).  1 = accepted: the trial buffer holds z + asoc d_soc with its
// assembly, D.alpha / D.az are those of the correction.  0: the Newton direction of the
// iteration is rebuilt (the correction overwrote it) and the backtracking goes on.
// The phases are fused (factorise + solve), so a correction costs a full pass.
OBCA_PHASE int ph_soc_try(double tht_first) {
    Shared &sh = g_sh; Drv &D = sh.drv; const Opts &o = sh.o; gdbl *const st = sh.sol.sl.st;
    const double alpha = D.alpha, th = D.th, phi = D.phi, gd = D.gd;
    double th_old = 0, th_tr = tht_first, asoc = alpha, azs = D.az; int acc = 0;
    // the correction's direction goes to its own buffer; (dt, nu) and the step scalars of the iteration's direction are kept aside
    gdbl *const d_iter = sh.inst.d;
    PAR(lane) { if (lane == 0) { sh.inst.d = sh.soc.dsoc; for (int i = 0; i < 5; i++) sh.soc.coef_keep[i] = sh.coef[i]; sh.soc.S_keep = sh.S; } }
    LDS_SYNC();
    for (int ps = 0; ps < sh.soc.max_soc && !acc && (ps == 0 || th_tr <= 0.99 * th_old); ps++) {
        th_old = th_tr;
        ph_soc_accumulate(asoc, ps == 0);
        ph_soc_assemble(D.mu, D.dw, D.dc_val);
        int a_ = sh.A.ok;
        if (a_) a_ = ph_soc_riccati(o.rho_term);
        if (a_) { ph_soc_direction(D.mu, D.dw, D.dc_val, o.rho_term, D.tau); a_ = sh.S.ok; }
        if (!a_) break;
        asoc = sh.S.ap; azs = sh.S.az;
        ph_soc_fused(D.mu, D.dc_val, asoc, fmin(asoc, azs), azs, o.kappa_sigma, D.dw);
        sh.soc.nsoc++;
        const double ft = sh.An.f, tht = sh.An.th1, pht = ft - D.mu * sh.An.bar;
        if (!(ft == ft && tht == tht)) break;
        th_tr = tht;
        if (pht == pht && tht < D.th_max) {
            int okf = 1; const int nf = D.nf;
            for (int i = 0; i < nf && okf; i++) if (!(tht < filt_get(sh, st, i, 0) || pht < filt_get(sh, st, i, 1))) okf = 0;
            if (okf) {
                const int sw = gd < 0 && alpha * D.pw_gd > o.delta * D.pw_th;
                const int armijo = pht <= phi + o.eta_phi * alpha * gd;
                if (th <= D.th_min && sw) { if (armijo) acc = 1; }
                else if (tht <= (1 - o.gamma_theta) * th || pht <= phi - o.gamma_phi * th) {
                    acc = 1;
                    if (!(sw && armijo) && nf < OB_FILT) {
                        PAR(lane) { if (lane == 0) { const double f0 = (1 - o.gamma_theta) * th, f1 = phi - o.gamma_phi * th;
                                                     if (nf < OB_FILT_LDS) { sh.filt[nf][0] = f0; sh.filt[nf][1] = f1; } else { st[SL_FILT + 2 * nf] = f0; st[SL_FILT + 2 * nf + 1] = f1; } } }
                        SYNC();
                        D.nf = nf + 1;
                    }
                }
            }
        }
    }
    if (acc) { PAR(lane) { if (lane == 0) sh.inst.d = d_iter; } LDS_SYNC(); D.alpha = asoc; D.az = azs; sh.soc.nsoc_acc++; return 1; }
    // not accepted: the backtracking goes on along the iteration's own direction.  Its stage part sits untouched in d; its x part -- the forward sweep's trajectory in LDS, which
    // the correction's sweep overwrote -- is a copy of entries of d (direction_main: d.x[k] = s_k[0..3], d.u[k] = s_{k+1}[4..5]) and is put back from there; (dt, nu) and the step
    // scalars come back from where they were kept.  The stage / Riccati records hold the correction's system, which nothing reads before the next assembly overwrites them.
    sh.soc.nrebuild++;      // (the count of rejected corrections; no pass of its own since round 6)
    {
        const Lay &l = sh.l; const int N = sh.c.N; const gdbl *d = d_iter;
        PAR(lane) {
            if (lane == 0) { sh.inst.d = d_iter; for (int i = 0; i < 5; i++) sh.coef[i] = sh.soc.coef_keep[i]; sh.S = sh.soc.S_keep; }
            for (int k = lane; k <= N; k += OB_NT) {
#pragma unroll
                for (int i = 0; i < 4; i++) g_traj[(size_t)k * 6 + i] = d[l.x + 4 * k + i];
                g_traj[(size_t)k * 6 + 4] = k ? (double)d[l.u + 2 * (k - 1)] : 0.0; g_traj[(size_t)k * 6 + 5] = k ? (double)d[l.u + 2 * (k - 1) + 1] : 0.0;
            }
        }
        SYNC();
    }
    return 0;
}
// (a call, not inlined into solve_instance's two sites -- first attempt and retry --: the driver's code is in the kernel once, 30 KB less on the instruction cache two CUs share;
//  its operands are named through g_sh here, so that they stay LDS accesses)
OBCA_PHASE void ipm_attempt() {
    Shared &sh = g_sh; Drv &D = sh.drv; const Opts &o = sh.o; Result &R = sh.sol.R; Slice &sl = sh.sol.sl;
    gdbl *const st = sl.st;
    const AsmOut &A = sh.A;
    sh.soc.nsoc = 0; sh.soc.nsoc_acc = 0; sh.soc.nrecalc = 0; sh.soc.nrebuild = 0;
    D.mu = o.mu_init; D.dw_last = 0; D.nf = 0; D.it = 0; D.nreg = 0; D.th_min = 0; D.th_max = 0; D.f = 0; D.pinf = 0; D.dinf = 0; D.status = ST_USERLIMIT;
    sh.soc.xpass0 = 0;      // (SL_XPASS is cumulative over the slices of an attempt like SL_NREG: the ordering kernel ranks by both)
    sh.soc.nrest = 0; sh.soc.reset_th = 0;
    if (sl.resume) {
        sh.soc.xpass0 = (int)st[SL_XPASS]; sh.soc.nrest = (int)st[SL_NREST] & 15; sh.soc.reset_th = ((int)st[SL_NREST] >> 4) & 1;
        D.it = (int)st[SL_IT]; D.nf = (int)st[SL_NF]; D.nreg = (int)st[SL_NREG]; D.mu = st[SL_MU]; D.dw_last = st[SL_DWLAST]; D.th_min = st[SL_THMIN];
        D.th_max = st[SL_THMAX];
        D.pinf = st[SL_PINF];
        if ((int)st[SL_HAVE]) { PAR(lane) { if (lane == 0) asm_unpack(sh.A, st + SL_ASM); } }
        PAR(lane) { const int nl = D.nf < OB_FILT_LDS ? D.nf : OB_FILT_LDS; for (int i = lane; i < 2 * nl; i += OB_NT) (&sh.filt[0][0])[i] = st[SL_FILT + i]; }
        SYNC();
        D.have_asm = (int)st[SL_HAVE];
        sl.resume = 0;
    // (IPOPT's default initial multipliers, an option here: Opts lsq_init)
    } else { PH(ph_init(o.bound_push, o.bound_frac)); D.have_asm = 0; if (sh.soc.lsq_init) ph_recalc_y(1); }
    D.tau = fmax(o.tau_min, 1 - D.mu);
    D.p_start = D.it + D.nreg;
    D.dc_mu = -1.0; D.dc_val = 0;
    // D.have_asm = 1: sh.A already holds the assembly of the current iterate, left behind by the accepted trial of the previous iteration (ph_fused)
    for (;;) {
        OB_NOHOIST();
        // out of budget: park the loop state, a later launch continues
        if (sl.budget > 0 && sl.used + (D.it + D.nreg - D.p_start) + sh.soc.nsoc + sh.soc.nrecalc >= sl.budget) {
            PAR(lane) {
                if (lane == 0) { st[SL_IT] = D.it; st[SL_NF] = D.nf; st[SL_NREG] = D.nreg; st[SL_MU] = D.mu; st[SL_DWLAST] = D.dw_last; st[SL_THMIN] = D.th_min; st[SL_THMAX] = D.th_max; st[SL_PINF] = D.pinf; st[SL_HAVE] = D.have_asm; st[SL_XPASS] = sh.soc.xpass0 + sh.soc.nsoc + sh.soc.nrecalc; st[SL_NREST] = sh.soc.nrest + 16 * sh.soc.reset_th; if (D.have_asm) asm_pack(st + SL_ASM, sh.A); }
                const int nl = D.nf < OB_FILT_LDS ? D.nf : OB_FILT_LDS;                 // (entries beyond the LDS part are in the record already)
                for (int i = lane; i < 2 * nl; i += OB_NT) st[SL_FILT + i] = (&sh.filt[0][0])[i];
            }
            D.status = ST_SUSPENDED; break;
        }
        if (D.mu != D.dc_mu) { D.dc_val = o.dc_bar * ob_pow(D.mu, o.kappa_c); D.dc_mu = D.mu; }   // a pow is a ~3k-clock dependent chain: keep it while mu stays
        PROF(sh.inst, PF_OTHER); if (!D.have_asm) PH(ph_assemble(D.mu, 0.0, D.dc_val, 0));
        D.have_asm = 0;
        if (D.it == 0 || sh.soc.reset_th) { D.th_min = 1e-4 * fmax(1.0, A.th1); D.th_max = 1e4 * fmax(1.0, A.th1); sh.soc.reset_th = 0; }
        D.f = A.f; D.pinf = A.pinf; D.dinf = A.dinf;
#ifdef OBCA_EMU      // OBCA_EMU_TRACE=1: one line per iteration in the format of the CPU checker's `verbose` option (test infrastructure), so that two traces can be laid side by side
        if (getenv("OBCA_EMU_TRACE")) printf("it %3d f=% .8e pinf=%.2e dinf=%.2e cinf=%.2e mu=%.1e dw=%.1e t=%.4f\n", D.it, A.f, A.pinf, A.dinf, A.cinf0, D.mu, D.dw_last, (double)sh.inst.z[sh.l.t]);
#endif
        {
            const double sd = fmax(o.s_max, (A.sumy + A.sumz) / (A.nm + A.nb)) / o.s_max;
            const double sc = fmax(o.s_max, A.sumz / A.nb) / o.s_max;
            const double E0 = fmax(A.dinf / sd, fmax(A.pinf, A.cinf0 / sc));
            if (E0 <= o.tol && A.pinf <= o.constr_viol_tol && A.dinf <= o.dual_inf_tol && A.cinf0 <= o.compl_inf_tol) { D.status = ST_OPTIMAL; break; }
            if (D.it >= o.max_iter) { D.status = ST_USERLIMIT; break; }
            if (!(A.f == A.f) || !(A.pinf == A.pinf) || !(A.dinf == A.dinf)) { D.status = ST_ERROR; break; }
            D.sd = sd; D.sc = sc;
        }
        // barrier update: mu <- max(tol/10, min(kappa_mu mu, mu^theta_mu)) while the barrier problem is solved to kappa_eps mu
        D.mu_changed = 0;
        D.cm = cinf_mu(A, D.mu);
        for (;;) {
            const double Emu = fmax(D.dinf / D.sd, fmax(D.pinf, D.cm / D.sc));
            if (Emu <= o.kappa_eps * D.mu && D.mu > o.tol / 10) {
                D.mu = fmax(o.tol / 10, fmin(o.kappa_mu * D.mu, ob_pow(D.mu, o.theta_mu)));
                D.tau = fmax(o.tau_min, 1 - D.mu); D.nf = 0; D.mu_changed = 1;
                D.dc_val = o.dc_bar * ob_pow(D.mu, o.kappa_c); D.dc_mu = D.mu;
                // complementarity error w.r.t. the new mu: from the extreme products of the assembly at hand (round 2 re-assembled for it)
                D.cm = cinf_mu(A, D.mu);
            } else break;
        }
        // search direction with inertia correction (IPOPT Algorithm IC)
        D.dw = 0; D.ok = 0;
        for (D.tr = 0; D.tr < 60; D.tr++) {
            PROF(sh.inst, PF_OTHER); if (D.tr > 0 || D.mu_changed) PH(ph_assemble(D.mu, D.dw, D.dc_val, 0));
            int a_ = A.ok;
            PROF(sh.inst, PF_OTHER); if (a_) { PH(a_ = ph_riccati(o.rho_term)); }
            sh.ft_done = 0;
            PROF(sh.inst, PF_OTHER); if (a_) { PH(ph_direction(D.mu, D.dw, D.dc_val, o.rho_term, D.tau, o.kappa_sigma)); a_ = sh.S.ok; }
            if (a_) { D.ok = 1; break; }
#ifdef OBCA_EMU
            if (getenv("OBCA_EMU_TRACE")) printf("   rung: it %d dw %.1e failed in %s (riccati stage %d)\n", D.it, D.dw, !A.ok ? "assembly" : (!sh.ric_ok ? "riccati" : "border"), g_emu_ric_fail_stage);
#endif
            D.nreg++;
            if (D.dw == 0) D.dw = D.dw_last == 0 ? o.dw0 : fmax(o.dw_min, o.kw_dec * D.dw_last);
            else D.dw *= (D.dw_last == 0 ? o.kw_inc0 : o.kw_inc);
            if (D.dw > o.dw_max) break;
        }
        // where IPOPT would enter its restoration phase (inertia ladder exhausted here, failed line search below): repair the degenerate obstacle blocks -- if there are
        // any -- restart the barrier, empty the filter and go on (obca_opts.restoration; restore_blocks above)
#define OB_RESTORE_AND_CONTINUE { sh.soc.nrest++; sh.soc.reset_th = 1; D.mu = o.mu_init; D.tau = fmax(o.tau_min, 1 - D.mu); D.nf = 0; D.dw_last = 0; D.have_asm = 0; continue; }
        if (!D.ok) { if (sh.soc.restoration && sh.soc.nrest < OB_MAX_RESTORE && ph_restore(o.bound_push) > 0) OB_RESTORE_AND_CONTINUE; D.status = ST_ERROR; break; }
        if (D.dw > 0) D.dw_last = D.dw;
        {
            const double th = A.th1, gd = sh.S.gd;
            D.th = th; D.phi = A.f - D.mu * A.bar; D.gd = gd; D.az = sh.S.az; D.pw_th = 0; D.pw_gd = 0;
            double amin;
            if (gd < 0) {
                amin = fmin(o.gamma_theta, o.gamma_phi * th / (-gd));
                // once per iteration (also the switching condition of every trial).  (Round 6 measured the two pows side by side in two lanes of one call: SLOWER -- the driver's
                // clocks per pass 23.7 k -> 29.9 k, `value` -2.5 %: with wave-uniform arguments the library pow takes scalar branches around its special cases, with per-lane
                // arguments it executes them all.  profiles/r06_ab_pow_in_two_lanes.txt)
                D.pw_th = ob_pow(th, o.s_theta); D.pw_gd = ob_pow(-gd, o.s_phi);
                if (th <= D.th_min) amin = fmin(amin, o.delta * D.pw_th / D.pw_gd);
            } else amin = o.gamma_theta;
            D.amin = amin * o.gamma_alpha;
        }
        D.alpha = sh.S.ap; D.acc = 0;
        PROF_DRV(sh.inst, PF_RIC_P1);
        while (D.alpha >= D.amin) {
            // the trial point z + alpha d goes to the second iterate buffer together with
            // its assembly (mu as is, delta_w = 0: what the next iteration starts from)
            PROF(sh.inst, PF_OTHER);
            // first trial: its block part ran with the direction (ph_direction2_trial)
            if (sh.ft_done) { sh.ft_done = 0; PH(ph_fused_stage(D.mu, D.dc_val, D.alpha, fmin(D.alpha, D.az), D.az, o.kappa_sigma, D.dw)); }
            else PH(ph_fused(D.mu, D.dc_val, D.alpha, fmin(D.alpha, D.az), D.az, o.kappa_sigma, D.dw));
            const double ft = sh.An.f, tht = sh.An.th1, pht = ft - D.mu * sh.An.bar, alpha = D.alpha, th = D.th, phi = D.phi, gd = D.gd;
            if (ft == ft && tht == tht && pht == pht && tht < D.th_max) {
                int okf = 1; const int nf = D.nf;
                for (int i = 0; i < nf && okf; i++) if (!(tht < filt_get(sh, st, i, 0) || pht < filt_get(sh, st, i, 1))) okf = 0;
                if (okf) {
                    const int sw = gd < 0 && alpha * D.pw_gd > o.delta * D.pw_th;
                    const int armijo = pht <= phi + o.eta_phi * alpha * gd;
                    if (th <= D.th_min && sw) { if (armijo) { D.acc = 1; break; } }
                    else if (tht <= (1 - o.gamma_theta) * th || pht <= phi - o.gamma_phi * th) {
                        D.acc = 1;
                        if (!(sw && armijo) && nf < OB_FILT) {
                            PAR(lane) { if (lane == 0) { const double f0 = (1 - o.gamma_theta) * th, f1 = phi - o.gamma_phi * th;
                                                         if (nf < OB_FILT_LDS) { sh.filt[nf][0] = f0; sh.filt[nf][1] = f1; } else { st[SL_FILT + 2 * nf] = f0; st[SL_FILT + 2 * nf + 1] = f1; } } }
                            SYNC();
                            D.nf = nf + 1;
                        }
                        break;
                    }
                }
            }
            // second-order correction: first trial step only (alpha is still the full step sh.S.ap), and only if it did not reduce theta
            if (sh.soc.max_soc > 0 && alpha == sh.S.ap && ft == ft && tht == tht && tht >= th) {
                if (ph_soc_try(tht)) { D.acc = 1; break; }
            }
            D.alpha = 0.5 * alpha;
        }
#ifdef OBCA_EMU
        if (getenv("OBCA_EMU_TRACE")) printf("   ls: alpha_max %.3e accepted %.3e soc %d dw %.1e nreg %d\n", sh.S.ap, D.acc ? D.alpha : 0.0, sh.soc.nsoc_acc, D.dw, D.nreg);
#endif
        if (!D.acc) { if (sh.soc.restoration && sh.soc.nrest < OB_MAX_RESTORE && ph_restore(o.bound_push) > 0) OB_RESTORE_AND_CONTINUE; D.status = ST_ERROR; break; }   // IPOPT would enter restoration here
#undef OB_RESTORE_AND_CONTINUE
        // accepted: the trial buffer becomes the iterate, its assembly the current one
        PAR(lane) { if (lane == 0) { Inst &I = sh.inst; gdbl *t_ = I.z; I.z = I.zn; I.zn = t_; sh.A = sh.An; } }
        LDS_SYNC();
        D.have_asm = 1;
        PROF_DRV(sh.inst, PF_RIC_P2);
        // recalc_y = "yes": least-squares multipliers at a (nearly) feasible iterate.  Whether the estimate is kept or not, the
        if (sh.soc.recalc_y && sh.A.pinf < OB_RECALC_FEAS_TOL) { ph_recalc_y(0); D.have_asm = 0; }
                                                                                                // call overwrote the stage / obstacle / Riccati records with
                                                                                                // the least-squares system: the next iteration assembles afresh
        D.it++;
    }
    // (a correction and a multiplier re-estimate are full passes each)
    sl.used += D.it + D.nreg - D.p_start + sh.soc.nsoc + sh.soc.nrecalc;
    R.status = D.status; R.iters = D.it; R.nreg = D.nreg; R.obj = D.f; R.pinf = D.pinf; R.dinf = D.dinf; R.mu = D.mu;
#if defined(OBCA_PROFILE) && !defined(OBCA_EMU)      // diagnostic counters of the IPOPT switches (slots behind the phase clocks): corrections tried / accepted, rebuilds, multiplier re-estimates
    if (LANE0) { sh.prof[13] += sh.soc.nsoc; sh.prof[14] += sh.soc.nrebuild + 1e-3 * sh.soc.nsoc_acc; sh.prof[15] += sh.soc.nrecalc; }
#endif
}

// Full solve of one instance (pointers already in g_sh.inst): first attempt, and on Error/UserLimit one re-solve from the last
// iterate (ParkingSignedDist.jl:256-290).  info[8] = {status, iterations, objective, pinf, dinf, mu, #regularisations, exitflag}
// Slicing: `st` is the instance's slice record, mode 1 resumes from it, budget > 0 limits the passes of this launch (info[0] = 3 when the
// solve was parked; the iterate buffer then holds the point to continue from).
OBCA_FN void solve_instance(int N, const Opts &o_arg, double *info, gdbl *st = nullptr, int mode = 0, int budget = 0, int max_soc = 0, int recalc_y = 0, int lsq_init = 0, int restoration = 0) {
    Shared &sh = g_sh;
    PAR(lane) {
        for (int i = lane; i < OB_HDR; i += OB_NT) sh.hdr[i] = sh.inst.prob[i];
        // (Shared::soc.csoc is set by the caller, like the pointers of Shared::inst)
        if (lane == 0) { sh.o = o_arg; sh.soc.max_soc = sh.soc.csoc ? max_soc : 0; sh.soc.recalc_y = recalc_y; sh.soc.lsq_init = lsq_init; sh.soc.restoration = restoration; }
    }
    SYNC();
    PAR(lane) {
        if (lane <= OB_NOBMAX) sh.roff[lane] = (int)sh.hdr[PH_ROFF + lane];
        if (lane < OB_NOBMAX) sh.vOb[lane] = (int)sh.hdr[PH_VOB + lane];
        if (lane == 0) {
            Consts &c = sh.c;
            c.N = N; c.Ts = sh.hdr[PH_TS]; c.L = sh.hdr[PH_L]; c.iL = 1.0 / c.L; c.off = sh.hdr[PH_OFF];
            for (int i = 0; i < 4; i++) { c.g[i] = sh.hdr[PH_G + i]; c.xl[i] = sh.hdr[PH_XL + i]; c.xu[i] = sh.hdr[PH_XU + i]; c.x0[i] = sh.hdr[PH_X0 + i]; c.xF[i] = sh.hdr[PH_XF + i]; }
            c.fixTime = (int)sh.hdr[PH_FIX]; c.nOb = (int)sh.hdr[PH_NOB]; c.M = (int)sh.hdr[PH_M];
            c.dist = (int)sh.hdr[PH_DIST];
            c.wa = (c.fixTime || c.dist) ? 0.5 : 0.1; c.wpsi = c.fixTime ? 1e-2 : 1e-4;      // ParkingDist.jl:87
            make_layout(c.N, c.nOb, c.M, sh.l);
            int vmx = 0; for (int j = 0; j < c.nOb; j++) { int v = (int)sh.hdr[PH_VOB + j]; if (v > vmx) vmx = v; }
            sh.vm2 = vmx <= 2; sh.vmc = vmx <= 2 ? 0 : (vmx <= OB_VMID ? 1 : 2);
            sh.ft_ok = (c.N + 1) * c.nOb <= OB_KEEP * OB_NT; sh.ft_done = 0;
        }
    }
    init_unpack_table(sh);
    SYNC();
    init_ric_table(sh);
    SYNC();
    // exit flag: ParkingSignedDist.jl:256-290 (Optimal -> 1; else one retry from the last iterate; if that fails too the reference's own
    // acceptance test decides) and ParkingDist.jl:245-289 (the test runs before the retry; after a failed retry it is inverted)
    // (this function's own state lives in LDS as well -- Shared::sol -- for the reason given at ipm_attempt)
    Sol &X = sh.sol;
    X.home = sh.inst.z;
    X.sl.st = st; X.sl.resume = mode == 1; X.sl.budget = budget; X.sl.used = 0;
    X.att = 0; X.it_prev = 0; X.nreg_prev = 0;
    if (mode == 1) { X.att = (int)st[SL_ATT]; X.it_prev = (int)st[SL_ITPREV]; X.nreg_prev = (int)st[SL_NREGPREV]; }
    X.R.status = ST_ERROR; X.R.iters = 0; X.R.nreg = 0; X.R.obj = X.R.pinf = X.R.dinf = X.R.mu = 0;
    X.ef = 0; X.iters = 0; X.nreg = 0; X.retry = X.att;
    if (X.att == 0) {
        ipm_attempt();
        X.iters = X.R.iters; X.nreg = X.R.nreg;
        if (X.R.status != ST_SUSPENDED) {
            X.ef = (X.R.status == ST_OPTIMAL); X.retry = !X.ef;
            if (X.retry && sh.c.dist && ph_ref_constraints(0)) { X.ef = 1; X.retry = 0; }
            if (X.retry) { X.att = 1; X.it_prev = X.R.iters; X.nreg_prev = X.R.nreg; }
        }
    }
    if (X.retry && X.R.status != ST_SUSPENDED) {
        ipm_attempt();
        X.iters = X.it_prev + X.R.iters; X.nreg = X.nreg_prev + X.R.nreg;
        if (X.R.status == ST_OPTIMAL) X.ef = 1;
        else if (X.R.status != ST_SUSPENDED) { const int feas = ph_ref_constraints(sh.c.dist ? 0 : 1); X.ef = sh.c.dist ? !feas : feas; }
    }
    if (X.R.status == ST_SUSPENDED) {
        PAR(lane) { if (lane == 0) { st[SL_ATT] = X.att; st[SL_ITPREV] = X.it_prev; st[SL_NREGPREV] = X.nreg_prev; } }
        X.ef = 0;
    }
    // the accepted trial points alternate between the two iterate buffers; results and parked solves live in the instance's own
    if (sh.inst.z != X.home) ph_bring_home();
    PAR(lane) {
        if (lane == 0) { info[0] = X.R.status; info[1] = X.iters; info[2] = X.R.obj; info[3] = X.R.pinf; info[4] = X.R.dinf; info[5] = X.R.mu; info[6] = X.nreg; info[7] = X.ef; }
    }
    SYNC();
}

