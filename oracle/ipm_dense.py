"""
TEST INFRASTRUCTURE ONLY (oracle).  Dense textbook primal-dual interior-point method
(Waechter & Biegler 2006 "Algorithm A": monotone barrier, filter line search, inertia
correction) on   min f(v)  s.t. c(v)=0, lb<=v<=ub.   O(n^3) per iteration: short horizons only.
Used to cross-check the structured oracle / HIP path (same local optimum, same objective).
Option values mirror the reference's IPOPT call (ParkingSignedDist.jl:41-43: tol=1e-5,
max_iter=200, min_hessian_perturbation=1e-12, jacobian_regularization_value=1e-7) and IPOPT
defaults for everything it does not override [ext].
"""
import numpy as np
import scipy.linalg as sla


def inertia_ldl(K):
    lu, d, perm = sla.ldl(K, lower=True)
    n = K.shape[0]
    pos = neg = zero = 0
    i = 0
    while i < n:
        if i + 1 < n and d[i + 1, i] != 0.0:
            ev = np.linalg.eigvalsh(d[i:i + 2, i:i + 2])
            for e in ev:
                if e > 0: pos += 1
                elif e < 0: neg += 1
                else: zero += 1
            i += 2
        else:
            e = d[i, i]
            if e > 0: pos += 1
            elif e < 0: neg += 1
            else: zero += 1
            i += 1
    return pos, neg, zero


class Opts:
    tol = 1e-5; max_iter = 200
    mu_init = 0.1; kappa_eps = 10.0; kappa_mu = 0.2; theta_mu = 1.5; tau_min = 0.99
    bound_push = 1e-2; bound_frac = 1e-2
    dw_min = 1e-12; dw0 = 1e-4; dw_max = 1e40; kw_inc0 = 100.0; kw_inc = 8.0; kw_dec = 1.0 / 3
    dc_bar = 1e-7; kappa_c = 0.25
    gamma_theta = 1e-5; gamma_phi = 1e-8; delta = 1.0; s_theta = 1.1; s_phi = 2.3; eta_phi = 1e-8
    gamma_alpha = 0.05; s_max = 100.0; kappa_sigma = 1e10
    constr_viol_tol = 1e-4; dual_inf_tol = 1.0; compl_inf_tol = 1e-4
    verbose = False


def solve(nlp, v0, o=Opts()):
    n, m = nlp.n, nlp.m
    lb, ub, mult = nlp.lb, nlp.ub, nlp.mult
    free = np.ones(n, bool)
    if getattr(nlp, "fixTime", 0):
        free[nlp.it] = False
    IL = np.isfinite(lb) & free; IU = np.isfinite(ub) & free
    v = v0.copy()
    # push into the interior (IPOPT sec 3.6)
    pL = np.minimum(o.bound_push * np.maximum(1, np.abs(lb)), o.bound_frac * (ub - lb))
    pU = np.minimum(o.bound_push * np.maximum(1, np.abs(ub)), o.bound_frac * (ub - lb))
    both = IL & IU
    v[both] = np.minimum(np.maximum(v[both], lb[both] + pL[both]), ub[both] - pU[both])
    oL = IL & ~IU; oU = IU & ~IL
    v[oL] = np.maximum(v[oL], lb[oL] + o.bound_push * np.maximum(1, np.abs(lb[oL])))
    v[oU] = np.minimum(v[oU], ub[oU] - o.bound_push * np.maximum(1, np.abs(ub[oU])))
    zL = np.where(IL, 1.0, 0.0); zU = np.where(IU, 1.0, 0.0)
    y = np.zeros(m)
    f, g, c, J, H = nlp.eval_all(v, y)
    # least-squares multipliers
    Kls = np.block([[np.eye(n), J.T], [J, np.zeros((m, m))]])
    for i in np.where(~free)[0]:
        Kls[i, :] = 0; Kls[:, i] = 0; Kls[i, i] = 1
    try:
        sol = np.linalg.solve(Kls, -np.concatenate([(g - mult * zL + mult * zU) * free, np.zeros(m)]))
        y = sol[n:]
        if np.max(np.abs(y)) > 1e3: y[:] = 0
    except np.linalg.LinAlgError:
        y[:] = 0
    mu = o.mu_init
    tau = max(o.tau_min, 1 - mu)
    filt = []
    dw_last = 0.0
    theta0 = np.abs(c).sum()
    th_min = 1e-4 * max(1, theta0); th_max = 1e4 * max(1, theta0)
    nb = IL.sum() + IU.sum()

    def barrier(vv, fval):
        return fval - mu * (mult[IL] * np.log(vv[IL] - lb[IL])).sum() - mu * (mult[IU] * np.log(ub[IU] - vv[IU])).sum()

    def err(mu_):
        f_, g_, c_, J_, H_ = nlp.eval_all(v, y)
        sd = max(o.s_max, (np.abs(y).sum() + np.abs(zL).sum() + np.abs(zU).sum()) / (m + nb)) / o.s_max
        sc = max(o.s_max, (np.abs(zL).sum() + np.abs(zU).sum()) / nb) / o.s_max
        rd = (g_ + J_.T @ y - mult * zL + mult * zU) * free
        cl = np.where(IL, (v - lb) * zL - mu_, 0.0); cu = np.where(IU, (ub - v) * zU - mu_, 0.0)
        dinf, pinf = np.abs(rd).max(), np.abs(c_).max()
        cinf = max(np.abs(cl[IL]).max(initial=0), np.abs(cu[IU]).max(initial=0))
        return max(dinf / sd, pinf, cinf / sc), (dinf, pinf, cinf), (f_, g_, c_, J_, H_)

    status = "UserLimit"
    it = 0
    stats = dict(iters=0, reg=0, restor=0)
    while it < o.max_iter:
        E0, (dinf, pinf, cinf0), ev = err(0.0)
        f, g, c, J, H = ev
        if o.verbose:
            print(f"it {it:3d} f={f: .6e} pinf={pinf:.2e} dinf={dinf:.2e} mu={mu:.1e} dw={dw_last:.1e}")
        if E0 <= o.tol and pinf <= o.constr_viol_tol and dinf <= o.dual_inf_tol and cinf0 <= o.compl_inf_tol:
            status = "Optimal"; break
        while True:
            Emu, _, _ = err(mu)
            if Emu <= o.kappa_eps * mu and mu > o.tol / 10:
                mu = max(o.tol / 10, min(o.kappa_mu * mu, mu ** o.theta_mu))
                tau = max(o.tau_min, 1 - mu)
                filt = []
            else:
                break
        dL = np.where(IL, v - lb, 1.0); dU = np.where(IU, ub - v, 1.0)
        Sig = mult * (np.where(IL, zL / dL, 0) + np.where(IU, zU / dU, 0))
        gphi = (g - mu * mult * np.where(IL, 1 / dL, 0) + mu * mult * np.where(IU, 1 / dU, 0))
        r1 = (gphi + J.T @ y) * free
        # inertia-corrected solve
        dw = 0.0; dc = 0.0
        ok = False
        trial = 0
        while True:
            K = np.zeros((n + m, n + m))
            K[:n, :n] = H + np.diag(Sig + dw)
            K[:n, n:] = J.T; K[n:, :n] = J
            K[n:, n:] = -dc * np.eye(m)
            for i in np.where(~free)[0]:
                K[i, :] = 0; K[:, i] = 0; K[i, i] = 1
            pos, neg, zero = inertia_ldl(K)
            if pos == n and neg == m and zero == 0:
                ok = True; break
            stats["reg"] += 1
            if zero > 0 or True:
                dc = o.dc_bar * mu ** o.kappa_c
            if dw == 0:
                dw = o.dw0 if dw_last == 0 else max(o.dw_min, o.kw_dec * dw_last)
            else:
                dw *= o.kw_inc0 if dw_last == 0 else o.kw_inc
            if dw > o.dw_max: break
        if not ok:
            status = "Error"; break
        if dw > 0: dw_last = dw
        sol = np.linalg.solve(K, -np.concatenate([r1, c]))
        dv, dy = sol[:n], sol[n:]
        dzL = np.where(IL, mu / dL - zL - zL / dL * dv, 0.0)
        dzU = np.where(IU, mu / dU - zU + zU / dU * dv, 0.0)
        # fraction to boundary
        amax = 1.0
        neg_ = IL & (dv < 0)
        if neg_.any(): amax = min(amax, (-tau * dL[neg_] / dv[neg_]).min())
        pos_ = IU & (dv > 0)
        if pos_.any(): amax = min(amax, (tau * dU[pos_] / dv[pos_]).min())
        az = 1.0
        q = IL & (dzL < 0)
        if q.any(): az = min(az, (-tau * zL[q] / dzL[q]).min())
        q = IU & (dzU < 0)
        if q.any(): az = min(az, (-tau * zU[q] / dzU[q]).min())
        theta = np.abs(c).sum(); phi = barrier(v, f)
        gd = gphi @ (dv * free)
        if gd < 0:
            amin = min(o.gamma_theta, o.gamma_phi * theta / (-gd))
            if theta <= th_min:
                amin = min(amin, o.delta * theta ** o.s_theta / (-gd) ** o.s_phi)
        else:
            amin = o.gamma_theta
        amin *= o.gamma_alpha
        alpha = amax
        accepted = False
        while alpha >= amin:
            vt = v + alpha * dv
            ft, ct = nlp.fc(vt)
            tht = np.abs(ct).sum()
            if np.isfinite(ft) and np.isfinite(tht) and tht < th_max:
                pht = barrier(vt, ft)
                okf = all((tht < tf) or (pht < pf) for tf, pf in filt)
                if okf:
                    sw = gd < 0 and alpha * (-gd) ** o.s_phi > o.delta * theta ** o.s_theta
                    if theta <= th_min and sw:
                        if pht <= phi + o.eta_phi * alpha * gd:
                            accepted = True; break
                    else:
                        if tht <= (1 - o.gamma_theta) * theta or pht <= phi - o.gamma_phi * theta:
                            accepted = True
                            if not (sw and pht <= phi + o.eta_phi * alpha * gd):
                                filt.append(((1 - o.gamma_theta) * theta, phi - o.gamma_phi * theta))
                            break
            alpha *= 0.5
        if not accepted:
            stats["restor"] += 1
            status = "RestorationNeeded"; break
        v = v + alpha * dv
        ay = min(alpha, az)          # alpha_for_y = min
        y = y + ay * dy
        zL = zL + az * dzL; zU = zU + az * dzU
        dL = np.where(IL, v - lb, 1.0); dU = np.where(IU, ub - v, 1.0)
        zL = np.where(IL, np.clip(zL, mu / (o.kappa_sigma * dL), o.kappa_sigma * mu / dL), 0)
        zU = np.where(IU, np.clip(zU, mu / (o.kappa_sigma * dU), o.kappa_sigma * mu / dU), 0)
        it += 1
    stats["iters"] = it
    return v, y, status, stats
