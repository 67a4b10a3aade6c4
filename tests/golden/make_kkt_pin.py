"""
Generates tests/golden/kkt_pin.npz: an INDEPENDENT optimality certificate of the oracle's solutions at the benchmark size (N = 80 parking, N = 60 quadcopter).

The reference's arithmetic (JuMP + IPOPT) cannot run here and ships no golden vectors, and a third-party NLP solver on the full N = 80
problem (2 424 variables, 1 376 equality rows; scipy SLSQP is dense O(n^3) per iteration) does not finish in an hour (tests/golden/make_pin_full.py is
that attempt).  What CAN be certified independently of the oracle's structured algebra: that the primal-dual point it returns satisfies the optimality
conditions of the NLP as restated flat in oracle/nlp_ref.py / nlp_ref_quad.py (torch autograd derivatives: nothing shared with the oracle's closed-form
derivatives, condensation or Riccati recursion but the problem statement):

  first order   r = grad f + J'y - zL + zU with the ORACLE's multipliers (y of the dynamics / terminal / steering / obstacle rows, zL, zU of the bounds) and the
                AUTOGRAD gradient and Jacobian; |r|_inf, the constraint violation |c|_inf and the complementarity max z_i * dist_i must meet IPOPT's termination
                thresholds (dual_inf_tol-scaled tol = 1e-5 .. , constr_viol_tol = 1e-4, compl_inf_tol = 1e-4); all multipliers of bounds >= 0;
  second order  the AUTOGRAD Hessian of the Lagrangian reduced to the null space of the Jacobian of the equality rows and the active bounds (z_i > dist_i)
                is positive definite  =>  a strict local minimiser (what IPOPT's inertia test certifies at every iteration).

Stored per instance: the primal point (so that the tests compare the oracle AND the HIP path against a fixed array), objective, |c|, |r|, complementarity,
smallest / largest eigenvalue of the reduced Hessian, active-set size.
Run from the repo root (about 3 min on 8 cores):  python tests/golden/make_kkt_pin.py
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.dirname(os.path.abspath(__file__))


def certificate(nlp, v, y, zL, zU):
    """v, y: primal point and equality multipliers in the ordering of `nlp`; zL, zU: bound multipliers per variable (0 where there is no bound),
    already multiplied by the bound's multiplicity (timeScale stands for N+1 copies)"""
    import torch
    g = torch.autograd.functional.jacobian(nlp.f, torch.tensor(v)).numpy()
    cval = nlp.c(torch.tensor(v)).numpy()
    J = torch.autograd.functional.jacobian(nlp.c, torch.tensor(v), vectorize=True).numpy()
    lb, ub = nlp.lb, nlp.ub; n = len(v); mult = getattr(nlp, "mult", np.ones(n))
    r = g + J.T @ y - zL + zU
    hasL = np.isfinite(lb) & (ub > lb); hasU = np.isfinite(ub) & (ub > lb)
    dL = np.where(hasL, v - lb, np.inf); dU = np.where(hasU, ub - v, np.inf)
    compl = max(np.where(hasL, zL * dL / mult, 0).max(), np.where(hasU, zU * dU / mult, 0).max())
    act = np.flatnonzero((hasL & (zL / mult > dL)) | (hasU & (zU / mult > dU)))
    yt = torch.tensor(y)
    H = torch.autograd.functional.hessian(lambda w: nlp.f(w) + (yt * nlp.c(w)).sum(), torch.tensor(v), vectorize=True).numpy()
    Ea = np.zeros((len(act), n)); Ea[np.arange(len(act)), act] = 1.0
    Jall = np.vstack([J, Ea])
    U, s, Vt = np.linalg.svd(Jall, full_matrices=True)
    rank = int((s > 1e-9 * s[0]).sum())
    Z = Vt[rank:].T
    ev = np.linalg.eigvalsh(Z.T @ H @ Z)
    return dict(obj=float(nlp.f(torch.tensor(v)).item()), cviol=float(np.abs(cval).max()), kkt=float(np.abs(r).max()), y_inf=float(np.abs(y).max()),
                z_min=float(min(zL[hasL].min(), zU[hasU].min() if hasU.any() else np.inf)), compl_max=float(compl), bound_viol=float(max((-dL[hasL]).max(), (-dU[hasU]).max() if hasU.any() else -1)),
                n=n, m=int(J.shape[0]), n_active=int(len(act)), rank=rank, redhess_min=float(ev.min()), redhess_max=float(ev.max()))


def parking_job(args):
    import torch
    torch.set_num_threads(1)
    from nlp_ref import ParkingNLP
    import oracle as O
    tag, i, x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS, uWS = args
    t0 = time.time()
    r = O.parking_signed_dist(x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, uWS, full=True)
    assert r["exitflag"] == 1
    z = r["zfull"]; Lz = O.layout(N, v); nOb = len(np.ravel(v)); M = int(np.sum(v)); N1 = N + 1
    nlp = ParkingNLP(x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2])
    vs = np.zeros(nlp.n); zL = np.zeros(nlp.n); zU = np.zeros(nlp.n)
    seg = lambda k, cnt: z[Lz[k]:Lz[k] + cnt]
    vs[nlp.ix] = seg("x", 4 * N1)[4:]; vs[nlp.it] = z[Lz["t"]]; vs[nlp.iu] = seg("u", 2 * N); vs[nlp.il] = seg("lam", M * N1); vs[nlp.im] = seg("mu", 4 * nOb * N1)
    vs[nlp.isl] = seg("sl", nOb * N1); vs[nlp.iss] = seg("ss", N); vs[nlp.iso] = seg("so", nOb * N1)
    zL[nlp.ix] = seg("zxL", 4 * N1)[4:]; zU[nlp.ix] = seg("zxU", 4 * N1)[4:]
    zL[nlp.ix][2::4] = 0; zU[nlp.ix][2::4] = 0                                   # psi is unbounded
    zxl = zL[nlp.ix].copy(); zxl[2::4] = 0; zL[nlp.ix] = zxl; zxu = zU[nlp.ix].copy(); zxu[2::4] = 0; zU[nlp.ix] = zxu
    zL[nlp.it] = (N + 1) * z[Lz["ztL"]]; zU[nlp.it] = (N + 1) * z[Lz["ztU"]]        # N+1 copies of the timeScale bounds
    zL[nlp.iu] = seg("zuL", 2 * N); zU[nlp.iu] = seg("zuU", 2 * N); zL[nlp.il] = seg("zlam", M * N1); zL[nlp.im] = seg("zmu", 4 * nOb * N1)
    zL[nlp.iss] = seg("zssL", N); zU[nlp.iss] = seg("zssU", N); zL[nlp.iso] = seg("zso", nOb * N1)
    y = np.concatenate([seg("pi", 4 * N), seg("nu", 4), seg("yg", N), seg("yo", 4 * nOb * N1)])
    c = certificate(nlp, vs, y, zL, zU)
    print(tag, i, {k: (float("%.3g" % v_) if isinstance(v_, float) else v_) for k, v_ in c.items()}, "oracle obj", r["obj"], "iters", r["iters"], "mu", r["mu"], "%.0fs" % (time.time() - t0), flush=True)
    return dict(tag=tag, idx=i, x0=x0, xF=xF, Ts=Ts, xWS=xWS, uWS=uWS, xp=r["xp"], up=r["up"], t=r["t"], lp=r["lp"], np_=r["np"], sl=r["sl"], oracle_obj=r["obj"], iters=r["iters"], **c)


def quad_job(args):
    import torch
    torch.set_num_threads(1)
    from nlp_ref_quad import QuadNLP
    import oracle_quad as Q
    i, x0, xF, N, Ts, R, ob, xWS = args
    t0 = time.time()
    r = Q.quadcopter_signed_dist_full(x0, xF, N, Ts, R, ob, xWS, 1.0)
    assert r["exitflag"] == 1
    nlp = QuadNLP(x0, xF, N, Ts, R, ob)
    # the oracle's v is x (12 (N+1), x_0 first) | u | t | lam | s | so; nlp_ref_quad drops x_0
    keep = np.arange(12, len(r["v"]))
    vs = r["v"][keep]; zL = r["zL"][keep].copy(); zU = r["zU"][keep].copy()
    zL[nlp.it] *= (N + 1); zU[nlp.it] *= (N + 1)                         # N+1 copies of the timeScale bounds
    zL[~np.isfinite(nlp.lb)] = 0; zU[~np.isfinite(nlp.ub)] = 0
    c = certificate(nlp, vs, r["y"], zL, zU)
    print("quad", i, {k: (float("%.3g" % v_) if isinstance(v_, float) else v_) for k, v_ in c.items()}, "oracle obj", r["obj"], "iters", r["iters"], "%.0fs" % (time.time() - t0), flush=True)
    return dict(tag="quad", idx=i, x0=x0, xF=xF, Ts=Ts, xWS=xWS, xp=r["xp"], up=r["up"], t=r["t"], lp=r["lp"], slack=r["slack"], oracle_obj=r["obj"], iters=r["iters"], R=R, ob=ob, **c)


def main():
    import multiprocessing as mp
    from obca_amd import scenarios as S
    which = [w for w in sys.argv[1:] if not w.startswith("--")] or ["cfg2", "cfg3", "quad"]
    jobs = []
    if "cfg2" in which:
        N = 80; bt = S.make_batch(S.BACKWARDS, 8, N)
        for i in range(8 if "--one" not in sys.argv else 1):
            xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
            jobs.append((parking_job, ("cfg2", i, bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS, bt["uWS"][i])))
    if "cfg3" in which:
        g = np.load(os.path.join(OUT, "oracle_cfg3.npz")); N = int(g["N"]); A, b, v = S.scenario_hrep(S.PARALLEL)
        for i in range(4):
            jobs.append((parking_job, ("cfg3", i, g["x0"][i], g["xF"][i], N, float(g["Ts"][i]), S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, g["xWS"][i], g["uWS"][i])))
    if "quad" in which:
        N = 60; q = S.make_quad_batch(2, N)
        for i in range(2):
            jobs.append((quad_job, (i, q["x0"][i], q["xF"][i], N, q["Ts"], q["R"], q["ob"], q["xWS"][i])))
    with mp.get_context("fork").Pool(min(6, len(jobs))) as pool:
        res = [pool.apply_async(fn, (a,)) for fn, a in jobs]
        res = [r.get() for r in res]
    name = "kkt_pin.npz" if len(which) == 3 and "--one" not in sys.argv else "kkt_pin_" + "_".join(which) + ".npz"
    np.savez(os.path.join(OUT, name), records=np.array(res, dtype=object))
    print("wrote", name, len(res), "records")


if __name__ == "__main__":
    main()
