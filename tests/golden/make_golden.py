"""
Generates the fixtures under tests/golden/ (run from the repo root:  python tests/golden/make_golden.py [--dense]).

  dense_N8.npz      : the dense textbook IPM (oracle/ipm_dense.py, autograd derivatives, dense LDL^T) on the backwards
                      scenario at N=8 -- the independent cross-check of the structured oracle (takes ~1 min, hence a fixture)
  oracle_cfg2.npz   : structured C oracle on the first 8 instances of config 2 (backwards parking, N=80, seed 20260925)
  oracle_cfg3.npz   : structured C oracle on 6 instances of config 3 (parallel parking, 4 obstacles, N=80) with Hybrid A* warm starts
                      (obca_amd/planner.py); the fixture stores the warm starts too, so the test does not depend on the planner
  slsqp_N8.npz      : the same N=8 NLP solved by a THIRD-PARTY solver (scipy.optimize SLSQP, an SQP method unrelated to the oracle's
                      interior point) from the same warm start, autograd derivatives of oracle/nlp_ref.py (--scipy, ~40 s)
  oracle_quad_cfg4.npz : quadcopter oracle (oracle/obca_oracle_quad.c) on config 4: the shipped scenario and three jittered start / goal pairs
                      of scenarios.make_quad_batch at N=60 (--quad)
  dualws_known.npz  : poses + closed-form rectangle/half-plane distances for the DualMultWS known-answer test

The reference itself (Julia 0.6 + JuMP + IPOPT) cannot run in this environment and ships no golden vectors
(SURVEY.md section 8c): these fixtures pin the oracle against independent computations, not against IPOPT.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from obca_amd import scenarios as S   # noqa: E402
import oracle as O                    # noqa: E402
OUT = os.path.dirname(os.path.abspath(__file__))


def dense_case(N):
    import torch
    from nlp_ref import ParkingNLP
    import ipm_dense
    sc = S.BACKWARDS; A, b, v = S.scenario_hrep(sc)
    x0 = sc["x0"].copy()
    Ts, xWS, uWS = S.warm_start_backwards(x0, sc["xF"], N); xWS[0] = x0
    lWS, nWS, _ = O.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], S.EGO)
    nlp = ParkingNLP(x0, sc["xF"], N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2])
    v0 = nlp.pack_start(xWS, uWS, lWS, nWS)
    vs, y, st, stats = ipm_dense.solve(nlp, v0)
    x, t, u, lam, mu, sl, ss, so = nlp.unpack(torch.tensor(vs))
    np.savez(os.path.join(OUT, f"dense_N{N}.npz"), N=N, x0=x0, xF=sc["xF"], Ts=Ts, xWS=xWS, uWS=uWS, lWS=lWS, nWS=nWS,
             xp=x.numpy().T, up=u.numpy().T, t=float(t), lp=lam.numpy().T, np_=mu.numpy().reshape(N + 1, -1).T, sl=sl.numpy().T,
             obj=nlp.f(torch.tensor(vs)).item(), status=st, iters=stats["iters"])
    print("dense", N, st, stats)


def oracle_cases(sc, B, name, goal_jitter=False):
    N = 80
    bt = S.make_batch(sc, B, N, goal_jitter=goal_jitter)
    keys = ("xp", "up", "lp", "np", "sl")
    res = {k: [] for k in keys}; meta = dict(exitflag=[], iters=[], obj=[], t=[])
    for i in range(B):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        r = O.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"],
                                  bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        for k in keys: res[k].append(r[k])
        for k in meta: meta[k].append(r[k])
    np.savez(os.path.join(OUT, name), B=B, N=N, **{k: np.array(v) for k, v in res.items()}, **{k: np.array(v) for k, v in meta.items()})
    print(name, meta["exitflag"], meta["iters"])


def oracle_cfg3(B=6, N=80):
    bt = S.make_batch(S.PARALLEL, B, N)
    keys = ("xp", "up", "lp", "np", "sl")
    res = {k: [] for k in keys}; meta = dict(exitflag=[], iters=[], obj=[], t=[])
    for i in range(B):
        xWS = bt["xWS"][i]
        r = O.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"],
                                  bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        for k in keys: res[k].append(r[k])
        for k in meta: meta[k].append(r[k])
    np.savez(os.path.join(OUT, "oracle_cfg3.npz"), B=B, N=N, x0=bt["x0"], xF=bt["xF"], Ts=bt["Ts"], xWS=bt["xWS"], uWS=bt["uWS"],
             **{k: np.array(v) for k, v in res.items()}, **{k: np.array(v) for k, v in meta.items()})
    print("oracle_cfg3.npz", meta["exitflag"], meta["iters"])


def slsqp_case(N=8):
    import torch
    from nlp_ref import ParkingNLP
    from scipy.optimize import minimize, Bounds
    sc = S.BACKWARDS; A, b, v = S.scenario_hrep(sc); x0 = sc["x0"].copy()
    Ts, xWS, uWS = S.warm_start_backwards(x0, sc["xF"], N); xWS[0] = x0
    lWS, nWS, _ = O.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], S.EGO)
    nlp = ParkingNLP(x0, sc["xF"], N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2])
    f = lambda w: nlp.f(torch.tensor(w)).item()
    g = lambda w: torch.autograd.functional.jacobian(nlp.f, torch.tensor(w)).numpy()
    c = lambda w: nlp.c(torch.tensor(w)).numpy()
    J = lambda w: torch.autograd.functional.jacobian(nlp.c, torch.tensor(w)).numpy()
    lb = np.where(np.isfinite(nlp.lb), nlp.lb, -np.inf); ub = np.where(np.isfinite(nlp.ub), nlp.ub, np.inf)
    v0 = np.clip(nlp.pack_start(xWS, uWS, lWS, nWS), lb + 1e-3, ub - 1e-3)
    res = minimize(f, v0, jac=g, method="SLSQP", bounds=Bounds(lb, ub), constraints=[dict(type="eq", fun=c, jac=J)],
                   options=dict(maxiter=500, ftol=1e-12))
    x, t, u, lam, mu, sl, ss, so = nlp.unpack(torch.tensor(res.x))
    np.savez(os.path.join(OUT, f"slsqp_N{N}.npz"), N=N, x0=x0, xF=sc["xF"], Ts=Ts, xWS=xWS, uWS=uWS, lWS=lWS, nWS=nWS, status=res.status,
             nit=res.nit, obj=res.fun, cviol=np.abs(c(res.x)).max(), xp=x.numpy().T, up=u.numpy().T, t=float(t))
    print("slsqp", res.status, res.message, res.nit, res.fun)


def dualws_known():
    # single half-plane obstacle a'p <= beta (unit a), rectangle centre c = (X+1.35 cos, Y+1.35 sin), half extents (2.35, 1):
    # d = max(0, -(a'c - beta) ... ) -- the obstacle is {p : a'p <= beta}; the car is outside it when a'c - support > beta
    rng = np.random.default_rng(7)
    n = 64
    poses = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n)], 1)
    ang = rng.uniform(-np.pi, np.pi, n)
    a = np.stack([np.cos(ang), np.sin(ang)], 1); beta = rng.uniform(-3, 3, n)
    hl, hw, off = 2.35, 1.0, 1.35
    d = np.zeros(n)
    for i in range(n):
        X, Y, psi = poses[i]
        c = np.array([X + off * np.cos(psi), Y + off * np.sin(psi)])
        e1 = np.array([np.cos(psi), np.sin(psi)]); e2 = np.array([-np.sin(psi), np.cos(psi)])
        supp = hl * abs(a[i] @ e1) + hw * abs(a[i] @ e2)       # support of the rectangle along -a
        d[i] = max(0.0, a[i] @ c - supp - beta[i])
    np.savez(os.path.join(OUT, "dualws_known.npz"), poses=poses, a=a, beta=beta, d=d)


def oracle_quad_cfg4(B=4, N=60):
    import oracle_quad as Q
    bt = S.make_quad_batch(B, N)
    keys = ("xp", "up", "lp"); res = {k: [] for k in keys}; meta = dict(exitflag=[], iters=[], obj=[], t=[])
    for i in range(B):
        r = Q.quadcopter_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"], bt["R"], bt["ob"], bt["xWS"][i], 1.0)
        for k in keys: res[k].append(r[k])
        for k in meta: meta[k].append(r[k])
    np.savez(os.path.join(OUT, "oracle_quad_cfg4.npz"), B=B, N=N, x0=bt["x0"], xF=bt["xF"], Ts=bt["Ts"], R=bt["R"], ob=bt["ob"], xWS=bt["xWS"],
             **{k: np.array(v) for k, v in res.items()}, **{k: np.array(v) for k, v in meta.items()})
    print("oracle_quad_cfg4.npz", meta["exitflag"], meta["iters"])


if __name__ == "__main__":
    if "--quad" in sys.argv:
        oracle_quad_cfg4(); sys.exit(0)
    dualws_known()
    oracle_cases(S.BACKWARDS, 8, "oracle_cfg2.npz")
    oracle_cfg3()
    if "--dense" in sys.argv:
        dense_case(8)
    if "--scipy" in sys.argv:
        slsqp_case(8)
