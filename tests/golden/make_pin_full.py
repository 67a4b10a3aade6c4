"""
Generates tests/golden/pin_full.npz: an INDEPENDENT pin of the oracle at the benchmark size.

For every instance -- 8 of BASELINE config 2 (backwards parking, N=80, the instances of oracle_cfg2.npz), 4 of config 3 (parallel parking, N=80, Hybrid A*
warm starts of oracle_cfg3.npz) and 2 of config 4 (quadcopter, N=60) -- the FULL NLP as restated in oracle/nlp_ref.py / nlp_ref_quad.py (flat, unstructured,
torch autograd derivatives: nothing shared with the structured oracle but the problem statement) is solved by a THIRD-PARTY solver, scipy.optimize SLSQP
(an active-set SQP method, unrelated to the oracle's interior point), started from the oracle's solution perturbed by 1e-3 relative noise, so that the
SQP iteration has to re-converge on its own.  Stored per instance: the third-party optimum (objective, trajectory, time scale), its constraint violation,
an independent first-order optimality residual of BOTH points (least-squares multipliers of the autograd Jacobian with sign constraints on the active
bounds), and the active sets of both points.

Run from the repo root:  python tests/golden/make_pin_full.py            (N = 80: one instance did NOT finish within 50 minutes on this machine -- SLSQP's dense
                                                                         QP is O(n^3) per iteration at n = 2 424; kept as the record of the attempt)
                          PIN_N=24 python tests/golden/make_pin_full.py cfg2   (n = 748: the eight config-2 starts at a horizon SLSQP can handle -> pin_cfg2_N24.npz)
The reference itself (Julia 0.6 + JuMP + IPOPT) cannot run here and ships no golden vectors: this pins the oracle's OPTIMUM at N=80 against an
independent solver; it does not pin IPOPT's iterates.
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.dirname(os.path.abspath(__file__))


def kkt_residual(g, J, v, lb, ub, tol_act=1e-6):
    """min over (y free, zL >= 0 on active lower bounds, zU >= 0 on active upper bounds) of |g + J'y - zL + zU|_inf (least squares, then the max norm)"""
    from scipy.optimize import lsq_linear
    aL = np.flatnonzero(np.isfinite(lb) & (v - lb <= tol_act)); aU = np.flatnonzero(np.isfinite(ub) & (ub - v <= tol_act))
    m = J.shape[0]; n = len(v)
    EL = np.zeros((n, len(aL))); EL[aL, np.arange(len(aL))] = -1.0
    EU = np.zeros((n, len(aU))); EU[aU, np.arange(len(aU))] = 1.0
    Amat = np.hstack([J.T, EL, EU])
    lo = np.concatenate([-np.inf * np.ones(m), np.zeros(len(aL) + len(aU))]); hi = np.inf * np.ones(Amat.shape[1])
    r = lsq_linear(Amat, -g, bounds=(lo, hi), method="bvls" if Amat.shape[1] < 4000 else "trf", tol=1e-13, max_iter=400)
    return float(np.abs(Amat @ r.x + g).max()), aL, aU


def solve_tp(nlp, vstar, seed, maxiter=300):
    import torch
    from scipy.optimize import minimize, Bounds
    f = lambda w: nlp.f(torch.tensor(w)).item()
    g = lambda w: torch.autograd.functional.jacobian(nlp.f, torch.tensor(w)).numpy()
    c = lambda w: nlp.c(torch.tensor(w)).numpy()
    J = lambda w: torch.autograd.functional.jacobian(nlp.c, torch.tensor(w), vectorize=True).numpy()
    lb = np.where(np.isfinite(nlp.lb), nlp.lb, -np.inf); ub = np.where(np.isfinite(nlp.ub), nlp.ub, np.inf)
    rng = np.random.default_rng(seed)
    v0 = vstar + 1e-3 * rng.standard_normal(len(vstar)) * np.maximum(1.0, np.abs(vstar))
    v0 = np.minimum(np.maximum(v0, lb), ub)
    t0 = time.time()
    res = minimize(f, v0, jac=g, method="SLSQP", bounds=Bounds(lb, ub), constraints=[dict(type="eq", fun=c, jac=J)], options=dict(maxiter=maxiter, ftol=1e-13))
    dt = time.time() - t0
    vt = res.x
    out = dict(status=res.status, nit=res.nit, obj=res.fun, cviol=float(np.abs(c(vt)).max()), seconds=dt, start_dist=float(np.abs(v0 - vstar).max()))
    for name, w in (("tp", vt), ("or", vstar)):
        kr, aL, aU = kkt_residual(g(w), J(w), w, lb, ub)
        out["kkt_" + name] = kr; out["aL_" + name] = aL; out["aU_" + name] = aU
    out["obj_or"] = f(vstar); out["cviol_or"] = float(np.abs(c(vstar)).max())
    return vt, out


def parking_job(args):
    import torch
    torch.set_num_threads(1)
    from nlp_ref import ParkingNLP
    import oracle as O
    tag, i, x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS, uWS = args
    r = O.parking_signed_dist(x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, uWS)
    assert r["exitflag"] == 1
    nlp = ParkingNLP(x0, xF, N, Ts, L, ego, XYb, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2])
    vs = np.zeros(nlp.n)
    vs[nlp.ix] = r["xp"].T[1:].reshape(-1); vs[nlp.it] = r["t"]; vs[nlp.iu] = r["up"].T.reshape(-1)
    vs[nlp.il] = r["lp"].T.reshape(-1); vs[nlp.im] = r["np"].T.reshape(-1); vs[nlp.isl] = r["sl"].T.reshape(-1)
    cv = nlp.c(torch.tensor(vs)).numpy(); nOb = nlp.nOb; o = 4 * N + 4
    vs[nlp.iss] = cv[o:o + N]; vs[nlp.iso] = np.maximum(cv[o + N:].reshape(N + 1, nOb, 4)[:, :, 3].reshape(-1), 0.0)      # slacks = row values
    vt, st = solve_tp(nlp, vs, 1000 + i)
    x, t, u, lam, mu, sl, ss, so = nlp.unpack(torch.tensor(vt))
    print(tag, i, "SLSQP", st["status"], st["nit"], "obj", st["obj"], "oracle", r["obj"], "cviol", st["cviol"], "kkt tp/or", st["kkt_tp"], st["kkt_or"],
          "dx", np.abs(x.numpy().T - r["xp"]).max(), "%.0fs" % st["seconds"], flush=True)
    return dict(tag=tag, idx=i, x0=x0, xF=xF, Ts=Ts, xWS=xWS, uWS=uWS, xp=x.numpy().T, up=u.numpy().T, t=float(t), sl=sl.numpy().T, oracle_obj=r["obj"], oracle_iters=r["iters"],
                oracle_xp=r["xp"], oracle_up=r["up"], oracle_t=r["t"], **{k: v_ for k, v_ in st.items() if not k.startswith("a")},
                active_mismatch=len(set(st["aL_tp"]) ^ set(st["aL_or"])) + len(set(st["aU_tp"]) ^ set(st["aU_or"])), n_active=len(st["aL_or"]) + len(st["aU_or"]))


def quad_job(args):
    import torch
    torch.set_num_threads(1)
    from nlp_ref_quad import QuadNLP
    import oracle_quad as Q
    i, x0, xF, N, Ts, R, ob, xWS = args
    r = Q.quadcopter_signed_dist(x0, xF, N, Ts, R, ob, xWS, 1.0)
    assert r["exitflag"] == 1
    nlp = QuadNLP(x0, xF, N, Ts, R, ob)
    vs = np.concatenate([r["xp"].T[1:].reshape(-1), r["up"].T.reshape(-1), [r["t"]], r["lp"].T.reshape(-1), r["slack"].T.reshape(-1), np.zeros(5 * (N + 1))])
    cv = nlp.c(torch.tensor(vs)).numpy()
    vs[nlp.iso] = np.maximum(cv[12 * N + 12:].reshape(N + 1, 5, 2)[:, :, 1].reshape(-1), 0.0)
    vt, st = solve_tp(nlp, vs, 2000 + i)
    x, u, t, lam, s, so = nlp.unpack(torch.tensor(vt))
    print("quad", i, "SLSQP", st["status"], st["nit"], "obj", st["obj"], "oracle", r["obj"], "cviol", st["cviol"], "kkt tp/or", st["kkt_tp"], st["kkt_or"],
          "du", np.abs(u.numpy().T - r["up"]).max(), "%.0fs" % st["seconds"], flush=True)
    return dict(tag="quad", idx=i, x0=x0, xF=xF, Ts=Ts, xWS=xWS, xp=x.numpy().T, up=u.numpy().T, t=float(t), oracle_obj=r["obj"], oracle_iters=r["iters"], oracle_xp=r["xp"],
                oracle_up=r["up"], oracle_t=r["t"], **{k: v_ for k, v_ in st.items() if not k.startswith("a")},
                active_mismatch=len(set(st["aL_tp"]) ^ set(st["aL_or"])) + len(set(st["aU_tp"]) ^ set(st["aU_or"])), n_active=len(st["aL_or"]) + len(st["aU_or"]))


def main():
    import multiprocessing as mp
    from obca_amd import scenarios as S
    which = sys.argv[1:] or ["cfg2", "cfg3", "quad"]
    jobs = []
    if "cfg2" in which:
        N = int(os.environ.get("PIN_N", "80")); bt = S.make_batch(S.BACKWARDS, 8, N)
        for i in range(8 if "--one" not in which else 1):
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
    with mp.get_context("fork").Pool(min(8, len(jobs))) as pool:
        res = [pool.apply_async(fn, (a,)) for fn, a in jobs]
        res = [r.get() for r in res]
    name = "pin_full.npz" if not sys.argv[1:] else "pin_" + "_".join(w for w in which if not w.startswith("--")) + ("_N" + os.environ["PIN_N"] if "PIN_N" in os.environ else "") + ".npz"
    np.savez(os.path.join(OUT, name), records=np.array(res, dtype=object))
    print("wrote", name, len(res), "records")


if __name__ == "__main__":
    main()
