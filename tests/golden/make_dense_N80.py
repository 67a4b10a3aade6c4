"""
Generates tests/golden/dense_N80.npz: the reference's UN-reformulated parking NLP at the benchmark size (N = 80) solved by oracle/ipm_ref80.py -- N + 1 time-scale variables with
their chain rows, x[:, 1] == x0 kept, every bound a constraint row with an IPOPT slack (7 690-dimensional KKT system for config 2), rows as obstHrep emits them with gradient-based
scaling, dense Bunch-Kaufman LDL' with counted inertia, IPOPT's Algorithm A with second-order correction, recalc_y and least-squares initial multipliers.  Nothing of it is shared
with the kernels' or the C oracle's structure (condensation, Riccati recursion, single time-scale variable, eliminated start state, unit-length rows, closed-form derivatives).

Instances: the first 8 of the config-2 bench batch (reverse parking, 3 obstacles / 5 rows), 4 of the config-3 batch (parallel parking, 4 obstacles / 6 rows, Hybrid A* warm
starts -- stored, so the tests do not depend on the planner) and 4 of the config-5 distribution whose extra polygons have sloped, unnormalised edges (rows of length up to 15).  The fixture keeps the inputs, the dense solution and its iteration log; tests/test_pin_cpu.py (C oracle) and
tests/test_gpu_parity.py (HIP path) must land on the same point.
Run from the repo root (about 4 minutes per instance on 8 cores):  python tests/golden/make_dense_N80.py
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    import torch
    torch.set_num_threads(int(os.environ.get("DENSE_THREADS", "8")))
    from obca_amd import scenarios as S
    import oracle as O
    import ipm_ref80 as R
    N = 80
    jobs = []
    b2 = S.make_batch(S.BACKWARDS, 8, N, seed=20260925)
    jobs += [("cfg2", S.BACKWARDS, b2, i) for i in range(8)]
    b3 = S.make_batch(S.PARALLEL, 4, N, seed=20260925, goal_jitter=True)
    jobs += [("cfg3", S.PARALLEL, b3, i) for i in range(4)]
    # config-5 instances with SLOPED polygon edges (obstHrep leaves them unnormalised: rows of length 1.3 .. 15.4 here) -- the kernels and the C oracle bring every row to unit
    # length, the dense solve takes the rows as they are and applies IPOPT's gradient-based scaling: the instances that exercise that difference (configs 2 / 3 have unit rows)
    b5 = S.make_mixed_batch(64, N, seed=20260925, min_obstacles=1)
    jobs += [("cfg5", None, b5, i) for i in (8, 16, 18, 35)]
    keys = "tag idx x0 xF Ts xWS uWS xp up ts obj iters reg soc soc_acc recalc attempts exitflag seconds obj_scaling rows_scaled".split()
    rec = {k: [] for k in keys}
    fn = os.path.join(OUT, "dense_N80.npz")
    if os.path.exists(fn):      # resume: keep what is there
        old = np.load(fn, allow_pickle=True)
        if "idx" in old.files:
            rec = {k: list(old[k]) for k in keys}
        else:
            rec = {k: (list(old[k]) if k != "idx" else [j if j < 8 else j - 8 for j in range(len(old["tag"]))]) for k in keys}
    done = set(zip(map(str, rec["tag"]), map(int, rec["idx"])))
    for tag, sc, bt, i in jobs:
        if (tag, i) in done:
            continue
        if sc is None:
            A, b, v = np.asarray(bt["A"][i], float), np.asarray(bt["b"][i], float), np.ravel(bt["vOb"][i]).astype(int)
        else:
            A, b, v = S.scenario_hrep(sc)
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = O.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], S.EGO)
        nlp = R.RefNLP(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2])
        v0 = nlp.start(xWS, bt["uWS"][i], lWS, nWS)
        t0 = time.time(); vs, ef, info = R.solve(nlp, v0); dt = time.time() - t0
        x, ts, u, lam, mu, sl = nlp.split(torch.tensor(vs))
        print(tag, i, "exitflag", ef, info, "%.0f s" % dt, flush=True)
        for k, val in (("tag", tag), ("idx", i), ("x0", bt["x0"][i]), ("xF", bt["xF"][i]), ("Ts", bt["Ts"][i]), ("xWS", xWS), ("uWS", bt["uWS"][i]), ("xp", x.numpy().T.copy()), ("up", u.numpy().T.copy()),
                       ("ts", ts.numpy().copy()), ("obj", info["obj"]), ("iters", info["iters"]), ("reg", info["reg"]), ("soc", info["soc"]), ("soc_acc", info["soc_acc"]),
                       ("recalc", info["recalc"]), ("attempts", info["attempts"]), ("exitflag", ef), ("seconds", dt), ("obj_scaling", info["obj_scaling"]), ("rows_scaled", info["rows_scaled"])):
            rec[k].append(val)
        np.savez(fn, N=N, **{k: np.array(val) for k, val in rec.items()})


if __name__ == "__main__":
    main()
