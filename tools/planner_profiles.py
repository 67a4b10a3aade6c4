"""Warm starts from this repository's search settings against the cost constants of the reference's Hybrid A* (hybrid_a_star.jl:60-63; planner.REFERENCE_COSTS): what
the difference does to the warm start and to the NLP solved from it (CPU oracle).  Quantifies the 'next-2' row of SURVEY 8f: the search is a re-design, and warm
starts differ from the reference's -- by how much, and does the optimum move?"""
import sys, os, time, json
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/oracle"]
import oracle as O
from obca_amd import scenarios as S, planner as PL
rng = np.random.default_rng(5); N = 80; out = {}
for sc, n in ((S.BACKWARDS, 32), (S.PARALLEL, 16)):
    rows = []
    for i in range(n):
        x0 = np.array([rng.uniform(-10, 10), rng.uniform(6.5, 9.5), rng.uniform(-0.2, 0.2), 0.0]); xF = sc["xF"]; res = {}
        for name, kw in (("ours", {}), ("ref", PL.REFERENCE_COSTS)):
            t = time.time(); w = PL.warm_start(sc, x0, xF, N, **kw); dt = time.time() - t
            if w is None: res[name] = None; continue
            Ts, xWS, uWS = w; xWS = xWS.copy(); xWS[0] = x0; A, b, v = S.scenario_hrep(sc)
            r = O.parking_signed_dist(x0, xF, N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, uWS)
            res[name] = (r["exitflag"], r["iters"], r["obj"], np.hypot(np.diff(xWS[:, 0]), np.diff(xWS[:, 1])).sum(), int((np.diff(np.sign(xWS[1:-1, 3])) != 0).sum()), dt, xWS, r["xp"])
        if res["ours"] and res["ref"]:
            a, b_ = res["ours"], res["ref"]
            rows.append((a[0], b_[0], a[1], b_[1], a[2], b_[2], a[3], b_[3], a[4], b_[4], np.abs(a[6][:, :2] - b_[6][:, :2]).max(), np.abs(a[7][:2] - b_[7][:2]).max(), a[5], b_[5]))
    T = np.array(rows); k = "ours_vs_reference_costs"
    out[sc["name"]] = dict(instances_planned_by_both=len(T), of=n, converged=[float(T[:, 0].mean()), float(T[:, 1].mean())], mean_iterations=[float(T[:, 2].mean()), float(T[:, 3].mean())],
                           mean_nlp_objective=[float(T[:, 4].mean()), float(T[:, 5].mean())], same_optimum_within_1e_4=float((np.abs(T[:, 4] - T[:, 5]) < 1e-4 * np.abs(T[:, 4])).mean()),
                           mean_path_length_m=[float(T[:, 6].mean()), float(T[:, 7].mean())], mean_direction_switches=[float(T[:, 8].mean()), float(T[:, 9].mean())],
                           mean_max_xy_gap_between_warm_starts_m=float(T[:, 10].mean()), mean_max_xy_gap_between_solutions_m=float(T[:, 11].mean()), mean_plan_seconds=[float(T[:, 12].mean()), float(T[:, 13].mean())])
print(json.dumps(out, indent=1))
