"""public validate() API (obca_amd/validate.py) on oracle solutions: the reference's acceptance tests and the full checkers"""
import numpy as np
import pytest
from conftest import golden
from obca_amd import scenarios as S, validate as V


def test_validate_parking_accepts_golden_solutions_and_rejects_perturbed_ones():
    g = golden("oracle_cfg3.npz"); N = int(g["N"]); A, b, v = S.scenario_hrep(S.PARALLEL)
    for i in range(int(g["B"])):
        ts = np.full(N + 1, g["t"][i])
        ok, viol = V.validate_parking(g["x0"][i], g["xF"][i], N, g["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, g["xp"][i], g["up"][i],
                                      ts, g["lp"][i], g["np"][i], g["sl"][i])
        assert ok, viol
        xp = g["xp"][i].copy(); xp[0, N // 2] += 0.01          # 1 cm off the dynamics
        ok, viol = V.validate_parking(g["x0"][i], g["xF"][i], N, g["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xp, g["up"][i],
                                      ts, g["lp"][i], g["np"][i], g["sl"][i])
        assert not ok and viol["dyn"] > 1e-3


def test_validate_quadcopter_on_oracle_solution():
    import oracle_quad as Q
    N = 30; Ts = S.quad_sample_time(N); xWS = S.quad_warm_start(S.QUAD_X0, S.QUAD_XF, N)
    for dist in (0, 1):
        r = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, S.QUAD_OB, xWS, 1.0, dist=dist)
        assert r["exitflag"] == 1
        ok, w = V.validate_quadcopter(r["xp"], r["up"], r["timeScale"], Q.X0, Q.XF, Ts, r["lp"], S.QUAD_OB, Q.EGO_R)
        assert ok, w
        up = r["up"].copy(); up[0, 3] = 8.0
        assert not V.validate_quadcopter(r["xp"], up, r["timeScale"], Q.X0, Q.XF, Ts, r["lp"], S.QUAD_OB, Q.EGO_R)[0]
