"""Quadcopter path: the HIP solver source (obca_quad_solver.h) compiled as a host emulation, against the quadcopter oracle."""
import ctypes as C
import numpy as np
import pytest
import packing as P
from test_emu_cpu import EOpts, dp


@pytest.fixture(scope="module")
def Q():
    import oracle_quad
    oracle_quad.lib()
    return oracle_quad


VIA = [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)]


def test_quad_layouts_agree(Q, emu):
    for N in (3, 16, 60):
        a = Q.layout(N); b = P.quad_layout(N)
        out = np.zeros(16, np.int32); cnt = emu.emu_quad_layout(C.c_int(N), out.ctypes.data_as(C.POINTER(C.c_int)))
        assert cnt == 14 and out[:14].tolist() == [b[k] for k in "x u t lam s so n pi nu yo m zL zU len".split()]
        assert all(a[k] == b[k] for k in "x u t lam s so n m".split())
        assert all(a[k] + b["n"] == b[k] for k in ("pi", "nu", "yo"))      # the oracle numbers the multipliers from 0


def test_emu_quad_newton_direction_matches_oracle(Q, emu):
    rng = np.random.default_rng(2)
    N, Ts, R, ob = 7, 0.3, 0.25, Q.OB_CLAMPED
    x0 = Q.X0.copy(); x0[9:12] = [0.1, -0.2, 0.15]       # non-zero stage-1 rates exercise the single-index quirk
    L = Q.layout(N); n, m = L["n"], L["m"]
    xWS = Q.warm_start(x0, Q.XF, N)
    v = np.zeros(n)
    X = xWS.copy(); X[1:] += 0.05 * rng.standard_normal((N, 12)); X[1:, 3:6] = 0.1 * rng.standard_normal((N, 3)); X[0] = x0
    v[L["x"]:L["x"] + 12 * (N + 1)] = X.reshape(-1)
    v[L["u"]:L["u"] + 4 * N] = rng.uniform(3, 6, 4 * N); v[L["t"]] = 1.1
    for k, cnt in (("lam", 30), ("s", 5), ("so", 5)):
        v[L[k]:L[k] + cnt * (N + 1)] = rng.uniform(0.1, 1, cnt * (N + 1))
    y = rng.standard_normal(m); zL = rng.uniform(0.1, 2, n); zU = rng.uniform(0.1, 2, n)
    mu, dw, dc = 0.1, 500.0, 1e-6
    ok, dv, dy, errs = Q.newton(N, Ts, R, x0, Q.XF, ob, v, y, zL, zU, mu, dw, dc)
    z = np.concatenate([v, y, zL, zU]); assert len(z) == P.quad_layout(N)["len"]
    prob = P.pack_quad_problem(x0, Q.XF, N, Ts, R, ob, xWS, 1.0)
    d = np.zeros(n + m); aux = np.zeros(11)
    ok2 = emu.emu_quad_newton(C.c_int(N), dp(prob), dp(z), C.c_double(mu), C.c_double(dw), C.c_double(dc), C.c_double(1e3), C.c_double(0.99), dp(d), dp(aux))
    assert ok == 1 and ok2 == 1
    assert np.abs(d[:n] - dv).max() < 1e-10 * max(1.0, np.abs(dv).max())
    assert np.abs(d[n:] - dy).max() < 1e-10 * max(1.0, np.abs(dy).max())
    assert np.allclose(aux[:3], errs, rtol=1e-10)


@pytest.mark.parametrize("N,dws", [(16, 1), (30, 1), (16, 0), (100, 0)], ids=["N16", "N30", "reference_start", "reference_start_N100_beyond_64"])
def test_emu_quad_full_solve_matches_oracle(Q, emu, N, dws):
    """dws = 0: the reference's own start (lambda = 0.05, block restoration before the first iteration); N = 100: a horizon of the size
    mainQuadcopter.jl's A* path produces (N_as >= 80), beyond the former limit of 64"""
    Ts = round(0.25 * 80 / N * 100) / 100
    xWS = Q.warm_start(Q.X0, Q.XF, N, VIA if N < 60 else [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)])
    r = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dual_ws=dws)
    oo = Q.default_opts(); eo = EOpts()
    for f, _ in EOpts._fields_:
        if hasattr(oo, f):      # (max_soc, recalc_y: parking options, not in the quadcopter checker's record; zero)
            setattr(eo, f, getattr(oo, f))
    L = P.quad_layout(N); prob = P.pack_quad_problem(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dual_ws=dws)
    z = np.zeros(L["len"]); info = np.zeros(8)
    emu.emu_quad_solve(C.c_int(N), dp(prob), C.byref(eo), dp(z), dp(info))
    assert r["exitflag"] == 1 and info[0] == 0 and info[7] == 1
    assert int(info[1]) == r["iters"] and int(info[6]) == r["nreg"]
    assert abs(info[2] - r["obj"]) < 1e-9 * abs(r["obj"])
    xp = z[L["x"]:L["u"]].reshape(N + 1, 12).T; up = z[L["u"]:L["t"]].reshape(N, 4).T
    assert np.abs(xp - r["xp"]).max() < 1e-4 and np.abs(up - r["up"]).max() < 1e-5 and abs(z[L["t"]] - r["t"]) < 1e-9


def test_emu_quadcopter_dist_variant_matches_oracle(Q, emu):
    """QuadcopterDist.jl formulation (no slack variable, x[10] in [-1.5, 3]) through the same device source"""
    N = 30; Ts = round(0.25 * 80 / N * 100) / 100
    xWS = Q.warm_start(Q.X0, Q.XF, N, VIA)
    r = Q.quadcopter_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    oo = Q.default_opts(); eo = EOpts()
    for f, _ in EOpts._fields_:
        if hasattr(oo, f):      # (max_soc, recalc_y: parking options, not in the quadcopter checker's record; zero)
            setattr(eo, f, getattr(oo, f))
    L = P.quad_layout(N); prob = P.pack_quad_problem(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dist=1)
    z = np.zeros(L["len"]); info = np.zeros(8)
    emu.emu_quad_solve(C.c_int(N), dp(prob), C.byref(eo), dp(z), dp(info))
    assert r["exitflag"] == 1 and info[7] == 1 and int(info[1]) == r["iters"]
    assert abs(info[2] - r["obj"]) < 1e-8 * abs(r["obj"]) and np.abs(z[L["s"]:L["so"]]).max() == 0
    xp = z[L["x"]:L["u"]].reshape(N + 1, 12).T
    assert np.abs(xp - r["xp"]).max() < 1e-4


def test_emu_quad_ipopt_switches_match_oracle_options(Q, emu):
    """opts.max_soc = 4 (IPOPT A-5.5 .. A-5.9), opts.lsq_init (least-squares initial multipliers) and opts.obj_scaling (gradient-based objective scaling, factor 100 / 2 100)
    -- together obca_quadcopter_reference_opts -- through the device
    source against the oracle's options: same iteration and regularisation counts, same optimum, for each switch alone and for both -- and the switches are
    exercised (the counts differ from the solve without them)"""
    from obca_amd import scenarios as S
    N = 60; q = S.make_quad_batch(3, N, seed=20260925, random_endpoints=True)
    L = P.quad_layout(N)
    for i in range(3):
        res = {}
        for (msoc, lsq, osc) in ((0, 0, 0), (4, 0, 0), (0, 1, 0), (0, 0, 1), (4, 1, 1)):
            oo = Q.default_opts(); oo.max_soc = msoc; oo.lsq_init = lsq; oo.obj_scaling = osc; eo = EOpts()
            for f, _ in EOpts._fields_:
                if hasattr(oo, f):
                    setattr(eo, f, getattr(oo, f))
            eo.max_soc = msoc; eo.lsq_init = lsq; eo.obj_scaling = osc
            r = Q.quadcopter_signed_dist(q["x0"][i], q["xF"][i], N, q["Ts"], q["R"], q["ob"], q["xWS"][i], q["timeWS"], opts=oo)
            prob = P.pack_quad_problem(q["x0"][i], q["xF"][i], N, q["Ts"], q["R"], q["ob"], q["xWS"][i], q["timeWS"])
            z = np.zeros(L["len"]); info = np.zeros(8)
            emu.emu_quad_solve(C.c_int(N), dp(prob), C.byref(eo), dp(z), dp(info))
            assert r["exitflag"] == 1 and info[7] == 1 and int(info[1]) == r["iters"] and int(info[6]) == r["nreg"], (i, msoc, lsq, osc, info, r["iters"], r["nreg"])
            assert abs(info[2] - r["obj"]) < 1e-9 * abs(r["obj"]) and abs(z[L["t"]] - r["t"]) < 1e-9
            up = z[L["u"]:L["t"]].reshape(N, 4).T
            assert np.abs(up - r["up"]).max() < 1e-5
            res[(msoc, lsq, osc)] = int(info[1])
        assert res[(4, 0, 0)] != res[(0, 0, 0)] and res[(0, 1, 0)] != res[(0, 0, 0)] and res[(0, 0, 1)] != res[(0, 0, 0)], (i, res)


@pytest.mark.parametrize("s_max", [1e-2, 1e-4], ids=["s_max_0.01", "s_max_0.0001"])
def test_quad_termination_scaling_factors_follow_the_oracle(Q, emu, s_max):
    """IPOPT's s_d, s_c made active by a small s_max (with the default 100 they are 1 on these instances): kernel source and oracle stop at the same iteration -- the check that
    found nothing wrong here and would have found the parking kernels' never-stored sums (tests/test_emu_cpu.py)"""
    N = 16; Ts = round(0.25 * 80 / N * 100) / 100
    xWS = Q.warm_start(Q.X0, Q.XF, N, VIA)
    oo = Q.default_opts(); oo.s_max = s_max; eo = EOpts()
    for f, _ in EOpts._fields_:
        if hasattr(oo, f):
            setattr(eo, f, getattr(oo, f))
    r = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, opts=oo)
    r1 = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    L = P.quad_layout(N); prob = P.pack_quad_problem(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dual_ws=1)
    z = np.zeros(L["len"]); info = np.zeros(8)
    emu.emu_quad_solve(C.c_int(N), dp(prob), C.byref(eo), dp(z), dp(info))
    assert r["exitflag"] == 1 and info[7] == 1 and int(info[1]) == r["iters"], (info[1], r["iters"], r1["iters"])
    assert abs(info[2] - r["obj"]) < 1e-9 * abs(r["obj"])
    assert r["iters"] != r1["iters"]         # the factors were active (they also enter the barrier update: the count may go either way)


def test_quad_solves_keep_their_bits_under_finite_poison(emu):
    """tests/emu/obca_emu.cpp fills the quadcopter solver's work buffers and its LDS block with a pattern before the solve (OBCA_EMU_POISON): horizons, formulations and switches
    drawn at random, 1e30 / -1e30 / NaN leave every bit of the result (NaN alone would not do: it hides behind fmax)"""
    import os
    import emu_solver as E
    from obca_amd import scenarios as S
    rng = np.random.default_rng(20260926)
    try:
        for draw in range(10):
            N = int(rng.choice([8, 12, 16, 21])); q = S.make_quad_batch(2, N, seed=int(rng.integers(1, 1000)))
            kw = dict(max_soc=4, lsq_init=1, obj_scaling=1) if rng.integers(0, 2) else dict()
            kw["dist"] = bool(rng.integers(0, 2)); kw["dual_ws"] = bool(rng.integers(0, 2))
            solve = lambda: E.quadcopter_signed_dist_batch(q["x0"][1:], q["xF"][1:], N, q["Ts"], q["R"], q["ob"], q["xWS"][1:], q["timeWS"], **kw)
            os.environ.pop("OBCA_EMU_POISON", None)
            ref = solve()
            for value in ("1e30", "-1e30", "nan"):
                os.environ["OBCA_EMU_POISON"] = "3"; os.environ["OBCA_EMU_POISON_VALUE"] = value
                o = solve()
                assert np.array_equal(o["info"], ref["info"], equal_nan=True) and np.array_equal(o["xp"], ref["xp"], equal_nan=True), (draw, N, kw, value, o["info"], ref["info"])
    finally:
        os.environ.pop("OBCA_EMU_POISON", None); os.environ.pop("OBCA_EMU_POISON_VALUE", None)
