"""CPU tests of the quadcopter oracle (QuadcopterSignedDist path): Newton direction vs dense autograd KKT, full solve, feasibility."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def Q():
    import oracle_quad
    oracle_quad.lib()
    return oracle_quad


@pytest.mark.parametrize("dist", [0, 1], ids=["signed_dist", "dist"])
def test_quad_newton_direction_vs_dense_autograd(Q, dist):
    torch = pytest.importorskip("torch")
    from nlp_ref_quad import QuadNLP
    rng = np.random.default_rng(2)
    N, Ts, R, ob = 7, 0.3, 0.25, Q.OB_CLAMPED
    x0 = Q.X0.copy(); x0[9:12] = [0.1, -0.2, 0.15]       # non-zero stage-1 rates exercise the single-index quirk
    nlp = QuadNLP(x0, Q.XF, N, Ts, R, ob, dist=bool(dist)); L = Q.layout(N); n, m = L["n"], L["m"]
    assert n == nlp.n + 12 and m == nlp.m
    xWS = Q.warm_start(x0, Q.XF, N)
    v = np.zeros(n)
    X = xWS.copy(); X[1:] += 0.05 * rng.standard_normal((N, 12)); X[1:, 3:6] = 0.1 * rng.standard_normal((N, 3)); X[0] = x0
    v[L["x"]:L["x"] + 12 * (N + 1)] = X.reshape(-1)
    v[L["u"]:L["u"] + 4 * N] = rng.uniform(3, 6, 4 * N); v[L["t"]] = 1.1
    for k, cnt in (("lam", 30), ("s", 5), ("so", 5)):
        v[L[k]:L[k] + cnt * (N + 1)] = rng.uniform(0.1, 1, cnt * (N + 1))
    y = rng.standard_normal(m); zL = rng.uniform(0.1, 2, n); zU = rng.uniform(0.1, 2, n)
    mu, dw, dc = 0.1, 500.0, 1e-6
    ok, dv, dy, errs = Q.newton(N, Ts, R, x0, Q.XF, ob, v, y, zL, zU, mu, dw, dc, dist=dist)
    assert ok == 1
    if dist:
        assert np.abs(dv[L["s"]:L["so"]]).max() == 0          # the frozen slack does not move
    vv = v[12:]; zLr = zL[12:].copy(); zUr = zU[12:].copy()
    f, g, c, J, H = nlp.eval_all(vv, y)
    IL = np.isfinite(nlp.lb); IU = np.isfinite(nlp.ub); zLr[~IL] = 0; zUr[~IU] = 0
    dL = np.where(IL, vv - nlp.lb, 1.0); dU = np.where(IU, nlp.ub - vv, 1.0)
    Sig = nlp.mult * (np.where(IL, zLr / dL, 0) + np.where(IU, zUr / dU, 0))
    gphi = g - mu * nlp.mult * np.where(IL, 1 / dL, 0) + mu * nlp.mult * np.where(IU, 1 / dU, 0)
    dcv = np.concatenate([np.zeros(12 * N + 12), dc * np.ones(10 * (N + 1))])
    K = np.block([[H + np.diag(Sig + dw), J.T], [J, -np.diag(dcv)]])
    sol = np.linalg.solve(K, -np.concatenate([gphi + J.T @ y, c]))
    ev = np.linalg.eigvalsh(K)
    assert (ev > 0).sum() == nlp.n and (ev < 0).sum() == nlp.m
    assert np.abs(dv[12:] - sol[:nlp.n]).max() < 1e-9 * max(1, np.abs(sol[:nlp.n]).max())
    assert np.abs(dy - sol[nlp.n:]).max() < 1e-8 * max(1, np.abs(sol[nlp.n:]).max())
    rd = g + J.T @ y - nlp.mult * zLr + nlp.mult * zUr
    assert abs(errs[0] - np.abs(rd).max()) < 1e-9 * np.abs(rd).max() and abs(errs[1] - np.abs(c).max()) < 1e-12


def test_quad_shipped_scenario_solves_and_is_feasible(Q):
    """mainQuadcopter.jl scenario (boxes as clamped by the plot call); way-point warm start instead of 3-D A*"""
    N = 60; Ts = round(0.25 * 80 / N * 100) / 100          # mainQuadcopter.jl:131
    via = [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)]
    xWS = Q.warm_start(Q.X0, Q.XF, N, via)
    r = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    assert r["exitflag"] == 1 and r["status"] == 0 and r["slack"].sum() < 1e-3        # :229-234, :285-288
    xp, up, ts, lp = r["xp"], r["up"], r["timeScale"], r["lp"]
    assert xp.shape == (12, N + 1) and up.shape == (4, N) and lp.shape == (30, N + 1)
    assert np.abs(xp[:, 0] - Q.X0).max() == 0 and np.abs(xp[:, N] - Q.XF).max() < 1e-4
    # a-posteriori feasibility in the spirit of constrSatisfaction.jl (tolerance 1e-3): bounds, dynamics, separation rows
    assert up.min() >= 1.2 - 1e-3 and up.max() <= 7.8 + 1e-3 and 0.5 - 1e-3 <= ts[0] <= 2 + 1e-3
    from nlp_ref_quad import XLB, XUB
    assert (xp >= XLB[:, None] - 1e-3).all() and (xp <= XUB[:, None] + 1e-3).all()
    import torch
    from nlp_ref_quad import QuadNLP
    nlp = QuadNLP(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED)
    v = np.concatenate([xp.T[1:].reshape(-1), up.T.reshape(-1), [ts[0]], lp.T.reshape(-1), r["slack"].T.reshape(-1), np.zeros(5 * (N + 1))])
    c = nlp.c(torch.tensor(v)).numpy()
    assert np.abs(c[:12 * N + 12]).max() < 1e-3                     # dynamics + terminal
    cob = c[12 * N + 12:].reshape(N + 1, 5, 2)
    assert np.abs(cob[:, :, 0]).max() < 1e-3 and cob[:, :, 1].min() > -1e-3     # |A'lam|=1 ; separation >= R (row slack = 0 here)
    # the ball really clears every box: Euclidean point-box distance >= R
    for k in range(N + 1):
        for j in range(5):
            hi = Q.OB_CLAMPED[j, :3]; lo = -Q.OB_CLAMPED[j, 3:]
            d = np.linalg.norm(xp[:3, k] - np.clip(xp[:3, k], lo, hi))
            assert d >= Q.EGO_R - 2e-3, (k, j, d)


def test_quad_reference_start_converges_through_block_restoration(Q):
    """the reference's start lambda = 0.05 (QuadcopterSignedDist.jl:204-208) has A'lambda = 0: the row |A'lambda|^2 == 1 has a zero gradient
    there (rank-deficient Jacobian; IPOPT leaves it through its restoration phase).  The block restoration (closed-form minimiser of the block's
    constraint violation at fixed positions) runs before the first iteration and the solve converges to the optimum of the dual-warm-start run"""
    for N, Ts, via in ((20, 1.0, [(2.25, 1.5, 0.3), (7.25, 4.5, 2.5)]), (60, 0.33, [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)])):
        xWS = Q.warm_start(Q.X0, Q.XF, N, via)
        r0 = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dual_ws=0)
        r1 = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, dual_ws=1)
        assert r0["exitflag"] == 1 and r1["exitflag"] == 1
        assert abs(r0["obj"] - r1["obj"]) < 1e-9 * abs(r1["obj"]) and abs(r0["t"] - r1["t"]) < 1e-9


def test_quadcopter_dist_variant_solves_collision_free(Q):
    """QuadcopterDist.jl (next-1 row): no slack variable, x[10] in [-1.5, 3], exit flag 0/1 only"""
    N = 60; Ts = round(0.25 * 80 / N * 100) / 100
    via = [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)]
    xWS = Q.warm_start(Q.X0, Q.XF, N, via)
    r = Q.quadcopter_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    assert r["exitflag"] == 1 and np.abs(r["slack"]).max() == 0
    xp = r["xp"]
    assert np.abs(xp[:, N] - Q.XF).max() < 1e-4 and xp[9].min() >= -1.5 - 1e-6 and xp[9].max() <= 3 + 1e-6
    for k in range(N + 1):
        for j in range(5):
            hi = Q.OB_CLAMPED[j, :3]; lo = -Q.OB_CLAMPED[j, 3:]
            assert np.linalg.norm(xp[:3, k] - np.clip(xp[:3, k], lo, hi)) >= Q.EGO_R - 1e-4
    r2 = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    assert abs(r["obj"] - r2["obj"]) < 1e-2 * abs(r2["obj"])      # the signed-distance optimum has ~zero slack, so the two optima nearly coincide


def test_quad_oracle_optimum_matches_third_party_sqp_fixture(Q):
    """a short hop (N=8) solved by scipy SLSQP (tests/golden/make_slsqp_quad.py): same optimal value.  The cost has no term on the path itself
    (inputs, rates, time only), so the minimiser is not unique along it: the objective and the time scale are compared, not the positions."""
    from conftest import golden
    g = golden("slsqp_quad_N8.npz"); N = int(g["N"])
    assert float(g["cviol"]) < 1e-7
    r = Q.quadcopter_signed_dist(g["x0"], g["xF"], N, float(g["Ts"]), Q.EGO_R, Q.OB_CLAMPED, g["xWS"], 1.0)
    assert r["exitflag"] == 1
    assert abs(r["obj"] - float(g["obj"])) < 1e-4 * abs(float(g["obj"])) and r["obj"] >= float(g["obj"]) - 1e-9
    assert abs(r["t"] - float(g["t"])) < 1e-5 and np.abs(r["up"] - g["up"]).max() < 5e-3


def test_quad_oracle_reproduces_golden_config4(Q):
    """the committed config-4 fixture is what the current oracle produces (guards the fixture the GPU test relies on)"""
    from conftest import golden
    g = golden("oracle_quad_cfg4.npz"); B, N = int(g["B"]), int(g["N"])
    for i in range(B):
        r = Q.quadcopter_signed_dist(g["x0"][i], g["xF"][i], N, float(g["Ts"]), float(g["R"]), g["ob"], g["xWS"][i], 1.0)
        assert r["exitflag"] == g["exitflag"][i] == 1 and r["iters"] == g["iters"][i]
        assert np.array_equal(r["xp"], g["xp"][i]) and np.array_equal(r["up"], g["up"][i]) and r["t"] == g["t"][i]


def test_quadcopter_second_order_correction_option():
    """IPOPT's second-order correction as an option of the quadcopter oracle (opts.max_soc; the reference's quadcopter call runs IPOPT's default, 4): the same
    solved instances, fewer iterations on most, the same optimum up to the flatness of the cost -- and not worth its passes: the census in DESIGN.md section 9"""
    import oracle_quad as Q
    from obca_amd import scenarios as S
    q = S.make_quad_batch(4, 60, seed=20260925, random_endpoints=True)
    base = [Q.quadcopter_signed_dist(q["x0"][i], q["xF"][i], 60, q["Ts"], q["R"], q["ob"], q["xWS"][i], q["timeWS"]) for i in range(4)]
    o = Q.default_opts(); o.max_soc = 4
    soc = [Q.quadcopter_signed_dist(q["x0"][i], q["xF"][i], 60, q["Ts"], q["R"], q["ob"], q["xWS"][i], q["timeWS"], opts=o) for i in range(4)]
    assert all(r["exitflag"] == 1 for r in base + soc)
    assert sum(r["iters"] for r in soc) < sum(r["iters"] for r in base)
    assert all(abs(a["obj"] - b["obj"]) <= 2e-3 * max(1.0, abs(a["obj"])) for a, b in zip(base, soc))


@pytest.mark.parametrize("dist", [0, 1], ids=["signed_dist", "dist"])
def test_quad_least_squares_multipliers_vs_dense_autograd(Q, dist):
    """opts.lsq_init: IPOPT's initial multipliers = the least-squares estimate  [I J'; J 0] [w; y] = -[grad f - zL + zU; 0]  (identity on every primal variable of the
    reference's model: the N + 1 timeScale variables the solver carries as one t contribute N + 1 to its diagonal entry).  The oracle's structured solve against a
    dense solve on the autograd Jacobian; and the option changes the path of a solve, not its optimum"""
    pytest.importorskip("torch")
    from nlp_ref_quad import QuadNLP
    rng = np.random.default_rng(3)
    N, Ts, R, ob = 7, 0.3, 0.25, Q.OB_CLAMPED
    x0 = Q.X0.copy(); x0[9:12] = [0.1, -0.2, 0.15]
    nlp = QuadNLP(x0, Q.XF, N, Ts, R, ob, dist=bool(dist)); L = Q.layout(N); n, m = L["n"], L["m"]
    xWS = Q.warm_start(x0, Q.XF, N)
    v = np.zeros(n)
    X = xWS.copy(); X[1:] += 0.05 * rng.standard_normal((N, 12)); X[1:, 3:6] = 0.1 * rng.standard_normal((N, 3)); X[0] = x0
    v[L["x"]:L["x"] + 12 * (N + 1)] = X.reshape(-1)
    v[L["u"]:L["u"] + 4 * N] = rng.uniform(3, 6, 4 * N); v[L["t"]] = 1.1
    for k, cnt in (("lam", 30), ("s", 5), ("so", 5)):
        v[L[k]:L[k] + cnt * (N + 1)] = rng.uniform(0.1, 1, cnt * (N + 1))
    zL = rng.uniform(0.1, 2, n); zU = rng.uniform(0.1, 2, n)
    ok, yls = Q.lsq_multipliers(N, Ts, R, x0, Q.XF, ob, v, zL, zU, dist=dist)
    assert ok == 1
    vv = v[12:]; zLr = zL[12:].copy(); zUr = zU[12:].copy()
    f, g, c, J, H = nlp.eval_all(vv, np.zeros(m))
    zLr[~np.isfinite(nlp.lb)] = 0; zUr[~np.isfinite(nlp.ub)] = 0
    gz = g - nlp.mult * zLr + nlp.mult * zUr
    K = np.block([[np.diag(nlp.mult.astype(float)), J.T], [J, np.zeros((m, m))]])
    sol = np.linalg.solve(K, -np.concatenate([gz, np.zeros(m)]))
    assert np.abs(yls - sol[nlp.n:]).max() < 1e-10 * max(1.0, np.abs(sol[nlp.n:]).max())
    if not dist:
        o = Q.default_opts(); o.lsq_init = 1
        xw = Q.warm_start(Q.X0, Q.XF, 16, [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)])
        a = Q.quadcopter_signed_dist(Q.X0, Q.XF, 16, 1.25, Q.EGO_R, Q.OB_CLAMPED, xw, 1.0)
        b = Q.quadcopter_signed_dist(Q.X0, Q.XF, 16, 1.25, Q.EGO_R, Q.OB_CLAMPED, xw, 1.0, opts=o)
        assert a["exitflag"] == 1 and b["exitflag"] == 1 and a["iters"] != b["iters"] and abs(a["obj"] - b["obj"]) < 1e-3 * abs(a["obj"])
        r0 = Q.quadcopter_signed_dist(Q.X0, Q.XF, 16, 1.25, Q.EGO_R, Q.OB_CLAMPED, xw, 1.0, opts=o, dual_ws=0)      # the reference's own start: singular system, y stays 0
        r1 = Q.quadcopter_signed_dist(Q.X0, Q.XF, 16, 1.25, Q.EGO_R, Q.OB_CLAMPED, xw, 1.0, dual_ws=0)
        assert r0["exitflag"] == 1 and r0["iters"] == r1["iters"] and r0["obj"] == r1["obj"]


def test_quad_gradient_based_objective_scaling_option(Q):
    """opts.obj_scaling: IPOPT's default nlp_scaling_method (gradient-based, max gradient 100).  On QuadcopterSignedDist the largest objective gradient at the reference's start is
    the slack penalty 1e2 + 2e3 * 1 = 2 100: the algorithm runs on f / 21 (fewer inertia rungs, termination 21 x looser in unscaled terms: the optimum moves by ~1e-4 relative);
    on QuadcopterDist (no slack variable) the largest gradient is 0.25 + 10 t = 10.25 < 100: factor 1, the solve must not change by a bit"""
    N = 30; Ts = round(0.25 * 80 / N * 100) / 100
    xWS = Q.warm_start(Q.X0, Q.XF, N, [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)])
    o = Q.default_opts(); o.obj_scaling = 1
    a = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    b = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, opts=o)
    assert a["exitflag"] == 1 and b["exitflag"] == 1 and (a["iters"], a["nreg"]) != (b["iters"], b["nreg"])
    assert abs(a["obj"] - b["obj"]) < 2e-3 * abs(a["obj"]) and b["obj"] >= a["obj"] - 1e-6 * abs(a["obj"])      # the unscaled objective is what is reported; the looser solve stops a little higher
    c = Q.quadcopter_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0)
    d = Q.quadcopter_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, opts=o)
    assert c["exitflag"] == 1 and (c["iters"], c["nreg"], c["obj"]) == (d["iters"], d["nreg"], d["obj"]) and np.array_equal(c["xp"], d["xp"])
