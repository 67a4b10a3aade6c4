"""The HIP solver source, compiled as a host emulation (64 lanes as a loop), against the oracle: kernel LOGIC without a GPU."""
import ctypes as C
import numpy as np
import pytest
from obca_amd import scenarios as S
import packing as P

D = C.POINTER(C.c_double)
dp = lambda a: a.ctypes.data_as(D)


class EOpts(C.Structure):      # obca::Opts (obca_solver.h); every field but the padding exists under the same name in the oracle's option record
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int)] + \
        [(n, C.c_double) for n in ("mu_init kappa_eps kappa_mu theta_mu tau_min bound_push bound_frac dw_min dw0 dw_max "
                                   "kw_inc0 kw_inc kw_dec dc_bar kappa_c gamma_theta gamma_phi delta s_theta s_phi eta_phi "
                                   "gamma_alpha s_max kappa_sigma constr_viol_tol dual_inf_tol compl_inf_tol rho_term").split()] + \
        [("max_soc", C.c_int), ("recalc_y", C.c_int), ("lsq_init", C.c_int), ("obj_scaling", C.c_int), ("restoration", C.c_int)]


def copy_opts(oo):
    eo = EOpts()
    for n, _ in EOpts._fields_:
        if True:
            setattr(eo, n, getattr(oo, n))
    return eo


def test_layouts_agree(oracle):
    for N, v in ((5, [2, 2, 1]), (80, [2, 2, 1, 1]), (12, [4]), (7, [1] * 10)):
        a = oracle.layout(N, v); b = P.layout(N, len(v), sum(v))
        assert all(a[k] == b[k] for k in a)


import pytest


@pytest.mark.parametrize("dist", [0, 1], ids=["signed_dist", "dist"])
def test_emu_newton_direction_matches_oracle(oracle, emu, backwards, dist):
    rng = np.random.default_rng(3)
    for N in (2, 3, 9):
        sc = S.BACKWARDS; A, b, v = backwards["A"], backwards["b"], backwards["vOb"]; nOb = len(v); M = int(v.sum())
        x0 = np.array([-5, 9.0, -0.1, 0.]); Ts, xWS, uWS = S.warm_start_backwards(x0, sc["xF"], N); Ts = 0.7
        L = P.layout(N, nOb, M)
        z = np.zeros(L["len"])
        X = xWS.copy(); X[1:] += 0.05 * rng.standard_normal((N, 4)); X[0] = x0
        z[L["x"]:L["u"]] = X.reshape(-1)
        z[L["u"]:L["t"]] = np.clip(uWS + 0.05 * rng.standard_normal((N, 2)), -0.3, 0.3).reshape(-1)
        z[L["t"]] = 0.95
        z[L["lam"]:L["sl"]] = rng.uniform(0.1, 1, L["sl"] - L["lam"])
        z[L["sl"]:L["so"]] = rng.uniform(0.1, 1, L["so"] - L["sl"]) if dist else 0.01 * rng.standard_normal(L["so"] - L["sl"])
        z[L["so"]:L["ss"]] = rng.uniform(0.1, 1, L["ss"] - L["so"])
        z[L["ss"]:L["pi"]] = rng.uniform(-0.3, 0.3, N)
        z[L["pi"]:L["zxL"]] = rng.standard_normal(L["zxL"] - L["pi"])
        z[L["zxL"]:] = rng.uniform(0.1, 2, L["len"] - L["zxL"])
        mu, dw, dc = 0.05, 5.0, 1e-7
        ok, d, errs = oracle.newton(N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, 0, x0, sc["xF"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], z, mu, dw, dc, dist=dist)
        prob = P.pack_problem(x0, sc["xF"], N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, dist=dist)
        d2 = np.zeros_like(z); aux = np.zeros(10)
        ok2 = emu.emu_newton(C.c_int(N), dp(prob), dp(z), C.c_int(L["len"]), C.c_double(mu), C.c_double(dw), C.c_double(dc),
                             C.c_double(1e3), C.c_double(0.99), dp(d2), dp(aux))
        assert ok == 1 and ok2 == 1
        nd = L["zxL"]
        assert np.abs(d[:nd] - d2[:nd]).max() < 1e-9 * max(1.0, np.abs(d[:nd]).max())
        if dist:
            assert np.abs(d[L["sl"]:L["so"]]).max() > 0       # the norm-row slack moves
        assert np.allclose(aux[:3], errs, rtol=1e-10)
        # the fused line-search step (trial point -> second iterate buffer, assembled there): at alpha = 0 the primal point and its f / th1 / bar are the assembly's;
        # at alpha > 0 the trial point is z + alpha d with y + min(alpha, az) dy, and its assembly equals a stand-alone assembly of that point
        for alpha in (0.0, 0.5 * min(aux[7], 1.0)):        # (aux[7]: the fraction-to-boundary step length)
            aux2 = np.zeros(16); d3 = np.zeros_like(z); zn = np.zeros_like(z)
            ok3 = emu.emu_newton_fused(C.c_int(N), dp(prob), dp(z), C.c_int(L["len"]), C.c_double(mu), C.c_double(dw), C.c_double(dc), C.c_double(1e3), C.c_double(0.99),
                                       C.c_double(alpha), C.c_double(1e10), dp(d3), dp(aux2), dp(zn))
            assert ok3 == 1 and np.array_equal(d3, d2)
            npr = L["pi"]; ay = min(alpha, aux2[8])
            assert np.abs(zn[:npr] - (z[:npr] + alpha * d2[:npr])).max() < 1e-14 * max(1.0, np.abs(zn[:npr]).max())
            assert np.abs(zn[npr:nd] - (z[npr:nd] + ay * d2[npr:nd])).max() < 1e-14 * max(1.0, np.abs(zn[npr:nd]).max())      # (the kernel forms them with one rounding: fma)
            if alpha == 0.0:
                assert np.allclose(aux2[10:13], aux[4:7], rtol=1e-12)
            d4 = np.zeros_like(z); aux3 = np.zeros(10)
            emu.emu_newton(C.c_int(N), dp(prob), dp(zn), C.c_int(L["len"]), C.c_double(mu), C.c_double(0.0), C.c_double(dc), C.c_double(1e3), C.c_double(0.99), dp(d4), dp(aux3))
            assert np.allclose(aux2[10:13], aux3[4:7], rtol=1e-13, atol=0) and np.allclose(aux2[13:16], aux3[0:3], rtol=1e-12)
            # bound multipliers of the trial point: z + az dz, clamped to [mu / (ks s), ks mu / s] (ks = 1e10: not active here)
            assert (zn[nd:] > 0).all()


@pytest.mark.parametrize("dist,N", [(0, 20), (1, 20), (0, 21), (0, 99)], ids=["signed_dist", "dist", "odd_horizon", "beyond_the_composed_pairs"])
def test_emu_full_solve_matches_oracle(oracle, emu, backwards, dist, N):
    B = 3 if N <= 21 else 1
    bt = S.make_batch(S.BACKWARDS, B, N)
    v = bt["vOb"]; nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
    oo = oracle.default_opts(); eo = copy_opts(oo)
    assert emu.emu_opts_size() == C.sizeof(eo)
    for i in range(B):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        r = oracle.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                                       xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i], lWS, nWS, dist=dist)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                              xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, dist=dist)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=bt["A"])
        zo = np.zeros_like(z0); info = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
        xp, up, t, lp, npp, sl = P.unpack_solution(zo, N, nOb, M)
        assert int(info[7]) == r["exitflag"] == 1 and int(info[1]) == r["iters"]
        assert np.abs(xp - r["xp"]).max() < 1e-8 and np.abs(up - r["up"]).max() < 1e-8 and abs(t - r["t"]) < 1e-10
        assert np.abs(lp - r["lp"]).max() < 1e-7 and np.abs(npp - r["np"]).max() < 1e-7


def test_emu_dualws_matches_oracle(oracle, emu, backwards):
    A, b, v = backwards["A"], backwards["b"], backwards["vOb"]
    rng = np.random.default_rng(5)
    g = np.array([2.35, 1.0, 2.35, 1.0])
    for _ in range(20):
        X, Y, psi = rng.uniform(-10, 10), rng.uniform(5.5, 10), rng.uniform(-np.pi, np.pi)
        lo, no, do = oracle.dualmult_ws(0, v, A, b, [X], [Y], [psi], S.EGO)
        r0 = 0
        for j, vj in enumerate(v):
            a1 = np.ascontiguousarray(A[r0:r0 + vj, 0]); a2 = np.ascontiguousarray(A[r0:r0 + vj, 1]); bj = np.ascontiguousarray(b[r0:r0 + vj])
            lam = np.zeros(4); mu = np.zeros(4); d = C.c_double(0)
            cs, sn = np.cos(psi), np.sin(psi)
            emu.emu_dualws(C.c_int(int(vj)), dp(a1), dp(a2), dp(bj), dp(g), C.c_double(X + 1.35 * cs), C.c_double(Y + 1.35 * sn),
                           C.c_double(cs), C.c_double(sn), dp(lam), dp(mu), C.byref(d))
            assert abs(d.value - do[0, j]) < 1e-10 and np.abs(lam[:vj] - lo[0, r0:r0 + vj]).max() < 1e-9
            assert np.abs(mu - no[0, 4 * j:4 * j + 4]).max() < 1e-9
            r0 += vj


@pytest.mark.parametrize("dist", [0, 1], ids=["signed_dist", "dist"])
def test_emu_exit_flag_after_failed_attempts_follows_the_reference(oracle, emu, backwards, dist):
    """both attempts hit the iteration limit: ParkingSignedDist then asks its acceptance test (infeasible -> 0, :278-283); ParkingDist asks it
    before the retry and INVERTS it afterwards (infeasible -> exitflag 1, ParkingDist.jl:277-282) -- reproduced, not fixed"""
    N = 20; bt = S.make_batch(S.BACKWARDS, 1, N)
    v = bt["vOb"]; nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
    oo = oracle.default_opts(); oo.max_iter = 3; eo = copy_opts(oo)
    xWS = bt["xWS"][0].copy(); xWS[0] = bt["x0"][0]
    lWS, nWS, _ = oracle.dualmult_ws(N, v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
    r = oracle.parking_signed_dist(bt["x0"][0], bt["xF"][0], N, bt["Ts"][0], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                                   xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][0], lWS, nWS, opts=oo, dist=dist)
    prob = P.pack_problem(bt["x0"][0], bt["xF"][0], N, bt["Ts"][0], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                          xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, dist=dist)
    z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][0], lWS, nWS, A=bt["A"])
    zo = np.zeros_like(z0); info = np.zeros(8)
    emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
    assert r["status"] == 1 and int(info[0]) == 1 and int(info[1]) == r["iters"] == 6
    assert r["exitflag"] == int(info[7]) == (1 if dist else 0)
    # the acceptance test itself: C restatement == the verbatim Python restatement, on an optimal and on an unfinished solution
    from obca_amd import validate as K
    ro = oracle.parking_signed_dist(bt["x0"][0], bt["xF"][0], N, bt["Ts"][0], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                                    xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][0], lWS, nWS, dist=dist)
    for sol in (ro, r):
        a = K.parking_constraints_ref(bt["x0"][0], bt["xF"][0], N, bt["Ts"][0], bt["L"], bt["ego"], bt["XYbounds"], nOb, v, bt["A"], bt["b"],
                                      sol["xp"], sol["up"], sol["lp"], sol["np"], np.full(N + 1, sol["t"]), 0, 0 if dist else 1)
        b_ = oracle.ref_constraints(bt["x0"][0], bt["xF"][0], N, bt["Ts"][0], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                                    sol["xp"], sol["up"], sol["t"], sol["lp"], sol["np"], 0, 0 if dist else 1)
        assert a == b_
    assert ro["exitflag"] == 1


@pytest.mark.parametrize("budget,max_iter", [(1, 3000), (5, 3000), (4, 7)], ids=["every_pass", "five_passes", "retry_attempt"])
def test_emu_sliced_solve_is_bit_identical(oracle, emu, backwards, budget, max_iter):
    """time slicing (two-launch schedule): a solve parked every `budget` factorisation passes and resumed from its slice record walks through
    exactly the same iterates as an uninterrupted solve -- also when the cut falls into the second attempt (max_iter small forces the retry)"""
    N = 20; bt = S.make_batch(S.BACKWARDS, 2, N)
    v = bt["vOb"]; nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
    oo = oracle.default_opts(); oo.max_iter = max_iter; eo = copy_opts(oo)
    for i in range(2):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"],
                              xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=bt["A"])
        za = np.zeros_like(z0); ia = np.zeros(8); zb = np.zeros_like(z0); ib = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(za), dp(ia))
        launches = emu.emu_solve_sliced(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), C.c_int(budget), dp(zb), dp(ib))
        passes = int(ia[1] + ia[6])
        assert launches > 1 and launches >= int(ia[1]) // budget          # the cut falls between iterations
        assert np.array_equal(ia, ib) and np.array_equal(za, zb)


def test_emu_wide_obstacles_match_oracle(oracle, emu):
    """obstacles with 5..8 half-space rows (obstHrep.jl emits one row per polygon edge for any vertex count): the OB_VMAX = 8 instantiation of the
    (stage, obstacle) block code against the oracle -- Newton direction sizes, DualMultWS and the full solve"""
    N = 16
    bt = S.make_mixed_batch(6, N, seed=11, rows=(5, 8), max_extra=4)
    oo = oracle.default_opts(); eo = copy_opts(oo)
    done = 0
    for i in range(6):
        v = np.asarray(bt["vOb"][i]); A = bt["A"][i]; b = bt["b"][i]
        if v.max() <= 4:
            continue
        nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, dWS = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        # DualMultWS sub-problem of the widest obstacle through the emulated kernel code
        j = int(np.argmax(v)); r0 = int(v[:j].sum()); vj = int(v[j]); k = 3
        rl = P.row_lengths(A)                      # the kernels (and the oracle) work on unit-length rows; lambda is handed back in the caller's row scaling
        a1 = np.ascontiguousarray(A[r0:r0 + vj, 0] / rl[r0:r0 + vj]); a2 = np.ascontiguousarray(A[r0:r0 + vj, 1] / rl[r0:r0 + vj]); bj = np.ascontiguousarray(b[r0:r0 + vj] / rl[r0:r0 + vj])
        lam = np.zeros(8); mu = np.zeros(4); d = C.c_double(0); cs, sn = np.cos(xWS[k, 2]), np.sin(xWS[k, 2]); g = np.array([2.35, 1.0, 2.35, 1.0])
        emu.emu_dualws(C.c_int(vj), dp(a1), dp(a2), dp(bj), dp(g), C.c_double(xWS[k, 0] + 1.35 * cs), C.c_double(xWS[k, 1] + 1.35 * sn), C.c_double(cs), C.c_double(sn),
                       dp(lam), dp(mu), C.byref(d))
        assert abs(d.value - dWS[k, j]) < 1e-9 and np.abs(lam[:vj] / rl[r0:r0 + vj] - lWS[k, r0:r0 + vj]).max() < 1e-8
        r = oracle.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, A, b,
                                       xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i], lWS, nWS)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=A)
        zo = np.zeros_like(z0); info = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
        xp, up, t, lp, npp, sl = P.unpack_solution(zo, N, nOb, M, A=A)
        assert int(info[7]) == r["exitflag"] == 1 and int(info[1]) == r["iters"]
        assert np.abs(xp - r["xp"]).max() < 1e-7 and np.abs(up - r["up"]).max() < 1e-7 and abs(t - r["t"]) < 1e-9
        assert np.abs(lp - r["lp"]).max() < 1e-5
        done += 1
    assert done >= 2


def test_emu_rows_of_any_length_give_the_same_solve(oracle, emu):
    """the kernels' packing (obca_amd/packing.py = the C ABI's batch_upload_range) brings the half-space rows to unit length and hands lambda back in the caller's
    scaling: rows scaled by 0.02 .. 1e3 through the emulated kernels -- DualMultWS + interior point -- give the same states, inputs, iteration counts, and the oracle's
    lambda for the scaled rows"""
    import emu_solver as E
    N, B = 20, 2
    bt = S.make_batch(S.BACKWARDS, B, N, seed=7)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    s = np.array([250.0, 0.02, 1e3, 7.0, 0.3])
    a = (bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"])
    w = (xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    o0 = E.parking_signed_dist_batch(*a, bt["A"], bt["b"], *w); o1 = E.parking_signed_dist_batch(*a, bt["A"] * s[:, None], bt["b"] * s, *w)
    assert (o0["exitflag"] == 1).all() and (o0["iters"] == o1["iters"]).all() and np.abs(o0["xp"] - o1["xp"]).max() < 1e-9 and np.abs(o0["up"] - o1["up"]).max() < 1e-9
    for i in range(B):
        assert np.abs(o0["lp"][i] - o1["lp"][i] * s[:, None]).max() < 1e-8 * max(1.0, np.abs(o0["lp"][i]).max())
        r = oracle.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"] * s[:, None], bt["b"] * s,
                                       xWS[i, :, 0], xWS[i, :, 1], xWS[i, :, 2], 0, xWS[i], bt["uWS"][i])
        assert r["iters"] == o1["iters"][i] and np.abs(r["xp"] - o1["xp"][i]).max() < 1e-7 and np.abs(r["lp"] - o1["lp"][i]).max() < 1e-5 * max(1.0, np.abs(r["lp"]).max())


def test_second_order_correction_in_the_kernels_follows_the_oracle(oracle):
    """obca_opts.max_soc > 0 (IPOPT's second-order correction, A-5.5..A-5.9): the kernels' correction steps -- c_soc accumulated from the iterate and the rejected trial, the
    system of the iterate re-assembled with it, the trial along the correction direction, the Newton direction rebuilt when no correction is accepted -- against the
    oracle's max_soc option on config-3 instances: the same iterations (corrections change the count on some of them), the same optimum."""
    import emu_solver as E
    bt = S.make_batch(S.PARALLEL, 24, 80, seed=20260925, goal_jitter=True)
    A, b, v = S.scenario_hrep(S.PARALLEL)
    base, soc = oracle.default_opts(), oracle.default_opts(); soc.max_soc = 4

    def args(i):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        return xWS, (bt["x0"][i], bt["xF"][i], 80, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
    # the instances are chosen by what the option does on the oracle (the batch itself follows the planner's settings): up to four on which the corrections change the
    # iteration count, and two on which they do not
    pairs = {i: (oracle.parking_signed_dist(*args(i)[1], opts=base), oracle.parking_signed_dist(*args(i)[1], opts=soc)) for i in range(24)}
    moved = [i for i in pairs if pairs[i][0]["iters"] != pairs[i][1]["iters"]][:4]; same = [i for i in pairs if pairs[i][0]["iters"] == pairs[i][1]["iters"]][:2]
    changed = tried = 0
    for i in moved + same:
        xWS, a = args(i); r0, r1 = pairs[i]
        e = E.parking_signed_dist_batch(bt["x0"][i:i + 1], bt["xF"][i:i + 1], 80, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[None, :, 0], xWS[None, :, 1],
                                        xWS[None, :, 2], 0, xWS[None], bt["uWS"][i:i + 1], max_soc=4)
        assert e["exitflag"][0] == r1["exitflag"] == 1 and e["iters"][0] == r1["iters"]
        assert np.abs(e["xp"][0] - r1["xp"]).max() < 1e-8 and abs(e["obj"][0] - r1["obj"]) < 1e-9 * abs(r1["obj"])
        changed += r0["iters"] != r1["iters"]; tried += e["nsoc"][0, 0] > 0
    assert changed >= 2 and tried >= 4, (moved, same, changed, tried)


def test_recalc_y_in_the_kernels_follows_the_oracle(oracle, emu):
    """obca_opts.recalc_y = 1 (recalc_y = "yes", ParkingSignedDist.jl:41): once the accepted iterate's constraint violation is below 1e-6 the equality multipliers are replaced
    by their least-squares estimate -- the structured solve with H := I, zero constraint right-hand side, z-form gradients (template parameter LSQ of the phases).  Against
    the oracle's option: the same iterations, and the multipliers (pi, nu, y_g, y_o) of the final iterate -- the re-estimated ones -- to 1e-11."""
    import emu_solver as E
    N = 80; sc = S.BACKWARDS
    bt = S.make_batch(sc, 5, N, seed=20260925)
    A, b, v = S.scenario_hrep(sc); v = np.ravel(v).astype(int); nOb, M = len(v), int(v.sum()); L = P.layout(N, nOb, M)
    oo = oracle.default_opts(); oo.recalc_y = 1; eo = copy_opts(oo)
    assert eo.recalc_y == 1 and eo.max_soc == 0
    n_re = 0; ys = slice(L["pi"], L["zxL"])
    for i in range(5):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], S.EGO)
        a = (bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i], lWS, nWS)
        r = oracle.parking_signed_dist(*a, opts=oo, full=True); r0 = oracle.parking_signed_dist(*a, full=True)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i][:N], lWS, nWS); zo = np.zeros_like(z0); info = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
        assert int(info[1]) == r["iters"] and int(info[7]) == r["exitflag"] == 1
        assert np.abs(zo[ys] - r["zfull"][ys]).max() < 1e-11 and np.abs(zo[:L["nprimal"]] - r["zfull"][:L["nprimal"]]).max() < 1e-9
        if emu.emu_last_recalc() > 0:
            n_re += 1
            assert np.abs(r["zfull"][ys] - r0["zfull"][ys]).max() > 0          # the estimate is not the multiplier the iteration carried (it agrees with it to ~1e-9 at the solution)
            assert np.abs(r["zfull"][ys] - r0["zfull"][ys]).max() < 1e-6
    assert n_re >= 3


def test_least_squares_initial_multipliers_in_the_kernels_follow_the_oracle(oracle):
    """obca_opts.lsq_init = 1 (IPOPT's default initialisation of the equality multipliers: least-squares estimate at the starting point, constr_mult_init_max = 1e3): the same
    LSQ phases as recalc_y, here far from a solution -- the iteration counts move by up to 50 % against y0 = 0, and the kernels' follow the oracle's option exactly."""
    import emu_solver as E
    bt = S.make_batch(S.PARALLEL, 6, 80, seed=20260925, goal_jitter=True)
    A, b, v = S.scenario_hrep(S.PARALLEL)
    oo = oracle.default_opts(); oo.lsq_init = 1
    changed = 0
    for i in (1, 2, 3, 5):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        a = (bt["x0"][i], bt["xF"][i], 80, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        r0 = oracle.parking_signed_dist(*a); r1 = oracle.parking_signed_dist(*a, opts=oo)
        e = E.parking_signed_dist_batch(bt["x0"][i:i + 1], bt["xF"][i:i + 1], 80, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[None, :, 0], xWS[None, :, 1],
                                        xWS[None, :, 2], 0, xWS[None], bt["uWS"][i:i + 1], lsq_init=1)
        assert e["exitflag"][0] == r1["exitflag"] == 1 and e["iters"][0] == r1["iters"]
        assert np.abs(e["xp"][0] - r1["xp"]).max() < 1e-8 and abs(e["obj"][0] - r1["obj"]) < 1e-9 * abs(r1["obj"])
        changed += r0["iters"] != r1["iters"]
    assert changed >= 3


@pytest.mark.parametrize("rows", [(3, 4), (5, 8)], ids=["rows_up_to_4", "rows_up_to_8"])
def test_ipopt_switches_on_wide_obstacles_follow_the_oracle(oracle, emu, rows):
    """max_soc, recalc_y and lsq_init together on instances whose obstacles have up to 4 / up to 8 half-space rows: the OB_VMID and OB_VMAX instantiations of the correction
    and least-squares phases (the other switch tests run on <= 2 rows) against the oracle with the same options"""
    N = 24
    bt = S.make_mixed_batch(8, N, seed=23, rows=rows, max_extra=4)
    oo = oracle.default_opts(); oo.max_soc = 4; oo.recalc_y = 1; oo.lsq_init = 1; oo.restoration = 1; eo = copy_opts(oo)
    base = oracle.default_opts()
    done = changed = 0
    for i in range(8):
        v = np.asarray(bt["vOb"][i]); A = bt["A"][i]; b = bt["b"][i]
        if v.max() < rows[0]:
            continue
        nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        a = (bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i], lWS, nWS)
        r = oracle.parking_signed_dist(*a, opts=oo); r0 = oracle.parking_signed_dist(*a, opts=base)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=A)
        zo = np.zeros_like(z0); info = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
        xp, up, t, lp, npp, sl = P.unpack_solution(zo, N, nOb, M, A=A)
        assert int(info[7]) == r["exitflag"] and int(info[1]) == r["iters"], (i, info[1], r["iters"])
        if r["exitflag"] == 1:
            assert np.abs(xp - r["xp"]).max() < 1e-7 and np.abs(up - r["up"]).max() < 1e-7 and abs(t - r["t"]) < 1e-9
        done += 1; changed += r["iters"] != r0["iters"]
    assert done >= 3 and changed >= 2


def test_bounded_sincos_of_the_kernels_is_accurate(emu):
    """sincos_bounded / tan_bounded (obca_model.h: Cody-Waite reduction + fdlibm kernels, straight-line) against 80-bit references: <= 1.6 ulp on |x| <= 1e5 (headings, steering
    and Euler angles are a few radians), tan <= 2.5 ulp on the steering range, exact at 0, NaN in -> NaN out"""
    emu.emu_tan.restype = C.c_double
    rng = np.random.default_rng(3)
    for scale, tol in ((1.0, 1.6), (10.0, 1.6), (1e3, 1.6), (1e5, 1.6)):
        x = rng.uniform(-scale, scale, 20000); s = np.zeros_like(x); c = np.zeros_like(x); sv, cv = C.c_double(), C.c_double()
        for i, xi in enumerate(x):
            emu.emu_sincos(C.c_double(xi), C.byref(sv), C.byref(cv)); s[i] = sv.value; c[i] = cv.value
        xl = x.astype(np.longdouble); rs, rc = np.sin(xl), np.cos(xl)
        us = np.abs(s.astype(np.longdouble) - rs) / np.spacing(np.abs(rs.astype(float))); uc = np.abs(c.astype(np.longdouble) - rc) / np.spacing(np.abs(rc.astype(float)))
        assert us.max() <= tol and uc.max() <= tol, (scale, float(us.max()), float(uc.max()))
    x = rng.uniform(-0.7, 0.7, 20000); t = np.array([emu.emu_tan(C.c_double(xi)) for xi in x]); rt = np.tan(x.astype(np.longdouble))
    assert (np.abs(t.astype(np.longdouble) - rt) / np.spacing(np.abs(rt.astype(float)))).max() <= 2.5
    sv, cv = C.c_double(), C.c_double()
    emu.emu_sincos(C.c_double(0.0), C.byref(sv), C.byref(cv)); assert sv.value == 0.0 and cv.value == 1.0
    emu.emu_sincos(C.c_double(float("nan")), C.byref(sv), C.byref(cv)); assert sv.value != sv.value and cv.value != cv.value


def test_a_discarded_recalc_y_estimate_leaves_no_stale_records(oracle, emu):
    """ph_recalc_y overwrites the stage / obstacle / Riccati records with the least-squares system before it knows whether the estimate is kept.  When it is NOT kept (forced here
    through the emulation's hook: an estimate is attempted at EVERY accepted iterate and discarded after the overwrite) the next iteration must assemble afresh: the solve with recalc_y on and every estimate discarded is then bit for bit the
    solve without the option (round-3 advisor finding: have_asm used to stay 1 on that path and the stale least-squares records were factorised as the Newton system)."""
    N = 80; sc = S.BACKWARDS
    bt = S.make_batch(sc, 4, N, seed=20260925)
    A, b, v = S.scenario_hrep(sc); v = np.ravel(v).astype(int); nOb, M = len(v), int(v.sum()); L = P.layout(N, nOb, M)
    o_off = copy_opts(oracle.default_opts()); oo = oracle.default_opts(); oo.recalc_y = 1; o_on = copy_opts(oo)
    entered = 0
    for i in range(4):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], S.EGO)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], S.L_WHEELBASE, S.EGO, S.XYBOUNDS, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i][:N], lWS, nWS)
        za = np.zeros_like(z0); ia = np.zeros(8); zb = np.zeros_like(z0); ib = np.zeros(8); zc = np.zeros_like(z0); ic = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(o_off), dp(za), dp(ia))
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(o_on), dp(zc), dp(ic)); entered += emu.emu_last_recalc() > 0
        emu.emu_force_recalc_failure(1)
        try:
            emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(o_on), dp(zb), dp(ib))
        finally:
            emu.emu_force_recalc_failure(0)
        assert int(ia[7]) == int(ib[7]) == 1 and int(ia[1]) == int(ib[1]) and np.array_equal(za, zb), (i, ia, ib)
    assert entered >= 3          # (the option does reach its estimate on these instances, so the forced failures above were real)


def test_instances_at_the_obstacle_and_row_limits_follow_the_oracle(oracle):
    """OBCA_NOBMAX = 16 obstacles / OBCA_MMAX = 64 half-space rows per instance (round 4; 10 / 40 before): the problem header, the item loops and the row offsets at their limits --
    the kernel source in the host emulation against the oracle on instances with 14-16 obstacles of 3-4 rows"""
    import emu_solver as E
    N = 40
    bt = S.make_mixed_batch(24, N, seed=5, max_extra=13, rows=(3, 4), max_rows=64)
    big = [i for i in range(24) if len(bt["vOb"][i]) >= 14][:3]
    assert len(big) == 3 and max(len(bt["vOb"][i]) for i in big) == 16 and max(int(np.sum(bt["vOb"][i])) for i in big) > 40
    for i in big:
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        a = (bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"][i], bt["A"][i], bt["b"][i], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        r = oracle.parking_signed_dist(*a)
        e = E.parking_signed_dist_batch(bt["x0"][i:i + 1], bt["xF"][i:i + 1], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"][i], bt["A"][i], bt["b"][i],
                                        xWS[None, :, 0], xWS[None, :, 1], xWS[None, :, 2], 0, xWS[None], bt["uWS"][i:i + 1])
        assert e["exitflag"][0] == r["exitflag"] == 1 and e["iters"][0] == r["iters"], (i, e["exitflag"][0], r["exitflag"], e["iters"][0], r["iters"])
        assert np.abs(e["xp"][0] - r["xp"]).max() < 1e-7


@pytest.mark.parametrize("dist", [0, 1], ids=["signed_dist", "dist"])
@pytest.mark.parametrize("s_max", [1e-2, 1e-4], ids=["s_max_0.01", "s_max_0.0001"])
def test_termination_scaling_factors_in_the_kernels_follow_the_oracle(oracle, emu, backwards, s_max, dist):
    """IPOPT's s_d, s_c (mean multiplier magnitude over s_max, at least 1) scale the optimality error of the termination test.  With the default s_max = 100 both are 1 on nearly
    every instance, which is why parity never noticed that the kernels' multiplier sums were not stored (round 4 - round 5); a small s_max makes them bite:
    the solve then ends earlier, at the same iteration as the oracle's."""
    N, B = 20, 3
    bt = S.make_batch(S.BACKWARDS, B, N)
    v = bt["vOb"]; nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
    oo = oracle.default_opts(); oo.s_max = s_max; eo = copy_opts(oo)
    o1 = oracle.default_opts()
    fewer = 0
    for i in range(B):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        lWS, nWS, _ = oracle.dualmult_ws(N, v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        args = (bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i], lWS, nWS)
        r = oracle.parking_signed_dist(*args, opts=oo, dist=dist); r1 = oracle.parking_signed_dist(*args, opts=o1, dist=dist)
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], v, bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, dist=dist)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=bt["A"])
        zo = np.zeros_like(z0); info = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(zo), dp(info))
        xp, up, t, lp, npp, sl = P.unpack_solution(zo, N, nOb, M)
        assert int(info[7]) == r["exitflag"] == 1 and int(info[1]) == r["iters"], (i, info[1], r["iters"], r1["iters"])
        assert np.abs(xp - r["xp"]).max() < 1e-8 and abs(info[2] - r["obj"]) < 1e-9 * max(1.0, abs(r["obj"]))
        fewer += r["iters"] < r1["iters"]
    assert fewer >= 1          # the factors were active: with them the oracle itself stops earlier than with s_max = 100


def test_a_solve_reads_nothing_it_has_not_written_whatever_the_pattern(emu, backwards):
    """The work buffers in HBM, the static and the dynamic LDS block are filled with a pattern before the solve (OBCA_EMU_POISON, tests/emu/obca_emu.cpp): the result keeps its
    bits.  NaN alone is not enough -- it passes through fmax / fmin and fails every comparison silently, which is how a read of two never-written LDS words survived the NaN
    poisoning of round 5 and was found only when another process's kernels left large finite numbers there."""
    import os
    N, B = 20, 2
    bt = S.make_batch(S.BACKWARDS, B, N)
    import emu_solver as E
    def solve(**kw):
        xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
        return E.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], **kw)
    try:
        for kw in (dict(), dict(max_soc=4, recalc_y=1, lsq_init=1, restoration=1), dict(dist=True)):
            os.environ.pop("OBCA_EMU_POISON", None)
            ref = solve(**kw)
            assert (ref["exitflag"] == 1).all()
            for value in ("nan", "1e30", "-1e30", "1e-30", "0.5", "-3.0", "1e300"):
                os.environ["OBCA_EMU_POISON"] = "7"; os.environ["OBCA_EMU_POISON_VALUE"] = value
                o = solve(**kw)
                assert np.array_equal(o["info"], ref["info"]) and np.array_equal(o["xp"], ref["xp"]) and np.array_equal(o["up"], ref["up"]), (kw, value, o["iters"], ref["iters"])
    finally:
        os.environ.pop("OBCA_EMU_POISON", None); os.environ.pop("OBCA_EMU_POISON_VALUE", None)


def test_random_problems_keep_their_bits_under_finite_poison(emu):
    """The check above on 36 seeded random draws of (horizon, scenario incl. 1-16 obstacles of up to 8 rows, formulation, fixed / variable time, option set, iteration limit
    small enough to end the first attempt -> retry path): the pattern 1e30 / -1e30 in work buffers and LDS leaves info, states and inputs unchanged."""
    import os
    import emu_solver as E
    rng = np.random.default_rng(20260926)
    d0 = E.default_opts
    checked = 0
    try:
        for draw in range(36):
            N = int(rng.choice([5, 8, 13, 21, 34, 55])); kind = int(rng.integers(0, 3)); dist = bool(rng.integers(0, 2)); fix = int(rng.integers(0, 2))
            kw = dict(max_soc=4, recalc_y=1, lsq_init=1, restoration=1) if rng.integers(0, 2) else dict()
            max_iter = int(rng.choice([200, 200, 6]))
            if kind == 2:
                bt = S.make_mixed_batch(2, N, seed=int(rng.integers(1, 1000)), min_obstacles=1, max_extra=13, rows=(3, 8), max_rows=64); v, A, b = bt["vOb"][1], bt["A"][1], bt["b"][1]; i = 1
            else:
                bt = S.make_batch(S.BACKWARDS if kind == 0 else S.PARALLEL, 2, N, seed=int(rng.integers(1, 1000))); v, A, b = bt["vOb"], bt["A"], bt["b"]; i = 1
            xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]; sl = slice(i, i + 1); Ts = np.broadcast_to(bt["Ts"], (2,))[sl]

            def patched(max_iter=max_iter):
                o = d0(); o.max_iter = max_iter; return o
            E.default_opts = patched

            def solve():
                return E.parking_signed_dist_batch(bt["x0"][sl], bt["xF"][sl], N, Ts, bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[sl, :, 0], xWS[sl, :, 1], xWS[sl, :, 2], fix, xWS[sl], bt["uWS"][sl],
                                                   dist=dist, **kw)
            os.environ.pop("OBCA_EMU_POISON", None)
            ref = solve()
            for value in ("1e30", "-1e30"):
                os.environ["OBCA_EMU_POISON"] = "7"; os.environ["OBCA_EMU_POISON_VALUE"] = value
                o = solve()
                assert np.array_equal(o["info"], ref["info"], equal_nan=True) and np.array_equal(o["xp"], ref["xp"], equal_nan=True) and np.array_equal(o["up"], ref["up"], equal_nan=True), \
                    (draw, N, kind, dist, fix, kw, max_iter, value, o["info"], ref["info"])
            checked += 1
    finally:
        E.default_opts = d0
        os.environ.pop("OBCA_EMU_POISON", None); os.environ.pop("OBCA_EMU_POISON_VALUE", None)
    assert checked == 36


def test_random_problems_the_kernels_follow_the_oracle(oracle):
    """120 seeded random draws of (horizon 10-100, backwards / parallel / 1-16 obstacles of up to 8 rows, formulation, fixed or variable time, option set): kernel source
    (emulation) and oracle agree in exit flag and iteration count -- a sweep of 4 000 such draws found one count off by two on the same solution (round 5) -- and in the
    trajectory to 1e-6."""
    import emu_solver as E
    off = 0; worst = 0.0; solved = 0
    for seed in range(1000, 1120):
        rng = np.random.default_rng(seed)
        N = int(rng.choice([10, 20, 33, 48, 64, 80, 100])); kind = int(rng.integers(0, 3)); dist = bool(rng.integers(0, 2)) if kind != 2 else False; fix = int(rng.integers(0, 4) == 0)
        ref = bool(rng.integers(0, 2)); kw = dict(max_soc=4, recalc_y=1, lsq_init=1, restoration=1) if ref else dict()
        if kind == 2:
            bt = S.make_mixed_batch(2, N, seed=int(rng.integers(1, 10000)), min_obstacles=1, max_extra=int(rng.choice([7, 13])), rows=(3, 8) if rng.integers(0, 2) else (3, 4), max_rows=64)
            v, A, b = bt["vOb"][1], bt["A"][1], bt["b"][1]
        else:
            bt = S.make_batch(S.BACKWARDS if kind == 0 else S.PARALLEL, 2, N, seed=int(rng.integers(1, 10000))); v, A, b = bt["vOb"], bt["A"], bt["b"]
        i = 1; xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]; sl = slice(i, i + 1); Ts = np.broadcast_to(bt["Ts"], (2,))
        e = E.parking_signed_dist_batch(bt["x0"][sl], bt["xF"][sl], N, Ts[sl], bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[sl, :, 0], xWS[sl, :, 1], xWS[sl, :, 2], fix, xWS[sl], bt["uWS"][sl],
                                        dist=dist, **kw)
        oo = oracle.default_opts()
        if ref:
            oo.max_soc = 4; oo.recalc_y = 1; oo.lsq_init = 1; oo.restoration = 1
        r = oracle.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, float(Ts[i]), bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[i, :, 0], xWS[i, :, 1], xWS[i, :, 2], fix, xWS[i], bt["uWS"][i],
                                       opts=oo, dist=int(dist))
        assert int(e["exitflag"][0]) == r["exitflag"], (seed, N, kind, dist, fix, ref)
        if int(e["iters"][0]) != r["iters"]:
            off += 1
        elif r["exitflag"] == 1:
            solved += 1; worst = max(worst, float(np.abs(e["xp"][0] - r["xp"]).max()))
    assert off <= 1 and solved >= 110 and worst < 1e-6, (off, solved, worst)


def test_random_problems_sliced_solves_are_bit_identical(oracle, emu):
    """the two-launch schedule on 80 seeded random draws of (horizon, scenario, formulation, option set, slice length 1-11 passes, iteration limit small enough to cut inside the
    retry attempt): the solve parked and resumed launch after launch returns the bits of the uninterrupted one (600 such draws were run once in round 5: none differed)"""
    import emu_solver as E
    launches_total = 0
    for seed in range(5000, 5080):
        rng = np.random.default_rng(seed)
        N = int(rng.choice([8, 13, 20, 33, 48])); kind = int(rng.integers(0, 3)); dist = int(rng.integers(0, 2)) if kind != 2 else 0
        ref = bool(rng.integers(0, 2)); budget = int(rng.integers(1, 12)); max_iter = int(rng.choice([3000, 3000, 7, 12]))
        if kind == 2:
            bt = S.make_mixed_batch(2, N, seed=int(rng.integers(1, 10000)), min_obstacles=1, max_extra=13, rows=(3, 8), max_rows=64); v, A, b = np.asarray(bt["vOb"][1]), bt["A"][1], bt["b"][1]
        else:
            bt = S.make_batch(S.BACKWARDS if kind == 0 else S.PARALLEL, 2, N, seed=int(rng.integers(1, 10000))); v, A, b = bt["vOb"], bt["A"], bt["b"]
        i = 1; nOb = len(v); M = int(np.sum(v)); L = P.layout(N, nOb, M)
        eo = E.default_opts(); eo.max_iter = max_iter
        if ref:
            eo.max_soc = 4; eo.recalc_y = 1; eo.lsq_init = 1; eo.restoration = 1
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]; Ts = float(np.broadcast_to(bt["Ts"], (2,))[i])
        lWS, nWS, _ = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
        prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, Ts, bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, dist=dist)
        z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=A)
        za = np.zeros_like(z0); ia = np.zeros(8); zb = np.zeros_like(z0); ib = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(za), dp(ia))
        launches_total += emu.emu_solve_sliced(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), C.c_int(budget), dp(zb), dp(ib))
        assert np.array_equal(ia, ib, equal_nan=True) and np.array_equal(za, zb, equal_nan=True), (seed, N, kind, dist, ref, budget, max_iter, ia, ib)
    assert launches_total > 800


def test_block_restoration_in_the_kernels_follows_the_oracle(oracle):
    """obca_opts.restoration in the kernel source (restore_blocks, obca_solver_ipm.h) against the oracle's: corridor instances whose wedges intrude 0.15 m into the warm start
    (DualMultWS gives lambda = 0 on the penetrating poses: degenerate blocks), reference configuration with the restoration at the start -- solved on both sides, the same
    iteration counts (one knife edge in six tolerated: these are 60-120 iteration solves; measured on the first 12 instances: 11 equal, one 117 against 115 to the same objective), the same trajectories; and a sliced solve (parked every 5 passes) keeps the bits."""
    import emu_solver as E
    N, B = 80, 64
    bt = S.make_corridor_batch(B, N, seed=11, clearance=(-0.15, 0.2))
    same = 0
    for i in (0, 1, 2, 3, 4, 6):          # (every one of them has degenerate blocks at its start; 2 fails without the restoration)
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        oo = oracle.default_opts(); oo.max_soc = 4; oo.recalc_y = 1; oo.lsq_init = 1; oo.restoration = 1
        a = (bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"][i], bt["A"][i], bt["b"][i], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        r = oracle.parking_signed_dist(*a, opts=oo)
        e = E.parking_signed_dist_batch(bt["x0"][i:i + 1], bt["xF"][i:i + 1], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"][i], bt["A"][i], bt["b"][i],
                                        xWS[None, :, 0], xWS[None, :, 1], xWS[None, :, 2], 0, xWS[None], bt["uWS"][i:i + 1], max_soc=4, recalc_y=1, lsq_init=1, restoration=1)
        assert e["exitflag"][0] == r["exitflag"] == 1, (i, e["exitflag"][0], r["exitflag"])
        if e["iters"][0] == r["iters"]:
            same += 1; assert np.abs(e["xp"][0] - r["xp"]).max() < 1e-7, i
        else:
            assert abs(e["obj"][0] - r["obj"]) < 1e-4 * max(1.0, abs(r["obj"])), (i, e["iters"][0], r["iters"])
    assert same >= 5, same
    # parked and resumed: the restoration count and the pending threshold reset travel in the slice record
    i = 13; v = np.asarray(bt["vOb"][i]); A, b = bt["A"][i], bt["b"][i]; nOb = len(v); M = int(v.sum()); L = P.layout(N, nOb, M)
    xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
    lWS, nWS, _ = oracle.dualmult_ws(N, v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], bt["ego"])
    prob = P.pack_problem(bt["x0"][i], bt["xF"][i], N, float(bt["Ts"][i]), bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0)
    z0 = P.pack_start(N, nOb, M, xWS, bt["uWS"][i], lWS, nWS, A=A)
    emu = E.load()
    for mode in (1, 2):
        eo = E.default_opts(); eo.max_soc = 4; eo.recalc_y = 1; eo.lsq_init = 1; eo.restoration = mode
        za = np.zeros_like(z0); ia = np.zeros(8); zb = np.zeros_like(z0); ib = np.zeros(8)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), dp(za), dp(ia))
        emu.emu_solve_sliced(C.c_int(N), dp(prob), dp(z0), C.c_int(L["len"]), C.byref(eo), C.c_int(5), dp(zb), dp(ib))
        assert ia[7] == 1 and np.array_equal(ia, ib) and np.array_equal(za, zb), (mode, ia, ib)
