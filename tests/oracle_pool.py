"""TEST INFRASTRUCTURE: the CPU oracle on many instances in parallel (spawned workers: safe next to a live HIP runtime)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init():
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _opts(O, sw):
    """oracle options with the IPOPT switches set explicitly: sw = (max_soc, recalc_y, lsq_init[, restoration]) or None for the defaults (all off)"""
    o = O.default_opts()
    if sw:
        o.max_soc, o.recalc_y, o.lsq_init = (int(v) for v in sw[:3])
        o.restoration = int(sw[3]) if len(sw) > 3 else 0
    return o


IPOPT = (4, 1, 1, 1)      # the reference's IPOPT configuration as obca_reference_opts sets it: max_soc = 4, recalc_y = "yes", least-squares initial multipliers, block restoration


def _parking_chunk(args):
    _init()
    import numpy as np
    import oracle as O
    (lo, x0, xF, N, Ts, L, ego, XYb, vOb, A, b, xWS, uWS, sw) = args
    out = []; o = _opts(O, sw)
    for i in range(len(x0)):
        r = O.parking_signed_dist(x0[i], xF[i], N, Ts[i], L, ego, XYb, vOb, A, b, xWS[i][:, 0], xWS[i][:, 1], xWS[i][:, 2], 0, xWS[i], uWS[i], opts=o)
        out.append((lo + i, r["exitflag"], r["iters"], r["obj"], r["xp"], r["up"], r["t"]))
    return out


def parking_oracle_all(bt, xWS, workers=None, chunk=8, switches=None):
    """oracle solution of every instance of a shared-obstacle batch: list of (index, exitflag, iters, obj, xp, up, t)"""
    import multiprocessing as mp
    import oracle as O
    O.build()
    B = len(bt["x0"]); N = xWS.shape[1] - 1
    jobs = [(lo, bt["x0"][lo:lo + chunk], bt["xF"][lo:lo + chunk], N, bt["Ts"][lo:lo + chunk], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
             xWS[lo:lo + chunk], bt["uWS"][lo:lo + chunk], switches) for lo in range(0, B, chunk)]
    workers = workers or min(os.cpu_count() or 1, 64)
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_parking_chunk, jobs)
    return sorted([r for ch in res for r in ch], key=lambda r: r[0])


def _quad_chunk(args):
    _init()
    import oracle_quad as Q
    (lo, x0, xF, N, Ts, R, ob, xWS, max_soc, lsq_init, obj_scaling) = args
    out = []; o = Q.default_opts(); o.max_soc = int(max_soc); o.lsq_init = int(lsq_init); o.obj_scaling = int(obj_scaling)
    for i in range(len(x0)):
        r = Q.quadcopter_signed_dist(x0[i], xF[i], N, Ts, R, ob, xWS[i], 1.0, opts=o)
        out.append((lo + i, r["exitflag"], r["iters"], r["nreg"], r["obj"], r["up"], r["t"]))
    return out


def quad_oracle_all(bt, workers=None, chunk=8, max_soc=0, lsq_init=0, obj_scaling=0):
    """quadcopter oracle on every instance of a batch of scenarios.make_quad_batch: list of (index, exitflag, iters, nreg, obj, up, t); max_soc = 4, lsq_init = 1, obj_scaling = 1: with IPOPT's
    second-order correction, least-squares initial multipliers and gradient-based objective scaling (the option set of obca_quadcopter_reference_opts)"""
    import multiprocessing as mp
    import oracle_quad as Q
    Q.lib()
    B = len(bt["x0"]); N = bt["xWS"].shape[1] - 1
    jobs = [(lo, bt["x0"][lo:lo + chunk], bt["xF"][lo:lo + chunk], N, bt["Ts"], bt["R"], bt["ob"], bt["xWS"][lo:lo + chunk], max_soc, lsq_init, obj_scaling) for lo in range(0, B, chunk)]
    workers = workers or min(os.cpu_count() or 1, 64)
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_quad_chunk, jobs)
    return sorted([r for ch in res for r in ch], key=lambda r: r[0])


def _mixed_chunk(args):
    _init()
    import oracle as O
    (lo, x0, xF, N, Ts, L, ego, XYb, vOb, A, b, xWS, uWS, sw) = args
    out = []; o = _opts(O, sw)
    for i in range(len(x0)):
        r = O.parking_signed_dist(x0[i], xF[i], N, Ts[i], L, ego, XYb, vOb[i], A[i], b[i], xWS[i][:, 0], xWS[i][:, 1], xWS[i][:, 2], 0, xWS[i], uWS[i], opts=o)
        out.append((lo + i, r["exitflag"], r["iters"], r["obj"], r["xp"]))
    return out


def mixed_oracle_all(bt, xWS, workers=None, chunk=8, switches=None):
    """parking oracle on every instance of a batch with per-instance obstacle sets (scenarios.make_mixed_batch)"""
    import multiprocessing as mp
    import oracle as O
    O.build()
    B = len(bt["x0"]); N = xWS.shape[1] - 1
    jobs = [(lo, bt["x0"][lo:lo + chunk], bt["xF"][lo:lo + chunk], N, bt["Ts"][lo:lo + chunk], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"][lo:lo + chunk],
             bt["A"][lo:lo + chunk], bt["b"][lo:lo + chunk], xWS[lo:lo + chunk], bt["uWS"][lo:lo + chunk], switches) for lo in range(0, B, chunk)]
    workers = workers or min(os.cpu_count() or 1, 64)
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_mixed_chunk, jobs)
    return sorted([r for ch in res for r in ch], key=lambda r: r[0])
