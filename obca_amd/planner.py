"""
Host-side warm-start planner: ctypes front end of libobca_plan.so (obca_amd/csrc/obca_planner.cpp, a compact Hybrid A*) and the
conversion of its path into the (rx, ry, ryaw, xWS, uWS, Ts) arrays the signed-distance entry points take.

Mirrors the step before the hot path in the reference: /root/reference/AutonomousParking/main.jl:216-252 (hybrid A* -> rx, ry, ryaw ->
velocity profile -> down-sampling to the horizon N); SURVEY.md section 8f "next-2".  CPU only, no GPU involved.
"""
import ctypes as C
import os
import subprocess
import numpy as np
from . import scenarios as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "obca_planner.cpp")
_SRC_REF = os.path.join(_HERE, "csrc", "obca_planner_ref.cpp")      # REFERENCE mode: the reference's Hybrid A* restated
_LIB = os.path.join(_HERE, "csrc", "libobca_plan.so")
_D = C.POINTER(C.c_double); _I = C.POINTER(C.c_int)
_lib = None


def build_library(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(_SRC_REF)):
        from .buildflags import GXX
        subprocess.check_call(GXX + ["-O2", "-pthread", "-I" + os.path.join(_HERE, "..", "include"), "-o", _LIB, _SRC, _SRC_REF])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_library())
    return _lib


DEFAULT_OPTS = dict(xy_res=0.25, yaw_res_deg=7.5, step=0.6, max_steer=0.6, steer_samples=2, margin=0.1, goal_xy_tol=0.3,
                    goal_yaw_tol_deg=8.0, reverse_cost=1.5, switch_cost=2.0, steer_cost=0.3, max_expansions=400000, analytic=0.85, steer_change_cost=0.2,
                    h_weight=1.0, rs_heuristic=0, nh_res=0.0, nh_yaw_res_deg=7.5)
# the cost constants of the reference's search (hybrid_a_star.jl:60-63: SB_COST 10, BACK_COST 0, STEER_CHANGE_COST 10, STEER_COST 0; arc length x 1 forwards, x BACK_COST
# backwards -- a tiny positive value here keeps reverse arcs from being free, which an A* with a consistent heuristic needs), its grids (:46-54: 0.3 m, 5 deg, 5 steer
# commands) and its full-lock analytic expansion: `hybrid_astar(..., **REFERENCE_COSTS)` / `warm_start(..., **REFERENCE_COSTS)`
REFERENCE_COSTS = dict(reverse_cost=0.05, switch_cost=10.0, steer_cost=0.0, steer_change_cost=10.0, analytic=1.0)
REFERENCE_GRID = dict(xy_res=0.3, yaw_res_deg=5.0, steer_samples=2)      # (too coarse for this planner's exact collision test in the 6 m bay of the parallel scenario)
# analytic: Reeds-Shepp expansion towards the goal (0 = off), the value is the fraction of the steering lock its arcs use.  The reference uses
# the full lock (1.0); a warm start that rides the steering bound costs the interior point iterations (config 3, 96 instances on the oracle:
# mean 43.4 / worst 400 iterations at 1.0, 37.4 / 71 at 0.85, 40.6 / 90 without the expansion), hence 0.85.


def hybrid_astar(start, goal, vOb, A, b, ego=S.EGO, L=S.L_WHEELBASE, XYbounds=S.XYBOUNDS, **kw):
    """start, goal: (x, y, yaw); obstacles as H-rep rows (vOb = rows per obstacle).  Returns (path (K,3), dir (K,), expansions) or None."""
    o = dict(DEFAULT_OPTS); o.update(kw)
    opts = np.array([o[k] for k in DEFAULT_OPTS], float)
    vOb = np.ascontiguousarray(vOb, np.int32); A = np.ascontiguousarray(A, float); b = np.ascontiguousarray(b, float)
    cap = 20000; path = np.zeros((cap, 3)); dr = np.zeros(cap, np.int32); nexp = C.c_int(0)
    s = np.ascontiguousarray(start, float)[:3].copy(); g = np.ascontiguousarray(goal, float)[:3].copy()
    e = np.ascontiguousarray(ego, float); xy = np.ascontiguousarray(XYbounds, float)
    n = _load().obca_plan_hybrid_astar2(s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_int(len(vOb)), vOb.ctypes.data_as(_I), A.ctypes.data_as(_D),
                                        b.ctypes.data_as(_D), e.ctypes.data_as(_D), C.c_double(L), xy.ctypes.data_as(_D), opts.ctypes.data_as(_D), C.c_int(len(opts)),
                                        path.ctypes.data_as(_D), dr.ctypes.data_as(_I), C.c_int(cap), C.byref(nexp))
    if n < 0:
        raise ValueError({-1: "bad arguments", -2: "start or goal pose collides"}[n])
    if n == 0:
        return None
    return path[:n].copy(), dr[:n].copy(), nexp.value


REFERENCE_OPTS = dict(xy_res=0.3, yaw_res_deg=5.0, motion_step=0.1, steer_samples=5, max_steer=0.6, wheelbase=2.7, switch_cost=10.0, reverse_cost=0.0,
                      steer_change_cost=10.0, steer_cost=0.0, h_cost=1.0, vehicle_radius=1.0, max_expansions=2000000)      # hybrid_a_star.jl:44-68


def reference_hybrid_astar(start, goal, ox, oy, **kw):
    """REFERENCE mode: the reference's Hybrid A* (hybrid_a_star.jl: calc_hybrid_astar_path) on its point-cloud obstacles (scenarios.reference_obstacle_points).
    Returns (path (K,3): rx, ry, ryaw at 0.1 m spacing, expansions) or None."""
    o = dict(REFERENCE_OPTS); o.update(kw)
    opts = np.array([o[k] for k in REFERENCE_OPTS], float)
    ox = np.ascontiguousarray(ox, float); oy = np.ascontiguousarray(oy, float)
    s = np.ascontiguousarray(start, float)[:3].copy(); g = np.ascontiguousarray(goal, float)[:3].copy()
    cap = 20000; path = np.zeros((cap, 3)); nexp = C.c_int(0)
    n = _load().obca_plan_reference_hybrid_astar(s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_int(len(ox)), ox.ctypes.data_as(_D), oy.ctypes.data_as(_D),
                                                 opts.ctypes.data_as(_D), path.ctypes.data_as(_D), C.c_int(cap), C.byref(nexp))
    if n < 0:
        raise ValueError("bad arguments")
    return None if n == 0 else (path[:n].copy(), nexp.value)


def reference_warm_start(sc, x0, xF, sampleN=3, motion_step=0.1, a_max=0.3, **kw):
    """main.jl:216-252 as it stands: the reference search on the scenario's point cloud, the speed profile from the path differences (Ts / sampleN per 0.1 m step),
    veloSmooth at 0.3 m/s^2, the steering angle from the yaw differences, every sampleN-th sample.  The horizon follows from the path length: N = samples - 1.
    Returns (N, Ts, xWS (N+1,4), uWS (N,2), path) or None."""
    ox, oy = S.reference_obstacle_points(sc)
    r = reference_hybrid_astar(np.asarray(x0, float)[:3], np.asarray(xF, float)[:3], ox, oy, **kw)
    if r is None:
        return None
    P = r[0]; rx, ry, ryaw = P[:, 0], P[:, 1], P[:, 2]
    Ts = sc["Ts"]; dts = Ts / sampleN
    rv = np.zeros(len(rx)); rv[:-1] = (np.diff(rx) * np.cos(ryaw[:-1]) + np.diff(ry) * np.sin(ryaw[:-1])) / dts      # main.jl:222-229
    v, a = velo_smooth(rv, a_max, dts)                                                                             # :230-231
    delta = np.arctan(np.diff(ryaw) * S.L_WHEELBASE / motion_step * np.sign(v[:-1]))                                # :233
    sl = slice(None, None, sampleN)
    xWS = np.stack([rx[sl], ry[sl], ryaw[sl], v[sl]], axis=1)                                                       # :237-248
    uWS = np.stack([delta[sl], a[sl]], axis=1)
    N = xWS.shape[0] - 1
    return N, Ts, xWS, uWS[:N], P


def reference_quad_obstacle_points():
    """the obstacle point lists mainQuadcopter.jl:59-105 builds for its A* call (room scaled by 10): the first wall x 20..25, y 0..105, z 6..55 and the second wall x 70..75
    with its window y 40..50, z 20..30 left open (left, right, top and bottom pieces)"""
    P = []
    r = lambda a, b: np.arange(a, b + 1, dtype=float)
    def block(xs, ys, zs):
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij"); P.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
    block(r(20, 25), r(0, 105), r(6, 55))
    for xx in r(70, 75):      # (the reference's loop order: per x the four pieces; the order of the points does not enter the search)
        block([xx], r(0, 40), r(0, 55)); block([xx], r(50, 105), r(0, 55)); block([xx], r(40, 50), r(30, 55)); block([xx], r(40, 50), r(0, 20))
    Q = np.concatenate(P, axis=0)
    return Q[:, 0].copy(), Q[:, 1].copy(), Q[:, 2].copy()


def reference_astar3d(start, goal, ox, oy, oz, room_min=(0.0, 0.0, 0.0), room_max=(105.0, 105.0, 55.0), reso=1.0):
    """REFERENCE mode of the quadcopter's search: QuadcopterNavigation/a_star_3D.jl restated (obca_plan_reference_astar3d).  Returns (way-points (K, 3), expansions, cost) or
    None; K = the reference's length(rx) (its path repeats the goal cell once: get_final_path, a_star_3D.jl:243-250)."""
    s = np.ascontiguousarray(start, float)[:3].copy(); g = np.ascontiguousarray(goal, float)[:3].copy()
    ox = np.ascontiguousarray(ox, float); oy = np.ascontiguousarray(oy, float); oz = np.ascontiguousarray(oz, float)
    lo = np.ascontiguousarray(room_min, float); hi = np.ascontiguousarray(room_max, float)
    cap = 8192; path = np.zeros((cap, 3)); nexp = C.c_int(0); cost = C.c_double(0)
    n = _load().obca_plan_reference_astar3d(s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_int(len(ox)), ox.ctypes.data_as(_D), oy.ctypes.data_as(_D), oz.ctypes.data_as(_D),
                                            lo.ctypes.data_as(_D), hi.ctypes.data_as(_D), C.c_double(reso), path.ctypes.data_as(_D), C.c_int(cap), C.byref(nexp), C.byref(cost))
    if n < 0:
        raise ValueError("bad arguments")
    return None if n == 0 else (path[:n].copy(), nexp.value, cost.value)


def reference_quad_warm_start(x0=None, xF=None, Ts=0.25):
    """mainQuadcopter.jl:107-138 as it stands: the A* call on the room scaled by 10 (start / goal = 10 x the positions, grid 1.0), N_as = length(rx) - 1,
    Ts_as = round(Ts 80 / N_as, 2) (:131), warm start = the way-points / 10 with every other state 0 (:134-136), inputs 0.5 (ignored by the solve), timeWS = 1.
    Returns (N_as, Ts_as, xWS (N_as + 1, 12), uWS (N_as, 4), way-points in metres) or None."""
    x0 = S.QUAD_X0 if x0 is None else np.asarray(x0, float); xF = S.QUAD_XF if xF is None else np.asarray(xF, float)
    ox, oy, oz = reference_quad_obstacle_points()
    r = reference_astar3d(10.0 * x0[:3], 10.0 * xF[:3], ox, oy, oz)
    if r is None:
        return None
    wp = r[0] / 10.0; N = len(wp) - 1
    Ts_as = np.round(Ts * 80 / N * 100) / 100
    xWS = np.zeros((N + 1, 12)); xWS[:, :3] = wp
    return N, float(Ts_as), xWS, 0.5 * np.ones((N, 4)), wp


def reeds_shepp(start, goal, R, step=0.2):
    """shortest Reeds-Shepp path between two poses (x, y, yaw) for turning radius R: returns (path (K,3), dir (K,), word, segment lengths, total)"""
    s = np.ascontiguousarray(start, float)[:3].copy(); g = np.ascontiguousarray(goal, float)[:3].copy()
    cap = 20000; path = np.zeros((cap, 3)); dr = np.zeros(cap, np.int32); word = C.create_string_buffer(8); seg = np.zeros(5); tot = C.c_double(0)
    n = _load().obca_plan_reeds_shepp(s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_double(R), C.c_double(step), path.ctypes.data_as(_D),
                                      dr.ctypes.data_as(_I), C.c_int(cap), word, seg.ctypes.data_as(_D), C.byref(tot))
    if n < 0:
        raise ValueError("bad arguments")
    w = word.value.decode()
    return path[:n].copy(), dr[:n].copy(), w, seg[:len(w)].copy(), tot.value


def collides(pose, vOb, A, b, ego=S.EGO, XYbounds=S.XYBOUNDS, margin=0.0):
    vOb = np.ascontiguousarray(vOb, np.int32); A = np.ascontiguousarray(A, float); b = np.ascontiguousarray(b, float)
    e = np.ascontiguousarray(ego, float); xy = np.ascontiguousarray(XYbounds, float)
    return bool(_load().obca_plan_collides(C.c_double(pose[0]), C.c_double(pose[1]), C.c_double(pose[2]), C.c_int(len(vOb)), vOb.ctypes.data_as(_I),
                                           A.ctypes.data_as(_D), b.ctypes.data_as(_D), e.ctypes.data_as(_D), xy.ctypes.data_as(_D), C.c_double(margin)))


def path_to_warm_start(path, dr, N, xF=None, v_nom=0.5, L=S.L_WHEELBASE, smooth=False):
    """resample the planner path uniformly in arc length to N+1 stages (main.jl:237-252 down-samples the smoothed profile instead) and
    derive the state / input warm start: speed +-v_nom (0 at both ends and at direction switches), steering from the path curvature.
    xF (optional) replaces the last pose so that the warm start ends on the NLP's terminal state."""
    P = np.asarray(path, float).copy(); dr = np.asarray(dr, float)
    P[:, 2] = np.unwrap(P[:, 2])
    if xF is not None:
        P[-1, :2] = xF[:2]; P[-1, 2] = P[-2, 2] + ((xF[2] - P[-2, 2] + np.pi) % (2 * np.pi) - np.pi)
    seg = np.hypot(np.diff(P[:, 0]), np.diff(P[:, 1])); cum = np.concatenate([[0], np.cumsum(seg)]); tot = cum[-1]
    ss = np.linspace(0, tot, N + 1)
    X = np.interp(ss, cum, P[:, 0]); Y = np.interp(ss, cum, P[:, 1]); yaw = np.interp(ss, cum, P[:, 2])
    idx = np.clip(np.searchsorted(cum, ss, side="left"), 1, len(cum) - 1)
    d = dr[idx]; d[0] = dr[min(1, len(dr) - 1)]
    Ts = tot / (N * v_nom)
    v = d * v_nom; v[0] = 0; v[-1] = 0
    for i in range(1, N):
        if d[i] != d[i + 1]:
            v[i] = 0
    if smooth:      # the reference's pipeline (main.jl:222-231): raw speed of every interval, then veloSmooth with 0.3 m/s^2
        rv = np.concatenate([d[1:] * v_nom, [0.0]])
        v, _ = velo_smooth(rv, 0.3, Ts)
    a = np.clip(np.diff(v) / Ts, -0.4, 0.4)
    dpsi = np.diff(yaw); dsv = np.maximum(np.diff(ss), 1e-9) * np.where(d[1:] == 0, 1, d[1:])
    delta = np.clip(np.arctan(L * dpsi / dsv), -0.6, 0.6)
    return Ts, np.stack([X, Y, yaw, v], 1), np.stack([delta, a], 1)


def velo_smooth(v, amax, Ts):
    """Velocity smoother of the warm-start pipeline (behaviour of AutonomousParking/veloSmooth.jl:29-109, used at main.jl:230-231): the planner's
    speed profile is piecewise constant at +-v_nom (0 at the end); every jump of the profile is replaced by a ramp of slope amax --
    0 <-> +-v_nom jumps by a ramp of round(v_nom / amax / Ts) samples that ends (starts) at the jump, +-v_nom <-> -+v_nom jumps by a ramp of
    twice that length centred on it -- and at every sample the candidate with the smallest speed is kept.  Returns (v_smooth (n,), a (n-1,))
    with a = diff(v_smooth) / Ts.  v[0] sets the nominal speed (veloSmooth.jl:44-47), so the profile must start in motion."""
    v = np.asarray(v, float).ravel(); n = len(v); v1 = abs(v[0])
    pad = 19                                                    # veloSmooth.jl:31-41: 19 leading / 21 trailing zeros around the profile
    vex = np.zeros(n + 41); vex[pad + 1:pad + 1 + n] = v        # 1-based like the reference: vex[k], k = 1 .. n+40
    bar = np.zeros((4, n + 41)); bar[:, pad + 1:pad + 1 + n] = v
    acc = int(round(v1 / amax / Ts))
    dv = np.zeros(n + 41); dv[1:n + 40] = np.diff(vex[1:n + 41])  # dv[k] = vex[k+1] - vex[k]
    ks = np.arange(1, n + 40)
    cut1, cut2 = 0.25 * v1, 1.25 * v1
    up1 = [k for k in ks if cut1 < dv[k] < cut2]; up2 = [k for k in ks if dv[k] > cut2]
    dn1 = [k for k in ks if -cut2 < dv[k] < -cut1]; dn2 = [k for k in ks if dv[k] < -cut2]
    if up1 and up1[0] == pad: up1[0] += 1                       # a jump on the very first sample starts its ramp on it (:55-60)
    if dn1 and dn1[0] == pad: dn1[0] += 1
    ramp = np.linspace(0.0, v1, acc + 1)

    def put(row, lo, vals):                                     # bar[row, lo : lo+len-1] = vals, clipped to the padded array
        for i, x in enumerate(vals):
            if 1 <= lo + i <= n + 40: bar[row, lo + i] = x
    for k in up1:                                               # rise by v_nom: start from rest, or come to rest from reverse (:63-69)
        if vex[k] > cut1 or vex[k + 1] > cut1: put(0, k, ramp)
        elif vex[k] < -cut1 or vex[k + 1] < -cut1: put(0, k - acc + 1, ramp - v1)
    for k in dn1:                                               # fall by v_nom: come to rest, or start in reverse (:71-77)
        if vex[k] > cut1 or vex[k + 1] > cut1: put(1, k - acc + 1, v1 - ramp)
        elif vex[k] < -cut1 or vex[k + 1] < -cut1: put(1, k, -ramp)
    full = np.linspace(-v1, v1, 2 * acc + 1)
    for k in up2: put(2, k - acc, full)                         # reverse -> forward (:79-81)
    for k in dn2: put(3, k - acc, -full)                        # forward -> reverse (:83-85)
    out = np.zeros(n)
    for i in range(pad + 1, pad + 1 + n):                       # (:87-104): a candidate of the wrong sign falls back to the raw profile
        c = np.where(bar[:, i] == 0, 0.0, np.where(np.sign(vex[i]) != np.sign(bar[:, i]), vex[i], bar[:, i]))
        out[i - pad - 1] = c.min() if vex[i] > 0 else c.max()
    return out, np.diff(out) / Ts


# search settings per scenario: the 6 m bay of the parallel scenario leaves 0.65 m at either end of the car, which needs a fine grid;
# nominal speeds follow the reference's sampling times (0.6 s and 0.9 s per 0.3 m of path, main.jl:46-50,66 and the scenario tables)
SCENARIO_OPTS = {"backwards": (dict(), 0.5),
                 "parallel": (dict(step=0.2, xy_res=0.1, yaw_res_deg=3.0, margin=0.02, max_expansions=2000000, rs_heuristic=1, h_weight=1.5), 0.25)}
# (round 5: the Reeds-Shepp length as a second heuristic -- hybrid_a_star.jl:58 has the switch -- and the heuristic weighted 1.5 -- hybrid_a_star.jl:64 H_COST -- halve the
#  planning time of a parallel-parking start (64 k -> 21 k expansions) and leave the NLP as it was: 512 instances on the oracle, throughput / reference options: 36.6 / 38.6
#  iterations against 36.4 / 38.1, longest solve 200 / 175 passes against 193 / 183.  Heavier weights plan faster still -- weight 2.5 with a switch cost of 3: 4.1 x less time --
#  but their paths carry more direction switches and the hardest NLP of a batch then takes 350-400 passes instead of ~190, which a batched solve waits for: not the default;
#  `warm_start_many(..., h_weight=2.5, switch_cost=3.0)` for a pipeline that is bound by the planner.)


def warm_start(sc, x0, xF, N, smooth=False, **kw):
    """Hybrid A* warm start of one instance of a scenario table (S.BACKWARDS / S.PARALLEL): returns (Ts, xWS (N+1,4), uWS (N,2)) or None.
    smooth=True runs the planner's speed profile through velo_smooth as main.jl:222-231 does."""
    A, b, vrows = S.scenario_hrep(sc)
    o, v_nom = SCENARIO_OPTS.get(sc["name"], (dict(), 0.5))
    o = dict(o); o.update(kw)
    try:
        r = hybrid_astar(np.asarray(x0, float)[:3], np.asarray(xF, float)[:3], vrows, A, b, **o)
    except ValueError:          # start (or goal) pose in collision: no collision-free path exists
        return None
    if r is None:
        return None
    return path_to_warm_start(r[0], r[1], N, xF, v_nom=v_nom, smooth=smooth)


def effective_cpus():
    """host threads this process may really use: the affinity mask, cut by the cgroup CPU quota if there is one (a container may see 256 CPUs and be allowed a dozen)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def hybrid_astar_many(starts, goals, vOb, A, b, ego=S.EGO, L=S.L_WHEELBASE, XYbounds=S.XYBOUNDS, threads=0, cap=1024, **kw):
    """B searches in one obstacle field on the host threads of the library (obca_plan_hybrid_astar_batch): returns a list of (path, dir, expansions) / None
    (no path, or the start / goal pose collides)."""
    o = dict(DEFAULT_OPTS); o.update(kw)
    opts = np.array([o[k] for k in DEFAULT_OPTS], float)
    vOb = np.ascontiguousarray(vOb, np.int32); A = np.ascontiguousarray(A, float); b = np.ascontiguousarray(b, float)
    s = np.ascontiguousarray(np.asarray(starts, float)[:, :3]); g = np.ascontiguousarray(np.asarray(goals, float)[:, :3]); B = len(s)
    e = np.ascontiguousarray(ego, float); xy = np.ascontiguousarray(XYbounds, float)
    paths = np.zeros((B, cap, 3)); dirs = np.zeros((B, cap), np.int32); cnt = np.zeros(B, np.int32); nexp = np.zeros(B, np.int32)
    rc = _load().obca_plan_hybrid_astar_batch2(C.c_int(B), s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_int(len(vOb)), vOb.ctypes.data_as(_I), A.ctypes.data_as(_D),
                                               b.ctypes.data_as(_D), e.ctypes.data_as(_D), C.c_double(L), xy.ctypes.data_as(_D), opts.ctypes.data_as(_D), C.c_int(len(opts)),
                                               paths.ctypes.data_as(_D), dirs.ctypes.data_as(_I), C.c_int(cap), cnt.ctypes.data_as(_I), nexp.ctypes.data_as(_I), C.c_int(int(threads or 0)))
    if rc != 0:
        raise ValueError("bad arguments")
    out = []
    for i in range(B):
        if cnt[i] == -1:          # a path longer than cap nodes (or bad arguments, which the single call reports)
            try:
                out.append(hybrid_astar(s[i], g[i], vOb, A, b, ego, L, XYbounds, **kw))
            except ValueError:
                out.append(None)
        else:
            out.append((paths[i, :cnt[i]].copy(), dirs[i, :cnt[i]].copy(), int(nexp[i])) if cnt[i] > 0 else None)
    return out


def warm_start_many(sc, x0, xF, N, workers=None, smooth=False, **kw):
    """warm starts of a batch: the searches run on the host threads of the planner library (one call, no worker processes: safe next to a live HIP runtime, so
    every rank of a multi-GPU job plans its own slice after its device is up); resampling to the horizon is numpy per instance.  workers = threads (default: all)."""
    A, b, vrows = S.scenario_hrep(sc)
    o, v_nom = SCENARIO_OPTS.get(sc["name"], (dict(), 0.5))
    o = dict(o); o.update(kw)
    x0 = np.asarray(x0, float); xF = np.asarray(xF, float)
    if len(x0) == 0:
        return []
    res = hybrid_astar_many(x0[:, :3], xF[:, :3], vrows, A, b, threads=workers or effective_cpus(), **o)
    return [None if r is None else path_to_warm_start(r[0], r[1], N, xF[i], v_nom=v_nom, smooth=smooth) for i, r in enumerate(res)]


# ---------------------------------------------------------------- quadcopter: 3-D grid A* (a_star_3D.jl, mainQuadcopter.jl:108-138)
QUAD_ROOM = (10.0, 10.0, 5.0)


def astar3d(start, goal, boxes=None, clear=0.4, room=QUAD_ROOM, res=0.25):
    """way-points (K,3) from start to goal around the boxes inflated by `clear`, or None.  boxes: (nBox,6) [max; -min] (default: the scenario's)."""
    boxes = np.ascontiguousarray(S.QUAD_OB if boxes is None else boxes, float).reshape(-1, 6)
    s = np.ascontiguousarray(start, float)[:3].copy(); g = np.ascontiguousarray(goal, float)[:3].copy(); rm = np.ascontiguousarray(room, float)
    cap = 4096; path = np.zeros((cap, 3)); nexp = C.c_int(0)
    n = _load().obca_plan_astar3d(s.ctypes.data_as(_D), g.ctypes.data_as(_D), C.c_int(len(boxes)), boxes.ctypes.data_as(_D), C.c_double(clear),
                                  rm.ctypes.data_as(_D), C.c_double(res), path.ctypes.data_as(_D), C.c_int(cap), C.byref(nexp))
    if n == -1:
        raise ValueError("bad arguments")
    return None if n <= 0 else path[:n].copy()


def quad_warm_start(x0, xF, N, boxes=None, clear=0.4, res=0.25):
    """A* way-points resampled uniformly in arc length to N+1 stages, all other states 0 (mainQuadcopter.jl:134-138); None if no path."""
    wp = astar3d(x0[:3], xF[:3], boxes, clear, res=res)
    if wp is None:
        return None
    return S.quad_warm_start(x0, xF, N, via=[tuple(p) for p in wp[1:-1]])
