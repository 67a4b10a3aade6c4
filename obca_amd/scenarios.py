"""
Host-side scenario tables and synthetic warm starts for the OBCA parking path.

Restates, as plain numpy, the input construction either side of the hot path:
  * obst_hrep            <- /root/reference/AutonomousParking/obstHrep.jl:31-102
  * BACKWARDS / PARALLEL <- /root/reference/AutonomousParking/main.jl:43-73,99-108,151-162,210-213
  * warm starts: the reference takes them from Hybrid A* (main.jl:216-248), which is
    out of scope.  Here a line/arc/line Reeds-Shepp-like
    primitive plays that role; it is resampled to a fixed horizon N and the nominal
    sampling time is chosen so that the path speed is 0.5 m/s, the reference's
    nominal value (0.1 m x 3 per 0.6 s, main.jl:46-50,66).
Nothing here touches the GPU; it only produces the (x0, xF, H-rep, rx, ry, ryaw, xWS, uWS)
arrays the reference entry points take.
"""
import numpy as np

L_WHEELBASE = 2.7                      # main.jl:63
EGO = np.array([3.7, 1.0, 1.0, 1.0])   # main.jl:73
XYBOUNDS = np.array([-15.0, 15.0, 1.0, 10.0])  # main.jl:210


def obst_hrep(nOb, vOb, lOb):
    """vertices (clock-wise, vOb[i] per obstacle) -> stacked half-space rows A p <= b."""
    vOb = [int(v) for v in np.asarray(vOb).ravel()]
    assert nOb == len(lOb)
    rows_A, rows_b = [], []
    for i in range(nOb):
        for j in range(vOb[i] - 1):
            v1 = np.asarray(lOb[i][j], float); v2 = np.asarray(lOb[i][j + 1], float)
            if v1[0] == v2[0]:                       # vertical edge
                if v2[1] < v1[1]:
                    a, bb = [1.0, 0.0], v1[0]
                else:
                    a, bb = [-1.0, 0.0], -v1[0]
            elif v1[1] == v2[1]:                     # horizontal edge
                if v1[0] < v2[0]:
                    a, bb = [0.0, 1.0], v1[1]
                else:
                    a, bb = [0.0, -1.0], -v1[1]
            else:                                    # general edge y = s x + c
                s, c = np.linalg.solve(np.array([[v1[0], 1.0], [v2[0], 1.0]]), np.array([v1[1], v2[1]]))
                if v1[0] < v2[0]:
                    a, bb = [-s, 1.0], c
                else:
                    a, bb = [s, -1.0], -c
            rows_A.append(a); rows_b.append(bb)
    return np.array(rows_A, float), np.array(rows_b, float)


BACKWARDS = dict(
    name="backwards", Ts=0.6, Ts_fix=0.55, nOb=3, vOb=[3, 3, 2],
    lOb=[[[-20, 5], [-1.3, 5], [-1.3, -5]], [[1.3, -5], [1.3, 5], [20, 5]], [[20, 11], [-20, 11]]],
    xF=np.array([0.0, 1.3, np.pi / 2, 0.0]), x0=np.array([-6.0, 9.5, 0.0, 0.0]))

PARALLEL = dict(
    name="parallel", Ts=0.9, Ts_fix=0.95, nOb=4, vOb=[3, 3, 2, 2],
    lOb=[[[-20, 5], [-3.0, 5], [-3.0, 0]], [[3.0, 0], [3.0, 5], [20, 5]], [[-3, 2.5], [3, 2.5]],
         [[20, 11], [-20, 11]]],
    xF=np.array([-L_WHEELBASE / 2, 4.0, 0.0, 0.0]), x0=np.array([-6.0, 9.5, 0.0, 0.0]))


def reference_obstacle_points(sc):
    """the obstacle POINT CLOUD main.jl hands to its Hybrid A* (main.jl:111-133 backwards, :172-198 parallel): walls sampled every 0.1 m along x and every 1 m
    along y, the far side of the aisle every 1 m.  Returns (ox, oy)."""
    fr = lambda a, b: np.round(np.arange(int(round(a * 10)), int(round(b * 10)) + 1) / 10.0, 10)      # Julia's a:0.1:b
    ir = lambda a, b: np.arange(a, b + 1, dtype=float)                                                  # a:b
    P = []
    if sc["name"] == "backwards":
        P += [(x, 5.0) for x in fr(-12, -1.3)] + [(-1.3, y) for y in ir(-2, 5)]          # obstacle 1
        P += [(1.3, y) for y in ir(-2, 5)] + [(x, 5.0) for x in fr(1.3, 12)]             # obstacle 2
        P += [(x, 11.0) for x in ir(-12, 12)]                                            # obstacle 3
    else:
        P += [(x, 5.0) for x in fr(-12, -3.0)] + [(-3.0, y) for y in ir(-2, 5)]          # obstacle 1
        P += [(x, 2.5) for x in ir(-3, 3)]                                               # obstacle 2
        P += [(3.0, y) for y in ir(-2, 5)] + [(x, 5.0) for x in fr(3, 12)]               # obstacle 3
        P += [(x, 11.5) for x in ir(-12, 12)]                                            # obstacle 4
    P = np.array(P, float)
    return P[:, 0].copy(), P[:, 1].copy()


def scenario_hrep(sc):
    A, b = obst_hrep(sc["nOb"], sc["vOb"], sc["lOb"])
    vrows = np.asarray(sc["vOb"]) - 1          # vObMPC = vOb-1  (main.jl:101)
    return A, b, vrows


# --------------------------------------------------------------------------- paths
def _sample_path(segs, n_pts):
    """segs: list of ('line', p0(2), heading, length, direction) / ('arc', center(2), R, th0, th1, direction, turn)
    direction = +1 forward / -1 reverse.  Returns X,Y,yaw,dir,curvature sampled uniformly in arc length."""
    lens = []
    for s in segs:
        lens.append(s[3] if s[0] == "line" else abs(s[4] - s[3]) * s[2])
    tot = float(sum(lens))
    ss = np.linspace(0.0, tot, n_pts)
    X = np.zeros(n_pts); Y = np.zeros(n_pts); yaw = np.zeros(n_pts); dr = np.zeros(n_pts); kap = np.zeros(n_pts)
    cum = np.concatenate([[0.0], np.cumsum(lens)])
    for i, s_ in enumerate(ss):
        k = min(np.searchsorted(cum, s_, side="right") - 1, len(segs) - 1)
        k = max(k, 0)
        while lens[k] == 0 and k < len(segs) - 1:
            k += 1
        loc = s_ - cum[k]
        sg = segs[k]
        if sg[0] == "line":
            _, p0, hd, ln, d = sg
            X[i] = p0[0] + d * loc * np.cos(hd); Y[i] = p0[1] + d * loc * np.sin(hd)
            yaw[i] = hd; dr[i] = d; kap[i] = 0.0
        else:
            _, cen, R, th0, th1, d, side = sg
            th = th0 + (th1 - th0) * (loc / lens[k] if lens[k] > 0 else 0.0)
            # side=+1: left-turn circle  p = c + R(sin th, -cos th); side=-1: right-turn circle p = c + R(-sin th, cos th)
            X[i] = cen[0] + side * R * np.sin(th); Y[i] = cen[1] - side * R * np.cos(th)
            yaw[i] = th; dr[i] = d; kap[i] = side / R
    return X, Y, yaw, dr, kap, tot


def _finish(X, Y, yaw, dr, kap, tot, N, v_nom=0.5):
    Ts = tot / (N * v_nom)
    # speed: +-v_nom with zero at both ends; linear ramps limited by 0.4 m/s^2 handled by the NLP itself
    v = dr * v_nom
    v[0] = 0.0; v[-1] = 0.0
    for i in range(1, N):
        if dr[i] != dr[i - 1]:
            v[i] = 0.0
    a = np.diff(v) / Ts
    a = np.clip(a, -0.4, 0.4)
    delta = np.arctan(kap[:-1] * L_WHEELBASE)          # tan(delta)/L = curvature
    delta = np.clip(delta, -0.6, 0.6)
    xWS = np.stack([X, Y, yaw, v], 1)
    uWS = np.stack([delta, a], 1)
    return Ts, xWS, uWS


def _lane_change(p0, dy, R, direction):
    """two-arc lane change by the lateral offset dy (sign = towards +y / -y) starting at p0 with heading 0;
    direction=+1 forward (moves +x), -1 reverse (moves -x).  Returns the segments and the end point."""
    sg = 1.0 if dy >= 0 else -1.0
    th = np.arccos(max(-1.0, 1.0 - abs(dy) / (2 * R)))
    dx = 2 * R * np.sin(th)
    x0_, y0_ = p0
    tm = sg * direction * th          # heading at the inflection point
    segs = [("arc", (x0_, y0_ + sg * R), R, 0.0, tm, direction, sg),
            ("arc", (x0_ + direction * dx, y0_ + dy - sg * R), R, tm, 0.0, direction, -sg)]
    return segs, (x0_ + direction * dx, y0_ + dy)


def warm_start_backwards(x0, xF, N, R=4.5, lane=8.4):
    """[lane change to y = lane] -> line along the lane -> reverse quarter arc -> reverse line into the slot.
    The quarter arc of radius R clears the slot corners (needs lane - R > 3.6) and the top wall (needs lane - R < 4.25) only when it
    starts from this lane, so starts elsewhere in the aisle first shift to it (forward if there is room ahead, else in reverse)."""
    X0, Y0 = float(x0[0]), float(x0[1])
    xg, yg = float(xF[0]), float(xF[1])
    Ya = lane
    xa = xg + R                       # arc starts at (xg+R, Ya) heading 0, ends at (xg, Ya-R) heading pi/2
    ya = Ya - R
    segs = []
    dy = Ya - Y0
    px, py = X0, Y0
    if abs(dy) > 0.02:
        th = np.arccos(max(-1.0, 1.0 - abs(dy) / (2 * R))); dxn = 2 * R * np.sin(th)
        if X0 >= xa + dxn or X0 + dxn > 14.0:   # far right of the arc start (or no room ahead): shift while reversing
            sg, (px, py) = _lane_change((px, py), dy, R, -1.0)
            segs += sg
        else:                         # shift driving forward (after a straight run if there is room)
            if X0 < xa - dxn:
                segs.append(("line", (px, py), 0.0, xa - dxn - X0, 1.0)); px = xa - dxn
            sg, (px, py) = _lane_change((px, py), dy, R, +1.0)
            segs += sg
    else:
        Ya = Y0; ya = Ya - R
    d1 = 1.0 if px < xa else -1.0
    segs.append(("line", (px, py), 0.0, abs(xa - px), d1))
    segs.append(("arc", (xa, ya), R, 0.0, np.pi / 2, -1.0, -1.0))
    segs.append(("line", (xg, ya), np.pi / 2, max(ya - yg, 0.0), -1.0))
    X, Y, yaw, dr, kap, tot = _sample_path(segs, N + 1)
    Ts, xWS, uWS = _finish(X, Y, yaw, dr, kap, tot, N)
    return Ts, xWS, uWS


def warm_start_parallel(x0, xF, N, R=4.5):
    """line along the lane -> reverse S-curve (right-turn arc then left-turn arc) into the bay."""
    X0, Y0 = float(x0[0]), float(x0[1])
    xg, yg = float(xF[0]), float(xF[1])
    dy = Y0 - yg
    th = np.arccos(max(-1.0, 1.0 - dy / (2 * R)))
    dx = 2 * R * np.sin(th)
    xs = xg + dx
    d1 = 1.0 if X0 < xs else -1.0
    segs = [("line", (X0, Y0), 0.0, abs(xs - X0), d1)]
    # reversing from heading 0 with heading increasing: right-turn circle (side=-1), centre below the lane
    segs.append(("arc", (xs, Y0 - R), R, 0.0, th, -1.0, -1.0))
    # then heading decreasing back to 0 while reversing: left-turn circle (side=+1), centre above the goal
    segs.append(("arc", (xg, yg + R), R, th, 0.0, -1.0, +1.0))
    X, Y, yaw, dr, kap, tot = _sample_path(segs, N + 1)
    Ts, xWS, uWS = _finish(X, Y, yaw, dr, kap, tot, N)
    return Ts, xWS, uWS


def _draw_start(rng):
    return np.array([rng.uniform(-10, 10), rng.uniform(6.5, 9.5), rng.uniform(-0.2, 0.2), 0.0])


def sample_poses(sc, B, rng, goal_jitter=False):
    """start / goal poses of a synthetic batch per SURVEY.md section 8d: X0~U[-10,10], Y0~U[6.5,9.5] (main.jl:165-168), psi0~U[-0.2,0.2], v0=0; goal = the scenario's,
    X_F~U[-1.85,-0.85] with goal_jitter (BASELINE config 3)"""
    x0 = np.zeros((B, 4)); xF = np.zeros((B, 4))
    for i in range(B):
        x0[i] = _draw_start(rng)
        xF[i] = sc["xF"]
        if goal_jitter:
            xF[i, 0] = rng.uniform(-1.85, -0.85)
    return x0, xF


def plan_batch(sc, x0, xF, N, rng, planner=None, workers=None, smooth=False):
    """warm starts (the step before the path, host side) for given poses: geometric line/arc primitives for the backwards scenario, Hybrid A* (obca_amd/planner.py) for
    the parallel one, whose 6 m bay needs a multi-manoeuvre path; planner=True/False forces the choice.  A start pose for which the planner finds no path is re-drawn
    from `rng` (x0 is updated in place).  Returns Ts (B,), xWS (B,N+1,4), uWS (B,N,2)."""
    B = len(x0)
    use_planner = (sc["name"] != "backwards") if planner is None else bool(planner)
    Ts = np.zeros(B); xWS = np.zeros((B, N + 1, 4)); uWS = np.zeros((B, N, 2))
    if not use_planner:
        ws = warm_start_backwards if sc["name"] == "backwards" else warm_start_parallel
        for i in range(B):
            Ts[i], xWS[i], uWS[i] = ws(x0[i], xF[i], N)
    else:
        from . import planner as PL
        todo = list(range(B))
        while todo:
            res = PL.warm_start_many(sc, x0[todo], xF[todo], N, workers=workers, smooth=smooth)
            nxt = []
            for i, r in zip(todo, res):
                if r is None:
                    x0[i] = _draw_start(rng); nxt.append(i)
                else:
                    Ts[i], xWS[i], uWS[i] = r
            todo = nxt
    return Ts, xWS, uWS


def make_batch(sc, B, N=80, seed=20260925, goal_jitter=False, planner=None, workers=None, smooth=False):
    """Synthetic batch: sample_poses + plan_batch from one random stream (seed)."""
    rng = np.random.default_rng(seed)
    A, b, vrows = scenario_hrep(sc)
    x0, xF = sample_poses(sc, B, rng, goal_jitter)
    Ts, xWS, uWS = plan_batch(sc, x0, xF, N, rng, planner=planner, workers=workers, smooth=smooth)
    return dict(x0=x0, xF=xF, Ts=Ts, xWS=xWS, uWS=uWS, A=A, b=b, vOb=vrows, N=N, L=L_WHEELBASE,
                ego=EGO.copy(), XYbounds=XYBOUNDS.copy())


# ---------------------------------------------------------------- quadcopter scenario (mainQuadcopter.jl:36-54, 131-138)
QUAD_X0 = np.array([1, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])
QUAD_XF = np.array([9, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])
QUAD_R = 0.25
# the five boxes [xmax,ymax,zmax,-xmin,-ymin,-zmin] after the in-place clamping to the room done by the plot call that precedes
# the signed-distance solve in the reference's main: wall with a window + the room around it
QUAD_OB = np.array([[2.5, 10, 5, -2, 0, -0.6], [7.5, 10, 5, -7, -5, 0], [7.5, 4, 5, -7, 0, 0], [7.5, 5, 2, -7, -4, 0], [7.5, 5, 5, -7, -4, -3]], float)
QUAD_VIA = [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)]   # under the first wall, through the window of the second


def quad_sample_time(N):
    return round(0.25 * 80 / N * 100) / 100          # mainQuadcopter.jl:131


def quad_warm_start(x0, xF, N, via=QUAD_VIA):
    """positions along straight segments x0 -> via points -> xF at constant spacing, all other states 0 (the reference uses a 3-D
    A* path, mainQuadcopter.jl:134-138; out of scope here)."""
    pts = [np.asarray(x0, float)[:3]] + [np.asarray(p, float) for p in (via or [])] + [np.asarray(xF, float)[:3]]
    seg = np.array([np.linalg.norm(pts[i + 1] - pts[i]) for i in range(len(pts) - 1)]); cum = np.concatenate([[0], np.cumsum(seg)])
    xWS = np.zeros((N + 1, 12))
    for k, s in enumerate(np.linspace(0, cum[-1], N + 1)):
        i = min(np.searchsorted(cum, s, side="right") - 1, len(seg) - 1)
        a = (s - cum[i]) / seg[i] if seg[i] > 0 else 0.0
        xWS[k, :3] = pts[i] + a * (pts[i + 1] - pts[i])
    return xWS


def _draw_quad_endpoints(rng):
    return [rng.uniform(0.5, 1.6), rng.uniform(0.5, 9.5), rng.uniform(0.5, 4.5)], [rng.uniform(8.0, 9.5), rng.uniform(0.5, 9.5), rng.uniform(0.5, 4.5)]


def plan_quad_batch(x0, xF, N, rng, first_is_fixed=True):
    """3-D grid A* warm starts (obca_amd/planner.py, the reference's a_star_3D.jl step) for given end points; an instance without a path gets new end points from `rng`
    (x0 / xF updated in place).  Returns xWS (B,N+1,12)."""
    from . import planner as PL
    B = len(x0); xWS = np.zeros((B, N + 1, 12))
    for i in range(B):
        while True:
            w = PL.quad_warm_start(x0[i], xF[i], N)
            if w is not None:
                xWS[i] = w; break
            if i == 0 and first_is_fixed:
                raise RuntimeError("no path for the shipped quadcopter scenario")
            x0[i, :3], xF[i, :3] = _draw_quad_endpoints(rng)
    return xWS


def make_quad_batch(B, N=60, seed=20260925, jitter=0.3, random_endpoints=False):
    """B instances of the quadcopter scenario.  Default: the shipped start / goal jittered uniformly by +-jitter (instance 0 exact) with the
    way-point warm start.  random_endpoints=True: start anywhere in front of the first wall, goal anywhere behind the second, warm start from
    the 3-D grid A* of obca_amd/planner.py (the reference's a_star_3D.jl step); instance 0 stays the shipped one."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(QUAD_X0, (B, 1)); xF = np.tile(QUAD_XF, (B, 1))
    if not random_endpoints:
        if B > 1:
            x0[1:, :3] += rng.uniform(-jitter, jitter, (B - 1, 3)); xF[1:, :3] += rng.uniform(-jitter, jitter, (B - 1, 3))
        xWS = np.stack([quad_warm_start(x0[i], xF[i], N) for i in range(B)])
    else:
        from . import planner as PL
        xWS = np.zeros((B, N + 1, 12))
        for i in range(B):
            while True:
                if i:
                    x0[i, :3], xF[i, :3] = _draw_quad_endpoints(rng)
                w = PL.quad_warm_start(x0[i], xF[i], N)
                if w is not None:
                    xWS[i] = w; break
    return dict(x0=x0, xF=xF, N=N, Ts=quad_sample_time(N), R=QUAD_R, ob=QUAD_OB.copy(), xWS=xWS, timeWS=1.0)


# ---------------------------------------------------------------- BASELINE config 5: mixed obstacle counts
def make_mixed_batch(B, N=80, seed=20260925, max_extra=7, min_obstacles=3, rows=(3, 4), max_rows=40):
    """config-5 style batch: the backwards-parking scenario plus 0..max_extra extra convex obstacles per instance (triangles = 3 rows,
    quadrilaterals = 4 rows, clockwise vertices through obstHrep) placed in the block left of the slot where the car never goes, so every
    instance stays solvable while nOb runs from 3 to 10 and M from 5 to 33: irregular per-instance H-rep packing, 1-4 rows per obstacle.
    min_obstacles=1 (BASELINE.json configs[4] as written: "1-10 obstacles per instance"): the obstacle count is drawn from U{1..10}; counts
    below three keep only the first one / two obstacles of the scenario (left block; both blocks of the slot, no wall above the road).
    rows=(lo, hi): edge count of the extra polygons (default triangles and quadrilaterals; up to OBCA_VMAX = 8 edges; the instance keeps at
    most max_rows = 40 rows in total by default; OBCA_MMAX = 64, OBCA_NOBMAX = 16 are the library's limits: max_extra = 13, max_rows = 64 reach them)."""
    rng = np.random.default_rng(seed)
    base = make_batch(BACKWARDS, B, N, seed=seed)
    sc = BACKWARDS
    vl, Al, bl = [], [], []
    for i in range(B):
        if min_obstacles < 3:
            ntot = int(rng.integers(max(1, min_obstacles), 3 + max_extra + 1)); nbase = min(3, ntot); nex = ntot - nbase
        else:
            nbase = 3; nex = int(rng.integers(0, max_extra + 1))
        lOb = [list(map(list, o)) for o in sc["lOb"][:nbase]]; vOb = list(sc["vOb"][:nbase])
        for _ in range(nex):
            cx, cy, r = rng.uniform(-13, -4), rng.uniform(2.0, 4.0), rng.uniform(0.3, 0.8)
            nv = int(rng.integers(rows[0], rows[1] + 1))                   # triangle or quadrilateral by default
            if sum(vOb) - len(vOb) + nv > max_rows:
                break
            gap = 0.5 if nv <= 4 else 0.9 * np.pi / nv
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv))[::-1]             # clockwise
            while np.min(np.diff(np.concatenate([ang[::-1], [ang[-1] + 2 * np.pi]]))) < gap or (nv > 4 and np.max(np.diff(np.concatenate([ang[::-1], [ang[-1] + 2 * np.pi]]))) > 0.95 * np.pi):
                ang = np.sort(rng.uniform(0, 2 * np.pi, nv))[::-1]
            pts = [[cx + r * np.cos(a), cy + r * np.sin(a)] for a in ang]
            lOb.append(pts + [pts[0]]); vOb.append(nv + 1)                 # closed polygon: nv edges = nv rows
        A, b = obst_hrep(len(vOb), vOb, lOb)
        vl.append(np.asarray(vOb) - 1); Al.append(A); bl.append(b)
    base.update(vOb=vl, A=Al, b=bl)
    return base


def make_corridor_batch(B, N=80, seed=20260925, n_extra=(3, 7), clearance=(0.0, 0.2)):
    """config-5 style batch whose extra obstacles BIND: the backwards-parking scenario plus n_extra[0]..n_extra[1] wedges per instance that narrow the road from its two
    sides -- triangles standing on the upper wall (y = 11) or on the blocks beside the slot (y = 5), sloped rows through obstHrep -- with their tips `clearance` metres
    beside the car body of the instance's own warm start, on alternating sides along the path.  The corridor stays one piece (no obstacle to pass on either side), the
    warm start threads it, and the optimum -- shorter, smoother -- leans on the tips: separation rows are active where `make_mixed_batch`'s decoys never are.  A wedge is
    kept only if no pose of the warm start overlaps it (separating-axis test; a negative clearance[0] lets the tips intrude that far into the warm start's swept body).
    nOb runs from 3 to 3 + n_extra[1] (wedges that would block a later pose of the path are dropped), M = 5 + 3 per wedge.
    Measured on the oracle with the reference's IPOPT configuration, 64 instances (round 5): clearance (0, 0.2): 64 / 64 solved in 45 iterations (decoys: 41), 0.23 wedges
    touched per solution; tips intruding up to 0.05 / 0.15 / 0.3 m into the warm start: 50 / 44 / 23 of 64 solved, 110-155 iterations -- a warm start that penetrates
    several obstacles is beyond this interior point without IPOPT's restoration phase, which is why the default keeps the warm start clear."""
    rng = np.random.default_rng(seed)
    base = make_batch(BACKWARDS, B, N, seed=seed)
    sc = BACKWARDS
    f, l, r, rt = EGO                                      # front, left, rear, right extents from the rear axle
    vl, Al, bl = [], [], []

    def touches(poly, xs, ys, yaws, margin):
        """does the car rectangle at any pose overlap the convex polygon (separating axes: the car's two axes and the polygon's edge normals; vertices clockwise)"""
        P = np.asarray(poly)
        c, s = np.cos(yaws), np.sin(yaws)
        u = c[:, None] * (P[None, :, 0] - xs[:, None]) + s[:, None] * (P[None, :, 1] - ys[:, None])       # polygon vertices in the car frames
        w = -s[:, None] * (P[None, :, 0] - xs[:, None]) + c[:, None] * (P[None, :, 1] - ys[:, None])
        apart = (u.min(1) > f + margin) | (u.max(1) < -r - margin) | (w.min(1) > l + margin) | (w.max(1) < -rt - margin)
        corners = np.array([[f, l], [-r, l], [-r, -rt], [f, -rt]])
        cx = xs[:, None] + c[:, None] * corners[None, :, 0] - s[:, None] * corners[None, :, 1]
        cy = ys[:, None] + s[:, None] * corners[None, :, 0] + c[:, None] * corners[None, :, 1]
        n = len(P)
        for e in range(n):
            p, q = P[e], P[(e + 1) % n]
            nx, ny = -(q[1] - p[1]), (q[0] - p[0])         # outward normal of a clockwise edge
            nn = np.hypot(nx, ny); nx, ny = nx / nn, ny / nn
            apart |= (nx * (cx - p[0]) + ny * (cy - p[1])).min(1) > margin
        return bool((~apart).any())

    for i in range(B):
        xs, ys, yaws = base["xWS"][i, :, 0], base["xWS"][i, :, 1], base["xWS"][i, :, 2]
        lOb = [list(map(list, o)) for o in sc["lOb"]]; vOb = list(sc["vOb"])
        nex = int(rng.integers(n_extra[0], n_extra[1] + 1))
        road = np.flatnonzero((ys > 6.2) & (np.abs(np.sin(yaws)) < 0.5))      # poses driving along the road; the slot's own walls bind further down
        ks = np.sort(rng.choice(road, size=min(nex, len(road)), replace=False)) if len(road) else []
        up = rng.random() < 0.5
        for k in ks:
            for attempt in range(10):
                clr = rng.uniform(*clearance); hw = rng.uniform(0.35, 0.8)
                xt = xs[k] + np.cos(yaws[k]) * rng.uniform(0.0, 3.0) + rng.uniform(-0.3, 0.3)
                if abs(xt) < 2.2 and not up:               # not in front of the slot's mouth
                    continue
                if up:                                      # wedge hanging from the upper wall, tip above the car's left / upper side
                    yt = max(ys[k], ys[min(k + 3, N)]) + max(l, rt) + clr + 0.35 * abs(np.sin(yaws[k])) * f
                    pts = [[xt - hw, 10.97], [xt + hw + 1e-3, 10.97], [xt + 2e-3, yt]]          # clockwise
                    ok = yt < 10.6
                else:                                       # wedge standing on the blocks beside the slot
                    yt = min(ys[k], ys[min(k + 3, N)]) - max(l, rt) - clr - 0.35 * abs(np.sin(yaws[k])) * f
                    pts = [[xt + hw + 1e-3, 5.03], [xt - hw, 5.03], [xt + 2e-3, yt]]            # clockwise
                    ok = yt > 5.4
                if ok and XYBOUNDS[0] + 1 < xt - hw and xt + hw < XYBOUNDS[1] - 1 and not touches(pts, xs, ys, yaws, min(0.0, clearance[0]) - 0.02):
                    lOb.append(pts + [pts[0]]); vOb.append(4); up = not up
                    break
        A, b = obst_hrep(len(vOb), vOb, lOb)
        vl.append(np.asarray(vOb) - 1); Al.append(A); bl.append(b)
    base.update(vOb=vl, A=Al, b=bl)
    return base
