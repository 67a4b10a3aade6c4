"""
A-posteriori feasibility checkers for returned trajectories -- the public validate() API.  Pure numpy,
no GPU, no oracle: they look only at a solution.

parking_constraints_ref : restatement of /root/reference/AutonomousParking/ParkingConstraints.jl:29-149 VERBATIM, including its
    quirks: in variable-time mode all four dynamics residuals are written to c3[0,i] (only the last, the speed
    row, survives, :76-79); c6 is overwritten per obstacle so only the LAST obstacle is checked (:108-130); the c6[3] row
    ignores the slack (:127-128); the steering-rate check divides by timeScale[0] only (:88).  It is the reference's own
    acceptance test (tolerance 5e-5, :133-139) and decides exitflag after failed attempts (ParkingSignedDist.jl:278-283, ParkingDist.jl).
parking_constraints_full : a correct checker of every constraint class of ParkingSignedDist.jl:100-207 (with the slack),
    returning the individual maxima.
validate_parking : (ok, violations) of one returned parking solution; validate_quadcopter: restatement of
    /root/reference/QuadcopterNavigation/constrSatisfaction.jl:25-204 (tolerance 1e-3; single-index gyroscopic quirk included).
Shapes follow the reference: x (4,N+1), u (2,N), l (M,N+1), n (4nOb,N+1), timeScale (N+1,) .
"""
import numpy as np

DMIN = 0.05


def _geom(ego):
    ego = np.asarray(ego, float).ravel()
    W_ev, L_ev = ego[1] + ego[3], ego[0] + ego[2]
    return np.array([L_ev / 2, W_ev / 2, L_ev / 2, W_ev / 2]), (ego[0] + ego[2]) / 2 - ego[2]


def _dyn(x, u, ts, Ts, L):
    q = ts * Ts
    s = x[3] + q / 2 * u[1]
    phi = x[2] + q / 2 * x[3] * np.tan(u[0]) / L
    return np.array([x[0] + q * s * np.cos(phi), x[1] + q * s * np.sin(phi), x[2] + q * s * np.tan(u[0]) / L, x[3] + q * u[1]])


def parking_constraints_ref(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, x, u, l, n, timeScale, fixTime, sd):
    x0 = np.ravel(x0); xF = np.ravel(xF); vOb = [int(v) for v in np.ravel(vOb)]
    A = np.asarray(A, float).reshape(-1, 2); b = np.ravel(b)
    timeScale = np.ravel(timeScale)
    c0 = np.zeros(5)
    c0[0] = np.max(np.abs(u[0, :])) - 0.6
    c0[1] = np.max(np.abs(u[1, :])) - 0.4
    c0[2] = np.max(np.abs(timeScale - 1)) - 0.2
    c0[3] = -np.min(l)
    c0[4] = -np.min(n)
    c1 = np.abs(x[:, 0] - x0)
    c2 = np.abs(x[:, N] - xF)
    c3 = np.zeros((4, N))
    for i in range(N):
        if fixTime == 1:
            c3[:, i] = x[:, i + 1] - _dyn(x[:, i], u[:, i], 1.0, Ts, L)
        else:
            r = x[:, i + 1] - _dyn(x[:, i], u[:, i], timeScale[i], Ts, L)
            c3[0, i] = r[3]          # ParkingConstraints.jl:76-79: every row is stored in c3[1,i]; the last assignment wins
    if fixTime == 1:
        c5 = np.max(np.abs(np.diff(np.concatenate([[0.0], u[0, :]]))) / Ts) - 0.6
        c4 = 0.0
    else:
        c4 = np.max(np.abs(np.diff(timeScale)))
        c5 = np.max(np.abs(np.diff(np.concatenate([[0.0], u[0, :]]))) / (timeScale[0] * Ts)) - 0.6
    g, off = _geom(ego)
    c6 = np.zeros((4, N + 1))
    for i in range(N + 1):
        r0 = 0
        for j in range(nOb):
            Aj = A[r0:r0 + vOb[j]]; bj = b[r0:r0 + vOb[j]]; lj = l[r0:r0 + vOb[j], i]; nj = n[4 * j:4 * j + 4, i]
            r0 += vOb[j]
            p = Aj.T @ lj
            cs, sn = np.cos(x[2, i]), np.sin(x[2, i])
            if sd == 1:
                c6[0, i] = abs(p[0] ** 2 + p[1] ** 2) - 1
            else:
                c6[0, i] = p[0] ** 2 + p[1] ** 2 - 1
            c6[1, i] = abs(nj[0] - nj[2] + cs * p[0] + sn * p[1])
            c6[2, i] = abs(nj[1] - nj[3] - sn * p[0] + cs * p[1])
            c6[3, i] = -(-g @ nj + (x[0, i] + cs * off) * p[0] + (x[1, i] + sn * off) * p[1] - bj @ lj) + DMIN
    e = [np.max(c0) <= 5e-5, np.max(c1) <= 5e-5, np.max(c2) <= 5e-5, np.max(np.abs(c3)) <= 5e-5, c4 <= 5e-5, c5 <= 5e-5,
         np.max(c6) <= 5e-5]
    return 1 if sum(e) == 7 else 0


def parking_constraints_full(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, x, u, l, n, timeScale, fixTime, sl=None):
    """max violation of each constraint class (all <= tol means feasible).  sl (nOb,N+1) is the parking slack (if None the
    obstacle row is evaluated with the best slack, i.e. the row is reported as the required slack)."""
    x0 = np.ravel(x0); xF = np.ravel(xF); vOb = [int(v) for v in np.ravel(vOb)]
    A = np.asarray(A, float).reshape(-1, 2); b = np.ravel(b); XYb = np.ravel(XYbounds)
    ts = np.ones(N + 1) if fixTime else np.ravel(timeScale)
    out = {}
    out["u_bounds"] = max(np.max(np.abs(u[0])) - 0.6, np.max(np.abs(u[1])) - 0.4)
    out["x_bounds"] = max(np.max(XYb[0] - x[0]), np.max(x[0] - XYb[1]), np.max(XYb[2] - x[1]), np.max(x[1] - XYb[3]),
                          np.max(-1 - x[3]), np.max(x[3] - 2))
    out["ts_bounds"] = max(np.max(0.8 - ts), np.max(ts - 1.2)) if not fixTime else 0.0
    out["ts_chain"] = np.max(np.abs(np.diff(ts)))
    out["dual_pos"] = max(-np.min(l), -np.min(n))
    out["start"] = np.max(np.abs(x[:, 0] - x0)); out["end"] = np.max(np.abs(x[:, N] - xF))
    out["dyn"] = max(np.max(np.abs(x[:, i + 1] - _dyn(x[:, i], u[:, i], ts[i], Ts, L))) for i in range(N))
    du = np.diff(np.concatenate([[0.0], u[0]]))
    out["steer_rate"] = np.max(np.abs(du) / (ts[:N] * Ts)) - 0.6
    g, off = _geom(ego)
    cn = ce = cd = 0.0
    need = np.zeros((nOb, N + 1))
    for i in range(N + 1):
        r0 = 0
        cs, sn = np.cos(x[2, i]), np.sin(x[2, i])
        for j in range(nOb):
            Aj = A[r0:r0 + vOb[j]]; bj = b[r0:r0 + vOb[j]]; lj = l[r0:r0 + vOb[j], i]; nj = n[4 * j:4 * j + 4, i]
            r0 += vOb[j]
            p = Aj.T @ lj
            cn = max(cn, abs(p @ p - 1))
            ce = max(ce, abs(nj[0] - nj[2] + cs * p[0] + sn * p[1]), abs(nj[1] - nj[3] - sn * p[0] + cs * p[1]))
            row = -g @ nj + (x[0, i] + cs * off) * p[0] + (x[1, i] + sn * off) * p[1] - bj @ lj
            need[j, i] = DMIN - row
            if sl is not None:
                cd = max(cd, DMIN - (row + sl[j, i]))
    out["norm"] = cn; out["rot"] = ce
    out["sep"] = cd if sl is not None else 0.0
    out["penetration"] = float(np.max(need))      # >0 : the trajectory needs positive slack somewhere (min-penetration mode)
    return out


def feasible(viol, tol=5e-5):
    return all(v <= tol for k, v in viol.items() if k != "penetration")


def validate_parking(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, xp, up, timeScale, lp, np_, sl=None, fixTime=0, tol=5e-5, dist=False):
    """(ok, violations).  dist=False: min-penetration solution (every row with its slack, |A'lam| == 1); dist=True: collision-free
    solution (|A'lam| <= 1, separation rows without slack)."""
    nOb = len(np.ravel(vOb)); ts = np.broadcast_to(np.ravel(timeScale), (N + 1,)) if np.size(timeScale) > 1 else np.full(N + 1, float(timeScale))
    v = parking_constraints_full(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, xp, up, lp, np_, ts, fixTime,
                                 np.zeros((nOb, N + 1)) if (dist or sl is None) else sl)
    if dist:
        Am = np.asarray(A, float).reshape(-1, 2); r0 = 0; worst = 0.0
        for vj in [int(q) for q in np.ravel(vOb)]:
            p = Am[r0:r0 + vj].T @ lp[r0:r0 + vj]; worst = max(worst, float(np.max((p ** 2).sum(0) - 1))); r0 += vj
        v["norm"] = worst
    return feasible(v, tol), v


_Q = dict(mass=0.5, g=9.81, kF=0.0611, kM=0.0015, I=(3.9e-3, 4.4e-3, 4.9e-3), arm=0.225)
_QXL = np.array([0, 0, 0, -3, -0.2, -0.2, -1, -1, -1, -1.5, -1, -1.0]); _QXU = np.array([10, 10, 5, 3, 0.2, 0.2, 1, 1, 1, 3, 1, 1.0])


def validate_quadcopter(x, u, timeScale, x0, xF, Ts, lam, ob, R, tol=1e-3):
    """constrSatisfaction(x,u,timeScale,x0,xF,Ts,lambda,ob1..ob5,R): x (12,N+1), u (4,N), lam (30,N+1), ob (5,6).  Returns (ok, worst)."""
    x = np.asarray(x, float); u = np.asarray(u, float); lam = np.asarray(lam, float); ob = np.asarray(ob, float).reshape(5, 6)
    N = x.shape[1] - 1; ts = np.broadcast_to(np.ravel(timeScale), (N + 1,)); q = _Q; w = {}
    w["start"] = np.abs(x[:, 0] - np.ravel(x0)).max(); w["end"] = np.abs(x[:, -1] - np.ravel(xF)).max()
    w["u_bounds"] = max((1.2 - u).max(), (u - 7.8).max()); w["x_bounds"] = max((_QXL[:, None] - x[:, :N]).max(), (x[:, :N] - _QXU[:, None]).max())
    X = x[:, :N]; U = (u ** 2).sum(0); s4, c4, s5, c5, s6, c6 = np.sin(X[3]), np.cos(X[3]), np.sin(X[4]), np.cos(X[4]), np.sin(X[5]), np.cos(X[5])
    g0 = x[9:12, 0]                                   # x[10], x[11], x[12] with a single index = stage 1
    G = np.stack([X[6], X[7], X[8], c5 * X[9] + s5 * X[11], s5 * s4 / c4 * X[9] + X[10] - c5 * s4 / c4 * X[11], -s5 / c4 * X[9] + c5 / c4 * X[11],
                  q["kF"] / q["mass"] * U * (s4 * c5 * s6 + s5 * c6), q["kF"] / q["mass"] * U * (-s4 * c5 * c6 + s5 * s6),
                  (q["kF"] * U * c4 * c5 - q["mass"] * q["g"]) / q["mass"],
                  (q["arm"] * q["kF"] * (u[1] ** 2 - u[3] ** 2) - (q["I"][2] - q["I"][1]) * g0[1] * g0[2]) / q["I"][0],
                  (q["arm"] * q["kF"] * (u[2] ** 2 - u[0] ** 2) - (q["I"][0] - q["I"][2]) * g0[0] * g0[2]) / q["I"][1],
                  (q["kM"] * (u[0] ** 2 - u[1] ** 2 + u[2] ** 2 - u[3] ** 2) - (q["I"][1] - q["I"][0]) * g0[0] * g0[1]) / q["I"][2]])
    w["dyn"] = np.abs(x[:, 1:] - X - ts[:N] * Ts * G).max(); w["ts_chain"] = np.abs(np.diff(ts)).max()
    w["dual_pos"] = -lam.min()
    L5 = lam.reshape(5, 6, N + 1)[:, :, :N]; qv = L5[:, :3] - L5[:, 3:]
    w["norm"] = ((qv ** 2).sum(1) - 1).max()
    sep = -(ob[:, :, None] * L5).sum(1) + (x[None, :3, :N] * qv).sum(1) - R
    w["sep"] = -sep.min()
    bad = w["start"] > tol or w["end"] > tol or w["u_bounds"] > 0 or w["x_bounds"] > 0 or w["dyn"] > tol or w["ts_chain"] > tol or \
        w["dual_pos"] > tol or w["norm"] > tol or w["sep"] > tol
    return (not bad), w
