"""
TEST INFRASTRUCTURE ONLY.  A-posteriori feasibility checkers for parking solutions.

parking_constraints_ref : restatement of /root/reference/AutonomousParking/ParkingConstraints.jl:29-149 VERBATIM, including its
    quirks: in variable-time mode all four dynamics residuals are written to c3[0,i] (only the last, the speed
    row, survives, :76-79); c6 is overwritten per obstacle so only the LAST obstacle is checked (:108-130); the c6[3] row
    ignores the slack (:127-128); the steering-rate check divides by timeScale[0] only (:88).  It is the reference's own
    acceptance test (tolerance 5e-5, :133-139) and decides exitflag after the second attempt (ParkingSignedDist.jl:278-283).
parking_constraints_full : a correct checker of every constraint class of ParkingSignedDist.jl:100-207 (with the slack),
    returning the individual maxima; used by the tests as the real feasibility criterion.
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
