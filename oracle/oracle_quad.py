"""TEST INFRASTRUCTURE ONLY: ctypes front end of oracle/obca_oracle_quad.c (quadcopter signed-distance path)."""
import ctypes as C
import os
import subprocess
import numpy as np
from oracle import Opts   # same option struct layout (tol, max_iter, ..., lsq_init, verbose)

_HERE = os.path.dirname(os.path.abspath(__file__))
_D = C.POINTER(C.c_double)
_LIB = None
LAYOUT_FIELDS = "x u t lam s so n pi nu yo m".split()
# shipped scenario (mainQuadcopter.jl:36-54), boxes as literals [xmax,ymax,zmax,-xmin,-ymin,-zmin]
OB_LITERAL = np.array([[2.5, 12, 7, -2, 2, -0.6], [7.5, 12, 7, -7, -5, 2], [7.5, 4, 7, -7, 2, 2], [7.5, 5, 2, -7, -4, 2], [7.5, 5, 7, -7, -4, -3]], float)
# ... and as clamped in place by plotTrajQuadcopter before the signed-distance call
OB_CLAMPED = np.array([[2.5, 10, 5, -2, 0, -0.6], [7.5, 10, 5, -7, -5, 0], [7.5, 4, 5, -7, 0, 0], [7.5, 5, 2, -7, -4, 0], [7.5, 5, 5, -7, -4, -3]], float)
X0 = np.array([1, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0.0]); XF = np.array([9, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])
EGO_R = 0.25


def build_native():
    """the -O3 -march=native build bench.py's cpu_baseline leg times (oracle.build_native: `make native`)"""
    import oracle as _O
    _O.build_native()
    return os.path.join(_O.native_dir(), "libobca_oracle_quad.so")


def lib():
    global _LIB
    if _LIB is None:
        if os.environ.get("OBCA_ORACLE_NATIVE") == "1":
            so = build_native()
        else:
            so = os.path.join(_HERE, "libobca_oracle_quad.so"); src = os.path.join(_HERE, "obca_oracle_quad.c")
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = C.CDLL(so)
    return _LIB


def default_opts():
    o = Opts(); lib().obca_oracle_quad_default_opts(C.byref(o)); return o


def layout(N):
    out = np.zeros(16, np.int32)
    n = lib().obca_oracle_quad_layout(C.c_int(N), out.ctypes.data_as(C.POINTER(C.c_int)))
    return dict(zip(LAYOUT_FIELDS, out[:n].tolist()))


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64); return a, a.ctypes.data_as(_D)


def warm_start(x0, xF, N, via=None):
    """positions along straight segments x0 -> via points -> xF, all other states 0 (mainQuadcopter.jl:134-138 uses 3-D A*)."""
    pts = [np.asarray(x0, float)[:3]] + [np.asarray(p, float) for p in (via or [])] + [np.asarray(xF, float)[:3]]
    seg = np.array([np.linalg.norm(pts[i + 1] - pts[i]) for i in range(len(pts) - 1)]); cum = np.concatenate([[0], np.cumsum(seg)])
    xWS = np.zeros((N + 1, 12))
    for k, s in enumerate(np.linspace(0, cum[-1], N + 1)):
        i = min(np.searchsorted(cum, s, side="right") - 1, len(seg) - 1)
        a = (s - cum[i]) / seg[i] if seg[i] > 0 else 0.0
        xWS[k, :3] = pts[i] + a * (pts[i + 1] - pts[i])
    return xWS


def quadcopter_signed_dist(x0, xF, N, Ts, R, ob, xWS, timeWS=1.0, opts=None, dual_ws=1, dist=0):
    """mirrors QuadcopterSignedDist(x0,xF,N,Ts,R,ob1..ob5,xWS,uWS,timeWS) (QuadcopterSignedDist.jl:25); xWS (N+1,12) here.
    dist=1: QuadcopterDist (QuadcopterDist.jl:25): no slack variable, x[10] in [-1.5, 3], no sum-slack exit flag."""
    a = [_d(v) for v in (x0, xF, np.reshape(ob, (5, 6)), np.asarray(xWS, float)[:N + 1])]
    xp = np.zeros((N + 1, 12)); up = np.zeros((N, 4)); ts = np.zeros(N + 1); lp = np.zeros((N + 1, 30)); sl = np.zeros((N + 1, 5))
    ef = C.c_int(0); info = np.zeros(8)
    rc = lib().obca_oracle_quadcopter_signed_dist(C.c_int(N), C.c_double(Ts), C.c_double(R), a[0][1], a[1][1], a[2][1], a[3][1], C.c_double(timeWS), C.c_int(int(bool(dual_ws)) | (int(bool(dist)) << 1)),
                                                  C.byref(opts) if opts is not None else None, xp.ctypes.data_as(_D), up.ctypes.data_as(_D),
                                                  ts.ctypes.data_as(_D), lp.ctypes.data_as(_D), sl.ctypes.data_as(_D), C.byref(ef), info.ctypes.data_as(_D))
    assert rc == 0
    return dict(xp=xp.T.copy(), up=up.T.copy(), timeScale=ts, exitflag=ef.value, lp=lp.T.copy(), slack=sl.T.copy(), status=int(info[0]),
                iters=int(info[1]), obj=info[2], pinf=info[3], dinf=info[4], mu=info[5], nreg=int(info[6]), t=info[7])


def quadcopter_signed_dist_full(x0, xF, N, Ts, R, ob, xWS, timeWS=1.0, opts=None, dual_ws=1):
    """the same solve + the full primal-dual iterate (v, y, zL, zU) in the oracle's layout (layout(N)): for the optimality certificate"""
    a = [_d(v) for v in (x0, xF, np.reshape(ob, (5, 6)), np.asarray(xWS, float)[:N + 1])]
    xp = np.zeros((N + 1, 12)); up = np.zeros((N, 4)); ts = np.zeros(N + 1); lp = np.zeros((N + 1, 30)); sl = np.zeros((N + 1, 5))
    ef = C.c_int(0); info = np.zeros(8); L = layout(N); full = np.zeros(3 * L["n"] + L["m"])
    rc = lib().obca_oracle_quadcopter_signed_dist_full(C.c_int(N), C.c_double(Ts), C.c_double(R), a[0][1], a[1][1], a[2][1], a[3][1], C.c_double(timeWS), C.c_int(int(bool(dual_ws))),
                                                       C.byref(opts) if opts is not None else None, xp.ctypes.data_as(_D), up.ctypes.data_as(_D),
                                                       ts.ctypes.data_as(_D), lp.ctypes.data_as(_D), sl.ctypes.data_as(_D), C.byref(ef), info.ctypes.data_as(_D), full.ctypes.data_as(_D))
    assert rc == 0
    n, m = L["n"], L["m"]
    return dict(xp=xp.T.copy(), up=up.T.copy(), exitflag=ef.value, lp=lp.T.copy(), slack=sl.T.copy(), iters=int(info[1]), obj=info[2], mu=info[5], t=info[7],
                v=full[:n], y=full[n:n + m], zL=full[n + m:2 * n + m], zU=full[2 * n + m:])


def quadcopter_dist(x0, xF, N, Ts, R, ob, xWS, timeWS=1.0, opts=None, dual_ws=1):
    return quadcopter_signed_dist(x0, xF, N, Ts, R, ob, xWS, timeWS, opts, dual_ws, dist=1)


def lsq_multipliers(N, Ts, R, x0, xF, ob, v, zL, zU, dist=0):
    """(ok, y_ls): the least-squares multiplier estimate the option lsq_init starts from"""
    a = [_d(q) for q in (x0, xF, np.reshape(ob, (5, 6)), v, zL, zU)]
    yls = np.zeros(layout(N)["m"])
    ok = lib().obca_oracle_quad_lsq_multipliers(C.c_int(N), C.c_double(Ts), C.c_double(R), a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], a[5][1],
                                                yls.ctypes.data_as(_D), C.c_int(int(dist)))
    return ok, yls


def newton(N, Ts, R, x0, xF, ob, v, y, zL, zU, mu, dw, dc, rho=1e3, dist=0):
    a = [_d(q) for q in (x0, xF, np.reshape(ob, (5, 6)), v, y, zL, zU)]
    L = layout(N)
    dv = np.zeros(L["n"]); dy = np.zeros(L["m"]); errs = np.zeros(3)
    ok = lib().obca_oracle_quad_newton(C.c_int(N), C.c_double(Ts), C.c_double(R), a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], a[5][1], a[6][1],
                                       C.c_double(mu), C.c_double(dw), C.c_double(dc), C.c_double(rho), dv.ctypes.data_as(_D), dy.ctypes.data_as(_D),
                                       errs.ctypes.data_as(_D), C.c_int(int(dist)))
    return ok, dv, dy, errs
