"""
TEST INFRASTRUCTURE ONLY: ctypes front end of the C oracle (oracle/obca_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int)


class Opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int)] + \
        [(n, C.c_double) for n in ("mu_init kappa_eps kappa_mu theta_mu tau_min bound_push bound_frac dw_min dw0 dw_max "
                                   "kw_inc0 kw_inc kw_dec dc_bar kappa_c gamma_theta gamma_phi delta s_theta s_phi eta_phi "
                                   "gamma_alpha s_max kappa_sigma constr_viol_tol dual_inf_tol compl_inf_tol rho_term").split()] + \
        [("lsq_init", C.c_int), ("verbose", C.c_int), ("max_soc", C.c_int), ("recalc_y", C.c_int), ("obj_scaling", C.c_int), ("restoration", C.c_int)]


def build():
    so = os.path.join(_HERE, "libobca_oracle.so")
    src = os.path.join(_HERE, "obca_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def native_dir():
    """oracle/_native/<hash of the CPU model>: one native build per CPU model -- a copy made on another machine (the tree travels to the GPU box) must not be picked up"""
    import hashlib
    try:
        cpu = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        cpu = []
    return os.path.join(_HERE, "_native", hashlib.sha1("".join(cpu).encode()).hexdigest()[:12])


def build_native():
    """-O3 -march=native build for bench.py's cpu_baseline leg, compiled on the machine that times it (`make native`, oracle/Makefile holds both flag sets).
    The parity tests keep the portable -O2 build: -march=native lets gcc contract into FMAs, which moves the last bits of the iterates."""
    d = native_dir()
    subprocess.check_call(["make", "-C", _HERE, "-s", "native", "NATIVE_DIR=" + d])
    return os.path.join(d, "libobca_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build_native() if os.environ.get("OBCA_ORACLE_NATIVE") == "1" else build())
    return _LIB


def default_opts():
    o = Opts()
    lib().obca_oracle_default_opts(C.byref(o))
    return o


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_D)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I)


LAYOUT_FIELDS = ("x u t lam mu sl so ss pi nu yg yo zxL zxU zuL zuU ztL ztU zlam zmu zso zssL zssU zs1 nprimal len").split()


def layout(N, vOb):
    vOb_, pv = _i(vOb)
    out = np.zeros(32, np.int32)
    n = lib().obca_oracle_layout(N, len(vOb_), pv, out.ctypes.data_as(_I))
    assert n == len(LAYOUT_FIELDS)
    return dict(zip(LAYOUT_FIELDS, out[:n].tolist()))


def dualmult_ws(N, vOb, A, b, rx, ry, ryaw, ego):
    """-> lWS (N+1,M), nWS (N+1,4nOb), d (N+1,nOb)   (DualMultWS.jl:29-86; ego explicit)"""
    vOb_, pv = _i(vOb); nOb = len(vOb_); M = int(vOb_.sum())
    A_, pA = _d(A); b_, pb = _d(b); rx_, prx = _d(rx); ry_, pry = _d(ry); ryaw_, pyw = _d(ryaw); ego_, pe = _d(ego)
    lWS = np.zeros((N + 1, M)); nWS = np.zeros((N + 1, 4 * nOb)); d = np.zeros((N + 1, nOb))
    rc = lib().obca_oracle_dualmult_ws(N, nOb, pv, pA, pb, prx, pry, pyw, pe, lWS.ctypes.data_as(_D), nWS.ctypes.data_as(_D),
                                       d.ctypes.data_as(_D))
    assert rc == 0
    return lWS, nWS, d


def parking_signed_dist(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, lWS=None, nWS=None,
                        opts=None, dist=0, full=False):
    """Mirrors ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS) (ParkingSignedDist.jl:29).
    xWS (N+1,4), uWS (>=N,2) as in the reference.  Returns dict with xp (4,N+1), up (2,N), timeScale, exitflag, lp (M,N+1),
    np (4nOb,N+1), sl, info."""
    vOb_, pv = _i(vOb); nOb = len(vOb_); M = int(vOb_.sum())
    if lWS is None:
        lWS, nWS, _ = dualmult_ws(N, vOb, A, b, rx, ry, ryaw, ego)
    args = [_d(v) for v in (ego, XYbounds, x0, xF)]
    A_, pA = _d(A); b_, pb = _d(b); rx_, prx = _d(rx); ry_, pry = _d(ry); ryaw_, pyw = _d(ryaw)
    xw, pxw = _d(np.asarray(xWS, float)[:N + 1]); uw, puw = _d(np.asarray(uWS, float)[:N])
    lw, plw = _d(lWS); nw, pnw = _d(nWS)
    xp = np.zeros((N + 1, 4)); up = np.zeros((N, 2)); ts = np.zeros(N + 1); lp = np.zeros((N + 1, M)); npp = np.zeros((N + 1, 4 * nOb))
    slp = np.zeros((N + 1, nOb)); ef = C.c_int(0); info = np.zeros(8)
    fn = lib().obca_oracle_parking_dist if dist else lib().obca_oracle_parking_signed_dist
    extra = []
    if full:        # full primal-dual iterate (signed-distance formulation only): layout(N, vOb)
        assert not dist
        zfull = np.zeros(layout(N, vOb_)["len"]); fn = lib().obca_oracle_parking_signed_dist_full; extra = [zfull.ctypes.data_as(_D)]
    rc = fn(
        C.c_int(N), C.c_double(Ts), C.c_double(L), args[0][1], args[1][1], C.c_int(int(fixTime)), args[2][1], args[3][1],
        C.c_int(nOb), pv, pA, pb, prx, pry, pyw, pxw, puw, plw, pnw, C.byref(opts) if opts is not None else None,
        xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D), lp.ctypes.data_as(_D), npp.ctypes.data_as(_D),
        slp.ctypes.data_as(_D), C.byref(ef), info.ctypes.data_as(_D), *extra)
    assert rc == 0
    if full:
        return dict(zfull=zfull, exitflag=ef.value, iters=int(info[1]), obj=info[2], mu=info[5], xp=xp.T.copy(), up=up.T.copy(), t=info[7],
                    lp=lp.T.copy(), np=npp.T.copy(), sl=slp.T.copy())
    return dict(xp=xp.T.copy(), up=up.T.copy(), timeScale=ts, exitflag=ef.value, lp=lp.T.copy(), np=npp.T.copy(), sl=slp.T.copy(),
                status=int(info[0]), iters=int(info[1]), obj=info[2], pinf=info[3], dinf=info[4], mu=info[5], nreg=int(info[6]), t=info[7])


def parking_dist(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, lWS=None, nWS=None, opts=None):
    """Mirrors ParkingDist(...) (ParkingDist.jl:29): the collision-free sibling; `sl` of the result holds the slack of |A'lam|^2 <= 1."""
    return parking_signed_dist(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, lWS, nWS, opts, dist=1)


def ref_constraints(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, xp, up, t, lp, np_, fixTime, sd):
    """the C restatement of ParkingConstraints.jl used for the exit flag; xp (4,N+1), up (2,N), lp (M,N+1), np_ (4nOb,N+1)"""
    vOb_, pv = _i(vOb)
    a = [_d(v) for v in (ego, XYbounds, x0, xF, A, b, np.asarray(xp).T, np.asarray(up).T, np.asarray(lp).T, np.asarray(np_).T)]
    return lib().obca_oracle_ref_constraints(C.c_int(N), C.c_double(Ts), C.c_double(L), a[0][1], a[1][1], C.c_int(int(fixTime)), a[2][1], a[3][1],
                                             C.c_int(len(vOb_)), pv, a[4][1], a[5][1], a[6][1], a[7][1], C.c_double(t), a[8][1], a[9][1], C.c_int(int(sd)))


def newton_soc(N, Ts, L, ego, XYbounds, fixTime, x0, xF, vOb, A, b, rx, ry, ryaw, z, mu, dw, dc, csoc, rho=1e3, dist=0):
    """Newton system of z with the constraint values csoc (layout pi | nu | yg | yo) on the right-hand side: the step of a second-order correction"""
    vOb_, pv = _i(vOb); nOb = len(vOb_)
    a = [_d(v) for v in (ego, XYbounds, x0, xF, A, b, rx, ry, ryaw, z, csoc)]
    d = np.zeros_like(a[9][0])
    ok = lib().obca_oracle_newton_soc(C.c_int(N), C.c_double(Ts), C.c_double(L), a[0][1], a[1][1], C.c_int(int(fixTime)), a[2][1], a[3][1],
                                      C.c_int(nOb), pv, a[4][1], a[5][1], a[6][1], a[7][1], a[8][1], a[9][1], C.c_double(mu),
                                      C.c_double(dw), C.c_double(dc), C.c_double(rho), a[10][1], d.ctypes.data_as(_D), C.c_int(int(dist)))
    return ok, d


def lsq_multipliers(N, Ts, L, ego, XYbounds, fixTime, x0, xF, vOb, A, b, rx, ry, ryaw, z, dist=0):
    """least-squares multiplier step at z (recalc_y / lsq_init): returns (ok, d) with d[pi:zxL] = the increment of the equality multipliers"""
    vOb_, pv = _i(vOb); nOb = len(vOb_)
    a = [_d(v) for v in (ego, XYbounds, x0, xF, A, b, rx, ry, ryaw, z)]
    d = np.zeros_like(a[9][0])
    ok = lib().obca_oracle_lsq_multipliers(C.c_int(N), C.c_double(Ts), C.c_double(L), a[0][1], a[1][1], C.c_int(int(fixTime)), a[2][1], a[3][1],
                                           C.c_int(nOb), pv, a[4][1], a[5][1], a[6][1], a[7][1], a[8][1], a[9][1], d.ctypes.data_as(_D), C.c_int(int(dist)))
    return ok, d


def newton(N, Ts, L, ego, XYbounds, fixTime, x0, xF, vOb, A, b, rx, ry, ryaw, z, mu, dw, dc, rho=1e3, dist=0):
    vOb_, pv = _i(vOb); nOb = len(vOb_)
    a = [_d(v) for v in (ego, XYbounds, x0, xF, A, b, rx, ry, ryaw, z)]
    d = np.zeros_like(a[9][0]); errs = np.zeros(3)
    ok = lib().obca_oracle_newton(C.c_int(N), C.c_double(Ts), C.c_double(L), a[0][1], a[1][1], C.c_int(int(fixTime)), a[2][1], a[3][1],
                                  C.c_int(nOb), pv, a[4][1], a[5][1], a[6][1], a[7][1], a[8][1], a[9][1], C.c_double(mu),
                                  C.c_double(dw), C.c_double(dc), C.c_double(rho), d.ctypes.data_as(_D), errs.ctypes.data_as(_D), C.c_int(int(dist)))
    return ok, d, errs
