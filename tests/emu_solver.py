"""TEST INFRASTRUCTURE: the HIP solver source compiled as a host emulation (tests/emu/obca_emu.cpp), wrapped with the signature of
obca_amd.parking_signed_dist_batch so that host-side plumbing (sharding over ranks) can be exercised end-to-end on a machine without a GPU
with the kernels' own logic -- DualMultWS sub-problems + interior point -- rather than the oracle."""
import ctypes as C
import os
import subprocess
import numpy as np
import packing as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = C.POINTER(C.c_double)
dp = lambda a: a.ctypes.data_as(D)


class EOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int)] + \
        [(n, C.c_double) for n in ("mu_init kappa_eps kappa_mu theta_mu tau_min bound_push bound_frac dw_min dw0 dw_max "
                                   "kw_inc0 kw_inc kw_dec dc_bar kappa_c gamma_theta gamma_phi delta s_theta s_phi eta_phi "
                                   "gamma_alpha s_max kappa_sigma constr_viol_tol dual_inf_tol compl_inf_tol rho_term").split()] + \
        [("max_soc", C.c_int), ("recalc_y", C.c_int), ("lsq_init", C.c_int), ("obj_scaling", C.c_int), ("restoration", C.c_int)]


_VARIANTS = {None: ("libobca_emu.so", ["-O1"])}
try:      # the checking builds of the emulation (race log, UBSan, ASan: tests/test_emu_sanitize.py) -- CPU only; their flag table lives in its own file, which like that test is
          # listed in .gpurunignore: sanitizer builds have no business on the GPU box
    from emu_sanitize_variants import VARIANTS as _SAN
    _VARIANTS.update(_SAN)
except ImportError:
    pass
_loaded = {}


def build(variant=None):
    name, flags = _VARIANTS[variant]
    src = os.path.join(ROOT, "tests", "emu", "obca_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", name)
    deps = [src] + [os.path.join(ROOT, "obca_amd", "csrc", f) for f in ("obca_solver.h", "obca_solver_lanes.h", "obca_solver_assemble.h", "obca_solver_riccati.h", "obca_solver_direction.h", "obca_solver_ipm.h", "obca_model.h", "obca_quad_solver.h", "obca_quad_model.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        tmp = so + ".%d.tmp" % os.getpid()                  # (several xdist workers may build at once: compile aside, rename into place)
        from obca_amd.buildflags import GXX                  # warnings are errors here too (obca_amd/buildflags.py)
        subprocess.check_call(GXX + flags + ["-o", tmp, src, "-ldl"])
        os.replace(tmp, so)
    return so


def load(variant=None):
    if variant not in _loaded:
        _loaded[variant] = C.CDLL(build(variant))
    return _loaded[variant]


def default_opts():
    """the reference's IPOPT option set (ParkingSignedDist.jl:41-43 + IPOPT defaults), as obca_default_opts sets it"""
    o = EOpts()
    vals = dict(tol=1e-5, max_iter=200, mu_init=0.1, kappa_eps=10, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99, bound_push=1e-2, bound_frac=1e-2, dw_min=1e-12,
                dw0=1e-4, dw_max=1e40, kw_inc0=100, kw_inc=8, kw_dec=1.0 / 3, dc_bar=1e-7, kappa_c=0.25, gamma_theta=1e-5, gamma_phi=1e-8, delta=1,
                s_theta=1.1, s_phi=2.3, eta_phi=1e-8, gamma_alpha=0.05, s_max=100, kappa_sigma=1e10, constr_viol_tol=1e-4, dual_inf_tol=1,
                compl_inf_tol=1e-4, rho_term=1e3)
    for k, v in vals.items():
        setattr(o, k, v)
    return o


def parking_signed_dist_batch(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, dist=False, max_soc=0, recalc_y=0, lsq_init=0, restoration=0, **_):
    emu = load()
    x0 = np.reshape(x0, (-1, 4)); B = x0.shape[0]
    v = np.ravel(vOb).astype(int); nOb, M = len(v), int(v.sum()); Lz = P.layout(N, nOb, M)
    A = np.asarray(A, float).reshape(M, 2); b = np.ravel(np.asarray(b, float)); ego = np.ravel(np.asarray(ego, float))
    rl = P.row_lengths(A); An = A / rl[:, None]; bn = b / rl          # the kernels see unit-length rows (obca_hip.hip: batch_upload_range); lambda comes back rescaled
    g = np.array([(ego[0] + ego[2]) / 2, (ego[1] + ego[3]) / 2, (ego[0] + ego[2]) / 2, (ego[1] + ego[3]) / 2]); off = (ego[0] + ego[2]) / 2 - ego[2]
    eo = default_opts(); eo.max_soc = int(max_soc); eo.recalc_y = int(recalc_y); eo.lsq_init = int(lsq_init); eo.restoration = int(restoration); nsoc = np.zeros((B, 3), int)
    Tsv = np.broadcast_to(np.asarray(Ts, float), (B,))
    xp = np.zeros((B, 4, N + 1)); up = np.zeros((B, 2, N)); ts = np.zeros((B, N + 1)); ef = np.zeros(B, np.int32); info = np.zeros((B, 8))
    lps, nps, sls = [], [], []
    for i in range(B):
        rxi, ryi, rwi = np.ravel(rx[i])[:N + 1], np.ravel(ry[i])[:N + 1], np.ravel(ryaw[i])[:N + 1]
        lWS = np.zeros((N + 1, M)); nWS = np.zeros((N + 1, 4 * nOb))
        for k in range(N + 1):
            cs, sn = np.cos(rwi[k]), np.sin(rwi[k]); r0 = 0
            for j, vj in enumerate(v):
                a1 = np.ascontiguousarray(An[r0:r0 + vj, 0]); a2 = np.ascontiguousarray(An[r0:r0 + vj, 1]); bj = np.ascontiguousarray(bn[r0:r0 + vj])
                lam = np.zeros(8); mu = np.zeros(4); d = C.c_double(0)
                emu.emu_dualws(C.c_int(int(vj)), dp(a1), dp(a2), dp(bj), dp(g), C.c_double(rxi[k] + off * cs), C.c_double(ryi[k] + off * sn),
                               C.c_double(cs), C.c_double(sn), dp(lam), dp(mu), C.byref(d))
                lWS[k, r0:r0 + vj] = lam[:vj]; nWS[k, 4 * j:4 * j + 4] = mu; r0 += vj
        prob = P.pack_problem(x0[i], np.reshape(xF, (-1, 4))[i], N, Tsv[i], L, ego, XYbounds, v, A, b, rxi, ryi, rwi, fixTime, dist=int(dist))
        z0 = P.pack_start(N, nOb, M, np.asarray(xWS[i], float).reshape(-1, 4)[:N + 1], np.asarray(uWS[i], float).reshape(-1, 2)[:N], lWS, nWS)
        zo = np.zeros_like(z0)
        emu.emu_solve(C.c_int(N), dp(prob), dp(z0), C.c_int(Lz["len"]), C.byref(eo), dp(zo), dp(info[i]))
        acc_ = C.c_int(0); nsoc[i, 0] = emu.emu_last_soc(C.byref(acc_)); nsoc[i, 1] = acc_.value; nsoc[i, 2] = emu.emu_last_recalc()
        x_, u_, t_, lp_, np_, sl_ = P.unpack_solution(zo, N, nOb, M, A=A)
        xp[i] = x_; up[i] = u_; ts[i] = 1.0 if fixTime else t_; ef[i] = int(info[i, 7]); lps.append(lp_); nps.append(np_); sls.append(sl_)
    return dict(xp=xp, up=up, timeScale=ts, exitflag=ef, lp=lps, np=nps, sl=sls, info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int), nsoc=nsoc)


def quadcopter_signed_dist_batch(x0, xF, N, Ts, R, ob, xWS, timeWS, dual_ws=True, dist=False, max_soc=0, lsq_init=0, obj_scaling=0, **_):
    """the quadcopter kernel source (obca_quad_solver.h) as a host emulation, with the signature of obca_amd.quadcopter_signed_dist_batch"""
    emu = load()
    x0 = np.reshape(x0, (-1, 12)); B = x0.shape[0]; xF = np.reshape(xF, (-1, 12)); L = P.quad_layout(N); N1 = N + 1
    Tsv = np.broadcast_to(np.asarray(Ts, float), (B,)); tw = np.broadcast_to(np.asarray(timeWS, float), (B,))
    xp = np.zeros((B, 12, N1)); up = np.zeros((B, 4, N)); ts = np.zeros((B, N1)); ef = np.zeros(B, np.int32); info = np.zeros((B, 8)); lp = np.zeros((B, 30, N1)); sl = np.zeros((B, 5, N1))
    eo = default_opts(); eo.max_iter = 3000; eo.dw_min = 1e-10; eo.max_soc = int(max_soc); eo.lsq_init = int(lsq_init); eo.obj_scaling = int(obj_scaling)            # QuadcopterSignedDist.jl:28-31 (obca_quadcopter_default_opts)
    for i in range(B):
        prob = P.pack_quad_problem(x0[i], xF[i], N, Tsv[i], R, ob, np.asarray(xWS[i], float).reshape(N1, 12), tw[i], dual_ws=int(bool(dual_ws)), dist=int(bool(dist)))
        z = np.zeros(L["len"])
        emu.emu_quad_solve(C.c_int(N), dp(prob), C.byref(eo), dp(z), dp(info[i]))
        xp[i] = z[L["x"]:L["u"]].reshape(N1, 12).T; up[i] = z[L["u"]:L["t"]].reshape(N, 4).T; ts[i] = z[L["t"]]; ef[i] = int(info[i, 7])
        lp[i] = z[L["lam"]:L["s"]].reshape(N1, 30).T; sl[i] = z[L["s"]:L["so"]].reshape(N1, 5).T
    return dict(xp=xp, up=up, timeScale=ts, exitflag=ef, lp=lp, slack=sl, info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int))
