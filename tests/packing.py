"""
Host-side packing of OBCA parking instances into the flat fp64 buffers the HIP solver reads.

  prob[inst] = [ header (OB_HDR = 252 doubles: scalars, obstacle H-rep rows) | rx | ry | ryaw ]      (N+1 each)
  z[inst]    = one primal-dual iterate, stage-contiguous (= the reference's column-major x, u, l, n arrays):
               x 4(N+1) | u 2N | t | lam M(N+1) | mu 4nOb(N+1) | sl nOb(N+1) | so nOb(N+1) | ss N |
               pi 4N | nu 4 | yg N | yo 4nOb(N+1) | bound multipliers ...
The argument conventions are those of ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)
(/root/reference/AutonomousParking/ParkingSignedDist.jl:29).
"""
import numpy as np

OB_VMAX, OB_NOBMAX, OB_MMAX = 8, 16, 64      # obca_model.h
PH = dict(TS=0, L=1, G=2, OFF=6, XL=7, XU=11, X0=15, XF=19, FIX=23, NOB=24, M=25, VOB=26, ROFF=26 + OB_NOBMAX, DIST=26 + 2 * OB_NOBMAX + 1)      # obca_solver.h: PH_*
PH["A"] = PH["DIST"] + 1; PH["B"] = PH["A"] + 2 * OB_MMAX; OB_HDR = PH["B"] + OB_MMAX
LAYOUT_FIELDS = ("x u t lam mu sl so ss pi nu yg yo zxL zxU zuL zuU ztL ztU zlam zmu zso zssL zssU zs1 nprimal len").split()


def layout(N, nOb, M):
    N1 = N + 1
    sizes = [4 * N1, 2 * N, 1, M * N1, 4 * nOb * N1, nOb * N1, nOb * N1, N, 4 * N, 4, N, 4 * nOb * N1,
             4 * N1, 4 * N1, 2 * N, 2 * N, 1, 1, M * N1, 4 * nOb * N1, nOb * N1, N, N, nOb * N1]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    L = dict(zip(LAYOUT_FIELDS[:24], off[:24].tolist()))
    L["nprimal"] = int(off[8]); L["len"] = int(off[24])
    return L


def check_obstacles(vOb):
    vOb = np.asarray(vOb, dtype=np.int64).ravel()
    if len(vOb) < 1 or len(vOb) > OB_NOBMAX:
        raise ValueError(f"nOb must be in 1..{OB_NOBMAX}")
    if vOb.min() < 1 or vOb.max() > OB_VMAX:
        raise ValueError(f"rows per obstacle must be in 1..{OB_VMAX}")
    if vOb.sum() > OB_MMAX:
        raise ValueError(f"at most {OB_MMAX} half-space rows per instance")
    return vOb


def row_lengths(A):
    """|a_r| of the half-space rows (1 for a zero row): the kernels solve on a_r / |a_r|, b_r / |a_r|; lambda_r scales with |a_r|"""
    A = np.asarray(A, float).reshape(-1, 2); n = np.hypot(A[:, 0], A[:, 1])          # hypot as in the C packing
    return np.where(n > 0, n, 1.0)


def pack_problem(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, dist=0):
    vOb = check_obstacles(vOb)
    nOb, M = len(vOb), int(vOb.sum())
    A = np.asarray(A, float).reshape(M, 2); b = np.asarray(b, float).ravel()
    ego = np.asarray(ego, float).ravel(); XYb = np.asarray(XYbounds, float).ravel()
    p = np.zeros(OB_HDR + 3 * (N + 1))
    p[PH["TS"]] = Ts; p[PH["L"]] = L; p[PH["DIST"]] = float(int(dist))      # 1: ParkingDist.jl formulation
    W_ev, L_ev = ego[1] + ego[3], ego[0] + ego[2]                      # ParkingSignedDist.jl:182-188
    p[PH["G"]:PH["G"] + 4] = [L_ev / 2, W_ev / 2, L_ev / 2, W_ev / 2]
    p[PH["OFF"]] = (ego[0] + ego[2]) / 2 - ego[2]
    p[PH["XL"]:PH["XL"] + 4] = [XYb[0], XYb[2], -1e300, -1.0]          # :104-106
    p[PH["XU"]:PH["XU"] + 4] = [XYb[1], XYb[3], 1e300, 2.0]
    p[PH["X0"]:PH["X0"] + 4] = np.asarray(x0, float).ravel()
    p[PH["XF"]:PH["XF"] + 4] = np.asarray(xF, float).ravel()
    p[PH["FIX"]] = int(fixTime); p[PH["NOB"]] = nOb; p[PH["M"]] = M
    p[PH["VOB"]:PH["VOB"] + nOb] = vOb
    p[PH["ROFF"]:PH["ROFF"] + nOb + 1] = np.concatenate([[0], np.cumsum(vOb)])
    n = row_lengths(A)                                                  # unit-length rows, as the C ABI packs them (obca_hip.hip: batch_upload_range)
    p[PH["A"]:PH["A"] + 2 * M] = (A / n[:, None]).reshape(-1)
    p[PH["B"]:PH["B"] + M] = b / n
    p[OB_HDR:OB_HDR + N + 1] = np.asarray(rx, float).ravel()[:N + 1]
    p[OB_HDR + N + 1:OB_HDR + 2 * (N + 1)] = np.asarray(ry, float).ravel()[:N + 1]
    p[OB_HDR + 2 * (N + 1):] = np.asarray(ryaw, float).ravel()[:N + 1]
    return p


def pack_start(N, nOb, M, xWS, uWS, lWS, nWS, zlen=None, A=None):
    """warm start -> iterate buffer (reference :213-222: timeScale=1, x=xWS', u=uWS[1:N,:]', l=lWS', n=nWS').  A: the obstacle rows, if they are not of unit
    length (lWS comes in the caller's row scaling, the iterate holds lambda for unit rows)."""
    L = layout(N, nOb, M)
    z = np.zeros(zlen or L["len"])
    z[L["x"]:L["x"] + 4 * (N + 1)] = np.asarray(xWS, float)[:N + 1].reshape(-1)
    z[L["u"]:L["u"] + 2 * N] = np.asarray(uWS, float)[:N].reshape(-1)
    z[L["t"]] = 1.0
    lam = np.asarray(lWS, float).reshape(N + 1, M)
    z[L["lam"]:L["lam"] + M * (N + 1)] = (lam if A is None else lam * row_lengths(A)[None, :]).reshape(-1)
    z[L["mu"]:L["mu"] + 4 * nOb * (N + 1)] = np.asarray(nWS, float).reshape(-1)
    return z


def unpack_solution(z, N, nOb, M, A=None):
    """A: the obstacle rows, if not of unit length (lambda is handed back in the caller's row scaling)"""
    L = layout(N, nOb, M)
    xp = z[L["x"]:L["x"] + 4 * (N + 1)].reshape(N + 1, 4).T.copy()
    up = z[L["u"]:L["u"] + 2 * N].reshape(N, 2).T.copy()
    lp = z[L["lam"]:L["lam"] + M * (N + 1)].reshape(N + 1, M).T.copy()
    if A is not None:
        lp /= row_lengths(A)[:, None]
    npp = z[L["mu"]:L["mu"] + 4 * nOb * (N + 1)].reshape(N + 1, 4 * nOb).T.copy()
    sl = z[L["sl"]:L["sl"] + nOb * (N + 1)].reshape(N + 1, nOb).T.copy()
    return xp, up, float(z[L["t"]]), lp, npp, sl


# ---------------------------------------------------------------- quadcopter path (obca_amd/csrc/obca_quad_solver.h)
QPH_TS, QPH_R, QPH_X0, QPH_XF, QPH_OB, QPH_TWS, QPH_DWS, QPH_DIST, QPH_SIZE = 0, 1, 2, 14, 26, 56, 57, 58, 64
QUAD_NMAX = 128


def quad_layout(N):
    """iterate buffer of one quadcopter instance: v[n] | y[m] | zL[n] | zU[n] (q_make_layout in obca_quad_solver.h)"""
    N1 = N + 1; o = 0; L = {}
    for k, cnt in (("x", 12 * N1), ("u", 4 * N), ("t", 1), ("lam", 30 * N1), ("s", 5 * N1), ("so", 5 * N1)):
        L[k] = o; o += cnt
    L["n"] = o
    for k, cnt in (("pi", 12 * N), ("nu", 12), ("yo", 10 * N1)):
        L[k] = o; o += cnt
    L["m"] = o - L["n"]
    L["zL"] = o; o += L["n"]; L["zU"] = o; o += L["n"]; L["len"] = o
    return L


def quad_problem_len(N):
    return QPH_SIZE + 12 * (N + 1)


def pack_quad_problem(x0, xF, N, Ts, R, ob, xWS, timeWS, dual_ws=1, dist=0):
    """problem record of one quadcopter instance: header, then xWS stage-contiguous (N+1, 12). ob: 5 x 6 [max; -min]."""
    p = np.zeros(quad_problem_len(N))
    p[QPH_TS] = Ts; p[QPH_R] = R; p[QPH_X0:QPH_X0 + 12] = x0; p[QPH_XF:QPH_XF + 12] = xF
    p[QPH_OB:QPH_OB + 30] = np.asarray(ob, float).reshape(30); p[QPH_TWS] = timeWS; p[QPH_DWS] = float(int(dual_ws)); p[QPH_DIST] = float(int(dist))
    p[QPH_SIZE:] = np.asarray(xWS, float)[:N + 1].reshape(-1)
    return p
