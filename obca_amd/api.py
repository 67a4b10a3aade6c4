"""
ctypes binding of libobca_hip.so and the Python mirror of the reference's entry points for the signed-distance path:

    ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)
        -> xp (4,N+1), up (2,N), timeScale, exitflag, time, lp (M,N+1), np (4nOb,N+1)
        same positional arguments, shapes and exit-flag meaning as
        /root/reference/AutonomousParking/ParkingSignedDist.jl:29,297-313
    DualMultWS(N,nOb,vOb,A,b,rx,ry,ryaw, ego) -> lp (N+1,M), np (N+1,4nOb)
        /root/reference/AutonomousParking/DualMultWS.jl:29,81-84 (the reference reads `ego` from global scope, :39-45)

    QuadcopterSignedDist(x0,xF,N,Ts,R,ob1,ob2,ob3,ob4,ob5,xWS,uWS,timeWS)
        -> xp (12,N+1), up (4,N), timeScale, exitflag, time, lp (30,N+1), status string
        /root/reference/QuadcopterNavigation/QuadcopterSignedDist.jl:25,298

plus batched variants (leading batch dimension) that keep everything resident on the GPU between upload and download.
"""
import ctypes as C
import numbers
import os
import subprocess
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIBPATH = os.environ.get("OBCA_HIP_LIBRARY") or os.path.join(_CSRC, "libobca_hip.so")   # override: diagnostic builds
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int)
_lib = None


class ObcaError(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int)] + \
        [(n, C.c_double) for n in ("mu_init kappa_eps kappa_mu theta_mu tau_min bound_push bound_frac dw_min dw0 dw_max "
                                   "kw_inc0 kw_inc kw_dec dc_bar kappa_c gamma_theta gamma_phi delta s_theta s_phi eta_phi "
                                   "gamma_alpha s_max kappa_sigma constr_viol_tol dual_inf_tol compl_inf_tol rho_term").split()] + \
        [("max_soc", C.c_int), ("recalc_y", C.c_int), ("lsq_init", C.c_int), ("obj_scaling", C.c_int), ("restoration", C.c_int)]      # the IPOPT switches (include/obca_hip.h): 0 in default_opts(), 4 / 1 / 1 in ipopt_opts()


def library_path():
    return _LIBPATH


def build_library(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("obca_hip.hip", "obca_solver.h", "obca_solver_lanes.h", "obca_solver_assemble.h", "obca_solver_riccati.h", "obca_solver_direction.h", "obca_solver_ipm.h", "obca_model.h", "obca_quad_solver.h", "obca_quad_model.h")] + \
           [os.path.join(_HERE, "..", "include", "obca_hip.h"), os.path.join(_HERE, "buildflags.py")]      # (that file holds the compile flags)
    if not force and os.path.exists(_LIBPATH) and all(os.path.getmtime(_LIBPATH) >= os.path.getmtime(s) for s in srcs):
        return _LIBPATH
    from .buildflags import HIPCC      # the flags (warnings are errors) and why: obca_amd/buildflags.py
    cmd = HIPCC + ["-o", _LIBPATH, os.path.join(_CSRC, "obca_hip.hip")]
    subprocess.check_call(cmd)
    return _LIBPATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise ObcaError(f"{_LIBPATH} is missing: build it with obca_amd.build_library() / __graft_entry__.build(); "
                        "there is no CPU fallback")
    # The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and serialises the launches that share one: with more than four
    # batches in flight the extra streams bought nothing (rounds 3-6: 4 / 8 / 12 streams gave the same rate).  16 queues: +6 % on config 2, +8 % on config 3 with 16
    # batches in flight (profiles/r06_hw_queues.txt).  Read by the runtime at its first call in the process; a value the caller has set stays.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = C.CDLL(_LIBPATH)
    lib.obca_last_error.restype = C.c_char_p
    lib.obca_last_error.argtypes = [C.c_void_p]
    _lib = lib
    return lib


EXPORTS = ["obca_create", "obca_create_multi", "obca_device_count", "obca_visible_device_count", "obca_destroy", "obca_last_error", "obca_default_opts", "obca_reference_opts", "obca_device_name",
           "obca_dualmult_ws_batch", "obca_parking_signed_dist_batch", "obca_parking_dist_batch", "obca_batch_create", "obca_batch_destroy",
           "obca_batch_set_formulation", "obca_batch_shift_warm_start",
           "obca_batch_upload", "obca_batch_solve", "obca_batch_sync", "obca_batch_kernel_ms", "obca_batch_last_schedule", "obca_batch_download",
           "obca_batch_scratch_bytes",
           "obca_quadcopter_default_opts", "obca_quadcopter_reference_opts", "obca_quadcopter_signed_dist_batch", "obca_quadcopter_dist_batch", "obca_quad_batch_create", "obca_quad_batch_destroy",
           "obca_quad_batch_upload", "obca_quad_batch_solve", "obca_quad_batch_sync", "obca_quad_batch_kernel_ms",
           "obca_quad_batch_download", "obca_quad_batch_scratch_bytes"]


def selftest(device=0, repeats=4, opts=None):
    """Does this GPU return the same bits for the same inputs, whatever ran on it before?  The config-2 bench batch (1 024 instances, N = 80: every SIMD of the chip holds one) is
    solved `repeats` times as one device-resident batch and every download is compared bit for bit with the first; then a kernel leaves a large finite pattern in the LDS of every
    CU (obca_amd.diag.leave_pattern, libobca_diag.so) and the batch is solved once more.  Returns a dict: `differing` = (instance, run) pairs that differ, `after_pattern` = instances that differ
    after the pattern, `pattern_units` = per device (compute units the pattern kernel ran on, units that got the four workgroups which cover their whole LDS), `instances`, `runs`, `solved`, `device`.  Both counts are 0: the kernels contain no atomics and no order-dependent reductions, and read nothing they have not
    written (DESIGN.md sections 3 and 11 -- until the end of round 5 the multiplier sums of the parking kernels' termination test were read from LDS unwritten, and results changed with what other
    kernels had left there, e.g. when another process shared the GPU)."""
    from . import scenarios as S
    N, B = 80, 1024
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    ctx = device if isinstance(device, Context) else Context(device)
    b = Batch(ctx, B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])

    def differing(o, ref):
        return int(((o["info"] != ref["info"]).any(axis=1) | (np.abs(o["xp"] - ref["xp"]).reshape(B, -1).max(axis=1) > 0)).sum())
    ref = None; bad = 0
    for _ in range(max(2, repeats)):
        b.solve(opts=opts); o = b.download()
        if ref is None:
            ref = o; continue
        bad += differing(o, ref)
    from . import diag
    reach = diag.leave_pattern(ctx, 4, 1e30)      # (a diagnostic library of its own: libobca_diag.so)
    b.solve(opts=opts); after = differing(b.download(), ref)
    name = ctx.name()
    b.close()
    if not isinstance(device, Context):
        ctx.close()
    return dict(differing=bad, after_pattern=after, pattern_units=reach, instances=B, runs=max(2, repeats), solved=int((ref["exitflag"] == 1).sum()), device=name)


def warm_restart_opts():
    """options for a solve that starts from a (shifted) previous solution: small initial barrier and bound push, so the interior point
    does not first walk away from the active bounds (16 instead of 47 iterations on the config-2 batch)"""
    o = default_opts(); o.mu_init = 1e-4; o.bound_push = 1e-4; o.bound_frac = 1e-4
    return o


def default_opts():
    o = Opts()
    _load().obca_default_opts(C.byref(o))
    return o


def ipopt_opts():
    """the reference's IPOPT configuration as far as the kernels carry it: default options + second-order correction (IPOPT's default max_soc = 4), recalc_y = "yes"
    (ParkingSignedDist.jl:41), IPOPT's least-squares initial multipliers, and the block feasibility restoration that stands in for IPOPT's restoration phase (restoration = 1)"""
    o = Opts()
    _load().obca_reference_opts(C.byref(o))
    return o


def _d(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_D)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I)


class Context:
    """One device (`device=k`), an explicit list (`devices=[...]`) or every visible device (`devices="all"`): the host-pointer entry points
    of a multi-device context shard their batch over the devices through a work queue (include/obca_hip.h, obca_create_multi)."""

    def __init__(self, device=0, devices=None):
        lib = _load()
        self._h = C.c_void_p()
        if devices is None:
            rc = lib.obca_create(C.byref(self._h), C.c_int(int(device)))
            self.devices = [int(device)]
        else:
            lst = [] if isinstance(devices, str) else [int(d) for d in devices]
            arr = (C.c_int * max(1, len(lst)))(*lst)
            rc = lib.obca_create_multi(C.byref(self._h), arr if lst else None, C.c_int(len(lst)))
            self.devices = lst
        if rc != 0:
            raise ObcaError("obca_create failed: " + (lib.obca_last_error(None) or b"").decode())
        if devices is not None:
            self.devices = list(range(lib.obca_device_count(self._h))) if not self.devices else self.devices
        self.device = self.devices[0]

    def device_count(self):
        return int(_load().obca_device_count(self._h))

    def _check(self, rc, what):
        if rc != 0:
            raise ObcaError(f"{what} failed ({rc}): " + (_load().obca_last_error(self._h) or b"").decode())

    def name(self):
        buf = C.create_string_buffer(256)
        _load().obca_device_name(self._h, buf, 256)
        return buf.value.decode()

    def close(self):
        if self._h:
            _load().obca_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def _ctx(device=0):
    """cached context of one device index, of a tuple of indices, or of "all" visible devices"""
    if isinstance(device, Context):
        return device
    if isinstance(device, numbers.Integral):          # np.int64 local ranks, torch scalars' .item(), ...
        device = int(device)
    key = device if isinstance(device, (int, str)) else tuple(int(d) for d in device)
    if key not in _default_ctx:
        _default_ctx[key] = Context(device) if isinstance(key, int) else Context(devices=device)
    return _default_ctx[key]


def _norm_obstacles(B, vOb, A, b):
    """Accept one shared obstacle set (vOb 1-D, A (M,2)) or per-instance lists; return packed per-instance arrays."""
    if isinstance(vOb, (list, tuple)) and len(vOb) == B and np.ndim(vOb[0]) >= 1:
        nObs = np.array([len(np.ravel(v)) for v in vOb], np.int32)
        vflat = np.concatenate([np.ravel(v) for v in vOb]).astype(np.int32)
        Aflat = np.concatenate([np.asarray(a, float).reshape(-1, 2) for a in A])
        bflat = np.concatenate([np.ravel(np.asarray(x, float)) for x in b])
        return nObs, vflat, Aflat, bflat
    v = np.ravel(np.asarray(vOb)).astype(np.int32)
    M = int(v.sum())
    A = np.asarray(A, float).reshape(M, 2); b = np.ravel(np.asarray(b, float))
    return np.full(B, len(v), np.int32), np.tile(v, B), np.tile(A, (B, 1)), np.tile(b, B)


def _warm_start(xWS, uWS, B, N):
    """xWS (B, >= N+1, 4), uWS (B, >= N, 2) cut to the horizon -- checked here, because the C side reads N+1 / N rows through raw pointers"""
    xWS = np.asarray(xWS, float).reshape(B, -1, 4); uWS = np.asarray(uWS, float).reshape(B, -1, 2)
    if xWS.shape[1] < N + 1 or uWS.shape[1] < N:
        raise ObcaError(f"warm start too short: xWS has {xWS.shape[1]} stages (need N+1 = {N + 1}), uWS {uWS.shape[1]} (need N = {N})")
    return xWS[:, :N + 1], uWS[:, :N]


def _dual_start(lWS, nWS, Mt, nt, N):
    """optional dual warm start, packed per instance ((N+1) x M_i, (N+1) x 4 nOb_i blocks); sizes checked before raw pointers go to C"""
    if lWS is None or nWS is None:
        return None, None
    if not isinstance(lWS, np.ndarray):
        lWS = np.concatenate([np.ravel(x) for x in lWS])
    if not isinstance(nWS, np.ndarray):
        nWS = np.concatenate([np.ravel(x) for x in nWS])
    if lWS.size != Mt * (N + 1) or nWS.size != 4 * nt * (N + 1):
        raise ObcaError(f"dual warm start has the wrong size: lWS {lWS.size} (need {Mt * (N + 1)}), nWS {nWS.size} (need {4 * nt * (N + 1)})")
    return lWS, nWS


def _row_counts(nObs, vflat):
    """half-space rows per instance: segment sums of vflat (one numpy call -- a Python loop over 16 384 instances cost a third of the wrapper's time)"""
    nObs = np.asarray(nObs, np.int64)
    csum = np.concatenate([[0], np.cumsum(np.asarray(vflat, np.int64))])
    ends = np.cumsum(nObs)
    return csum[ends] - csum[ends - nObs]


class Batch:
    """Device-resident batch: upload once, solve (repeatedly), download."""

    def __init__(self, ctx, B, N):
        self.ctx, self.B, self.N = ctx, int(B), int(N)
        self._h = C.c_void_p()
        ctx._check(_load().obca_batch_create(ctx._h, C.c_int(self.B), C.c_int(self.N), C.byref(self._h)), "obca_batch_create")

    def upload(self, x0, xF, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, lWS=None, nWS=None, dist=False):
        self.ctx._check(_load().obca_batch_set_formulation(self._h, C.c_int(int(bool(dist)))), "obca_batch_set_formulation")
        """lWS/nWS, when given, are packed per instance as (N+1, M_i) / (N+1, 4 nOb_i) row-major blocks (= the reference's
        column-major l (M x N+1) and n (4nOb x N+1))."""
        B, N = self.B, self.N
        nObs, vflat, Aflat, bflat = _norm_obstacles(B, vOb, A, b)
        self.nObs, self.vflat = nObs, vflat
        self.Ms = _row_counts(nObs, vflat)
        Ts = np.broadcast_to(np.asarray(Ts, float), (B,))
        lWS, nWS = _dual_start(lWS, nWS, int(self.Ms.sum()), int(nObs.sum()), N)
        keep = [_d(Ts), _d(ego), _d(XYbounds), _d(np.reshape(x0, (B, 4))), _d(np.reshape(xF, (B, 4))), _i(nObs), _i(vflat),
                _d(Aflat), _d(bflat), _d(np.reshape(rx, (B, N + 1))), _d(np.reshape(ry, (B, N + 1))), _d(np.reshape(ryaw, (B, N + 1))),
                *(_d(w) for w in _warm_start(xWS, uWS, B, N)),
                _d(lWS), _d(nWS)]
        p = [k[1] for k in keep]
        rc = _load().obca_batch_upload(self._h, p[0], C.c_double(float(L)), p[1], p[2], C.c_int(int(fixTime)), p[3], p[4], p[5], p[6],
                                       p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15])
        self.ctx._check(rc, "obca_batch_upload")

    def solve(self, opts=None, sync=True):
        self.ctx._check(_load().obca_batch_solve(self._h, C.byref(opts) if opts is not None else None), "obca_batch_solve")
        if sync:
            self.sync()

    def sync(self):
        self.ctx._check(_load().obca_batch_sync(self._h), "obca_batch_sync")

    def shift_warm_start(self, shift, x0_new=None):
        """receding-horizon restart: the next solve starts from the last solution advanced by `shift` stages (kept on the device)."""
        keep = _d(np.reshape(x0_new, (self.B, 4))) if x0_new is not None else None
        rc = _load().obca_batch_shift_warm_start(self._h, C.c_int(int(shift)), keep[1] if keep else None)
        self.ctx._check(rc, "obca_batch_shift_warm_start")

    def kernel_ms(self):
        """(ipm_ms, dualws_ms) of the last solve, measured with HIP events on the launch stream."""
        a, b = C.c_float(0), C.c_float(0)
        self.ctx._check(_load().obca_batch_kernel_ms(self._h, C.byref(a), C.byref(b)), "obca_batch_kernel_ms")
        return a.value, b.value

    def last_schedule(self):
        """(ipm launches, slice passes) of the last solve: (1, 0) single launch, (2, q) the two-launch schedule with a q-pass first slice"""
        a, b = C.c_int(0), C.c_int(0)
        self.ctx._check(_load().obca_batch_last_schedule(self._h, C.byref(a), C.byref(b)), "obca_batch_last_schedule")
        return a.value, b.value

    def phase_cycles(self):
        """(B,16) per-phase shader-cycle counters of the last solve: exists in the profiling build only (OBCA_HIP_LIBRARY=.../libobca_hip_prof.so, tools/phase_profile.py)."""
        out = np.zeros((self.B, 16))
        if not hasattr(_load(), "obca_batch_debug_phase_cycles"):
            raise ObcaError("phase_cycles(): the loaded library is not the -DOBCA_PROFILE build")
        self.ctx._check(_load().obca_batch_debug_phase_cycles(self._h, out.ctypes.data_as(_D)), "obca_batch_debug_phase_cycles")
        return out

    def scratch_bytes(self):
        v = C.c_longlong(0)
        _load().obca_batch_scratch_bytes(self._h, C.byref(v))
        return v.value

    def download(self):
        B, N = self.B, self.N
        Mt, nt = int(self.Ms.sum()), int(self.nObs.sum())
        xp = np.zeros((B, N + 1, 4)); up = np.zeros((B, N, 2)); ts = np.zeros((B, N + 1)); ef = np.zeros(B, np.int32)
        lp = np.zeros(Mt * (N + 1)); npp = np.zeros(4 * nt * (N + 1)); sl = np.zeros(nt * (N + 1)); info = np.zeros((B, 8))
        rc = _load().obca_batch_download(self._h, xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D),
                                         ef.ctypes.data_as(_I), lp.ctypes.data_as(_D), npp.ctypes.data_as(_D), sl.ctypes.data_as(_D),
                                         info.ctypes.data_as(_D))
        self.ctx._check(rc, "obca_batch_download")
        return _unpack_parking(B, N, self.nObs, self.Ms, xp, up, ts, ef, lp, npp, sl, info)

    def close(self):
        if self._h:
            _load().obca_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parking_signed_dist_batch(x0, xF, N, Ts, L, ego, XYbounds, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, lWS=None, nWS=None,
                              opts=None, device=0, dist=False, buffers=None):
    """Batched ParkingSignedDist through the host-pointer entry point obca_parking_(signed_)dist_batch (what the Julia shim calls):
    x0,xF (B,4); rx,ry,ryaw (B,N+1); xWS (B,N+1,4); uWS (B,>=N,2); Ts scalar or (B,).
    Obstacles: one shared set (vOb 1-D, A (M,2), b (M,)) or per-instance lists.  lWS/nWS=None runs DualMultWS on the GPU.
    `device`: an index, a list of indices or "all" (the batch is then sharded over the devices, obca_create_multi), or a Context.
    `buffers`: a dict the caller keeps between calls; the output arrays live in it and are written again by the next call of the same shape (a
    16 384-instance call returns 280 MB: fresh arrays cost a page fault per 4 KB inside the C call, tools/pcie_rate.py measures both)."""
    x0 = np.ascontiguousarray(np.reshape(x0, (-1, 4)), float); B = x0.shape[0]
    ctx = _ctx(device)
    nObs, vflat, Aflat, bflat = _norm_obstacles(B, vOb, A, b)
    Ms = _row_counts(nObs, vflat)
    Mt, nt = int(Ms.sum()), int(nObs.sum())
    Tsv = np.ascontiguousarray(np.broadcast_to(np.asarray(Ts, float), (B,)))
    lWS, nWS = _dual_start(lWS, nWS, Mt, nt, N)
    keep = [_d(Tsv), _d(ego), _d(XYbounds), _d(x0), _d(np.reshape(xF, (B, 4))), _i(nObs), _i(vflat), _d(Aflat), _d(bflat),
            _d(np.reshape(rx, (B, N + 1))), _d(np.reshape(ry, (B, N + 1))), _d(np.reshape(ryaw, (B, N + 1))),
            *(_d(w) for w in _warm_start(xWS, uWS, B, N)), _d(lWS), _d(nWS)]
    p = [k[1] for k in keep]
    shapes = dict(xp=(B, N + 1, 4), up=(B, N, 2), ts=(B, N + 1), lp=(Mt * (N + 1),), npp=(4 * nt * (N + 1),), sl=(nt * (N + 1),), info=(B, 8))
    bufs = buffers if buffers is not None else {}
    for k, shp in shapes.items():
        if k not in bufs or bufs[k].shape != shp:
            bufs[k] = np.zeros(shp) if k in ("sl", "info") else np.empty(shp)
    if "ef" not in bufs or bufs["ef"].shape != (B,):
        bufs["ef"] = np.zeros(B, np.int32)
    xp, up, ts, ef, lp, npp, sl, info = (bufs[k] for k in ("xp", "up", "ts", "ef", "lp", "npp", "sl", "info"))
    lib = _load()
    t0 = time.perf_counter()
    if dist:
        rc = lib.obca_parking_dist_batch(ctx._h, C.c_int(B), C.c_int(int(N)), p[0], C.c_double(float(L)), p[1], p[2], C.c_int(int(fixTime)), p[3], p[4],
                                         p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], C.byref(opts) if opts is not None else None,
                                         xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D), ef.ctypes.data_as(_I), lp.ctypes.data_as(_D),
                                         npp.ctypes.data_as(_D), info.ctypes.data_as(_D))
    else:
        rc = lib.obca_parking_signed_dist_batch(ctx._h, C.c_int(B), C.c_int(int(N)), p[0], C.c_double(float(L)), p[1], p[2], C.c_int(int(fixTime)), p[3], p[4],
                                                p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15],
                                                C.byref(opts) if opts is not None else None, xp.ctypes.data_as(_D), up.ctypes.data_as(_D),
                                                ts.ctypes.data_as(_D), ef.ctypes.data_as(_I), lp.ctypes.data_as(_D), npp.ctypes.data_as(_D),
                                                sl.ctypes.data_as(_D), info.ctypes.data_as(_D))
    dt = time.perf_counter() - t0
    ctx._check(rc, "obca_parking_dist_batch" if dist else "obca_parking_signed_dist_batch")
    out = _unpack_parking(B, N, nObs, Ms, xp, up, ts, ef, lp, npp, sl, info)
    out["time"] = dt
    return out


def _unpack_parking(B, N, nObs, Ms, xp, up, ts, ef, lp, npp, sl, info):
    """C-ABI output arrays -> the reference's shapes: xp (B,4,N+1), up (B,2,N), per-instance lp (M,N+1) / np (4nOb,N+1) / sl (nOb,N+1): lists for ragged obstacle sets, (B, ., N+1) arrays for uniform ones"""
    if len(set(Ms.tolist())) == 1 and len(set(nObs.tolist())) == 1:          # uniform obstacle sets: one reshape, views per instance
        m, n = int(Ms[0]), int(nObs[0])
        L3 = lp.reshape(B, N + 1, m).transpose(0, 2, 1); N3 = npp.reshape(B, N + 1, 4 * n).transpose(0, 2, 1); S3 = sl.reshape(B, N + 1, n).transpose(0, 2, 1)
        lps, nps, sls = L3, N3, S3                                             # (B, M, N+1) views: lp[i] is instance i's (M, N+1) array, as in the ragged case -- no 3 x B Python objects
    else:
        lps, nps, sls = [], [], []
        ro = oo = 0
        for m, n in zip(Ms, nObs):
            lps.append(lp[ro * (N + 1):(ro + m) * (N + 1)].reshape(N + 1, m).T)
            nps.append(npp[4 * oo * (N + 1):4 * (oo + n) * (N + 1)].reshape(N + 1, 4 * n).T)
            sls.append(sl[oo * (N + 1):(oo + n) * (N + 1)].reshape(N + 1, n).T)
            ro += m; oo += n
    return dict(xp=np.transpose(xp, (0, 2, 1)), up=np.transpose(up, (0, 2, 1)), timeScale=ts, exitflag=ef,
                lp=lps, np=nps, sl=sls, info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int))


def ParkingSignedDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, opts=None, device=0):
    """Drop-in for ParkingSignedDist.jl:29 (one instance).  Returns (xp, up, timeScalep, exitflag, time, lp, np).
    opts=None runs the reference's IPOPT configuration (ipopt_opts(): recalc_y = "yes" as ParkingSignedDist.jl:41 sets it, IPOPT's default second-order correction and
    least-squares initial multipliers); pass default_opts() for the library's throughput defaults (include/obca_hip.h says what the difference costs and changes)."""
    assert int(nOb) == len(np.ravel(vOb))
    opts = ipopt_opts() if opts is None else opts
    r = parking_signed_dist_batch(np.reshape(x0, (1, 4)), np.reshape(xF, (1, 4)), N, Ts, L, ego, XYbounds, vOb, A, b,
                                  np.reshape(np.ravel(rx)[:N + 1], (1, -1)), np.reshape(np.ravel(ry)[:N + 1], (1, -1)),
                                  np.reshape(np.ravel(ryaw)[:N + 1], (1, -1)), fixTime, np.asarray(xWS, float)[None, :N + 1],
                                  np.asarray(uWS, float)[None, :N], opts=opts, device=device)
    ts = np.ones((1, N + 1)) if fixTime else r["timeScale"][0]          # ParkingSignedDist.jl:304-308
    return r["xp"][0], r["up"][0], ts, int(r["exitflag"][0]), r["time"], r["lp"][0], r["np"][0]


def ParkingDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS, opts=None, device=0):
    """Drop-in for ParkingDist.jl:29 (the collision-free sibling of ParkingSignedDist): same arguments, same 7-tuple.  opts=None: the reference's IPOPT configuration
    (ParkingDist.jl:41 sets recalc_y = "yes" as well), see ParkingSignedDist."""
    assert int(nOb) == len(np.ravel(vOb))
    opts = ipopt_opts() if opts is None else opts
    r = parking_signed_dist_batch(np.reshape(x0, (1, 4)), np.reshape(xF, (1, 4)), N, Ts, L, ego, XYbounds, vOb, A, b,
                                  np.reshape(np.ravel(rx)[:N + 1], (1, -1)), np.reshape(np.ravel(ry)[:N + 1], (1, -1)),
                                  np.reshape(np.ravel(ryaw)[:N + 1], (1, -1)), fixTime, np.asarray(xWS, float)[None, :N + 1],
                                  np.asarray(uWS, float)[None, :N], opts=opts, device=device, dist=True)
    ts = np.ones((1, N + 1)) if fixTime else r["timeScale"][0]
    return r["xp"][0], r["up"][0], ts, int(r["exitflag"][0]), r["time"], r["lp"][0], r["np"][0]


def dualmult_ws_batch(N, vOb, A, b, rx, ry, ryaw, ego, device=0):
    """Batched DualMultWS: rx,ry,ryaw (B,N+1) -> lWS list of (N+1,M), nWS list of (N+1,4nOb), d list of (N+1,nOb)."""
    rx = np.atleast_2d(np.asarray(rx, float)); B = rx.shape[0]
    ctx = _ctx(device)
    nObs, vflat, Aflat, bflat = _norm_obstacles(B, vOb, A, b)
    Ms = _row_counts(nObs, vflat)
    Mt, nt = int(Ms.sum()), int(nObs.sum())
    lw = np.zeros(Mt * (N + 1)); nw = np.zeros(4 * nt * (N + 1)); dd = np.zeros(nt * (N + 1))
    keep = [_d(ego), _i(nObs), _i(vflat), _d(Aflat), _d(bflat), _d(rx), _d(np.atleast_2d(ry)), _d(np.atleast_2d(ryaw))]
    p = [k[1] for k in keep]
    rc = _load().obca_dualmult_ws_batch(ctx._h, C.c_int(B), C.c_int(N), p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7],
                                        lw.ctypes.data_as(_D), nw.ctypes.data_as(_D), dd.ctypes.data_as(_D))
    ctx._check(rc, "obca_dualmult_ws_batch")
    ls, ns, ds = [], [], []
    ro = oo = 0
    for m, n in zip(Ms, nObs):
        ls.append(lw[ro * (N + 1):(ro + m) * (N + 1)].reshape(N + 1, m).copy())
        ns.append(nw[4 * oo * (N + 1):4 * (oo + n) * (N + 1)].reshape(N + 1, 4 * n).copy())
        ds.append(dd[oo * (N + 1):(oo + n) * (N + 1)].reshape(N + 1, n).copy())
        ro += m; oo += n
    return ls, ns, ds


def DualMultWS(N, nOb, vOb, A, b, rx, ry, ryaw, ego, device=0):
    """Drop-in for DualMultWS.jl:29 -> (lp (N+1,M), np (N+1,4nOb)); `ego` is explicit (the reference uses a global)."""
    assert int(nOb) == len(np.ravel(vOb))
    ls, ns, _ = dualmult_ws_batch(N, vOb, A, b, np.ravel(rx)[None, :N + 1], np.ravel(ry)[None, :N + 1],
                                  np.ravel(ryaw)[None, :N + 1], ego, device)
    return ls[0], ns[0]


# ---------------------------------------------------------------- quadcopter path (QuadcopterSignedDist.jl)
def quadcopter_default_opts():
    o = Opts()
    _load().obca_quadcopter_default_opts(C.byref(o))
    return o


def quadcopter_ipopt_opts():
    """the reference's IPOPT configuration of the quadcopter call as far as the kernel carries it (obca_quadcopter_reference_opts: max_soc = 4, least-squares
    initial multipliers, gradient-based objective scaling; recalc_y = "no" as QuadcopterSignedDist.jl:29 sets it): the default of the drop-ins QuadcopterSignedDist / QuadcopterDist"""
    o = Opts()
    _load().obca_quadcopter_reference_opts(C.byref(o))
    return o


class QuadBatch:
    """Device-resident batch of quadcopter signed-distance NLPs (obca_quad_batch_* in include/obca_hip.h)."""

    def __init__(self, ctx, B, N):
        self.ctx, self.B, self.N = ctx, int(B), int(N)
        self._h = C.c_void_p()
        ctx._check(_load().obca_quad_batch_create(ctx._h, C.c_int(self.B), C.c_int(self.N), C.byref(self._h)), "obca_quad_batch_create")

    def upload(self, x0, xF, Ts, R, ob, xWS, timeWS, dual_ws=True, dist=False):
        B, N = self.B, self.N
        Tsv = np.broadcast_to(np.asarray(Ts, float), (B,)).copy(); tw = np.broadcast_to(np.asarray(timeWS, float), (B,)).copy()
        obv = np.broadcast_to(np.asarray(ob, float).reshape(-1, 30) if np.size(ob) != 30 else np.asarray(ob, float).reshape(1, 30), (B, 30)).copy()
        xw = np.ascontiguousarray(np.asarray(xWS, float)[:, :N + 1]); assert xw.shape == (B, N + 1, 12)
        keep = [_d(Tsv), _d(np.reshape(x0, (B, 12))), _d(np.reshape(xF, (B, 12))), _d(obv), _d(xw), _d(tw)]
        p = [k[1] for k in keep]
        rc = _load().obca_quad_batch_upload(self._h, p[0], C.c_double(R), p[1], p[2], p[3], p[4], p[5], C.c_int(int(bool(dual_ws))), C.c_int(int(bool(dist))))
        self.ctx._check(rc, "obca_quad_batch_upload")

    def solve(self, opts=None, sync=True):
        self.ctx._check(_load().obca_quad_batch_solve(self._h, C.byref(opts) if opts is not None else None), "obca_quad_batch_solve")
        if sync:
            self.sync()

    def sync(self):
        self.ctx._check(_load().obca_quad_batch_sync(self._h), "obca_quad_batch_sync")

    def kernel_ms(self):
        a = C.c_float(0)
        self.ctx._check(_load().obca_quad_batch_kernel_ms(self._h, C.byref(a)), "obca_quad_batch_kernel_ms")
        return a.value

    def scratch_bytes(self):
        v = C.c_longlong(0)
        _load().obca_quad_batch_scratch_bytes(self._h, C.byref(v))
        return v.value

    def phase_cycles(self):
        out = np.zeros((self.B, 16))
        if not hasattr(_load(), "obca_quad_batch_debug_phase_cycles"):
            raise ObcaError("phase_cycles(): the loaded library is not the -DOBCA_PROFILE build")
        self.ctx._check(_load().obca_quad_batch_debug_phase_cycles(self._h, out.ctypes.data_as(_D)), "obca_quad_batch_debug_phase_cycles")
        return out

    def download(self):
        B, N = self.B, self.N
        xp = np.zeros((B, N + 1, 12)); up = np.zeros((B, N, 4)); ts = np.zeros((B, N + 1)); ef = np.zeros(B, np.int32)
        lp = np.zeros((B, N + 1, 30)); sl = np.zeros((B, N + 1, 5)); info = np.zeros((B, 8))
        rc = _load().obca_quad_batch_download(self._h, xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D), ef.ctypes.data_as(_I),
                                              lp.ctypes.data_as(_D), sl.ctypes.data_as(_D), info.ctypes.data_as(_D))
        self.ctx._check(rc, "obca_quad_batch_download")
        T = lambda a: np.transpose(a, (0, 2, 1)).copy()
        return dict(xp=T(xp), up=T(up), timeScale=ts, exitflag=ef, lp=T(lp), slack=T(sl), info=info, iters=info[:, 1].astype(int),
                    obj=info[:, 2], status=info[:, 0].astype(int))

    def close(self):
        if self._h:
            _load().obca_quad_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def quadcopter_signed_dist_batch(x0, xF, N, Ts, R, ob, xWS, timeWS, dual_ws=True, opts=None, device=0, dist=False):
    """Batched QuadcopterSignedDist / QuadcopterDist through the host-pointer entry points (what the Julia shim calls):
    x0,xF (B,12); ob (5,6) shared or (B,5,6); xWS (B,N+1,12); Ts, timeWS scalar or (B,).  `device` as in parking_signed_dist_batch."""
    x0 = np.ascontiguousarray(np.reshape(x0, (-1, 12)), float); B = x0.shape[0]
    ctx = _ctx(device)
    Tsv = np.broadcast_to(np.asarray(Ts, float), (B,)).copy(); tw = np.broadcast_to(np.asarray(timeWS, float), (B,)).copy()
    obv = np.broadcast_to(np.asarray(ob, float).reshape(-1, 30) if np.size(ob) != 30 else np.asarray(ob, float).reshape(1, 30), (B, 30)).copy()
    xw = np.ascontiguousarray(np.asarray(xWS, float)[:, :N + 1]); assert xw.shape == (B, N + 1, 12)
    keep = [_d(Tsv), _d(x0), _d(np.reshape(xF, (B, 12))), _d(obv), _d(xw), _d(tw)]
    p = [k[1] for k in keep]
    xp = np.empty((B, N + 1, 12)); up = np.empty((B, N, 4)); ts = np.empty((B, N + 1)); ef = np.zeros(B, np.int32)
    lp = np.empty((B, N + 1, 30)); sl = np.zeros((B, N + 1, 5)); info = np.zeros((B, 8))
    lib = _load(); o = C.byref(opts) if opts is not None else None
    t0 = time.perf_counter()
    if dist:
        rc = lib.obca_quadcopter_dist_batch(ctx._h, C.c_int(B), C.c_int(int(N)), p[0], C.c_double(float(R)), p[1], p[2], p[3], p[4], None, p[5],
                                            C.c_int(int(bool(dual_ws))), o, xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D),
                                            ef.ctypes.data_as(_I), lp.ctypes.data_as(_D), info.ctypes.data_as(_D))
    else:
        rc = lib.obca_quadcopter_signed_dist_batch(ctx._h, C.c_int(B), C.c_int(int(N)), p[0], C.c_double(float(R)), p[1], p[2], p[3], p[4], None, p[5],
                                                   C.c_int(int(bool(dual_ws))), o, xp.ctypes.data_as(_D), up.ctypes.data_as(_D), ts.ctypes.data_as(_D),
                                                   ef.ctypes.data_as(_I), lp.ctypes.data_as(_D), sl.ctypes.data_as(_D), info.ctypes.data_as(_D))
    dt = time.perf_counter() - t0
    ctx._check(rc, "obca_quadcopter_dist_batch" if dist else "obca_quadcopter_signed_dist_batch")
    T = lambda a: np.transpose(a, (0, 2, 1))
    return dict(xp=T(xp), up=T(up), timeScale=ts, exitflag=ef, lp=T(lp), slack=T(sl), info=info, iters=info[:, 1].astype(int),
                obj=info[:, 2], status=info[:, 0].astype(int), time=dt)


_QUAD_STATUS = {0: "Optimal", 1: "UserLimit", 2: "Error"}


def QuadcopterSignedDist(x0, xF, N, Ts, R, ob1, ob2, ob3, ob4, ob5, xWS, uWS, timeWS, opts=None, device=0, dual_ws=True):
    """Drop-in for QuadcopterSignedDist.jl:25 (one instance; xWS is (N+1,12) here, the reference's is 12 x (N+1) column-major,
    i.e. the same memory).  Returns (xp, up, timeScalep, exitflag, time, lp, status) like :298; uWS is ignored like :202."""
    ob = np.stack([np.ravel(o)[:6] for o in (ob1, ob2, ob3, ob4, ob5)])
    opts = quadcopter_ipopt_opts() if opts is None else opts      # the drop-in runs the reference's IPOPT configuration; the batched calls default to the throughput options
    r = quadcopter_signed_dist_batch(np.reshape(x0, (1, 12)), np.reshape(xF, (1, 12)), N, Ts, R, ob, np.asarray(xWS, float)[None, :N + 1],
                                     timeWS, dual_ws, opts, device)
    return r["xp"][0], r["up"][0], r["timeScale"][0], int(r["exitflag"][0]), r["time"], r["lp"][0], _QUAD_STATUS[int(r["status"][0])]


def QuadcopterDist(x0, xF, N, Ts, R, ob1, ob2, ob3, ob4, ob5, xWS, uWS, timeWS, opts=None, device=0, dual_ws=True):
    """Drop-in for QuadcopterDist.jl:25 (the collision-free sibling: no slack variable): same arguments and 7-tuple as QuadcopterSignedDist."""
    ob = np.stack([np.ravel(o)[:6] for o in (ob1, ob2, ob3, ob4, ob5)])
    opts = quadcopter_ipopt_opts() if opts is None else opts      # the drop-in runs the reference's IPOPT configuration; the batched calls default to the throughput options
    r = quadcopter_signed_dist_batch(np.reshape(x0, (1, 12)), np.reshape(xF, (1, 12)), N, Ts, R, ob, np.asarray(xWS, float)[None, :N + 1],
                                     timeWS, dual_ws, opts, device, dist=True)
    return r["xp"][0], r["up"][0], r["timeScale"][0], int(r["exitflag"][0]), r["time"], r["lp"][0], _QUAD_STATUS[int(r["status"][0])]
