"""
Multi-process sharding of one host batch (one process per GPU, `torch.distributed`: backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  SURVEY.md section 8e: problem instances are independent, so the only exchanges are ONE scatter of the inputs and ONE
gather of the outputs per batch; nothing touches the solve.

  rank `src` holds the whole batch (host arrays, leading dimension B)
    -> scatter_rows : rank r receives the contiguous slice [r*ceil(B/G), (r+1)*ceil(B/G))
    -> every rank solves its slice with the HIP path (obca_amd.parking_signed_dist_batch / quadcopter_signed_dist_batch on its device;
       inside a rank the C ABI cuts the slice into chunks and pipelines them, include/obca_hip.h)
    -> gather_rows  : rank `src` gets the reference's FULL result tuple per instance (xp, up, timeScale, exitflag, lp, np, + sl, info)

(Within ONE process the same split is done by a multi-device context -- obca_create_multi -- with a work queue instead of static slices;
this module is the route for launchers that start one process per GPU, e.g. `torchrun bench.py`.)
"""
import numpy as np


def shard_range(B, rank, world):
    per = -(-B // world)
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def balanced_permutation(keys, world):
    """Order in which a batch of UNEQUAL instances is handed to the contiguous slices of scatter_rows (BASELINE config 5: 1-10 obstacles per instance): the
    instances are bucketed by `keys` (any per-instance cost key, e.g. (nOb, M) as one number) and the buckets dealt round-robin, so that every rank receives the same mix.
    Returns `perm` (rank r gets the instances perm[r*ceil(B/G) : (r+1)*ceil(B/G)]) and its inverse (to put the gathered results back in the caller's order)."""
    keys = np.asarray(keys); B = len(keys); per = -(-B // world)
    order = np.argsort(keys, kind="stable")                         # buckets, in ascending cost
    perm = np.empty(B, np.int64); fill = [0] * world
    for j, i in enumerate(order):                                   # deal in snake order (0 .. G-1, G-1 .. 0): no rank always gets the dearer one of a round
        r = j % world if (j // world) % 2 == 0 else world - 1 - j % world
        while fill[r] >= min(per, B - r * per) and fill[r] >= 0:
            r = (r + 1) % world
        perm[r * per + fill[r]] = i; fill[r] += 1
    inv = np.empty(B, np.int64); inv[perm] = np.arange(B)
    return perm, inv


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def _dev(device):
    torch, dist = _dist()
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def scatter_rows(full, B, K, rank, world, src=0, device=None):
    """full: (B, K) float64 on rank `src` (ignored elsewhere) -> this rank's (n_local, K) slice.  One collective."""
    torch, dist = _dist()
    dev = _dev(device)
    per = -(-B // world)
    recv = torch.empty((per, K), dtype=torch.float64, device=dev)
    parts = None
    if rank == src:
        pad = np.zeros((per * world, K)); pad[:B] = np.asarray(full, float).reshape(B, K)
        t = torch.from_numpy(pad).to(dev)
        parts = [t[r * per:(r + 1) * per].contiguous() for r in range(world)]
    dist.scatter(recv, parts, src=src)
    lo, hi = shard_range(B, rank, world)
    return recv[:hi - lo].cpu().numpy()


def gather_rows(local, B, rank, world, dst=0, device=None):
    """local: (n_local, K) float64 of this rank's slice -> (B, K) on rank `dst`, None elsewhere.  One collective."""
    torch, dist = _dist()
    dev = _dev(device)
    per = -(-B // world)
    local = np.asarray(local, float)
    K = local.shape[1]
    pad = np.zeros((per, K)); pad[:local.shape[0]] = local
    t = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat(out, 0).cpu().numpy()[:B]


def gather_summaries(local, B, rank, world, device=None):
    """local: (n_local, K) summaries of this rank's slice -> (B, K) array on EVERY rank (one all_gather)."""
    torch, dist = _dist()
    per = -(-B // world)
    local = np.asarray(local, float)
    K = local.shape[1]
    pad = np.zeros((per, K)); pad[:local.shape[0]] = local
    t = torch.from_numpy(pad).to(_dev(device))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()[:B]


# ---------------------------------------------------------------- parking: ParkingSignedDist / ParkingDist over the ranks
def _parking_widths(N, nOb, M):
    N1 = N + 1
    return dict(inp=[("x0", 4), ("xF", 4), ("Ts", 1), ("rx", N1), ("ry", N1), ("ryaw", N1), ("xWS", 4 * N1), ("uWS", 2 * N)],
                out=[("xp", 4 * N1), ("up", 2 * N), ("timeScale", N1), ("exitflag", 1), ("lp", M * N1), ("np", 4 * nOb * N1), ("sl", nOb * N1), ("info", 8)])


def parking_signed_dist_sharded(batch, N, L, ego, XYbounds, vOb, A, b, fixTime, rank, world, src=0, solver=None, device=None, dist_formulation=False,
                                local_device=0):
    """One host batch, sharded over the ranks.  `batch` (rank `src` only; None elsewhere): dict with x0, xF (B,4), Ts (B,), rx, ry, ryaw
    (B,N+1), xWS (B,N+1,4), uWS (B,N,2).  The obstacle set (vOb, A, b) is shared by the batch and known to every rank (configs 2-4 of
    BASELINE.json).  Returns on rank `src` the dict obca_amd.parking_signed_dist_batch returns for the whole batch (None elsewhere).
    `solver`: callable with the signature of obca_amd.parking_signed_dist_batch (default: that function = the HIP path on `local_device`)."""
    torch, dist = _dist()
    vOb = np.ravel(np.asarray(vOb)).astype(int); nOb, M = len(vOb), int(vOb.sum()); N1 = N + 1
    W = _parking_widths(N, nOb, M)
    Bt = torch.zeros(1, dtype=torch.int64, device=_dev(device))
    if rank == src:
        Bt[0] = int(np.reshape(batch["x0"], (-1, 4)).shape[0])
    dist.broadcast(Bt, src=src)
    B = int(Bt.item())
    Kin = sum(w for _, w in W["inp"])
    full = None
    if rank == src:
        full = np.concatenate([np.asarray(batch[k], float).reshape(B, -1)[:, :w] if k != "uWS" else np.asarray(batch[k], float).reshape(B, -1, 2)[:, :N].reshape(B, -1)
                               for k, w in W["inp"]], axis=1)
    loc = scatter_rows(full, B, Kin, rank, world, src, device)
    n = loc.shape[0]
    f = {}; o = 0
    for k, w in W["inp"]:
        f[k] = loc[:, o:o + w]; o += w
    if solver is None:
        from . import api
        solver = lambda *a, **kw: api.parking_signed_dist_batch(*a, device=local_device, **kw)
    res = None
    if n > 0:
        r = solver(f["x0"], f["xF"], N, f["Ts"][:, 0], L, ego, XYbounds, vOb, A, b, f["rx"], f["ry"], f["ryaw"], fixTime,
                   f["xWS"].reshape(n, N1, 4), f["uWS"].reshape(n, N, 2), dist=dist_formulation)
        cols = [np.transpose(r["xp"], (0, 2, 1)).reshape(n, -1), np.transpose(r["up"], (0, 2, 1)).reshape(n, -1), np.asarray(r["timeScale"]).reshape(n, -1),
                np.asarray(r["exitflag"], float).reshape(n, 1), np.stack([np.asarray(x).T.reshape(-1) for x in r["lp"]]),
                np.stack([np.asarray(x).T.reshape(-1) for x in r["np"]]), np.stack([np.asarray(x).T.reshape(-1) for x in r["sl"]]), np.asarray(r["info"]).reshape(n, 8)]
        res = np.concatenate(cols, axis=1)
    else:
        res = np.zeros((0, sum(w for _, w in W["out"])))
    allr = gather_rows(res, B, rank, world, src, device)
    if rank != src:
        return None
    out = {}; o = 0
    for k, w in W["out"]:
        out[k] = allr[:, o:o + w]; o += w
    info = out["info"]
    return dict(xp=np.transpose(out["xp"].reshape(B, N1, 4), (0, 2, 1)), up=np.transpose(out["up"].reshape(B, N, 2), (0, 2, 1)), timeScale=out["timeScale"],
                exitflag=out["exitflag"][:, 0].astype(np.int32), lp=list(np.transpose(out["lp"].reshape(B, N1, M), (0, 2, 1))),
                np=list(np.transpose(out["np"].reshape(B, N1, 4 * nOb), (0, 2, 1))), sl=list(np.transpose(out["sl"].reshape(B, N1, nOb), (0, 2, 1))),
                info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int))


def parking_signed_dist_sharded_ragged(batch, N, L, ego, XYbounds, fixTime, rank, world, src=0, solver=None, device=None, local_device=0):
    """BASELINE config 5 over the ranks: every instance brings its OWN obstacle set.  `batch` (rank `src` only): the keys of parking_signed_dist_sharded plus the
    per-instance lists vOb, A, b.  The instances are dealt to the ranks by (nOb, M) buckets, round-robin (balanced_permutation), the obstacle sets travel with the
    rows (padded to OBCA_NOBMAX / OBCA_MMAX), results come back in the caller's order.  One scatter, one gather."""
    torch, dist = _dist()
    N1 = N + 1; NOB, MM = 16, 64                                   # OBCA_NOBMAX, OBCA_MMAX (include/obca_hip.h)
    inp = [("x0", 4), ("xF", 4), ("Ts", 1), ("rx", N1), ("ry", N1), ("ryaw", N1), ("xWS", 4 * N1), ("uWS", 2 * N), ("vOb", NOB), ("A", 2 * MM), ("b", MM)]
    outw = [("xp", 4 * N1), ("up", 2 * N), ("timeScale", N1), ("exitflag", 1), ("lp", MM * N1), ("np", 4 * NOB * N1), ("sl", NOB * N1), ("info", 8)]
    Bt = torch.zeros(1, dtype=torch.int64, device=_dev(device))
    full = None; inv = None
    if rank == src:
        B = int(np.reshape(batch["x0"], (-1, 4)).shape[0]); Bt[0] = B
        vo = np.zeros((B, NOB)); Aa = np.zeros((B, 2 * MM)); bb = np.zeros((B, MM))
        for i in range(B):
            v = np.ravel(batch["vOb"][i]).astype(int); m = int(v.sum())
            vo[i, :len(v)] = v; Aa[i, :2 * m] = np.ravel(np.asarray(batch["A"][i], float)); bb[i, :m] = np.ravel(np.asarray(batch["b"][i], float))
        cols = dict(vOb=vo, A=Aa, b=bb)
        full = np.concatenate([cols[k] if k in cols else (np.asarray(batch[k], float).reshape(B, -1, 2)[:, :N].reshape(B, -1) if k == "uWS" else np.asarray(batch[k], float).reshape(B, -1)[:, :w])
                               for k, w in inp], axis=1)
        perm, inv = balanced_permutation((vo > 0).sum(1) * 100 + vo.sum(1), world)
        full = full[perm]
    dist.broadcast(Bt, src=src)
    B = int(Bt.item())
    loc = scatter_rows(full, B, sum(w for _, w in inp), rank, world, src, device)
    n = loc.shape[0]
    f = {}; o = 0
    for k, w in inp:
        f[k] = loc[:, o:o + w]; o += w
    if solver is None:
        from . import api
        solver = lambda *a, **kw: api.parking_signed_dist_batch(*a, device=local_device, **kw)
    res = np.zeros((n, sum(w for _, w in outw)))
    if n > 0:
        vl, Al, bl = [], [], []
        for i in range(n):
            v = f["vOb"][i]; v = v[v > 0].astype(int); m = int(v.sum())
            vl.append(v); Al.append(f["A"][i, :2 * m].reshape(m, 2)); bl.append(f["b"][i, :m])
        r = solver(f["x0"], f["xF"], N, f["Ts"][:, 0], L, ego, XYbounds, vl, Al, bl, f["rx"], f["ry"], f["ryaw"], fixTime, f["xWS"].reshape(n, N1, 4), f["uWS"].reshape(n, N, 2))
        pad = lambda lst, rws: np.stack([np.concatenate([np.asarray(x).T.reshape(-1), np.zeros((rws - np.asarray(x).shape[0]) * N1)]) for x in lst])
        res = np.concatenate([np.transpose(r["xp"], (0, 2, 1)).reshape(n, -1), np.transpose(r["up"], (0, 2, 1)).reshape(n, -1), np.asarray(r["timeScale"]).reshape(n, -1),
                              np.asarray(r["exitflag"], float).reshape(n, 1), pad(r["lp"], MM), pad(r["np"], 4 * NOB), pad(r["sl"], NOB), np.asarray(r["info"]).reshape(n, 8)], axis=1)
    allr = gather_rows(res, B, rank, world, src, device)
    if rank != src:
        return None
    allr = allr[inv]                                                 # back to the caller's order
    out = {}; o = 0
    for k, w in outw:
        out[k] = allr[:, o:o + w]; o += w
    info = out["info"]; lp, npp, sl = [], [], []
    for i in range(B):
        v = np.ravel(batch["vOb"][i]).astype(int); m = int(v.sum()); no = len(v)
        lp.append(out["lp"][i, :m * N1].reshape(N1, m).T); npp.append(out["np"][i, :4 * no * N1].reshape(N1, 4 * no).T); sl.append(out["sl"][i, :no * N1].reshape(N1, no).T)
    return dict(xp=np.transpose(out["xp"].reshape(B, N1, 4), (0, 2, 1)), up=np.transpose(out["up"].reshape(B, N, 2), (0, 2, 1)), timeScale=out["timeScale"],
                exitflag=out["exitflag"][:, 0].astype(np.int32), lp=lp, np=npp, sl=sl, info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int))


# ---------------------------------------------------------------- quadcopter: QuadcopterSignedDist / QuadcopterDist over the ranks
def quadcopter_signed_dist_sharded(batch, N, R, ob, rank, world, src=0, solver=None, device=None, dist_formulation=False, dual_ws=True, local_device=0):
    """`batch` (rank `src`): dict with x0, xF (B,12), Ts (B,), timeWS (B,), xWS (B,N+1,12); `ob` (5,6) shared.  Returns on `src` the dict of
    obca_amd.quadcopter_signed_dist_batch for the whole batch: xp, up, timeScale, exitflag, lp, slack, info."""
    torch, dist = _dist()
    N1 = N + 1
    Bt = torch.zeros(1, dtype=torch.int64, device=_dev(device))
    if rank == src:
        Bt[0] = int(np.reshape(batch["x0"], (-1, 12)).shape[0])
    dist.broadcast(Bt, src=src)
    B = int(Bt.item())
    inp = [("x0", 12), ("xF", 12), ("Ts", 1), ("timeWS", 1), ("xWS", 12 * N1)]
    outw = [("xp", 12 * N1), ("up", 4 * N), ("timeScale", N1), ("exitflag", 1), ("lp", 30 * N1), ("slack", 5 * N1), ("info", 8)]
    full = None
    if rank == src:
        full = np.concatenate([np.broadcast_to(np.asarray(batch[k], float).reshape(-1, 1) if w == 1 else np.asarray(batch[k], float).reshape(B, -1)[:, :w], (B, w)) for k, w in inp], axis=1)
    loc = scatter_rows(full, B, sum(w for _, w in inp), rank, world, src, device)
    n = loc.shape[0]
    f = {}; o = 0
    for k, w in inp:
        f[k] = loc[:, o:o + w]; o += w
    if solver is None:
        from . import api
        solver = lambda *a, **kw: api.quadcopter_signed_dist_batch(*a, device=local_device, **kw)
    if n > 0:
        r = solver(f["x0"], f["xF"], N, f["Ts"][:, 0], R, ob, f["xWS"].reshape(n, N1, 12), f["timeWS"][:, 0], dual_ws=dual_ws, dist=dist_formulation)
        T = lambda a: np.transpose(a, (0, 2, 1)).reshape(n, -1)
        res = np.concatenate([T(r["xp"]), T(r["up"]), np.asarray(r["timeScale"]).reshape(n, -1), np.asarray(r["exitflag"], float).reshape(n, 1), T(r["lp"]), T(r["slack"]),
                              np.asarray(r["info"]).reshape(n, 8)], axis=1)
    else:
        res = np.zeros((0, sum(w for _, w in outw)))
    allr = gather_rows(res, B, rank, world, src, device)
    if rank != src:
        return None
    out = {}; o = 0
    for k, w in outw:
        out[k] = allr[:, o:o + w]; o += w
    U = lambda a, c: np.transpose(a.reshape(B, -1, c), (0, 2, 1))
    info = out["info"]
    return dict(xp=U(out["xp"], 12), up=U(out["up"], 4), timeScale=out["timeScale"], exitflag=out["exitflag"][:, 0].astype(np.int32), lp=U(out["lp"], 30),
                slack=U(out["slack"], 5), info=info, iters=info[:, 1].astype(int), obj=info[:, 2], status=info[:, 0].astype(int))
