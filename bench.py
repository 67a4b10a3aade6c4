#!/usr/bin/env python
"""
bench.py -- OBCA NLP solves/s on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one batch of synthetic input.  Default = BASELINE config 2: the reverse-parking NLP
(ParkingSignedDist, N=80, 3 obstacles / 5 half-space rows, variable time, fp64), 1 024 randomised start poses PER GPU (weak scaling).
`--config 3 | 4 | 5` runs the other BASELINE configs at their stated batch sizes divided by 8 GPUs (2 048 / 1 024 / 4 096 per GPU):
  3  parallel parking (4 obstacles / 6 rows), randomised (start, goal), Hybrid A* warm starts          (ParkingSignedDist)
  4  quadcopter, 5 boxes, N=60, y~U[1,9] z~U[1,4] endpoints, 3-D A* warm starts                          (QuadcopterSignedDist)
  5  parking with 1-10 obstacles of 1-4 rows per instance                                                 (ParkingSignedDist)

Data path: rank 0 generates ONE host batch for all ranks and scatters it (obca_amd/sharding.py: one scatter over RCCL), every rank uploads
its slice, the K timed steps run device-resident (inputs in HBM before the timed region; a step = device-side reset of the iterates +
DualMultWS kernel + interior-point kernel(s)), afterwards the full result tuples are gathered on rank 0 (one gather) and validated there.
No collective touches the solve.  `value` counts CONVERGED and VALIDATED solves of all ranks per second.

Steps are PIPELINED: every rank keeps --streams (default 4) device-resident copies of its batch, each on its own HIP stream, and step k runs
on copy k mod streams without a host synchronisation between steps (the K timed steps are bracketed by barrier + synchronize as the
contract says): solve times are heavy-tailed (median 27 factorisation passes, slowest of a batch 100-300), so a step that waits for its
last instance leaves the GPU idle for half of its duration.  After the timed region the SAME process measures `--sync-steps` synchronous
steps (one launch alone on the GPU, HIP events on the launch stream): that is the per-launch kernel time the roofline is computed from.

Extra objects in the JSON line:
  roofline     : dominant kernel (obca_parking_ipm_kernel / obca_quad_ipm_kernel).  SURVEY 8d Model B ("condensed variant: compute /
                 latency-bound on fp64 VALU; HBM bytes = I/O only"): achieved = EXECUTED fp64 flops of one launch (F_PASS x the passes the
                 kernel reports) / the launch's HIP-event duration; peak = 78.6 TFLOP/s (MI355X fp64 vector = fp64 matrix peak); no MFMA
                 instruction is issued by the parking kernel.  Also reported, separately named: `streamed_model_gbs` (the per-pass HBM
                 streaming model of DESIGN.md section 5 / that duration), `pipelined_*` (the same work / the wall time per step of the
                 pipelined region: a device-utilisation figure, not a kernel roofline) and `traffic` (PMC bytes per launch from the
                 committed rocprofv3 pass of `bench.py --streams 1 --steps 1` named in `traffic_source` -- not collected in this run).
  cpu_baseline : the CPU oracle (C restatement, NOT IPOPT) on a bounded sample of the same instances, on the box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260925
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = matrix peak (AMD datasheet; the microarch guide lists no fp64 figure)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CONFIGS = {
    2: dict(kind="parking", N=80, per_gpu=1024, name="BASELINE config 2: reverse-parking ParkingSignedDist NLP, N=80, 3 obstacles (5 half-space rows), variable time, "
            "randomised start poses, line/arc/line warm starts, fp64 interior point"),
    3: dict(kind="parking", N=80, per_gpu=2048, name="BASELINE config 3: parallel-parking ParkingSignedDist NLP, N=80, 4 obstacles (6 rows; the reference's scenario, "
            "main.jl:151), randomised (start, goal), Hybrid A* warm starts, 16 384 instances over 8 GPUs, fp64"),
    4: dict(kind="quad", N=60, per_gpu=1024, name="BASELINE config 4: QuadcopterSignedDist, 5 boxes, N=60, start y~U[1,9] z~U[1,4] and goal likewise, 3-D A* warm starts, "
            "8 192 instances over 8 GPUs, fp64"),
    5: dict(kind="parking", N=80, per_gpu=4096, name="BASELINE config 5: ParkingSignedDist with 1-10 obstacles of 1-4 rows per instance (irregular H-rep packing), N=80, "
            "32 768 instances over 8 GPUs, fp64 (the fp32 + refinement mode is a separate A/B, DESIGN.md)"),
}


def f_pass_parking(N, blocks_per_stage):
    """executed-algorithm fp64 flops per factorisation pass of one parking instance (SURVEY 8d Model B): (stage, obstacle)
    blocks x (700 condense + 600 back-substitute) + stages x 3000 (bicycle Hessians, costs) + Riccati backward N x 3500 + closed loop / forward
    N x 400 + line-search evaluations (111 per block + 1 per ... ~27e3 at N=80 / 3 obstacles)"""
    nb = (N + 1) * blocks_per_stage
    return nb * 1300 + (N + 1) * 3000 + N * 3500 + N * 400 + nb * 60 + (N + 1) * 150


def b_pass_parking(N, nOb, M):
    """per-pass HBM streaming model of DESIGN.md section 5 (iterate, direction, stage / Riccati / obstacle records streamed a fixed number of times
    per pass), scaled from the measured N=80 / 3 obstacles / 5 rows layout (56.5k doubles read + 29.55k written)"""
    zlen = lambda n, no, m: 28 * n + 22 + (2 * m + 19 * no) * (n + 1)
    return (56500 + 29550) * 8.0 * (zlen(N, nOb, M) + 204 * (N + 1) + 12 * nOb * (N + 1)) / (zlen(80, 3, 5) + 204 * 81 + 12 * 3 * 81)


F_PASS_QUAD = 60 * 33000 + 305 * 2400 + 61 * 2500 + 1.0e5   # Riccati 16-state sweep + 305 box blocks + stage derivatives + trial evaluations


def committed_pmc_traffic(kernel, tag=""):
    """HBM bytes PER LAUNCH of `kernel` from the committed rocprofv3 --pmc passes of `bench.py --streams 1 --steps 1 --sync-steps 1` (profiles/r02_pmc_*.csv,
    FETCH_SIZE and WRITE_SIZE in separate passes): 2 x FETCH_SIZE + WRITE_SIZE, KiB units, averaged over the launches in the trace.  The factor 2 and what the
    counters see were calibrated this round (tools/micro/fetch_calib.hip, profiles/r02_pmc_calib_*.csv): FETCH_SIZE reports half the bytes of wide, of coalesced
    8-byte and of one-double-per-128-byte-line loads alike (i.e. whole 128-byte lines), does NOT count re-reads served by the Infinity Cache (a 64 MiB buffer read
    8 times counts once), WRITE_SIZE is exact for streaming stores and counts 32 bytes per isolated 8-byte store.  NOT collected in this run."""
    import csv
    try:
        vals = {}; cnt = {}
        for name, fn in (("FETCH_SIZE", f"r02_pmc_{tag}fetch_size.csv"), ("WRITE_SIZE", f"r02_pmc_{tag}write_size.csv")):
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fn))):
                if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == name:
                    vals[name] = vals.get(name, 0.0) + float(r["Counter_Value"]) * 1024.0; cnt[name] = cnt.get(name, 0) + 1
        return 2.0 * vals["FETCH_SIZE"] / cnt["FETCH_SIZE"] + vals["WRITE_SIZE"] / cnt["WRITE_SIZE"], f"profiles/r02_pmc_{tag}fetch_size.csv + r02_pmc_{tag}write_size.csv (committed rocprofv3 --pmc passes of this command, not this run)"
    except Exception:
        return None, None


# ---------------------------------------------------------------- CPU baseline (oracle = test infrastructure, used here only as the timed CPU leg)
def _cpu_worker(args):
    k, per = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from obca_amd import scenarios as S
    bt = S.make_batch(S.BACKWARDS, per, 80, seed=SEED + 1000 * k)      # worker 0 = the first instances of rank 0's batch
    t0 = time.perf_counter(); ok = 0; its = 0
    for i in range(per):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        r = O.parking_signed_dist(bt["x0"][i], bt["xF"][i], 80, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"],
                                  bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        ok += r["exitflag"]; its += r["iters"]
    return ok, its, time.perf_counter() - t0


def cpu_baseline():
    """oracle (kind 'port') on all host cores (<=64), 256 instances of the config-2 distribution per core (~10-15 s each)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    cores = min(os.cpu_count() or 1, 64)
    per = 256
    n = per * cores
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(k, per) for k in range(cores)])
    wall = time.perf_counter() - t0
    ok = sum(r[0] for r in res); its = sum(r[1] for r in res)
    busy = max(r[2] for r in res)
    return dict(value=round(ok / busy, 2), unit="solves/s", cores=cores, kind="port",
                sample=f"{n} instances of the config-2 distribution (seed 20260925+1000k), {per} per core, one oracle/obca_oracle.c solve at a time per core "
                       f"(CPU restatement of the reference's IPOPT path, not IPOPT itself); "
                       f"{ok}/{n} converged, mean {its / n:.1f} iterations, wall {wall:.1f}s")


# ---------------------------------------------------------------- batch generation (rank 0) and the scatter
def make_host_batch(cfg, B, seed, hybrid=False):
    """the whole job's batch as a dict of (B, K) float64 arrays + the shared scalars"""
    from obca_amd import scenarios as S
    c = CONFIGS[cfg]; N = c["N"]
    if c["kind"] == "quad":
        q = S.make_quad_batch(B, N, seed=seed, random_endpoints=True)
        rows = dict(x0=q["x0"], xF=q["xF"], Ts=np.full((B, 1), q["Ts"]), timeWS=np.full((B, 1), q["timeWS"]), xWS=q["xWS"].reshape(B, -1))
        return rows, dict(R=q["R"], ob=q["ob"])
    if cfg == 2:
        bt = S.make_batch(S.BACKWARDS, B, N, seed=seed, planner=True, smooth=True) if hybrid else S.make_batch(S.BACKWARDS, B, N, seed=seed)
    elif cfg == 3:
        bt = S.make_batch(S.PARALLEL, B, N, seed=seed, goal_jitter=True)
    else:
        bt = S.make_mixed_batch(B, N, seed=seed, min_obstacles=1)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    rows = dict(x0=bt["x0"], xF=bt["xF"], Ts=bt["Ts"].reshape(B, 1), xWS=xWS.reshape(B, -1), uWS=bt["uWS"].reshape(B, -1))
    shared = dict(L=bt["L"], ego=bt["ego"], XYbounds=bt["XYbounds"])
    if cfg == 5:     # per-instance obstacle sets, padded to fixed widths so that they travel as rows too
        vo = np.zeros((B, 10)); Aa = np.zeros((B, 80)); bb = np.zeros((B, 40))
        for i in range(B):
            v = np.ravel(bt["vOb"][i]); vo[i, :len(v)] = v; Aa[i, :2 * v.sum()] = np.ravel(bt["A"][i]); bb[i, :v.sum()] = np.ravel(bt["b"][i])
        rows.update(vOb=vo, A=Aa, b=bb)
    else:
        shared.update(vOb=bt["vOb"], A=bt["A"], b=bt["b"])
    return rows, shared


def scatter_job(rows, shared, B_total, rank, world, dist, backend):
    """rank 0's batch -> this rank's slice (sharding.scatter_rows: ONE scatter over RCCL / gloo); the shared scalars travel as one small object"""
    from obca_amd import sharding
    if world == 1:
        return rows, shared
    meta = [None]
    if rank == 0:
        meta = [(shared, [(k, v.shape[1]) for k, v in rows.items()])]
    dist.broadcast_object_list(meta, src=0)
    shared, widths = meta[0]
    full = np.concatenate([rows[k] for k, _ in widths], axis=1) if rank == 0 else None
    loc = sharding.scatter_rows(full, B_total, sum(w for _, w in widths), rank, world, 0)
    out = {}; o = 0
    for k, w in widths:
        out[k] = loc[:, o:o + w]; o += w
    return out, shared


def obstacle_args(cfg, rows, shared, n):
    if cfg != 5:
        return shared["vOb"], shared["A"], shared["b"]
    vl, Al, bl = [], [], []
    for i in range(n):
        v = rows["vOb"][i]; v = v[v > 0].astype(int); m = int(v.sum())
        vl.append(v); Al.append(rows["A"][i, :2 * m].reshape(m, 2)); bl.append(rows["b"][i, :m])
    return vl, Al, bl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of 1.3 s at ~6.4 ms per step)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the BASELINE batch size of the config / 8 GPUs; config 2: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one GPU for a functional test)")
    ap.add_argument("--streams", type=int, default=4, help="device-resident copies of the batch, each on its own HIP stream (1 = synchronous steps)")
    ap.add_argument("--sync-steps", type=int, default=6, help="synchronous steps measured after the timed region for the per-launch kernel time of the roofline")
    ap.add_argument("--warm-start", default="primitive", choices=["primitive", "hybrid"], help="config 2 only: line/arc/line primitives (default) or the reference's "
                    "pipeline main.jl:216-248 -- Hybrid A* path, velocity smoother, resampling (planned on the host cores before the timed region)")
    ap.add_argument("--seed-offset", type=int, default=0, help="diagnostic: shift the seed of the job's batch")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.steps < 1:
        raise SystemExit("bench.py: --steps must be >= 1")
    cfg = a.config; C = CONFIGS[cfg]; N = C["N"]; quad = C["kind"] == "quad"
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and cfg == 2:
        cpu = cpu_baseline()          # before any HIP context exists in this process (fork-safe)
    rows = shared = None
    B = a.batch or C["per_gpu"]; B_total = B * world
    if rank == 0:
        rows, shared = make_host_batch(cfg, B_total, SEED + a.seed_offset, hybrid=(a.warm_start == "hybrid"))      # (config 3 / 4 plan their warm starts on the host cores here, before HIP is up)
    import torch
    import obca_amd
    from obca_amd import sharding, validate as V
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            local = local % max(1, torch.cuda.device_count())          # functional test: ranks may share a device
            torch.cuda.set_device(local)
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    rows, shared = scatter_job(rows, shared, B_total, rank, world, dist, a.backend)
    lo, hi = sharding.shard_range(B_total, rank, world); n = hi - lo
    assert n == B and rows["x0"].shape[0] == B
    nS = max(1, a.streams)
    batches = []
    if quad:
        for si in range(nS):
            ctx = obca_amd.Context(local)
            bq = obca_amd.QuadBatch(ctx, B, N)
            bq.upload(rows["x0"], rows["xF"], rows["Ts"][:, 0], shared["R"], shared["ob"], rows["xWS"].reshape(B, N + 1, 12), rows["timeWS"][:, 0])
            batches.append(bq)
    else:
        vOb, A, b = obstacle_args(cfg, rows, shared, B)
        xWS = rows["xWS"].reshape(B, N + 1, 4); uWS = rows["uWS"].reshape(B, N, 2)
        for si in range(nS):
            ctx = obca_amd.Context(local)              # one HIP stream per context
            bq = obca_amd.Batch(ctx, B, N)
            bq.upload(rows["x0"], rows["xF"], rows["Ts"][:, 0], shared["L"], shared["ego"], shared["XYbounds"], vOb, A, b,
                      xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, uWS)
            batches.append(bq)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        batches[w % nS].solve(sync=False)
    for bq in batches:
        bq.sync()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        batches[k % nS].solve(sync=False)   # queued behind the previous step of the same stream; overlaps the other streams
    for bq in batches:
        bq.sync()
    fence()
    dt = time.perf_counter() - t0
    # ---- per-launch kernel time: synchronous steps of copy 0, one launch alone on the GPU, HIP events on the launch stream (outside the timed region)
    ipm_ms, dws_ms = [], []
    for k in range(max(1, a.sync_steps)):
        batches[0].solve(sync=True)
        m = batches[0].kernel_ms()
        if quad:
            ipm_ms.append(m)
        else:
            ipm_ms.append(m[0]); dws_ms.append(m[1])
    fence()
    # ---- results: every copy solved the same inputs and must hold the same bits
    outs = [bq.download() for bq in batches[:min(nS, a.steps + a.warmup)]]
    out = outs[0]
    same = all(np.array_equal(o["info"], out["info"]) and np.array_equal(np.asarray(o["xp"]), np.asarray(out["xp"])) for o in outs[1:])
    # ---- gather the full result tuple of every instance on rank 0 (one gather), validate there: a solve counts only if exitflag == 1 AND the
    # returned trajectory passes the a-posteriori checker (obca_amd/validate.py, pure numpy, outside the timed region)
    T = lambda x: np.transpose(np.asarray(x), (0, 2, 1)).reshape(B, -1)
    if quad:
        packed = np.concatenate([T(out["xp"]), T(out["up"]), out["timeScale"], out["exitflag"].reshape(B, 1).astype(float), T(out["lp"]), out["info"]], axis=1)
    else:
        nOb_max, M_max = max(len(np.ravel(v)) for v in (vOb if cfg == 5 else [vOb])), max(int(np.sum(v)) for v in (vOb if cfg == 5 else [vOb]))
        if world > 1 and cfg == 5:      # ragged outputs travel padded to the job-wide maxima
            nOb_max, M_max = 10, 40
        pad = lambda lst, r: np.stack([np.concatenate([np.asarray(x).T.reshape(-1), np.zeros((r - np.asarray(x).shape[0]) * (N + 1))]) for x in lst])
        packed = np.concatenate([T(out["xp"]), T(out["up"]), out["timeScale"], out["exitflag"].reshape(B, 1).astype(float), pad(out["lp"], M_max),
                                 pad(out["np"], 4 * nOb_max), pad(out["sl"], nOb_max), out["info"]], axis=1)
    allp = packed if world == 1 else sharding.gather_rows(packed, B_total, rank, world, 0)
    allrows = rows
    if world > 1:      # the inputs of the other ranks' instances, for the validation on rank 0 (rank 0 made them; re-gather keeps the code path single)
        keys = list(rows.keys())
        g = sharding.gather_rows(np.concatenate([rows[k] for k in keys], axis=1), B_total, rank, world, 0)
        if rank == 0:
            allrows = {}; o = 0
            for k in keys:
                w = rows[k].shape[1]; allrows[k] = g[:, o:o + w]; o += w
    stats = torch.tensor([dt, float(np.mean(ipm_ms)), float(same)], dtype=torch.float64)
    if dist is not None:
        g = stats.cuda() if a.backend == "nccl" else stats.clone()
        tmax = g[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        smin = g[2:3].clone(); dist.all_reduce(smin, op=dist.ReduceOp.MIN)
        dt = float(tmax.item()); same = bool(smin.item() > 0.5)
    if rank == 0:
        Bt = B_total; N1 = N + 1
        if quad:
            o = 0; xp = allp[:, o:o + 12 * N1].reshape(Bt, N1, 12); o += 12 * N1; up = allp[:, o:o + 4 * N].reshape(Bt, N, 4); o += 4 * N
            ts = allp[:, o:o + N1]; o += N1; ef = allp[:, o].astype(int); o += 1; lp = allp[:, o:o + 30 * N1].reshape(Bt, N1, 30); o += 30 * N1; info = allp[:, o:o + 8]
            okv = np.zeros(Bt, bool)
            for i in np.flatnonzero(ef == 1):      # (exit flag 2 = solved but penetrating, QuadcopterSignedDist.jl:285-288: reported, not counted)
                okv[i] = V.validate_quadcopter(xp[i].T, up[i].T, ts[i], allrows["x0"][i], allrows["xF"][i], allrows["Ts"][i, 0], lp[i].T, shared["ob"], shared["R"])[0]
            conv_flag = int((ef == 1).sum())
        else:
            o = 0; xp = allp[:, o:o + 4 * N1].reshape(Bt, N1, 4); o += 4 * N1; up = allp[:, o:o + 2 * N].reshape(Bt, N, 2); o += 2 * N
            ts = allp[:, o:o + N1]; o += N1; ef = allp[:, o].astype(int); o += 1
            lpw, npw, slw = M_max * N1, 4 * nOb_max * N1, nOb_max * N1
            lpa = allp[:, o:o + lpw]; o += lpw; npa = allp[:, o:o + npw]; o += npw; sla = allp[:, o:o + slw]; o += slw; info = allp[:, o:o + 8]
            okv = np.zeros(Bt, bool)
            av, aA, ab = obstacle_args(cfg, allrows, shared, Bt)
            for i in np.flatnonzero(ef == 1):
                v = np.ravel(av[i] if cfg == 5 else av); m = int(v.sum()); no = len(v)
                okv[i] = V.validate_parking(allrows["x0"][i], allrows["xF"][i], N, allrows["Ts"][i, 0], shared["L"], shared["ego"], shared["XYbounds"], v,
                                            aA[i] if cfg == 5 else aA, ab[i] if cfg == 5 else ab, xp[i].T, up[i].T, ts[i], lpa[i, :m * N1].reshape(N1, m).T,
                                            npa[i, :4 * no * N1].reshape(N1, 4 * no).T, sla[i, :no * N1].reshape(N1, no).T, tol=1e-4)[0]
            conv_flag = int((ef == 1).sum())
        conv_all = int(okv.sum())
        iters = info[:, 1]; passes_all = float((info[:, 1] + info[:, 6]).sum())
        passes0 = float((out["info"][:, 1] + out["info"][:, 6]).sum())      # passes of ONE launch of rank 0's batch (the launch the HIP events timed)
        k_ms = float(np.median(ipm_ms))
        if quad:
            f_pass = F_PASS_QUAD; b_pass = None; kernel = "obca_quad_ipm_kernel"
        else:
            if cfg == 5:
                nb_mean = float(np.mean([len(np.ravel(v)) for v in vOb])); f_pass = f_pass_parking(N, nb_mean)
                b_pass = float(np.mean([b_pass_parking(N, len(np.ravel(v)), int(np.sum(v))) for v in vOb]))
            else:
                f_pass = f_pass_parking(N, len(np.ravel(vOb))); b_pass = b_pass_parking(N, len(np.ravel(vOb)), int(np.sum(vOb)))
            kernel = "obca_parking_ipm_kernel"
        tflops = passes0 * f_pass / (k_ms * 1e-3) / 1e12
        traffic, tsrc = committed_pmc_traffic(kernel, "" if cfg == 2 else "quad_") if (cfg in (2, 4) and B == 1024) else (None, None)
        roof = {"bound": "mfma", "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP64_PEAK_TFLOPS, 5),
                "traffic": traffic, "traffic_source": tsrc,
                "bound_detail": "fp64 arithmetic of the executed algorithm (SURVEY 8d Model B: the condensed KKT solve is compute / latency bound, HBM carries I/O only); "
                                "peak = MI355X fp64 vector peak = fp64 matrix (MFMA) peak, 78.6 TFLOP/s",
                "kernel": kernel, "kernel_ms": round(k_ms, 3), "kernel_ms_all": [round(x, 3) for x in ipm_ms],
                "kernel_timing": "HIP events on the launch stream around the interior-point launches of ONE synchronous step of rank 0's batch, measured in this process after the "
                                 "timed region (median of %d); nothing else runs on the GPU" % len(ipm_ms),
                "passes_per_launch": int(passes0), "flops_per_pass_model": f_pass,
                "pipelined_tflops": round(passes0 * f_pass / (dt / a.steps) / 1e12, 3), "pipelined_frac": round(passes0 * f_pass / (dt / a.steps) / 1e12 / FP64_PEAK_TFLOPS, 5),
                "pipelined_note": "the same work / (timed wall time / steps) with %d steps in flight: device utilisation of the timed region, not a kernel roofline" % nS}
        if traffic is not None:     # measured HBM bytes (committed PMC passes) over the kernel time / the pipelined step time of THIS run
            roof.update(hbm_measured_gbs_one_launch=round(traffic / (k_ms * 1e-3) / 1e9, 1), hbm_measured_frac_one_launch=round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        hbm_measured_gbs_pipelined=round(traffic / (dt / a.steps) / 1e9, 1), hbm_measured_frac_pipelined=round(traffic / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                        hbm_measured_note="PMC bytes per launch / kernel time, and / wall time per pipelined step: the kernels stream their per-instance state through HBM "
                                          "(it does not fit LDS), so this -- not the fp64 fraction -- is the roof the pipelined rate runs into (DESIGN.md section 5)")
        if b_pass is not None:
            roof.update(streamed_model_gbs=round(passes0 * b_pass / (k_ms * 1e-3) / 1e9, 1), streamed_model_frac_of_hbm=round(passes0 * b_pass / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        streamed_model_note="per-pass streaming model of DESIGN.md section 5 (%.3g bytes per pass), NOT a SURVEY 8d quantity and not measured" % b_pass)
            sched = batches[0].last_schedule()
            roof.update(ipm_launches_per_step=sched[0], slice_passes=sched[1], dualws_kernel_ms=round(float(np.median(dws_ms)), 3))
        line = {
            "metric": "OBCA NLP solves/sec (N=80, 3 obs, batch) at 1/2/4/8 MI355X vs IPOPT-CPU", "value": round(conv_all * a.steps / dt, 2), "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": C["name"] + (" [warm starts: Hybrid A* + velocity smoother, the reference's pipeline main.jl:216-248]" if (cfg == 2 and a.warm_start == "hybrid") else ""), "config": cfg, "batch_per_gpu": B, "horizon": N,
                       "sharding": f"one host batch of {B_total} instances made on rank 0, scattered over {world} rank(s) (one scatter), solved device-resident, full result tuples "
                                   f"gathered on rank 0 (one gather) and validated there; no collective inside the timed region",
                       "streams": nS, "timed_region_s": round(dt, 3), "converged": conv_all, "exitflag_ok": conv_flag, "instances": B_total,
                       "exitflag2": int((ef == 2).sum()), "copies_bit_identical": bool(same), "mean_iterations": round(float(iters.mean()), 2), "max_iterations": int(iters.max()),
                       "p95_iterations": float(np.percentile(iters, 95)), "mean_passes": round(passes_all / B_total, 2)},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
