#!/usr/bin/env python
"""
bench.py -- OBCA NLP solves/s on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE config 2 = the reverse-parking NLP
(ParkingSignedDist, N=80, 3 obstacles / 5 half-space rows, variable time, fp64) for 1 024 randomised start poses PER GPU
(weak scaling: rank r solves its own 1 024 instances, seed 20260925+r; no collective touches the solve).  Inputs (problem data
and warm starts) are resident in HBM before the timed region; a step = device-side reset of the iterates + DualMultWS kernel +
interior-point kernel(s).  `value` counts CONVERGED solves (exitflag 1) of all ranks per second.

Steps are PIPELINED: every rank keeps --streams (default 4) device-resident copies of its batch, each on its own HIP stream, and step k runs
on copy k mod streams without a host synchronisation between steps (the K timed steps are bracketed by barrier + synchronize as the
contract says).  The solve times of a batch are heavy-tailed -- the median instance needs 27 factorisation passes, the slowest of a batch
100-300 depending on the seed (tools/rank_tails.py) -- so a step that waits for its last instance leaves the GPU idle for half of its
duration; with several batches in flight the tail of one step overlaps the bulk of the next ones.  `--streams 1` gives the synchronous step
(one launch alone on the GPU; 87.5 k instead of ~128 k solves/s, and an 8-GPU run is then held back by the rank with the unluckiest batch).

  python bench.py --gpus 1 --steps 20 --warmup 4
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...

Extra objects in the JSON line:
  roofline     : dominant kernel = obca_parking_ipm_kernel: "achieved" = algorithmic HBM bytes (B_PASS per
                 factorisation pass x passes actually taken, read back from the kernel's iteration/regularisation counters) of the
                 launches in the timed region / the time they take: with --streams 1 the HIP-event duration of the kernel, with
                 pipelined steps (launches of several streams overlap, so a single launch's duration says nothing about the rate the
                 device sustains) the wall time of the region; kernel_ms is the HIP-event duration of one step's launches either way
                 (overlapped if pipelined -- the figure the rocprofv3 trace of the same command shows).  The executed-algorithm fp64
                 flop rate is reported next to it.
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

N_HORIZON = 80
BATCH_PER_GPU = 1024
SEED = 20260925
# executed-algorithm flop model per factorisation pass of one instance, N=80, 3 obstacles (rows 2,2,1):
#   obstacle blocks 243 x (700 condense + 600 back-substitute) + stages 81 x 3000 (bicycle Hessians, costs)
#   + Riccati backward 80 x 3500 + forward/closed-loop 80 x 400 + line-search evaluations ~27e3
F_PASS = 243 * 1300 + 81 * 3000 + 80 * 3500 + 80 * 400 + 27e3
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = matrix peak (AMD datasheet; the microarch guide lists no fp64 figure)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic HBM bytes per factorisation pass of one instance: the iterate, direction, assembled stage records,
# Riccati records and condensed obstacle records live in HBM (0.32 MB/instance does not fit LDS) and each is streamed a fixed number of
# times per pass: ~56.5k doubles read + ~29.5k doubles written
B_PASS = (56500 + 29550) * 8.0


def committed_pmc_traffic():
    """HBM bytes per step (all launches of obca_parking_ipm_kernel in the one-step trace) from the committed rocprofv3 --pmc passes of this same command
    (profiles/r01_pmc_*.csv; FETCH_SIZE doubled per the gfx950 calibration in MI355X_MICROARCH.md, checked on the D2D copy in the
    same trace).  Not collected live: bench.py cannot run under rocprofv3 by itself."""
    import csv
    try:
        vals = {}
        for name, fn in (("FETCH_SIZE", "r01_pmc_fetch_size.csv"), ("WRITE_SIZE", "r01_pmc_write_size.csv")):
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fn))):
                if r["Kernel_Name"].startswith("obca_parking_ipm_kernel") and r["Counter_Name"] == name:
                    vals[name] = vals.get(name, 0.0) + float(r["Counter_Value"]) * 1024.0       # one step = all IPM launches of the trace
        return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]
    except Exception:
        return None


def _cpu_worker(args):
    k, per = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from obca_amd import scenarios as S
    bt = S.make_batch(S.BACKWARDS, per, N_HORIZON, seed=SEED + 1000 * k)      # worker 0 = the first instances of rank 0's batch
    t0 = time.perf_counter(); ok = 0; its = 0
    for i in range(per):
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        r = O.parking_signed_dist(bt["x0"][i], bt["xF"][i], N_HORIZON, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"],
                                  bt["A"], bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        ok += r["exitflag"]; its += r["iters"]
    return ok, its, time.perf_counter() - t0


def cpu_baseline():
    """oracle (kind 'port') on all host cores (<=64), 256 instances of the config-2 distribution per core (~10-15 s each)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    per = 256
    n = per * cores
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(k, per) for k in range(cores)])
    wall = time.perf_counter() - t0
    ok = sum(r[0] for r in res); its = sum(r[1] for r in res)
    # workers regenerate the warm starts inside their timed region; subtract nothing: use the sum of pure solve times instead
    busy = max(r[2] for r in res)
    return dict(value=round(ok / busy, 2), unit="solves/s", cores=cores, kind="port",
                sample=f"{n} instances of the config-2 distribution (seed 20260925+1000k), {per} per core, one oracle/obca_oracle.c solve at a time per core "
                       f"(CPU restatement of the reference's IPOPT path, not IPOPT itself); "
                       f"{ok}/{n} converged, mean {its / n:.1f} iterations, wall {wall:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU (default: BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one GPU for a functional test)")
    ap.add_argument("--streams", type=int, default=4, help="device-resident copies of the batch, each on its own HIP stream: step k runs on copy k mod "
                    "streams and steps are not synchronised one by one, so the tail of one step (a few hard instances) overlaps the bulk of the "
                    "next ones; 1 = synchronous steps (one launch alone on the GPU)")
    ap.add_argument("--seed-offset", type=int, default=None, help="diagnostic: use the batch of this rank (seed 20260925 + offset) instead of the rank's own")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()          # before any HIP context exists in this process (fork-safe)
    import torch
    import obca_amd
    from obca_amd import scenarios as S
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
    B = a.batch
    bt = S.make_batch(S.BACKWARDS, B, N_HORIZON, seed=SEED + (rank if a.seed_offset is None else a.seed_offset))
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    batches = []
    for si in range(max(1, a.streams)):
        ctx = obca_amd.Context(local)              # one HIP stream per context
        bq = obca_amd.Batch(ctx, B, N_HORIZON)
        bq.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                  xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
        batches.append(bq)
    batch = batches[0]; nS = len(batches)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        batches[w % nS].solve()
    fence()
    t0 = time.perf_counter()
    ipm_ms = []; dws_ms = []
    for k in range(a.steps):
        bq = batches[k % nS]
        if nS == 1:
            bq.solve()                      # reset iterates + DualMultWS + IPM on the context's stream, then stream sync
            m = bq.kernel_ms(); ipm_ms.append(m[0]); dws_ms.append(m[1])
        else:
            bq.solve(sync=False)            # queued behind the previous step of the same stream; overlaps the other streams
    for bq in batches:
        bq.sync()
    fence()
    dt = time.perf_counter() - t0
    if nS > 1:
        used = sorted({k % nS for k in range(a.steps)} | {w % nS for w in range(a.warmup)})
        for si in used:
            m = batches[si].kernel_ms(); ipm_ms.append(m[0]); dws_ms.append(m[1])     # last launch of every stream that ran (overlapped durations)
    if a.steps + a.warmup == 0 or (a.steps == 0):
        raise SystemExit("bench.py: --steps must be >= 1")
    if not any(k % nS == 0 for k in range(a.steps)) and not any(w % nS == 0 for w in range(a.warmup)):
        batch.solve()                                   # (never with steps >= 1: step 0 runs on copy 0)
    out = batch.download()
    # a solve counts only if exitflag == 1 AND the returned trajectory passes the a-posteriori checker: every constraint class of
    # the NLP with its slack at IPOPT's constr_viol_tol (obca_amd/validate.py, pure numpy, outside the timed region)
    from obca_amd import validate as V
    okv = np.zeros(B, bool)
    for i in np.flatnonzero(out["exitflag"] == 1):
        okv[i] = V.validate_parking(bt["x0"][i], bt["xF"][i], N_HORIZON, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                    out["xp"][i], out["up"][i], out["timeScale"][i], out["lp"][i], out["np"][i], out["sl"][i], tol=1e-4)[0]
    conv = int(okv.sum())
    passes = float((out["info"][:, 1] + out["info"][:, 6]).sum())
    stats = torch.tensor([dt, float(conv), float(out["iters"].sum()), passes, float(np.mean(ipm_ms))], dtype=torch.float64)
    if dist is not None:
        g = stats.cuda() if a.backend == "nccl" else stats.clone()
        tmax = g[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = g[1:4].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dt = float(tmax.item()); conv_all, iters_all, passes_all = [float(v) for v in sums.cpu()]
    else:
        conv_all, iters_all, passes_all = float(conv), float(out["iters"].sum()), passes
    if rank == 0:
        k_ms = float(np.mean(ipm_ms))
        t_eff = k_ms * 1e-3 if nS == 1 else dt / a.steps       # time per step's worth of passes: the launch alone, or the pipelined region / K
        tflops = passes * F_PASS / t_eff / 1e12                # rank 0: flops of the passes of one step / that time
        gbs = passes * B_PASS / t_eff / 1e9                    # algorithmic HBM bytes of those passes / that time
        how = ("the HIP-event time of the step's IPM launches" if nS == 1 else
               "(timed wall time / steps): %d steps are in flight on their own streams, so the rate the device sustains is the region's, not one "
               "overlapped launch's" % nS)
        model_txt = ("per factorisation pass of one instance: B_PASS=%.3g algorithmic HBM bytes and F_PASS=%.3g executed fp64 flops (SURVEY 8d Model B), "
                     "x %d passes per step (iterations + inertia retries, read back from the kernel). The larger of the two fractions is reported as the "
                     "bound; neither is tight: the kernel is latency / issue bound (one wave per SIMD, 81 dependent stages per pass) and a batch ends "
                     "with its slowest instance. achieved = those bytes / %s. kernel_ms = HIP-event time of all IPM launches of one step (two-launch "
                     "schedule: slice + hardest-first completion)%s; traffic = PMC bytes of one step from profiles/r01_pmc_*.csv (committed, not live)"
                     % (B_PASS, F_PASS, int(passes), how, "" if nS == 1 else ", overlapped with the other steps in flight"))
        line = {
            "metric": "OBCA NLP solves/sec (N=80, 3 obs, batch) at 1/2/4/8 MI355X vs IPOPT-CPU", "value": round(conv_all * a.steps / dt, 2), "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: reverse-parking ParkingSignedDist NLP, N=80, 3 obstacles (5 half-space rows), "
                                   "variable time, 1024 randomised start poses per GPU, line/arc/line warm starts, fp64 interior point",
                       "batch_per_gpu": B, "horizon": N_HORIZON, "sharding": f"independent instances, {world} rank(s), no data-path collective", "streams": nS,
                       "converged": int(conv_all), "instances": B * world, "mean_iterations": round(iters_all / (B * world), 2),
                       "max_iterations_rank0": int(out["iters"].max()), "p95_iterations_rank0": float(np.percentile(out["iters"], 95)),
                       "exitflag1_rank0": int((out["exitflag"] == 1).sum()), "validated_rank0": int(okv.sum())},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "traffic": committed_pmc_traffic(),
                         "kernel": "obca_parking_ipm_kernel", "kernel_ms": round(k_ms, 3), "steps_in_flight": nS, "ipm_launches_per_step": batch.last_schedule()[0], "slice_passes": batch.last_schedule()[1], "dualws_kernel_ms": round(float(np.mean(dws_ms)), 3),
                         "fp64_tflops": round(tflops, 3), "fp64_frac": round(tflops / FP64_PEAK_TFLOPS, 5),
                         "model": model_txt},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
