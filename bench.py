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

Steps are PIPELINED: every rank keeps --streams (default 16) device-resident copies of its batch, each on its own HIP stream and -- GPU_MAX_HW_QUEUES=16, set below
unless the caller set it: the runtime's default of 4 serialises streams that share a hardware queue -- its own hardware queue, and step k runs
on copy k mod streams without a host synchronisation between steps (the K timed steps are bracketed by barrier + synchronize as the
contract says): solve times are heavy-tailed (median 27 factorisation passes, slowest of a batch 100-300), so a step that waits for its
last instance leaves the GPU idle for half of its duration.  After the timed region the SAME process measures `--sync-steps` synchronous
steps (one launch alone on the GPU, HIP events on the launch stream): that is the per-launch kernel time the roofline is computed from.

`value` is measured with the REFERENCE's solver configuration (obca_reference_opts: IPOPT's second-order correction max_soc = 4, recalc_y = "yes" as ParkingSignedDist.jl:41
sets it, least-squares initial multipliers; the quadcopter call: obca_quadcopter_reference_opts).  The library's throughput defaults (those switches off: a fifth fewer passes
per solve) are the secondary leg `config.fast_options`; `--fast-options` times them instead.

Extra objects in the JSON line:
  roofline     : dominant kernel (obca_parking_ipm_kernel / obca_quad_ipm_kernel), one launch alone on the GPU, HIP events on its stream.  SURVEY 8d Model B (the condensed KKT
                 solve these kernels are): algorithmic work = executed fp64 flops per factorisation pass x passes reported by the kernel, against 78.6 TFLOP/s (fp64 vector =
                 matrix peak) -> `bound` "mfma", `achieved`, `frac`; algorithmic HBM bytes = the I/O of the solves only (`algorithmic_bytes_io_only`, ~21.6 KB per solve).
                 `traffic` = bytes of one launch measured in THIS run: bench.py re-runs itself once per counter under `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace`
                 (separate passes, 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950); null if rocprofv3 is not available (`--no-pmc` skips it).
                 `implementation_*` = the kernel's own streaming model (what the code moves per pass because an instance's 0.2 MB of state does not fit its LDS share; NOT
                 algorithmic bytes), `traffic_over_io_only` = how far the measured traffic is above the algorithm's; `regime_of_value` relates the same work to the wall time
                 of a pipelined step (device utilisation in the regime `value` is measured in, not a kernel roofline).
  cpu_baseline : the CPU oracle (C restatement, NOT IPOPT; gcc -O3 -march=native, oracle/Makefile `native`) on a bounded sample of the same distribution, on the box's host
                 cores, with the option set of the timed GPU steps.  Every entry of config.other_configs carries its own.
  config       : besides the workload, what a caller of the drop-in sees (never `value`): `host_pointer_solves_per_s` = obca_parking_signed_dist_batch on 16 384
                 host-array instances, PCIe and (un)packing included (what a Julia ccall gets), `single_batch_sync_solves_per_s` = one batch issued and waited for,
                 `other_configs` = BASELINE configs 3, 4, 5 at their per-GPU batch sizes (reference options, throughput options, CPU leg).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # before anything initialises the HIP runtime: one hardware queue per stream of the pipelined steps (obca_amd/api.py)

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
    N x 400 + trial-point formation (60 per block, 150 per stage)"""
    nb = (N + 1) * blocks_per_stage
    return nb * 1300 + (N + 1) * 3000 + N * 3500 + N * 400 + nb * 60 + (N + 1) * 150


def b_pass_parking(N, nOb, M):
    """ALGORITHMIC HBM bytes per factorisation pass of one parking instance: the streaming model of DESIGN.md section 5 for the round-4 kernel (the obstacle part of the search
    direction is recomputed, never stored; the condensed obstacle contributions are summed in LDS, never written).  Per stage and pass, in doubles; zb = iterate part of the
    stage's (stage, obstacle) blocks = 2 v + 15 per block (lambda, mu, sl, slack, their multipliers):
      direction_obs        read zb
      fused line search    obstacle part: read zb, write zb;  stage part: read 26 (iterate) + 8 (step) + 3 (reference), write 26 (trial iterate) + 60 (stage record)
      backward sweep       read 60, write 72 (Riccati record)
      forward sweep + stage back-substitution   read 44 + 21 (stage record) + 72 (Riccati record: gains once, value-function rows once) + 7, write 8 (stage step)
    plus 0.16 stand-alone assemblies per pass (first iterate, barrier updates, inertia retries)."""
    N1 = N + 1
    zb = 2.0 * M + 15.0 * nOb
    rd = N1 * (zb + zb + (26 + 8 + 3) + 60 + (44 + 21 + 72 + 7))
    wr = N1 * (zb + (26 + 60) + 72 + 8)
    asm = 0.16 * N1 * ((zb + 26) + 60)
    return 8.0 * (rd + wr + asm)


def b_iter_quad(N):
    """ALGORITHMIC HBM bytes per interior-point ITERATION of one quadcopter instance (the streaming model of obca_quad_solver.h; doubles per stage: iterate 190 = 56 primal +
    22 multipliers + 2 x 56 bound multipliers, packed stage record 288, Riccati record 480 of its 768 slots, condensed box records 5 x 12):
      once per iteration   forward sweep reads 280 + 56;  stage back-substitution reads 360 (value-function rows of the next stage) and writes 78 (step);  block back-substitution
                           reads 130, writes 50;  trial evaluation reads 270;  update reads 270, writes 190                                                     = 1 684
      per backward sweep   reads the stage record 288, writes the Riccati record 480 (1.09 sweeps per iteration: 8 % fail the inertia test on the way)            =   768
      per assembly         block part reads 130, writes 60;  stage part reads 60 + 60, writes 288 (1.23 assemblies per iteration: rungs of the inertia ladder that are not
                           skipped on a block hint)                                                                                                              =   598
    (sweeps and assemblies per iteration: host emulation with counters on six bench instances, docs/HISTORY.md section 9; the kernel reports iterations and rungs only)"""
    return 8.0 * (N + 1) * (1684 + 1.09 * 768 + 1.23 * 598)


F_PASS_QUAD = 60 * 33000 + 305 * 2400 + 61 * 2500 + 1.0e5   # Riccati 16-state sweep + 305 box blocks + stage derivatives + trial evaluations


IO_BYTES_PER_SOLVE_QUAD = 29.2e3      # quadcopter N = 60, problem data in + result tuple out


def io_bytes_parking(N, vOb, ragged):
    """SURVEY 8d Model B: the HBM bytes the ALGORITHM needs per solve = its I/O (problem data and warm start in: x0, xF, rx, ry, ryaw, xWS, uWS, the H-rep; result tuple out:
    x, u, timeScale, lambda, mu, sl): 761 doubles in + 1 944 out = 21.6 KB at N = 80 with 3 obstacles / 5 rows; mean over the instances of a ragged batch"""
    def one(v):
        v = np.ravel(v); nOb = len(v); M = int(np.sum(v)); N1 = N + 1
        return 8.0 * ((8 + 3 * N1 + 4 * N1 + 2 * N + 3 * M + nOb + 10) + (4 * N1 + 2 * N + N1 + M * N1 + 4 * nOb * N1))
    return float(np.mean([one(v) for v in vOb])) if ragged else one(vOb)


def live_pmc_traffic(kernel, cfg, batch, fast=False):
    """HBM bytes PER LAUNCH of `kernel`, measured now: this script is run again, once per counter, under `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE and
    WRITE_SIZE cannot share a pass on gfx950) in its `--pmc-child` mode -- the same batch, one step + one synchronous step, nothing else.  Returns (bytes, note) with
    bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2: FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950, MI355X_MICROARCH.md / HBM, calibrated
    for 8-byte gathers by tools/micro/fetch_calib.hip), averaged over the launches of the child; (None, reason) if rocprofv3 cannot be run."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="obca_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--config", str(cfg), "--batch", str(batch)] + (["--fast-options"] if fast else [])
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            tot = n = 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == ctr:
                        tot += float(r["Counter_Value"]) * 1024.0; n += 1
            if n == 0:
                return None, f"no {ctr} rows for {kernel} in the rocprofv3 output"
            vals[ctr] = tot / n
        except Exception as e:      # noqa: BLE001 -- the bench line must not depend on the profiler
            return None, f"rocprofv3 --pmc {ctr} failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) around "
                                                            "`bench.py --pmc-child`, 2 x FETCH_SIZE + WRITE_SIZE per launch")


# ---------------------------------------------------------------- CPU baseline (oracle = test infrastructure, used here only as the timed CPU leg)
CPU_SAMPLE_PER_CORE = {2: 512, 3: 96, 4: 24, 5: 96}      # instances per host core: 2-5 s of oracle work per core and config (the default line carries all four)


def _cpu_worker(args):
    """one host core: `per` instances of the config's distribution, one oracle solve at a time, with the option set the GPU steps are timed with"""
    cfg, k, per, fast, rows, shared = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    N = CONFIGS[cfg]["N"]
    if rows is None:                                # configs without a planner: worker k draws its own instances (other seeds of the same distribution)
        rows, shared = make_host_batch(cfg, per, SEED + 1000 * k)
    t0 = time.perf_counter(); ok = 0; its = 0
    if CONFIGS[cfg]["kind"] == "quad":
        import oracle_quad as Q
        o = Q.default_opts()
        if not fast:
            o.max_soc = 4; o.lsq_init = 1; o.obj_scaling = 1      # obca_quadcopter_reference_opts
        for i in range(per):
            r = Q.quadcopter_signed_dist(rows["x0"][i], rows["xF"][i], N, float(rows["Ts"][i, 0]), shared["R"], shared["ob"], rows["xWS"][i].reshape(N + 1, 12), float(rows["timeWS"][i, 0]), opts=o)
            ok += int(r["exitflag"] == 1); its += r["iters"]
        return ok, its, time.perf_counter() - t0
    import oracle as O
    o = O.default_opts()
    if not fast:
        o.max_soc = 4; o.recalc_y = 1; o.lsq_init = 1; o.restoration = 1            # obca_reference_opts
    vOb, A, b = obstacle_args(cfg, rows, shared, per)
    for i in range(per):
        xWS = rows["xWS"][i].reshape(N + 1, 4); uWS = rows["uWS"][i].reshape(N, 2)
        r = O.parking_signed_dist(rows["x0"][i], rows["xF"][i], N, float(rows["Ts"][i, 0]), shared["L"], shared["ego"], shared["XYbounds"], vOb[i] if cfg == 5 else vOb,
                                  A[i] if cfg == 5 else A, b[i] if cfg == 5 else b, xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, uWS, opts=o)
        ok += int(r["exitflag"] == 1); its += r["iters"]
    return ok, its, time.perf_counter() - t0


def cpu_baseline(cfg, fast=False, rows=None, shared=None):
    """The oracle (kind 'port': the C restatement of the reference's path, NOT IPOPT) on the host cores this process may use (<= 64), CPU_SAMPLE_PER_CORE[cfg] instances of
    the config's distribution per core, with the SAME option set the timed GPU steps run (the reference's IPOPT configuration unless --fast-options).  Configs whose warm
    starts come from a planner (3, 4) take the first instances of the job's own batch (`rows`), the others draw fresh ones per core.  Runs before any HIP context exists in
    this process (fork)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    os.environ["OBCA_ORACLE_NATIVE"] = "1"      # the workers time the -O3 -march=native build (oracle/Makefile: `native`), compiled on the machine that runs it
    O.build_native()
    quad = CONFIGS[cfg]["kind"] == "quad"
    if quad:
        import oracle_quad as Q
        Q.build_native()
    cores = min(effective_cpus(), 64)            # the threads this process may really use (affinity mask and cgroup quota), not the CPUs the box shows
    per = CPU_SAMPLE_PER_CORE[cfg]
    if rows is not None:
        per = max(1, min(per, rows["x0"].shape[0] // cores))
    n = per * cores
    jobs = [(cfg, k, per, fast, None if rows is None else {q: v[k * per:(k + 1) * per] for q, v in rows.items()}, shared) for k in range(cores)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    ok = sum(r[0] for r in res); its = sum(r[1] for r in res)
    busy = max(r[2] for r in res)
    os.environ.pop("OBCA_ORACLE_NATIVE", None)
    return dict(value=round(ok / busy, 2), unit="solves/s", cores=cores, kind="port",
                sample=f"{n} instances of the config-{cfg} distribution, {per} per core, one oracle solve at a time per core (oracle/obca_oracle{'_quad' if quad else ''}.c: the CPU "
                       f"restatement of the reference's IPOPT path, not IPOPT itself; gcc -O3 -march=native), options: "
                       f"{'library throughput defaults' if fast else 'the reference IPOPT configuration (the option set of the timed GPU steps)'}; "
                       f"{ok}/{n} converged, mean {its / n:.1f} iterations, wall {wall:.1f}s")


# ---------------------------------------------------------------- batch generation (rank 0) and the scatter
def effective_cpus():
    from obca_amd import planner
    return planner.effective_cpus()


def needs_planner(cfg, hybrid):
    return cfg in (3, 4) or (cfg == 2 and hybrid)


def make_host_batch(cfg, B, seed, hybrid=False, world=1):
    """The whole job's batch on rank 0, as a dict of (B, K) float64 arrays + the shared scalars.  Configs whose warm starts come from a planner (Hybrid A* / 3-D A*: the
    step before the path, host side) only carry the problem DEFINITIONS here (start / goal poses): every rank plans the warm starts of its own slice after the scatter
    (complete_rows), on its share of the host cores -- rank 0 planning 16 384 parallel-parking paths would take minutes while the other ranks wait.  Config 5's unequal
    instances are dealt to the ranks by (nOb, M) buckets, round-robin (sharding.balanced_permutation)."""
    from obca_amd import scenarios as S, sharding
    c = CONFIGS[cfg]; N = c["N"]
    if needs_planner(cfg, hybrid) and world > 1:
        rng = np.random.default_rng(seed)
        if c["kind"] == "quad":
            x0 = np.tile(S.QUAD_X0, (B, 1)); xF = np.tile(S.QUAD_XF, (B, 1))
            for i in range(1, B):
                x0[i, :3], xF[i, :3] = S._draw_quad_endpoints(rng)
            return dict(x0=x0, xF=xF), dict(R=S.QUAD_R, ob=S.QUAD_OB.copy())
        sc = S.BACKWARDS if cfg == 2 else S.PARALLEL
        x0, xF = S.sample_poses(sc, B, rng, goal_jitter=(cfg == 3))
        A, b, v = S.scenario_hrep(sc)
        return dict(x0=x0, xF=xF), dict(L=S.L_WHEELBASE, ego=S.EGO.copy(), XYbounds=S.XYBOUNDS.copy(), vOb=v, A=A, b=b)
    if c["kind"] == "quad":      # (single rank: planned here, in one random stream -- the batch the GPU parity tests use)
        q = S.make_quad_batch(B, N, seed=seed, random_endpoints=True)
        return dict(x0=q["x0"], xF=q["xF"], Ts=np.full((B, 1), q["Ts"]), timeWS=np.full((B, 1), q["timeWS"]), xWS=q["xWS"].reshape(B, -1)), dict(R=q["R"], ob=q["ob"])
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
        if world > 1:
            perm, _ = sharding.balanced_permutation((vo > 0).sum(1) * 100 + vo.sum(1), world)
            rows = {k: v_[perm] for k, v_ in rows.items()}
    else:
        shared.update(vOb=bt["vOb"], A=bt["A"], b=bt["b"])
    return rows, shared


def complete_rows(cfg, rows, shared, seed, rank, world, hybrid=False):
    """every rank: the planner-made warm starts of ITS slice (the planner library's threads on 1 / local-world of the host cores; a pose without a path is re-drawn locally)"""
    if not needs_planner(cfg, hybrid) or world == 1:
        return rows
    from obca_amd import scenarios as S
    N = CONFIGS[cfg]["N"]; n = rows["x0"].shape[0]
    rng = np.random.default_rng(seed + 1000003 * (rank + 1))
    x0 = np.array(rows["x0"]); xF = np.array(rows["xF"])
    if CONFIGS[cfg]["kind"] == "quad":
        xWS = S.plan_quad_batch(x0, xF, N, rng, first_is_fixed=(rank == 0))
        return dict(x0=x0, xF=xF, Ts=np.full((n, 1), S.quad_sample_time(N)), timeWS=np.full((n, 1), 1.0), xWS=xWS.reshape(n, -1))
    sc = S.BACKWARDS if cfg == 2 else S.PARALLEL
    workers = max(1, effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    Ts, xWS, uWS = S.plan_batch(sc, x0, xF, N, rng, planner=True, workers=workers, smooth=(cfg == 2))
    xWS = xWS.copy(); xWS[:, 0, :] = x0
    return dict(x0=x0, xF=xF, Ts=Ts.reshape(n, 1), xWS=xWS.reshape(n, -1), uWS=uWS.reshape(n, -1))


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


# ---------------------------------------------------------------- pieces shared by the extra legs of the default line (outside the timed region, never `value`)
def validated_mask(cfg, rows, shared, out, N):
    """which instances of a downloaded batch (Batch.download / QuadBatch.download) count: exit flag 1 AND the a-posteriori checker accepts the trajectory at 1e-4"""
    from obca_amd import validate as V
    B = len(out["exitflag"]); ok = np.zeros(B, bool)
    if CONFIGS[cfg]["kind"] == "quad":
        for i in np.flatnonzero(out["exitflag"] == 1):
            ok[i] = V.validate_quadcopter(out["xp"][i], out["up"][i], out["timeScale"][i], rows["x0"][i], rows["xF"][i], rows["Ts"][i, 0], out["lp"][i], shared["ob"], shared["R"])[0]
        return ok
    vOb, A, b = obstacle_args(cfg, rows, shared, B)
    for i in np.flatnonzero(out["exitflag"] == 1):
        v = np.ravel(vOb[i] if cfg == 5 else vOb)
        ok[i] = V.validate_parking(rows["x0"][i], rows["xF"][i], N, rows["Ts"][i, 0], shared["L"], shared["ego"], shared["XYbounds"], v, A[i] if cfg == 5 else A, b[i] if cfg == 5 else b,
                                   out["xp"][i], out["up"][i], out["timeScale"][i], out["lp"][i], out["np"][i], out["sl"][i], tol=1e-4)[0]
    return ok


def device_batches(cfg, rows, shared, B, N, nS, local):
    """nS device-resident copies of one host batch, each with its own context / HIP stream"""
    import obca_amd
    out = []
    for _ in range(nS):
        ctx = obca_amd.Context(local)
        if CONFIGS[cfg]["kind"] == "quad":
            bq = obca_amd.QuadBatch(ctx, B, N)
            bq.upload(rows["x0"], rows["xF"], rows["Ts"][:, 0], shared["R"], shared["ob"], rows["xWS"].reshape(B, N + 1, 12), rows["timeWS"][:, 0])
        else:
            vOb, A, b = obstacle_args(cfg, rows, shared, B)
            xWS = rows["xWS"].reshape(B, N + 1, 4); uWS = rows["uWS"].reshape(B, N, 2)
            bq = obca_amd.Batch(ctx, B, N)
            bq.upload(rows["x0"], rows["xF"], rows["Ts"][:, 0], shared["L"], shared["ego"], shared["XYbounds"], vOb, A, b, xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, uWS)
        out.append(bq)
    return out


def pipelined_rate(batches, steps, warmup, opts=None):
    """`steps` pipelined steps over the copies (step k on copy k mod nS, no host synchronisation in between); seconds of the timed steps"""
    import torch
    nS = len(batches)
    for w in range(warmup):
        batches[w % nS].solve(opts=opts, sync=False)
    for bq in batches:
        bq.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        batches[k % nS].solve(opts=opts, sync=False)
    for bq in batches:
        bq.sync()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def config_opts(cfg, fast):
    """the option record of a config's timed steps: the reference's IPOPT configuration of the call (obca_reference_opts / obca_quadcopter_reference_opts) unless `fast`
    (None = the library's throughput defaults)"""
    import obca_amd
    if fast:
        return None
    return obca_amd.quadcopter_ipopt_opts() if CONFIGS[cfg]["kind"] == "quad" else obca_amd.ipopt_opts()


OPTION_NAMES = {("parking", False): "reference IPOPT configuration: max_soc = 4, recalc_y = yes, lsq_init = 1, block restoration = 1 (obca_reference_opts; ParkingSignedDist.jl:41-43 + IPOPT defaults)",
                ("parking", True): "library throughput defaults: max_soc = 0, recalc_y = no, y0 = 0 (obca_default_opts)",
                ("quad", False): "reference IPOPT configuration: max_soc = 4, lsq_init = 1, obj_scaling = 1, recalc_y = no (obca_quadcopter_reference_opts; QuadcopterSignedDist.jl:28-31 + IPOPT defaults)",
                ("quad", True): "library throughput defaults: max_soc = 0, y0 = 0, no objective scaling (obca_quadcopter_default_opts)"}


def other_host_batch(cfg, seed=SEED):
    """host batch of another BASELINE config at its per-GPU size.  Config 3's warm starts are Hybrid A* plans (host side, ~8 ms each per core): the default line plans 256
    (start, goal) pairs and repeats them 8 x to fill the 2 048 instances of the batch (`bench.py --config 3` plans all 2 048)."""
    B = CONFIGS[cfg]["per_gpu"]; t0 = time.perf_counter(); note = None
    if cfg == 3:
        rows, shared = make_host_batch(cfg, 256, seed)
        rows = {k: np.concatenate([v] * (B // 256), axis=0) for k, v in rows.items()}
        note = "256 planned (start, goal) pairs x 8"
    else:
        rows, shared = make_host_batch(cfg, B, seed)
    return rows, shared, round(time.perf_counter() - t0, 2), note


def other_config_line(cfg, local, prepared, cpu, steps=16, streams=8):
    """compact, driver-visible rate of another BASELINE config at its per-GPU batch size: `steps` pipelined steps with the reference's IPOPT configuration, every instance
    validated; the same with the library's throughput options beside it; warm starts planned before the timed steps (the planner is host code outside the path)"""
    C_ = CONFIGS[cfg]; N = C_["N"]; B = C_["per_gpu"]
    rows, shared, t_make, note = prepared
    quad = C_["kind"] == "quad"
    res = {}
    for fast in (False, True):
        o = config_opts(cfg, fast)
        bs = device_batches(cfg, rows, shared, B, N, streams, local)
        dt = pipelined_rate(bs, steps, streams, opts=o)
        bs[0].solve(opts=o, sync=True); k_ms = bs[0].kernel_ms(); k_ms = float(k_ms if quad else k_ms[0])
        out = bs[0].download(); ok = validated_mask(cfg, rows, shared, out, N)
        for bq in bs:
            bq.close()
        res[fast] = dict(solves_per_s=round(int(ok.sum()) * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 3), validated=int(ok.sum()), exitflag_ok=int((out["exitflag"] == 1).sum()),
                         mean_iterations=round(float(out["info"][:, 1].mean()), 2), mean_passes=round(float((out["info"][:, 1] + out["info"][:, 6]).mean()), 2), kernel_ms_one_launch=round(k_ms, 3))
    line = dict(config=cfg, workload=C_["name"], batch_per_gpu=B, steps=steps, streams=streams, options=OPTION_NAMES[(C_["kind"], False)])
    line.update(res[False])
    line.update(batch_made_in_s=t_make, batch_note=note, fast_options=dict(options=OPTION_NAMES[(C_["kind"], True)], **res[True]), cpu_baseline=cpu)
    return line


def options_leg(cfg, rows, shared, B, N, nS, local, steps, out_main, fast):
    """the SAME batch solved with the OTHER option set (fast = True: the library's throughput defaults, when the timed steps ran the reference's IPOPT configuration; and the
    other way round): pipelined rate, iterations / passes, and how many instances end somewhere else than with the option set of the timed steps; never `value`"""
    quad = CONFIGS[cfg]["kind"] == "quad"
    o = config_opts(cfg, fast)
    bs = device_batches(cfg, rows, shared, B, N, nS, local)
    dt = pipelined_rate(bs, steps, nS, opts=o)
    bs[0].solve(opts=o, sync=True); k_ms = bs[0].kernel_ms(); k_ms = float(k_ms if quad else k_ms[0])
    out = bs[0].download(); ok = validated_mask(cfg, rows, shared, out, N)
    for bq in bs:
        bq.close()
    both = (out["exitflag"] == 1) & (out_main["exitflag"] == 1)
    dx = np.array([np.abs(np.asarray(out["xp"][i]) - np.asarray(out_main["xp"][i])).max() for i in range(B)])
    du = np.array([np.abs(np.asarray(out["up"][i]) - np.asarray(out_main["up"][i])).max() for i in range(B)])
    df = np.abs(out["obj"] - out_main["obj"]) / np.maximum(1.0, np.abs(out_main["obj"]))
    dts = np.abs(out["timeScale"][:, 0] - out_main["timeScale"][:, 0])
    differs = both & ((df > 2.1e-3) | (dts > 1e-3)) if quad else both & ((dx > 1e-3) | (du > 1e-3) | (df > 1e-4) | (dts > 1e-4))
    return dict(options=OPTION_NAMES[("quad" if quad else "parking", fast)], solves_per_s=round(int(ok.sum()) * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 3),
                validated=int(ok.sum()), exitflag_ok=int((out["exitflag"] == 1).sum()), mean_iterations=round(float(out["info"][:, 1].mean()), 2),
                mean_passes=round(float((out["info"][:, 1] + out["info"][:, 6]).mean()), 2), kernel_ms_one_launch=round(k_ms, 3),
                exitflag_differs_from_timed_options=int((out["exitflag"] != out_main["exitflag"]).sum()), iterations_differ_from_timed_options=int((out["info"][:, 1] != out_main["info"][:, 1]).sum()),
                solution_differs_from_timed_options=int(differs.sum()), worst_dx=float(dx[both].max()) if both.any() else None, worst_rel_objective=float(df[both].max()) if both.any() else None,
                note=("solution_differs: instances solved by both option sets whose objective differs by more than 2.1e-3 relative or time scale by 1e-3: IPOPT's objective scaling (1 / 21 on "
                      "this NLP) terminates 21 x looser in unscaled terms than the throughput options, which do not scale; the cost has no term on the path, so the states of two solves of ONE "
                      "local solution differ by up to 1e-2 (worst_dx is reported only); never `value`") if quad else
                     "solution_differs: instances solved by both option sets whose states / inputs differ by more than 1e-3, time scale by 1e-4 or objective by 1e-4 relative "
                     "(the path's stated tolerance, SURVEY 8c): the NLP is non-convex, another iteration path may end in another local solution; never `value`")


def single_process(a):
    """ONE process, every visible GPU: the route a Julia caller takes (julia/OBCAHip.jl: MultiContext).  The host-pointer entry point cuts each call into chunks and the
    worker lanes of all devices pull them from one queue (include/obca_hip.h: obca_create_multi); inputs and outputs are host arrays, PCIe is inside the timed region."""
    import obca_amd
    from obca_amd import validate as V
    cfg = a.config; C = CONFIGS[cfg]; N = C["N"]
    assert C["kind"] == "parking", "--single-process: parking configs (2, 3, 5)"
    from obca_amd import api
    ndev = max(1, int(api._load().obca_visible_device_count()))
    B = (a.batch or 16 * C["per_gpu"]) * max(1, ndev)      # 16 chunks of 1 024 per device and call: the work queue needs several chunks per lane to overlap their tails
    rows, shared = make_host_batch(cfg, B, SEED + a.seed_offset, hybrid=(a.warm_start == "hybrid"))
    vOb, A, b = obstacle_args(cfg, rows, shared, B)
    xWS = rows["xWS"].reshape(B, N + 1, 4); uWS = rows["uWS"].reshape(B, N, 2); keep = {}
    call = lambda: obca_amd.parking_signed_dist_batch(rows["x0"], rows["xF"], N, rows["Ts"][:, 0], shared["L"], shared["ego"], shared["XYbounds"], vOb, A, b,
                                                      xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, uWS, opts=config_opts(cfg, a.fast_options), device="all", buffers=keep)
    for _ in range(max(1, a.warmup // 4)):
        out = call()
    steps = max(1, a.steps // 10)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = call()
    dt = time.perf_counter() - t0
    ok = 0
    for i in np.flatnonzero(out["exitflag"] == 1):
        v = np.ravel(vOb[i] if cfg == 5 else vOb)
        ok += bool(V.validate_parking(rows["x0"][i], rows["xF"][i], N, rows["Ts"][i, 0], shared["L"], shared["ego"], shared["XYbounds"], v, A[i] if cfg == 5 else A, b[i] if cfg == 5 else b,
                                      out["xp"][i], out["up"][i], out["timeScale"][i], out["lp"][i], out["np"][i], out["sl"][i], tol=1e-4)[0])
    print(json.dumps({"metric": "OBCA NLP solves/sec (N=80, 3 obs, batch) at 1/2/4/8 MI355X vs IPOPT-CPU", "value": round(ok * steps / dt, 2), "unit": "solves/s", "n_gpus": ndev,
                      "steps": steps, "warmup": max(1, a.warmup // 4), "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f64", "data": "synthetic",
                      "config": {"workload": C["name"], "config": cfg, "mode": "single process, multi-device context (obca_create_multi), host-pointer entry point: host arrays in, host arrays out, "
                                 "PCIe and (un)packing inside the timed region", "instances_per_step": B, "converged": ok}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of 1.3 s at ~6.4 ms per step)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the BASELINE batch size of the config / 8 GPUs; config 2: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one GPU for a functional test)")
    ap.add_argument("--streams", type=int, default=16, help="device-resident copies of the batch, each on its own HIP stream (1 = synchronous steps)")
    ap.add_argument("--sync-steps", type=int, default=6, help="synchronous steps measured after the timed region for the per-launch kernel time of the roofline")
    ap.add_argument("--warm-start", default="primitive", choices=["primitive", "hybrid"], help="config 2 only: line/arc/line primitives (default) or the reference's "
                    "pipeline main.jl:216-248 -- Hybrid A* path, velocity smoother, resampling (planned on the host cores before the timed region)")
    ap.add_argument("--seed-offset", type=int, default=0, help="diagnostic: shift the seed of the job's batch")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-distinct", action="store_true", help="skip the run with a different batch on every stream behind config.distinct_batches (config 2)")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the host-pointer (PCIe-inclusive) call behind config.host_pointer_solves_per_s")
    ap.add_argument("--single-process", action="store_true", help="the Julia route: ONE process drives every visible GPU through a multi-device context (obca_create_multi) and the "
                    "host-pointer entry point; a step = one call on batch x devices host-array instances, PCIe included (parking configs)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: the run rocprofv3 wraps (one step + one synchronous step of the same batch, no output)")
    ap.add_argument("--fast-options", action="store_true", help="the TIMED steps run the library's throughput defaults (no second-order correction, no recalc_y, y0 = 0) instead of "
                    "the reference's IPOPT configuration (obca_reference_opts / obca_quadcopter_reference_opts), which is what `value` is measured with since round 5")
    ap.add_argument("--ipopt-options", action="store_true", help="(accepted for old job scripts: the reference's IPOPT configuration is the default now)")
    ap.add_argument("--no-ipopt-leg", "--no-options-leg", dest="no_ipopt_leg", action="store_true", help="skip config.fast_options (the same batch with the other option set)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip config.other_configs (compact rates of BASELINE configs 3, 4 and 5, each with its CPU leg)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.single_process:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (the contract's launch line), so that an 8-GPU run cannot be lost to a missing torchrun
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    if a.pmc_child:
        a.steps, a.warmup, a.streams, a.sync_steps, a.no_cpu_baseline, a.no_pmc, a.no_host_rate, a.no_distinct, a.no_ipopt_leg, a.no_other_configs = 1, 0, 1, 1, True, True, True, True, True, True
    if a.single_process:
        return single_process(a)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {a.gpus}, or without a launcher: bench.py starts its own ranks)")
    if a.steps < 1:
        raise SystemExit("bench.py: --steps must be >= 1")
    cfg = a.config; C = CONFIGS[cfg]; N = C["N"]; quad = C["kind"] == "quad"
    fast = bool(a.fast_options)
    rows = shared = None
    B = a.batch or C["per_gpu"]; B_total = B * world
    t_plan0 = time.perf_counter()
    if rank == 0:
        rows, shared = make_host_batch(cfg, B_total, SEED + a.seed_offset, hybrid=(a.warm_start == "hybrid"), world=world)
    t_plan = time.perf_counter() - t_plan0          # (world == 1: the planner runs inside make_host_batch)
    # ---- CPU legs (the oracle on the host cores), before any HIP context exists in this process (they fork): this config's, and those of the other configs of the default line
    cpu = None; other_prepared = {}; other_cpu = {}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg, fast, rows if needs_planner(cfg, a.warm_start == "hybrid") else None, shared)
    if rank == 0 and world == 1 and cfg == 2 and not a.no_other_configs:
        for c_ in (3, 4, 5):
            other_prepared[c_] = other_host_batch(c_)
            if not a.no_cpu_baseline:
                other_cpu[c_] = cpu_baseline(c_, False, other_prepared[c_][0] if needs_planner(c_, False) else None, other_prepared[c_][1])
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
            ndev_ = max(1, torch.cuda.device_count())
            local = local % ndev_          # functional test: ranks may share a device
            if world > ndev_:              # ... and then share its hardware queues: 16 batches in flight per DEVICE, not per rank (32 queues on one GPU measured worse than 4)
                a.streams = max(1, a.streams // ((world + ndev_ - 1) // ndev_))
            torch.cuda.set_device(local)
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    rows, shared = scatter_job(rows, shared, B_total, rank, world, dist, a.backend)
    t_plan0 = time.perf_counter()
    rows = complete_rows(cfg, rows, shared, SEED + a.seed_offset, rank, world, hybrid=(a.warm_start == "hybrid"))
    if world > 1:
        t_plan = time.perf_counter() - t_plan0      # this rank's slice, planned on its share of the host cores
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

    run_opts = config_opts(cfg, fast)      # the reference's IPOPT configuration of the config's call unless --fast-options
    for w in range(a.warmup):
        batches[w % nS].solve(opts=run_opts, sync=False)
    for bq in batches:
        bq.sync()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        batches[k % nS].solve(opts=run_opts, sync=False)   # queued behind the previous step of the same stream; overlaps the other streams
    for bq in batches:
        bq.sync()
    fence()
    dt = time.perf_counter() - t0
    # ---- per-launch kernel time: synchronous steps of copy 0, one launch alone on the GPU, HIP events on the launch stream (outside the timed region)
    ipm_ms, dws_ms, sync_s = [], [], []
    for k in range(max(1, a.sync_steps)):
        ts0 = time.perf_counter(); batches[0].solve(opts=run_opts, sync=True); sync_s.append(time.perf_counter() - ts0)
        m = batches[0].kernel_ms()
        if quad:
            ipm_ms.append(m)
        else:
            ipm_ms.append(m[0]); dws_ms.append(m[1])
    fence()
    if a.pmc_child:
        return
    # ---- what a caller of the drop-in gets (outside the timed region, never `value`): the host-pointer entry point on 16 384 host-array instances, PCIe included
    host_rate = None
    if rank == 0 and world == 1 and not quad and not a.no_host_rate:
        reps = max(1, 16384 // B); tile = lambda x: np.concatenate([np.asarray(x)] * reps, axis=0)
        hv, hA, hb = (vOb * reps, A * reps, b * reps) if cfg == 5 else (vOb, A, b)
        hx = tile(xWS); hx0, hxF, hTs, hu = tile(rows["x0"]), tile(rows["xF"]), tile(rows["Ts"][:, 0]), tile(uWS); keep = {}; best = bestc = None
        for rep_ in range(3):       # the caller keeps its output arrays between calls (fresh ones cost a page fault per 4 KB inside the C call)
            th0 = time.perf_counter()
            ho = obca_amd.parking_signed_dist_batch(hx0, hxF, N, hTs, shared["L"], shared["ego"], shared["XYbounds"], hv, hA, hb,
                                                    hx[:, :, 0], hx[:, :, 1], hx[:, :, 2], 0, hx, hu, opts=run_opts, device=local, buffers=keep)
            th = time.perf_counter() - th0; best = th if best is None else min(best, th); bestc = float(ho["time"]) if bestc is None else min(bestc, float(ho["time"]))
        nok = int((ho["exitflag"] == 1).sum())
        host_rate = dict(instances=B * reps, solves_per_s=round(nok / best, 1), seconds=round(best, 4), c_call_seconds=round(bestc, 4), c_call_solves_per_s=round(nok / bestc, 1),
                         note="solves_per_s: around the Python wrapper (input normalisation, per-instance views of the results); c_call_*: inside obca_parking_signed_dist_batch itself, "
                              "which is what a ccall from Julia pays")
    # ---- the batch lottery (outside `value`): the headline re-solves ONE batch of B instances on every stream; here every stream holds a DIFFERENT batch (other seeds),
    # so the timed steps cover streams x B distinct instances and their stragglers (the longest solve differs from batch to batch)
    distinct = None
    if rank == 0 and world == 1 and cfg == 2 and a.warm_start == "primitive" and not a.no_distinct:
        dbs, dval, dmaxp = [], [], []
        for si in range(nS):
            r2, s2 = make_host_batch(cfg, B, SEED + a.seed_offset + 7919 * (si + 1))
            x2 = r2["xWS"].reshape(B, N + 1, 4); u2 = r2["uWS"].reshape(B, N, 2)
            bq = obca_amd.Batch(obca_amd.Context(local), B, N)
            bq.upload(r2["x0"], r2["xF"], r2["Ts"][:, 0], s2["L"], s2["ego"], s2["XYbounds"], s2["vOb"], s2["A"], s2["b"], x2[:, :, 0], x2[:, :, 1], x2[:, :, 2], 0, x2, u2)
            dbs.append((bq, r2, s2))
        for w_ in range(nS):
            dbs[w_][0].solve(opts=run_opts, sync=False)
        for bq, _, _ in dbs:
            bq.sync()
        torch.cuda.synchronize(); td0 = time.perf_counter()
        for k in range(a.steps):
            dbs[k % nS][0].solve(opts=run_opts, sync=False)
        for bq, _, _ in dbs:
            bq.sync()
        torch.cuda.synchronize(); dtd = time.perf_counter() - td0
        for bq, r2, s2 in dbs:
            o2 = bq.download(); nv = 0
            for i in np.flatnonzero(o2["exitflag"] == 1):
                nv += bool(V.validate_parking(r2["x0"][i], r2["xF"][i], N, r2["Ts"][i, 0], s2["L"], s2["ego"], s2["XYbounds"], np.ravel(s2["vOb"]), s2["A"], s2["b"],
                                              o2["xp"][i], o2["up"][i], o2["timeScale"][i], o2["lp"][i], o2["np"][i], o2["sl"][i], tol=1e-4)[0])
            dval.append(nv); dmaxp.append(int((o2["iters"] + o2["info"][:, 6]).max()))
            bq.close()
        solved = sum(dval[k % nS] for k in range(a.steps))
        distinct = dict(solves_per_s=round(solved / dtd, 1), batches=nS, instances=nS * B, validated=dval, longest_solve_passes=dmaxp,
                        note="every stream holds a different batch of the same distribution (other seeds): the timed steps cover their stragglers too; never `value`")
    # ---- results: every copy solved the same inputs and must hold the same bits
    outs = [bq.download() for bq in batches[:min(nS, a.steps + a.warmup)]]
    out = outs[0]
    other_leg = None; others = None
    if rank == 0 and world == 1 and not a.no_ipopt_leg:
        other_leg = options_leg(cfg, rows, shared, B, N, nS, local, max(8, min(a.steps, 40)), out, not fast)
    if rank == 0 and world == 1 and cfg == 2 and not a.no_other_configs:
        others = [other_config_line(c_, local, other_prepared[c_], other_cpu.get(c_)) for c_ in (3, 4, 5)]
    same = all(np.array_equal(o["info"], out["info"]) and np.array_equal(np.asarray(o["xp"]), np.asarray(out["xp"])) for o in outs[1:])
    # ---- and the same bits after a kernel has left a large pattern in the registers, LDS and scratch of every CU (until the end of round 5 the parking
    # kernels' termination test read LDS words nothing had written).  A report in `config`, never a condition of the line.
    try:
        from obca_amd import diag      # (a diagnostic library of its own, libobca_diag.so: nothing of it is in the product library)
        diag.leave_pattern(batches[0].ctx, 15, 1e30)
        batches[0].solve(opts=run_opts, sync=True); o_ = batches[0].download()
        after_pattern = bool(np.array_equal(o_["info"], out["info"]) and np.array_equal(np.asarray(o_["xp"]), np.asarray(out["xp"])))
    except Exception as e:      # noqa: BLE001 -- a diagnostic must not cost the line
        after_pattern = "not run: %r" % (e,)
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
            f_pass = F_PASS_QUAD; kernel = "obca_quad_ipm_kernel"
            b_pass = b_iter_quad(N) * float(out["info"][:, 1].sum()) / max(1.0, passes0)      # per pass = per iteration x iterations / passes of this launch
        else:
            if cfg == 5:
                nb_mean = float(np.mean([len(np.ravel(v)) for v in vOb])); f_pass = f_pass_parking(N, nb_mean)
                b_pass = float(np.mean([b_pass_parking(N, len(np.ravel(v)), int(np.sum(v))) for v in vOb]))
            else:
                f_pass = f_pass_parking(N, len(np.ravel(vOb))); b_pass = b_pass_parking(N, len(np.ravel(vOb)), int(np.sum(vOb)))
            kernel = "obca_parking_ipm_kernel"
        tflops = passes0 * f_pass / (k_ms * 1e-3) / 1e12
        traffic, tsrc = (None, "skipped (--no-pmc)") if a.no_pmc or world > 1 else live_pmc_traffic(kernel, cfg, B, fast)
        fp64_frac = tflops / FP64_PEAK_TFLOPS
        # SURVEY 8d, Model B (the condensed variant these kernels are): executed flops; HBM bytes = I/O only -- the problem data in and the result tuple out, once per solve
        io_bytes = B * (IO_BYTES_PER_SOLVE_QUAD if quad else io_bytes_parking(N, vOb, cfg == 5))
        io_gbs = io_bytes / (k_ms * 1e-3) / 1e9
        impl_bytes = passes0 * b_pass                                 # what the IMPLEMENTATION streams per launch by its own model: not the algorithm's bytes
        impl_gbs = impl_bytes / (k_ms * 1e-3) / 1e9
        step_s = dt / a.steps
        roof = {"bound": "mfma",
                "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fp64_frac, 5),
                "traffic": traffic, "traffic_source": tsrc,
                "bound_detail": "SURVEY 8d Model B (condensed KKT solve): ALGORITHMIC work = executed fp64 flops per factorisation pass x the passes the kernel reports; algorithmic HBM bytes = the "
                                "I/O of the solves only (algorithmic_bytes_io_only), a fraction ~1e-4 of the HBM roof -- so the flop roof is the one `frac` is quoted on: fp64 vector = fp64 matrix "
                                "peak 78.6 TFLOP/s (the parking kernel issues no MFMA; the quadcopter sweep does).  `frac` is for ONE launch alone on the GPU (HIP events = rocprofv3 --stats); "
                                "a lone launch ends with its slowest instance (3-4 x the mean number of passes), see `regime_of_value` for the pipelined load `value` is measured at",
                "model": "SURVEY 8d Model B", "kernel": kernel, "kernel_ms": round(k_ms, 3), "kernel_ms_all": [round(x, 3) for x in ipm_ms],
                "kernel_timing": "HIP events on the launch stream around the interior-point launches of ONE synchronous step of rank 0's batch, measured in this process after the "
                                 "timed region (median of %d); nothing else runs on the GPU" % len(ipm_ms),
                "passes_per_launch": int(passes0), "flops_per_pass_model": f_pass, "fp64_tflops": round(tflops, 3), "fp64_frac": round(fp64_frac, 5),
                "algorithmic_bytes_io_only": int(io_bytes), "hbm_io_only_gbs": round(io_gbs, 3), "hbm_io_only_frac": round(io_gbs / HBM_PEAK_GBS, 7),
                "implementation_bytes": int(impl_bytes), "implementation_bytes_per_pass_model": b_pass, "implementation_hbm_gbs": round(impl_gbs, 1),
                "implementation_hbm_frac": round(impl_gbs / HBM_PEAK_GBS, 5),
                "implementation_note": "implementation_* = the kernel's own streaming model (every array of the per-instance state x the times a factorisation pass reads / writes it, DESIGN.md "
                                       "section 5): what the code moves because an instance's state (0.2 MB) does not fit the LDS share of its CU, NOT algorithmic bytes.  The working set of a "
                                       "resident batch (1 024 x 0.2 MB = 205 MB) is about the size of the 256 MiB Infinity Cache and FETCH_SIZE / WRITE_SIZE count fabric requests including MALL "
                                       "hits (MI355X_MICROARCH.md, HBM), so much of `traffic` never reaches HBM and the 8 TB/s roof does not bind it",
                "regime_of_value": {"note": "the same work over the wall time of a pipelined step (%d launches in flight): device utilisation in the regime `value` is measured in, not a kernel roofline" % nS,
                                    "fp64_tflops": round(passes0 * f_pass / step_s / 1e12, 3), "fp64_frac": round(passes0 * f_pass / step_s / 1e12 / FP64_PEAK_TFLOPS, 5),
                                    "implementation_hbm_gbs": round(impl_bytes / step_s / 1e9, 1), "implementation_hbm_frac": round(impl_bytes / step_s / 1e9 / HBM_PEAK_GBS, 5)}}
        if traffic is not None:     # measured bytes of one launch over the kernel time / the pipelined step time of THIS run
            roof.update(traffic_over_io_only=round(traffic / io_bytes, 1), traffic_over_implementation_model=round(traffic / impl_bytes, 3),
                        traffic_gbs_one_launch=round(traffic / (k_ms * 1e-3) / 1e9, 1), traffic_frac_of_hbm_peak_one_launch=round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            roof["regime_of_value"].update(traffic_gbs=round(traffic / step_s / 1e9, 1), traffic_frac_of_hbm_peak=round(traffic / step_s / 1e9 / HBM_PEAK_GBS, 4))
        if not quad:
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
                       "exitflag2": int((ef == 2).sum()), "copies_bit_identical": bool(same), "same_bits_after_foreign_pattern_on_the_cus": after_pattern, "mean_iterations": round(float(iters.mean()), 2), "max_iterations": int(iters.max()),
                       "p95_iterations": float(np.percentile(iters, 95)), "mean_passes": round(passes_all / B_total, 2),
                       "single_batch_sync_solves_per_s": round(conv_all / world / float(np.median(sync_s)), 1),
                       "single_batch_sync_note": "one batch issued and waited for (reset + DualMultWS + interior point, inputs resident): what a caller without several batches in flight gets",
                       "planning": None if not needs_planner(cfg, a.warm_start == "hybrid") else {
                           "seconds": round(t_plan, 2), "instances_per_rank": B, "host_cpus_visible": os.cpu_count(), "host_cpus_effective": effective_cpus(),
                           "end_to_end_solves_per_s": round(conv_all / (t_plan + dt / a.steps), 1),
                           "note": "warm starts of this config come from the host-side planner (Hybrid A* on the library's threads / 3-D A*), run ONCE before the timed region, "
                                   "every rank for its own slice; end_to_end = validated solves of one batch / (planning + one step): the planner, not the solve, bounds a "
                                   "pipeline that plans every instance afresh; never `value`"},
                       "distinct_batches": distinct,
                       "options": OPTION_NAMES[(C["kind"], fast)],
                       ("reference_options" if fast else "fast_options"): other_leg,
                       "other_configs": others,
                       "host_pointer": host_rate,
                       "host_pointer_note": "obca_parking_signed_dist_batch on host arrays (the entry point the Julia shim binds): packing, PCIe both ways, kernels, unpacking; never `value`"},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
