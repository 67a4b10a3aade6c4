"""world_size-2 gloo tests of the N>1 path: ONE host batch on rank 0 is scattered over the ranks, every rank solves its slice, rank 0 gets
the reference's full result tuple per instance (obca_amd/sharding.py).  On this GPU-less machine the slices are solved by the HIP solver
source compiled as a host emulation (tests/emu, the kernels' own logic: DualMultWS sub-problems + interior point); the GPU suite runs the
same two-rank flow through libobca_hip.so (tests/test_gpu_multi.py)."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import ROOT

N_T, B_T = 12, 5


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obca_amd import scenarios as S, sharding
    import emu_solver as E
    bt = None
    sc = S.make_batch(S.BACKWARDS, B_T, N_T)
    if rank == 0:
        xWS = sc["xWS"].copy(); xWS[:, 0, :] = sc["x0"]
        bt = dict(x0=sc["x0"], xF=sc["xF"], Ts=sc["Ts"], rx=xWS[:, :, 0], ry=xWS[:, :, 1], ryaw=xWS[:, :, 2], xWS=xWS, uWS=sc["uWS"])
    out = sharding.parking_signed_dist_sharded(bt, N_T, sc["L"], sc["ego"], sc["XYbounds"], sc["vOb"], sc["A"], sc["b"], 0, rank, world,
                                               solver=E.parking_signed_dist_batch)
    summ = sharding.gather_summaries(np.full((sharding.shard_range(B_T, rank, world)[1] - sharding.shard_range(B_T, rank, world)[0], 1), float(rank)), B_T, rank, world)
    dist.barrier()
    ret[rank] = (out, summ)
    dist.destroy_process_group()


def test_shard_ranges_cover_batch():
    from obca_amd.sharding import shard_range
    for B in (1, 5, 1024, 1000):
        for G in (1, 2, 3, 8):
            r = [shard_range(B, k, G) for k in range(G)]
            assert r[0][0] == 0 and r[-1][1] == B and all(r[k][1] == r[k + 1][0] for k in range(G - 1))


@pytest.mark.timeout(600)
def test_two_rank_gloo_scatter_solve_gather_full_tuple(emu):
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29533 + os.getpid() % 500, ret), nprocs=world, join=True)
    out, summ = ret[0]
    assert ret[1][0] is None                                    # only the source rank holds the gathered batch
    assert np.array_equal(summ[:, 0], [0, 0, 0, 1, 1]) and np.array_equal(ret[1][1], summ)       # contiguous slices: 3 + 2
    # the same batch solved in one process by the same solver: sharding must not change a bit of any output
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_solver as E
    from obca_amd import scenarios as S
    sc = S.make_batch(S.BACKWARDS, B_T, N_T)
    xWS = sc["xWS"].copy(); xWS[:, 0, :] = sc["x0"]
    ref = E.parking_signed_dist_batch(sc["x0"], sc["xF"], N_T, sc["Ts"], sc["L"], sc["ego"], sc["XYbounds"], sc["vOb"], sc["A"], sc["b"],
                                      xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, sc["uWS"])
    assert np.all(out["exitflag"] == 1) and np.array_equal(out["exitflag"], ref["exitflag"]) and np.array_equal(out["iters"], ref["iters"])
    for k in ("xp", "up", "timeScale", "info"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k])), k
    for k in ("lp", "np", "sl"):
        for i in range(B_T):
            assert np.array_equal(out[k][i], ref[k][i]), (k, i)
    assert out["xp"].shape == (B_T, 4, N_T + 1) and out["lp"][0].shape == (5, N_T + 1) and out["np"][0].shape == (12, N_T + 1)


# ---------------------------------------------------------------- the other tuples of the N > 1 path: quadcopter, and parking with per-instance obstacle sets
NQ_T, BQ_T = 10, 3
NR_T, BR_T = 10, 6


def _worker_quad(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obca_amd import scenarios as S, sharding
    import emu_solver as E
    q = S.make_quad_batch(BQ_T, NQ_T, seed=2)
    bt = dict(x0=q["x0"], xF=q["xF"], Ts=np.full(BQ_T, q["Ts"]), timeWS=np.full(BQ_T, 1.0), xWS=q["xWS"]) if rank == 0 else None
    out = sharding.quadcopter_signed_dist_sharded(bt, NQ_T, q["R"], q["ob"], rank, world, solver=E.quadcopter_signed_dist_batch)
    dist.barrier()
    ret[rank] = out
    dist.destroy_process_group()


def _ragged_batch():
    from obca_amd import scenarios as S
    bt = S.make_mixed_batch(BR_T, NR_T, seed=5, min_obstacles=1, max_extra=2)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    return bt, xWS


def _ragged_solver(x0, xF, N, Ts, L, ego, XYb, vl, Al, bl, rx, ry, ryaw, fixTime, xWS, uWS, **_):
    """per-instance obstacle sets through the emulated kernels: one call per instance, results stacked like obca_amd.parking_signed_dist_batch"""
    import emu_solver as E
    outs = [E.parking_signed_dist_batch(x0[i:i + 1], xF[i:i + 1], N, np.atleast_1d(Ts)[i:i + 1], L, ego, XYb, vl[i], Al[i], bl[i], rx[i:i + 1], ry[i:i + 1], ryaw[i:i + 1], fixTime,
                                        xWS[i:i + 1], uWS[i:i + 1]) for i in range(len(x0))]
    cat = lambda k: np.concatenate([o[k] for o in outs], axis=0)
    return dict(xp=cat("xp"), up=cat("up"), timeScale=cat("timeScale"), exitflag=cat("exitflag"), info=cat("info"), lp=[o["lp"][0] for o in outs], np=[o["np"][0] for o in outs],
                sl=[o["sl"][0] for o in outs])


def _worker_ragged(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obca_amd import sharding
    bt, xWS = _ragged_batch()
    b = dict(x0=bt["x0"], xF=bt["xF"], Ts=bt["Ts"], rx=xWS[:, :, 0], ry=xWS[:, :, 1], ryaw=xWS[:, :, 2], xWS=xWS, uWS=bt["uWS"], vOb=bt["vOb"], A=bt["A"], b=bt["b"]) if rank == 0 else None
    out = sharding.parking_signed_dist_sharded_ragged(b, NR_T, bt["L"], bt["ego"], bt["XYbounds"], 0, rank, world, solver=_ragged_solver)
    dist.barrier()
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_quadcopter_tuple(emu):
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker_quad, args=(world, 29633 + os.getpid() % 500, ret), nprocs=world, join=True)
    out = ret[0]
    assert ret[1] is None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_solver as E
    from obca_amd import scenarios as S
    q = S.make_quad_batch(BQ_T, NQ_T, seed=2)
    ref = E.quadcopter_signed_dist_batch(q["x0"], q["xF"], NQ_T, q["Ts"], q["R"], q["ob"], q["xWS"], 1.0)
    assert out["xp"].shape == (BQ_T, 12, NQ_T + 1) and out["up"].shape == (BQ_T, 4, NQ_T) and out["lp"].shape == (BQ_T, 30, NQ_T + 1) and out["slack"].shape == (BQ_T, 5, NQ_T + 1)
    assert (ref["exitflag"] >= 1).all()
    for k in ("xp", "up", "timeScale", "exitflag", "lp", "slack", "info"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k])), k


@pytest.mark.timeout(600)
def test_two_rank_gloo_ragged_obstacle_sets_dealt_round_robin(emu):
    """BASELINE config 5's tuple: per-instance obstacle sets travel with the rows, the instances are dealt to the ranks by (nOb, M) buckets, and rank 0 gets every
    instance's result back in the caller's order, bit-equal to solving the batch in one process"""
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker_ragged, args=(world, 29733 + os.getpid() % 500, ret), nprocs=world, join=True)
    out = ret[0]
    assert ret[1] is None
    bt, xWS = _ragged_batch()
    assert len(set(len(np.ravel(v)) for v in bt["vOb"])) >= 2            # really ragged
    ref = _ragged_solver(bt["x0"], bt["xF"], NR_T, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    assert (out["exitflag"] == 1).all()
    for k in ("xp", "up", "timeScale", "exitflag", "info"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k])), k
    for k in ("lp", "np", "sl"):
        for i in range(BR_T):
            assert out[k][i].shape == ref[k][i].shape and np.array_equal(out[k][i], ref[k][i]), (k, i)


def test_balanced_permutation_deals_every_rank_the_same_mix():
    from obca_amd.sharding import balanced_permutation, shard_range
    rng = np.random.default_rng(1)
    for B, G in ((4096, 8), (1000, 8), (10, 3), (7, 8), (1, 2)):
        keys = rng.integers(1, 11, B) * 100 + rng.integers(1, 41, B)
        perm, inv = balanced_permutation(keys, G)
        assert sorted(perm) == list(range(B)) and np.array_equal(perm[inv], np.arange(B))
        if B >= 1000:
            sums = [keys[perm[slice(*shard_range(B, r, G))]].sum() for r in range(G)]
            assert max(sums) - min(sums) <= 0.01 * max(sums)              # static slices of the unsorted batch differ by several per cent
