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
