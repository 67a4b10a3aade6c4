"""world_size-2 gloo test of the N>1 path: shard the batch, 'solve' each slice, gather the summaries on every rank."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import ROOT


def _worker(rank, world, port, B, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from obca_amd import scenarios as S, sharding
    import oracle as O
    N = 12
    bt = S.make_batch(S.BACKWARDS, B, N)
    lo, hi = sharding.shard_range(B, rank, world)
    loc = []
    for i in range(lo, hi):      # the test stands in for the GPU with the oracle; the sharding/gather code is the product's
        xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
        r = O.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"],
                                  bt["b"], xWS[:, 0], xWS[:, 1], xWS[:, 2], 0, xWS, bt["uWS"][i])
        loc.append([r["exitflag"], r["iters"], r["obj"], i])
    full = sharding.gather_summaries(np.array(loc, float).reshape(-1, 4), B, rank, world)
    dist.barrier()
    ret[rank] = full
    dist.destroy_process_group()


def test_shard_ranges_cover_batch():
    from obca_amd.sharding import shard_range
    for B in (1, 5, 1024, 1000):
        for G in (1, 2, 3, 8):
            r = [shard_range(B, k, G) for k in range(G)]
            assert r[0][0] == 0 and r[-1][1] == B and all(r[k][1] == r[k + 1][0] for k in range(G - 1))


@pytest.mark.timeout(300)
def test_two_rank_gloo_shard_and_gather():
    B, world = 5, 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29533 + os.getpid() % 500, B, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a.shape == (B, 4) and np.array_equal(a, b)
    assert np.array_equal(a[:, 3], np.arange(B)) and np.all(a[:, 0] == 1)
