"""GPU tests of the sharded / pipelined data path (run with -m gpu on the one-GPU box).  Everything goes through libobca_hip.so:
  * a host-pointer call cut into odd chunks over several worker lanes gives the same bits as one device-resident batch;
  * a multi-device context (the one GPU listed twice stands in for two devices) gives the same bits, full result tuple included;
  * two `torch.distributed` ranks (gloo rendezvous, both on the one GPU) scatter one host batch, solve their slices with the HIP path and
    gather the full tuple on rank 0: bit-equal to the single-rank result."""
import os
import sys
import numpy as np
import pytest
from conftest import ROOT, gpu_verdict
from obca_amd import scenarios as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def OA():
    import obca_amd
    obca_amd.Context(0).close()
    return obca_amd


def _resident(OA, bt, N):
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    ctx = OA.Context(0)
    b = OA.Batch(ctx, len(bt["x0"]), N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    b.solve()
    out = b.download(); b.close(); ctx.close()
    return out, xWS


def _same(a, b, B):
    if not (np.array_equal(a["exitflag"], b["exitflag"]) and np.array_equal(a["info"], b["info"])):
        w = np.argwhere(a["info"] != b["info"])
        raise AssertionError("info rows of %d instances differ (first: instance %d, %s | %s)" % (len(set(w[:, 0])), int(w[0, 0]) if len(w) else -1,
                             a["info"][w[0, 0]].tolist() if len(w) else "", b["info"][w[0, 0]].tolist() if len(w) else "") + gpu_verdict())
    for k in ("xp", "up", "timeScale"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    for k in ("lp", "np", "sl"):
        for i in range(B):
            assert np.array_equal(a[k][i], b[k][i]), (k, i)


@pytest.mark.parametrize("chunk,slots", [(37, 3), (64, 1), (1000, 2)])
def test_chunked_host_call_equals_resident_batch(OA, chunk, slots):
    N, B = 40, 150
    bt = S.make_mixed_batch(B, N, seed=3)                 # ragged obstacle sets: chunks see different shapes, packed offsets are global
    ref, xWS = _resident(OA, bt, N)
    os.environ["OBCA_CHUNK"] = str(chunk); os.environ["OBCA_SLOTS"] = str(slots)
    try:
        ctx = OA.Context(0)
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                           xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], device=ctx)
        out2 = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                            xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], device=ctx)      # cached lane batches reused
        ctx.close()
    finally:
        del os.environ["OBCA_CHUNK"]; del os.environ["OBCA_SLOTS"]
    assert (ref["exitflag"] == 1).mean() > 0.9
    _same(out, ref, B); _same(out2, ref, B)


def test_multi_device_context_full_tuple(OA, oracle):
    N, B = 80, 96
    bt = S.make_batch(S.BACKWARDS, B, N)
    ref, xWS = _resident(OA, bt, N)
    os.environ["OBCA_CHUNK"] = "16"
    try:
        ctx = OA.Context(devices=[0, 0])                  # two "devices": the work queue hands chunks to both
        assert ctx.device_count() == 2
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                           xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], device=ctx)
        ls, ns, ds = OA.dualmult_ws_batch(N, bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], bt["ego"], device=ctx)
        ctx.close()
    finally:
        del os.environ["OBCA_CHUNK"]
    _same(out, ref, B)
    assert (out["exitflag"] == 1).all()
    i = 77
    r = oracle.parking_signed_dist(bt["x0"][i], bt["xF"][i], N, bt["Ts"][i], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                   xWS[i, :, 0], xWS[i, :, 1], xWS[i, :, 2], 0, xWS[i], bt["uWS"][i])
    assert out["iters"][i] == r["iters"] and np.abs(out["xp"][i] - r["xp"]).max() < 1e-6 and np.abs(out["lp"][i] - r["lp"]).max() < 1e-5
    lo, no, do = oracle.dualmult_ws(N, bt["vOb"], bt["A"], bt["b"], xWS[i, :, 0], xWS[i, :, 1], xWS[i, :, 2], bt["ego"])
    assert np.abs(ds[i] - do).max() < 1e-9 and np.abs(ls[i] - lo).max() < 1e-8


def test_quadcopter_chunked_equals_resident(OA):
    N, B = 20, 40
    bt = S.make_quad_batch(B, N, seed=5)
    ctx = OA.Context(0)
    qb = OA.QuadBatch(ctx, B, N); qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"]); qb.solve(); ref = qb.download(); qb.close(); ctx.close()
    os.environ["OBCA_CHUNK"] = "7"
    try:
        ctx = OA.Context(devices=[0, 0])
        out = OA.quadcopter_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"], device=ctx)
        ctx.close()
    finally:
        del os.environ["OBCA_CHUNK"]
    for k in ("xp", "up", "timeScale", "exitflag", "lp", "slack", "info"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k])), k
    assert (out["exitflag"] >= 1).mean() > 0.9


def _rank(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks share the one GPU: rendezvous and collectives on gloo
    from obca_amd import sharding
    N, B = 80, 70
    sc = S.make_batch(S.BACKWARDS, B, N)
    bt = None
    if rank == 0:
        xWS = sc["xWS"].copy(); xWS[:, 0, :] = sc["x0"]
        bt = dict(x0=sc["x0"], xF=sc["xF"], Ts=sc["Ts"], rx=xWS[:, :, 0], ry=xWS[:, :, 1], ryaw=xWS[:, :, 2], xWS=xWS, uWS=sc["uWS"])
    out = sharding.parking_signed_dist_sharded(bt, N, sc["L"], sc["ego"], sc["XYbounds"], sc["vOb"], sc["A"], sc["b"], 0, rank, world, local_device=0)
    dist.barrier()
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_scatter_solve_gather(OA):
    """the N>1 launch path end to end through the HIP library: no oracle, no emulation"""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_rank, args=(world, 29611 + os.getpid() % 300, ret), nprocs=world, join=True)
    out = ret[0]
    assert ret[1] is None
    N, B = 80, 70
    sc = S.make_batch(S.BACKWARDS, B, N)
    ref, _ = _resident(OA, sc, N)
    _same(out, ref, B)
    assert (out["exitflag"] == 1).all() and out["xp"].shape == (B, 4, N + 1) and out["np"][0].shape == (12, N + 1)


def _nccl_rank(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from obca_amd import sharding
    rng = np.random.default_rng(1); full = rng.standard_normal((7, 5))
    loc = sharding.scatter_rows(full if rank == 0 else None, 7, 5, rank, world, 0)
    back = sharding.gather_rows(loc * 2.0, 7, rank, world, 0)
    summ = sharding.gather_summaries(loc[:, :2], 7, rank, world)
    meta = [("hello", 3)] if rank == 0 else [None]
    dist.broadcast_object_list(meta, src=0)
    dist.barrier()
    ret[rank] = (np.array_equal(loc, full), np.array_equal(back, 2.0 * full), np.array_equal(summ, full[:, :2]), meta[0])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharding_collectives_on_rccl(OA):
    """the collectives bench.py uses under torchrun (scatter, gather, all_gather, object broadcast) on the `nccl` = RCCL backend with device tensors: one rank on the
    one GPU of this box (two RCCL ranks cannot share a device); the 8-GPU runs are the driver's"""
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_nccl_rank, args=(1, 29711 + os.getpid() % 200, ret), nprocs=1, join=True)
    assert ret[0] == (True, True, True, ("hello", 3))


def test_caller_kept_output_buffers_are_reused(OA):
    """`buffers=`: the output arrays of one call are written again by the next call of the same shape (no fresh 280 MB per 16 384-instance call); a call of another
    shape gets new arrays; results equal the plain call's"""
    N, B = 40, 96
    bt = S.make_batch(S.BACKWARDS, B, N, seed=9)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    call = lambda b_, n, **kw: OA.parking_signed_dist_batch(b_["x0"][:n], b_["xF"][:n], N, b_["Ts"][:n], b_["L"], b_["ego"], b_["XYbounds"], b_["vOb"], b_["A"], b_["b"],
                                                            xWS[:n, :, 0], xWS[:n, :, 1], xWS[:n, :, 2], 0, xWS[:n], b_["uWS"][:n], **kw)
    ref = call(bt, B)
    keep = {}
    o1 = call(bt, B, buffers=keep); p1 = keep["xp"].ctypes.data
    o2 = call(bt, B, buffers=keep)
    assert keep["xp"].ctypes.data == p1 and np.array_equal(o1["xp"], ref["xp"]) and np.array_equal(o2["xp"], ref["xp"]) and np.array_equal(o2["up"], ref["up"])
    o3 = call(bt, B // 2, buffers=keep)
    assert keep["xp"].shape[0] == B // 2 and np.array_equal(o3["xp"], ref["xp"][:B // 2]) and (o3["exitflag"] == ref["exitflag"][:B // 2]).all()
