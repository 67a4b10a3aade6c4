"""
Batch sharding across the GPUs of one node: problem instances are independent, so rank r of G takes
the contiguous slice [r*ceil(B/G), (r+1)*ceil(B/G)) -- no collective touches the solve.  The only exchange is the gather of
the per-instance summaries (exit flag, iterations, objective) onto every rank, a single all_gather over RCCL/xGMI
(`nccl` backend on ROCm) or `gloo` in the CPU tests.
"""
import numpy as np


def shard_range(B, rank, world):
    per = -(-B // world)
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def gather_summaries(local, B, rank, world, device=None):
    """local: (n_local, K) float64 summaries of this rank's slice -> (B, K) array on every rank."""
    import torch
    import torch.distributed as dist
    per = -(-B // world)
    K = local.shape[1]
    pad = np.zeros((per, K)); pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    full = torch.cat(out, 0).cpu().numpy()
    return full[:B] if per * world >= B else full
