"""
Multi-GPU form of the hot path: one process per GPU, reads sharded by contiguous
ranges, ONE sum-allreduce of the integer count tables (RCCL over xGMI when the
tensors are on the GPU; `gloo` works for CPU tensors in tests), then the model
solve replicated on every rank and a purely local apply.  SURVEY.md 8(e).

Nothing here computes on the data path: K1 / K2 are the HIP kernels, the
collective is torch.distributed's.
"""
import numpy as np


def shard_range(nreads, rank, world):
    """Contiguous records [lo, hi) of rank `rank`; pairs stay together (even boundaries)."""
    per = -(-nreads // world)
    per += per & 1
    lo = min(rank * per, nreads)
    return lo, min(lo + per, nreads)


def allreduce_tables(buf):
    """In-place SUM over all ranks of the concatenated count tables
    [pos_errs | pos_total | dinuc_errs | dinuc_total] (int64)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def merge_rg_maps(local_names):
    """Global first-appearance read-group order from per-rank first-appearance lists
    (rank order == read order for contiguous shards).  Returns (global_names, remap) where
    remap[i] is the global id of this rank's local id i."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, list(local_names))
    else:
        gathered = [list(local_names)]
    order = {}
    for names in gathered:
        for nm in names:
            order.setdefault(nm, len(order))
    return list(order), np.array([order[nm] for nm in local_names], dtype=np.int64)


def max_over_ranks(value):
    """Largest `value` over all ranks (the global longest read -> S)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        backend = dist.get_backend()
        t = torch.tensor([int(value)], dtype=torch.int64,
                         device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())
    return int(value)
