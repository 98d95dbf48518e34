"""
Multi-GPU form of the hot path: one process per GPU, reads sharded by contiguous
ranges, ONE sum-allreduce of the integer count tables (RCCL over xGMI when the
tensors are on the GPU; `gloo` works for CPU tensors in tests), then the model
solve replicated on every rank and a purely local apply.  SURVEY.md 8(e).

Nothing here computes on the data path: K1 / K2 are the HIP kernels, the
collective is torch.distributed's.
"""
import sys

import numpy as np


def _dist():
    """torch.distributed when a process group can exist: a process that has not imported torch has none (and the
    single-GPU command line should not pay for the import before its files are being read)."""
    if 'torch' not in sys.modules:
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def shard_range(nreads, rank, world):
    """Contiguous records [lo, hi) of rank `rank`; pairs stay together (even boundaries)."""
    per = -(-nreads // world)
    per += per & 1
    lo = min(rank * per, nreads)
    return lo, min(lo + per, nreads)


def allreduce_tables(buf):
    """In-place SUM over all ranks of the concatenated count tables
    [pos_errs | pos_total | dinuc_errs | dinuc_total] (int64)."""
    dist = _dist()
    if dist is not None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def merge_rg_maps(local_names):
    """Global first-appearance read-group order from per-rank first-appearance lists
    (rank order == read order for contiguous shards).  Returns (global_names, remap) where
    remap[i] is the global id of this rank's local id i."""
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, list(local_names))
    else:
        gathered = [list(local_names)]
    order = {}
    for names in gathered:
        for nm in names:
            order.setdefault(nm, len(order))
    return list(order), np.array([order[nm] for nm in local_names], dtype=np.int64)


def broadcast_object(obj, src=0):
    """`obj` of rank `src` on every rank (any picklable Python object); without a process group: obj itself."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def all_gather_object(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (any picklable objects); without a process group: [obj]."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [obj]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, obj)
    return everyone


def world_rank():
    """(world size, rank) of the initialised process group, (1, 0) without one."""
    dist = _dist()
    if dist is not None:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def init_from_env():
    """One process per GPU under torch.distributed.run: bind cuda:LOCAL_RANK and join the RCCL group
    (backend "nccl" is RCCL on ROCm; KBBQ_DIST_BACKEND=gloo lets several ranks share one GPU for
    rehearsals).  No-op outside a launcher or when already initialised."""
    import os
    if 'RANK' not in os.environ or (int(os.environ.get('WORLD_SIZE', '1')) <= 1 and not os.environ.get('KBBQ_DIST_ALWAYS')):
        return world_rank()                  # (KBBQ_DIST_ALWAYS=1: join the group even as its only rank -- the RCCL path on a one-GPU box)
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return world_rank()
    backend = os.environ.get('KBBQ_DIST_BACKEND', 'nccl')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count() if backend != 'nccl' else local
        torch.cuda.set_device(local)
        bind_host_threads(local)
    # stdout is the recalibrated FASTQ: the communication libraries' own banners (Gloo's "connected to peer
    # ranks", RCCL's version line under NCCL_DEBUG) go to stderr -- they appear while the group is set up and
    # at the first collective, so both happen under the redirection
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
            warm = torch.zeros(1, device='cuda')
        else:
            dist.init_process_group(backend)
            warm = torch.zeros(1)
        dist.all_reduce(warm)
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return world_rank()


HOST_BINDING = {}        # what bind_host_threads did for this process (stage reports, bench.py's line)


def bind_host_threads(device):
    """A rank's host side under the launcher: its scan / fill / format threads are 1 / LOCAL_WORLD_SIZE of the CPUs the job
    may use (csrc/host_threads.h reads the launcher's variable itself) and run on the NUMA node its GPU hangs on
    (kbbq_bind_host_to_device; KBBQ_NUMA_BIND=0 leaves the affinity alone).  Returns and records {numa_node, cpus,
    host_threads}."""
    import ctypes
    import os
    from . import _native as N
    lib = N.load()
    node, ncpus = ctypes.c_int(-1), ctypes.c_int(0)
    if os.environ.get('KBBQ_NUMA_BIND', '1') != '0':
        try:
            N.check(lib.kbbq_bind_host_to_device(int(device), ctypes.byref(node), ctypes.byref(ncpus)))
        except Exception:                    # noqa: BLE001 -- binding is an optimisation: a box that refuses it still runs
            node.value = -1
    HOST_BINDING.update(numa_node=node.value, cpus=ncpus.value or len(os.sched_getaffinity(0)),
                        host_threads=int(lib.kbbq_host_threads(1 << 40)),
                        local_ranks=int(os.environ.get('KBBQ_LOCAL_RANKS') or os.environ.get('LOCAL_WORLD_SIZE') or 1))
    return dict(HOST_BINDING)


def barrier():
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()


def raise_first_error(exc=None, index=None):
    """Every rank calls this after a data-dependent step with the exception it caught (or None) and
    the GLOBAL index of the read that caused it.  If any rank has one, ALL ranks raise the error of the
    smallest read index -- what a single process walking the reads in order would have raised -- so no
    rank is left waiting in a collective."""
    dist = _dist()
    mine = None if exc is None else (int(index if index is not None else 0), type(exc).__name__, str(exc))
    if dist is not None and dist.get_world_size() > 1:
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
    else:
        everyone = [mine]
    found = [e for e in everyone if e is not None]
    if not found:
        return
    idx, name, msg = min(found, key=lambda e: e[0])
    if exc is not None and mine == (idx, name, msg):
        raise exc
    import builtins
    raise getattr(builtins, name, RuntimeError)(msg)


def in_rank_order(fn):
    """Run fn() on rank 0, then rank 1, ...: ordered output on a shared stdout."""
    world, rank = world_rank()
    for r in range(world):
        if r == rank:
            fn()
        barrier()


def max_over_ranks(value):
    """Largest `value` over all ranks (the global longest read -> S)."""
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        import torch
        backend = dist.get_backend()
        t = torch.tensor([int(value)], dtype=torch.int64,
                         device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())
    return int(value)
