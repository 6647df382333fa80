"""Multi-GPU: one process per GPU, partitions of the key space sharded over ranks, ONE all-reduce.

Every rank scans all reads and keeps the level-1 partitions p with p % world == rank (SimkaKeyCfg
shard_index/shard_count); count, merge and pair accumulation are then rank-local, exactly as one
simkaMerge process per partition is independent in the reference (ref: src/SimkaPotara.hpp:974-1124).
The only exchange is the reduction of the flat u64 accumulator buffer -- SimkaStatistics::operator+=
(ref: src/core/SimkaDistance.cpp:156-213) -- as a single all-reduce(sum): RCCL over xGMI with the
"nccl" backend on GPUs, gloo in the CPU tests.  Integer sums: the result is order-independent.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_of(rank=None, world=None):
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return rank, world


def allreduce_stats_host(flat_u64):
    """Sum a host copy of the flat statistics buffer over all ranks (gloo or nccl via a device bounce)."""
    t = torch.from_numpy(np.ascontiguousarray(flat_u64, dtype=np.uint64).view(np.int64).copy())
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu()
    return t.numpy().view(np.uint64).copy()


def _allreduce_device_words(ptr, n):
    class _Wrap:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}

    t = torch.as_tensor(_Wrap(), device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()


def allreduce_stats_device(ctx, totals_already_reduced=False):
    """In-place all-reduce of the ctx's DEVICE statistics buffer (no host bounce): the buffer is wrapped as an
    int64 CUDA tensor through __cuda_array_interface__ and handed to RCCL.  After allreduce_totals_device() only the
    head (header + pair arrays) is reduced, so the totals are not summed twice."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    ctx.sync()
    if totals_already_reduced:
        (hp, hn), _ = ctx.stats_device_ranges()
        _allreduce_device_words(hp, hn)
    else:
        ptr, n = ctx.stats_device_buffer()
        _allreduce_device_words(ptr, n)


def allreduce_totals_device(ctx):
    """-complex-dist, sharded: make the per-sample totals global BEFORE simka_merge."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    ctx.sync()
    _, (tp, tn) = ctx.stats_device_ranges()
    _allreduce_device_words(tp, tn)
