"""Multi-GPU: one process per GPU.  Two ways to split the job, both ending in ONE all-reduce of the flat u64 accumulators
(SimkaStatistics::operator+=, ref: src/core/SimkaDistance.cpp:156-213; RCCL over xGMI with the "nccl" backend, gloo in the
CPU tests; integer sums, so the result is order-independent):

* partition shards (shard_index/shard_count of the context): every rank scans all reads and keeps the level-1 partitions
  p with p % world == rank; count, merge and pair accumulation are rank-local, exactly as one simkaMerge process per
  partition is independent in the reference (ref: src/SimkaPotara.hpp:974-1124).  No exchange besides the all-reduce, but
  the scan is replicated (measured: one rank of 8 still needs 44 % of the single-GPU step on C2).

* sample shards + spectrum exchange (`exchange_spectra`): rank r counts the samples s with s % world == r over the WHOLE
  key space (the reference: one simkaCount job per sample), then the solid spectra move to the rank that owns their
  partition range with one all-to-all (the reference: every simkaMerge job reads partition p of every sample's solid/
  directory, ref: src/SimkaMerge.cpp:1164-1264), the merge is rank-local per partition range, and the accumulators are
  all-reduced.  Nothing is replicated; the exchange moves 12 bytes per SOLID k-mer, a small fraction of the counting traffic.
"""
import os

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
    if dist.get_backend() == "gloo":          # tests: ranks without RCCL (e.g. two processes on one GPU) bounce through the host
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()


def create_comm(rank, world, device):
    """The RCCL communicator of the C ABI (simka_comm_*), bootstrapped through torch.distributed: rank 0 makes the unique id,
    the process group broadcasts its 128 bytes, every rank creates.  A 1-word all-reduce checks it end to end."""
    from .api import Comm
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = Comm(box[0], world, rank, device)
    t = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", device))
    comm.allreduce_u64(t.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if int(t.item()) != world:
        raise RuntimeError("simka_comm: the check all-reduce returned %d on a world of %d" % (int(t.item()), world))
    return comm


def allreduce_stats_device(ctx, totals_already_reduced=False, comm=None):
    """In-place all-reduce of the ctx's DEVICE statistics buffer (no host bounce).  With `comm` (simka_amd.api.Comm) it is the C
    ABI's simka_stats_allreduce: ncclAllReduce on the context's stream, no synchronisation.  Without it (gloo tests) the
    buffer is wrapped as an int64 CUDA tensor through __cuda_array_interface__ and handed to torch.distributed.  After
    allreduce_totals_device() only the head (header + pair arrays) is reduced, so the totals are not summed twice."""
    if comm is not None:
        ctx.allreduce_stats(comm, "head" if totals_already_reduced else "all")
        return
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    ctx.sync()
    if totals_already_reduced:
        (hp, hn), _ = ctx.stats_device_ranges()
        _allreduce_device_words(hp, hn)
    else:
        ptr, n = ctx.stats_device_buffer()
        _allreduce_device_words(ptr, n)


def allreduce_totals_device(ctx, comm=None):
    """-complex-dist, sharded: make the per-sample totals global BEFORE simka_merge."""
    if comm is not None:
        ctx.allreduce_stats(comm, "totals")
        return
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    ctx.sync()
    _, (tp, tn) = ctx.stats_device_ranges()
    _allreduce_device_words(tp, tn)


# ---- sample shards + spectrum exchange ---------------------------------------------------------------------------------
def samples_of(rank, world, nb_samples):
    """Samples counted by `rank`: s % world == rank (ascending)."""
    return list(range(rank, nb_samples, world))


def partition_bounds(nparts, world):
    """Rank g merges the partitions [bounds[g], bounds[g+1])."""
    return [(nparts * g) // world for g in range(world + 1)]


def pack_spectra(local, nparts, world, nb_samples, rank, device=None):
    """Phase A, per rank.  local: {sample: (totals dict, part_counts u32[nparts], keys int64 tensor, counts int32 tensor)} for
    samples_of(rank).  Spectra are partition-major, so the records bound for rank g are ONE slice per sample.
    Returns (meta int32 [world, maxn, width], totals int64 [maxn, 6], keys_send, counts_send, send_splits)."""
    bounds = partition_bounds(nparts, world)
    width = max(bounds[g + 1] - bounds[g] for g in range(world))
    maxn = (nb_samples + world - 1) // world
    mine = samples_of(rank, world, nb_samples)
    meta = np.zeros((world, maxn, max(width, 1)), dtype=np.int32)
    totals = np.zeros((maxn, 6), dtype=np.int64)
    key_parts, count_parts, send_splits = [], [], []
    offs = {}
    for j, s in enumerate(mine):
        t, pc, _, _ = local[s]
        totals[j] = [t["nb_reads"], t["D"], t["N"], t["Q"], t["K_occ"], t["D_all"]]
        offs[s] = np.concatenate([[0], np.cumsum(pc.astype(np.int64))])
        for g in range(world):
            meta[g, j, : bounds[g + 1] - bounds[g]] = pc[bounds[g]: bounds[g + 1]].astype(np.int32)
    for g in range(world):
        n_g = 0
        for s in mine:
            lo, hi = int(offs[s][bounds[g]]), int(offs[s][bounds[g + 1]])
            if hi > lo:
                key_parts.append(local[s][2][lo:hi]); count_parts.append(local[s][3][lo:hi])
            n_g += hi - lo
        send_splits.append(n_g)
    ref = next(iter(local.values())) if local else None
    if key_parts:
        keys_send, counts_send = torch.cat(key_parts), torch.cat(count_parts)
    else:
        dev = device if device is not None else (ref[2].device if ref is not None else "cpu")
        keys_send, counts_send = torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    return meta, totals, keys_send, counts_send, send_splits


def recv_splits_of(meta_recv):
    """Phase B.  meta_recv int32 [world(source), maxn, width]: records each source rank sends to this rank."""
    return [int(meta_recv[r].astype(np.int64).sum()) for r in range(meta_recv.shape[0])]


def unpack_spectra(meta_recv, totals_all, keys_recv, counts_recv, nparts, world, nb_samples, rank):
    """Phase C.  -> [(sample, totals dict, part_counts u32[nparts] (zero outside this rank's range), keys, counts)] for ALL
    samples: what this rank imports before the merge.  totals_all int64 [world, maxn, 6]."""
    bounds = partition_bounds(nparts, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    out, pos = [], 0
    for r in range(world):
        for j, s in enumerate(samples_of(r, world, nb_samples)):
            pc = np.zeros(nparts, dtype=np.uint32)
            pc[lo:hi] = meta_recv[r, j, : hi - lo].astype(np.uint32)
            n = int(pc.astype(np.int64).sum())
            tt = totals_all[r, j]
            t = {"nb_reads": int(tt[0]), "D": int(tt[1]), "N": int(tt[2]), "Q": int(tt[3]), "K_occ": int(tt[4]), "D_all": int(tt[5])}
            out.append((s, t, pc, keys_recv[pos: pos + n], counts_recv[pos: pos + n]))
            pos += n
    assert pos == int(keys_recv.numel())
    return out


def _comm_device(t_dev):
    """gloo moves host tensors; nccl (RCCL) device tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else t_dev


def exchange_spectra(local, nparts, nb_samples, device):
    """All three phases with torch.distributed in between: one all-to-all of the per-range record counts, one all-gather of the
    per-sample totals, one all-to-all each for keys and counts (uneven splits)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    meta, totals, ks, cs, send_splits = pack_spectra(local, nparts, world, nb_samples, rank, device)
    cdev = _comm_device(device)
    meta_t = torch.from_numpy(meta).to(cdev)
    meta_r = torch.empty_like(meta_t)
    dist.all_to_all_single(meta_r, meta_t)
    tot_t = torch.from_numpy(totals).to(cdev)
    tot_list = [torch.empty_like(tot_t) for _ in range(world)]
    dist.all_gather(tot_list, tot_t)
    tot_all = torch.stack(tot_list)
    meta_recv = meta_r.cpu().numpy()
    recv_splits = recv_splits_of(meta_recv)
    ks, cs = ks.to(cdev), cs.to(cdev)
    kr = torch.empty(sum(recv_splits), dtype=torch.int64, device=cdev)
    cr = torch.empty(sum(recv_splits), dtype=torch.int32, device=cdev)
    dist.all_to_all_single(kr, ks, recv_splits, send_splits)
    dist.all_to_all_single(cr, cs, recv_splits, send_splits)
    return unpack_spectra(meta_recv, tot_all.cpu().numpy(), kr.to(device), cr.to(device), nparts, world, nb_samples, rank)


def _excl_cumsum_rows(m):
    """row-wise exclusive prefix sums of a 2-D integer array, as int64"""
    c = np.cumsum(m.astype(np.int64), axis=1)
    return c - m.astype(np.int64)


def _host_table(ctx, name, shape, np_dtype, device):
    """numpy array (zeroed) for a per-step host table of the exchange.  On GPUs: pinned memory, allocated once per context and
    shape -- the tables are megabytes (C3 on eight ranks: 27-54 MB each) and a pageable buffer would be pinned and unpinned by the
    runtime at every copy (the host stalls behind that)."""
    np_dtype = np.dtype(np_dtype)
    pin = torch.cuda.is_available() and torch.device(device).type == "cuda"
    if not pin:
        return np.zeros(shape, dtype=np_dtype)
    cache = ctx.__dict__.setdefault("_xchg_tables", {})
    key = (tuple(int(x) for x in shape), np_dtype.str)
    ent = cache.get(name)
    if ent is None or ent[0] != key:
        tdt = torch.int32 if np_dtype.itemsize == 4 else torch.int64
        ent = (key, torch.empty(key[0], dtype=tdt, pin_memory=True))
        cache[name] = ent
    a = ent[1].numpy().view(np_dtype)
    a[...] = 0
    return a


def pack_batch(ctx, mine, P, kw, world, nb_samples, device):
    """Send side of the batch exchange on one rank: the counted samples `mine` of `ctx` gathered into ONE destination-major buffer
    [g][my sample j][partitions of g].  One-word k-mers: the tables stay on the device (simka_pack_plan / simka_pack_run: with 2^19
    partitions their prefix sums on the host cost ten times the gather itself) and `meta` is an int32 DEVICE tensor; two-word k-mers:
    whole sorted runs, host tables (simka_gather_samples_device_wide), `meta` a numpy array.
    -> (meta [world, maxn, width], totals int64 [maxn, 6], keys, keys2 (kmer_size >= 32: low words), counts, send_splits)."""
    bounds = partition_bounds(P, world)
    width = max(bounds[g + 1] - bounds[g] for g in range(world))
    maxn = (nb_samples + world - 1) // world
    tot_send = np.zeros((maxn, 6), dtype=np.int64)
    if kw == 1 and torch.device(device).type == "cuda":
        for j, s_ in enumerate(mine):
            t = ctx.sample_totals(s_)
            tot_send[j] = [t["nb_reads"], t["D"], t["N"], t["Q"], t["K_occ"], t["D_all"]]
        send_splits = ctx.pack_plan(mine, world) if mine else [0] * world
        meta = torch.zeros((world, maxn, width), dtype=torch.int32, device=device)
        ks = torch.empty(sum(send_splits), dtype=torch.int64, device=device)
        cs = torch.empty(sum(send_splits), dtype=torch.int32, device=device)
        if mine:
            # `meta` was zero-filled on torch's current stream, simka_pack_run writes it on the context's stream: nothing orders the two
            # unless one of them is the legacy null stream (include/simka_hip.h, "stream ordering of caller buffers")
            torch.cuda.current_stream(device).synchronize()
            ctx.pack_run(ks, cs, meta)
        return meta, tot_send, ks, torch.empty(0, dtype=torch.int64, device=device), cs, send_splits
    meta = _host_table(ctx, "meta", (world, maxn, width), np.int32, device)
    send_splits = [0] * world
    if mine:
        pc, tot = ctx.samples_spectrum_info(mine)
        out_off = _host_table(ctx, "out_off", pc.shape, np.uint64, device)
        pos = 0
        for g in range(world):
            lo, hi = bounds[g], bounds[g + 1]
            seg = pc[:, lo:hi]
            rows = seg.astype(np.int64).sum(axis=1)
            starts = pos + np.concatenate([[0], np.cumsum(rows)[:-1]])
            out_off[:, lo:hi] = (_excl_cumsum_rows(seg) + starts[:, None]).astype(np.uint64)
            meta[g, : len(mine), : hi - lo] = seg.astype(np.int32)
            send_splits[g] = int(rows.sum())
            pos += send_splits[g]
        for j in range(len(mine)):
            t = tot[j]
            tot_send[j] = [t.nb_reads, t.nb_distinct, t.nb_kmers, t.sum_sq, t.kmer_occurrences, t.distinct_all]
        ks = torch.empty(pos, dtype=torch.int64, device=device)
        ks2 = torch.empty(pos if kw == 2 else 0, dtype=torch.int64, device=device)
        cs = torch.empty(pos, dtype=torch.int32, device=device)
        if kw == 2:
            ctx.gather_samples_device_wide(mine, out_off, ks, ks2, cs)
        else:
            ctx.gather_samples_device(mine, out_off, ks, cs)
    else:
        ks = torch.empty(0, dtype=torch.int64, device=device)
        ks2 = torch.empty(0, dtype=torch.int64, device=device)
        cs = torch.empty(0, dtype=torch.int32, device=device)
    return meta, tot_send, ks, ks2, cs, send_splits


def import_batch(ctx, rank, world, nb_samples, P, kw, meta_recv, tot_all, kr, kr2, cr, device):
    """Receive side on rank `rank`: the received block [source r][its sample j][my partitions] (run lengths meta_recv int32
    [world, maxn, width] -- a device tensor or a numpy array --, per-sample totals tot_all int64 [world, maxn, 6]) imported into `ctx`
    in ONE call (one-word k-mers: simka_import_block_device, tables on the device).  The context is reset first; the caller merges."""
    from .api import SampleTotals
    bounds = partition_bounds(P, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    w = hi - lo
    if kw == 1 and torch.device(device).type == "cuda":
        meta_dev = meta_recv if isinstance(meta_recv, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(meta_recv))
        meta_dev = meta_dev.to(device=device, dtype=torch.int32).contiguous()
        maxn = int(meta_dev.shape[1])
        slots = np.full(world * maxn, 0xffffffff, dtype=np.uint32)
        tot_in = (SampleTotals * (world * maxn))()
        for r in range(world):
            for j, s_ in enumerate(samples_of(r, world, nb_samples)):
                slots[r * maxn + j] = s_
                tt = tot_all[r, j]
                tot_in[r * maxn + j] = SampleTotals(int(tt[0]), int(tt[1]), int(tt[2]), int(tt[3]), int(tt[4]), int(tt[5]))
        ctx.reset()
        torch.cuda.current_stream(device).synchronize()      # meta_dev / kr / cr come from an asynchronous .to() / all_to_all on torch's stream
        ctx.import_block_device(slots, tot_in, lo, w, meta_dev.view(world * maxn, -1), P, kr, cr)
        return
    if isinstance(meta_recv, torch.Tensor):
        meta_recv = meta_recv.cpu().numpy()
    pc_in = _host_table(ctx, "pc_in", (nb_samples, max(w, 1)), np.uint32, device)
    off_in = _host_table(ctx, "off_in", (nb_samples, max(w, 1)), np.uint64, device)
    tot_in = (SampleTotals * nb_samples)()
    pos = 0
    for r in range(world):
        ss = samples_of(r, world, nb_samples)
        if not ss:
            continue
        seg = meta_recv[r, : len(ss), :w]
        rows = seg.astype(np.int64).sum(axis=1)
        starts = pos + np.concatenate([[0], np.cumsum(rows)[:-1]])
        pc_in[ss, :w] = seg.astype(np.uint32)
        off_in[ss, :w] = (_excl_cumsum_rows(seg) + starts[:, None]).astype(np.uint64)
        pos += int(rows.sum())
        for j, s in enumerate(ss):
            tt = tot_all[r, j]
            tot_in[s] = SampleTotals(int(tt[0]), int(tt[1]), int(tt[2]), int(tt[3]), int(tt[4]), int(tt[5]))
    assert pos == int(kr.numel())
    ctx.reset()
    if kw == 2:      # each sample's slice of my key-prefix range is one contiguous, sorted run of the received block
        s_rec = pc_in.astype(np.int64).sum(axis=1)
        s_off = off_in[:, 0].astype(np.int64) if w else np.zeros(nb_samples, dtype=np.int64)
        ctx.import_samples_device_wide(np.arange(nb_samples), tot_in, s_off, s_rec, kr, kr2, cr)
    else:
        ctx.import_samples_device(np.arange(nb_samples), tot_in, lo, pc_in[:, :max(w, 0)] if w else pc_in[:, :0], off_in[:, :w] if w else off_in[:, :0], P, kr, cr)


def count_exchange_merge(ctx, count_fn, nb_samples, device, comm=None):
    """One sample-sharded job on this rank: count my samples (count_fn(sample)), exchange, import every sample's slice of my
    partition range, merge, all-reduce the pair accumulators.  `ctx` is created with shard_count=1.  Single process: plain path.
    Batch ABI: one gather into a destination-major send buffer, three collectives (counts, keys, counts of k-mers) + the
    totals all-gather, one import of the received block."""
    from .api import SampleTotals
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ctx.reset()
    if world == 1 and not (dist.is_initialized() and os.environ.get("SIMKA_FORCE_EXCHANGE")):
        for s in range(nb_samples):
            count_fn(s)
        ctx.merge()
        return
    cdev = _comm_device(device)
    mine = samples_of(rank, world, nb_samples)
    for s in mine:
        count_fn(s)
    info = ctx.spectrum_info(mine[0]) if mine else (0, 0, 0)
    np_t = torch.tensor([info[1], info[2]], dtype=torch.int64, device=cdev)
    dist.all_reduce(np_t, op=dist.ReduceOp.MAX)          # ranks without samples learn the partition count and the key width
    P, kw = int(np_t[0].item()), int(np_t[1].item())      # kw = 2: kmer_size >= 32, high and low key words travel separately
    maxn = (nb_samples + world - 1) // world
    width = max(b - a_ for a_, b in zip(partition_bounds(P, world)[:-1], partition_bounds(P, world)[1:]))
    # ---- send side: destination-major layout [g][my sample j][partitions of g]
    meta, tot_send, ks, ks2, cs, send_splits = pack_batch(ctx, mine, P, kw, world, nb_samples, device)
    # ---- the exchange: run lengths (device to device with RCCL; through the host with gloo), totals, who sends how much to whom
    meta_t = (meta if isinstance(meta, torch.Tensor) else torch.from_numpy(meta)).to(cdev)
    meta_r = torch.empty_like(meta_t)
    dist.all_to_all_single(meta_r, meta_t)
    tot_t = torch.from_numpy(tot_send).to(cdev)
    tot_list = [torch.empty_like(tot_t) for _ in range(world)]
    dist.all_gather(tot_list, tot_t)
    sp_t = torch.tensor(send_splits, dtype=torch.int64, device=cdev)
    sp_list = [torch.empty_like(sp_t) for _ in range(world)]
    dist.all_gather(sp_list, sp_t)
    recv_splits = [int(sp_list[r][rank].item()) for r in range(world)]
    meta_recv = meta_r              # [source rank][its sample j][my partitions]
    kr2 = None
    if comm is not None:
        # the data path on RCCL through the C ABI: grouped ncclSend / ncclRecv between device buffers, on torch's current stream
        stream = torch.cuda.current_stream().cuda_stream
        kr = torch.empty(sum(recv_splits), dtype=torch.int64, device=device)
        cr = torch.empty(sum(recv_splits), dtype=torch.int32, device=device)
        ctx.sync()                                            # the gather ran on the context's stream
        comm.alltoallv(ks.data_ptr(), send_splits, kr.data_ptr(), recv_splits, 8, stream)
        comm.alltoallv(cs.data_ptr(), send_splits, cr.data_ptr(), recv_splits, 4, stream)
        if kw == 2:
            kr2 = torch.empty(sum(recv_splits), dtype=torch.int64, device=device)
            comm.alltoallv(ks2.data_ptr(), send_splits, kr2.data_ptr(), recv_splits, 8, stream)
        torch.cuda.synchronize()
    else:
        kr = torch.empty(sum(recv_splits), dtype=torch.int64, device=cdev)
        cr = torch.empty(sum(recv_splits), dtype=torch.int32, device=cdev)
        dist.all_to_all_single(kr, ks.to(cdev), recv_splits, send_splits)
        dist.all_to_all_single(cr, cs.to(cdev), recv_splits, send_splits)
        if kw == 2:
            kr2 = torch.empty(sum(recv_splits), dtype=torch.int64, device=cdev)
            dist.all_to_all_single(kr2, ks2.to(cdev), recv_splits, send_splits)
            kr2 = kr2.to(device)
        kr, cr = kr.to(device), cr.to(device)
    ks = ks2 = cs = None
    # ---- receive side: block layout [r][j][my partitions]; samples in ascending order for the import
    import_batch(ctx, rank, world, nb_samples, P, kw, meta_recv, torch.stack(tot_list).cpu().numpy(), kr, kr2, cr, device)
    ctx.merge()
    allreduce_stats_device(ctx, totals_already_reduced=True, comm=comm)      # imported totals are already global on every rank
