#!/usr/bin/env python3
"""Compute time per RANK of the two multi-GPU decompositions, measured on ONE GPU for G = 1, 2, 4, 8 ranks (collectives unmeasured).

    python scripts/shard_projection.py [workload=c3] [out.json]

(a) partition shards (north_star's split: rank g keeps the minimizer partitions p % G == g, every rank scans every read, ONE all-reduce
    of the N x N partials, ref: src/SimkaPotara.hpp:974-1124 one merge job per partition): the whole step of shard g of G, for EVERY g --
    the slowest rank bounds the step; its kernel table shows the replicated scan.
(b) sample shards (one count job per sample, one merge job per partition range -- the reference's own job structure, ref:
    src/SimkaPotara.hpp:813-1124): rank r counts the samples s % G == r (timed), exports their spectra to device buffers (timed; the bytes
    it would send to each peer are reported), and rank g imports the partition range [P g / G, P (g + 1) / G) of every sample's spectrum and
    merges it (timed, for g = 0, G / 2 and G - 1).  The all-to-all of the spectra and the all-reduce of the heads are NOT timed: a one-GPU box.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simka_amd
from simka_amd import dist as sdist
import bench

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c3"
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "shard_projection_%s.json" % wl_name)
wl = dict(bench.WORKLOADS[wl_name])
lib = simka_amd.load_library()
dev = torch.device("cuda:0")
_, reads = bench.gen_device_samples(lib, torch, wl, dev)
n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
kw = dict(kmer_size=k, abundance_min=wl["amin"], simple_dist=wl["simple"], complex_dist=bool(wl.get("complex")), max_kmers_per_sample=R * (L - k + 1))


def sync():
    torch.cuda.synchronize()


def kernel_ms(ctx):
    return {name: round(ms, 3) for name, (cnt, ms) in ctx.profile().items() if cnt}


res = {"workload": wl["desc"], "note": "compute per rank on ONE MI355X, collectives unmeasured; ms per step", "partition_shards": {}, "sample_shards": {}}
checksum = None
for G in (1, 2, 4, 8):
    # ---- (a) partition shards
    per_rank, tables = [], []
    for g in range(G):
        with simka_amd.SimkaContext(n, shard_index=g, shard_count=G, **kw) as ctx:
            def step():
                ctx.reset()
                for s in range(n):
                    ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
                ctx.merge()
                return ctx.stats()
            step(); sync()
            t0 = time.perf_counter(); step(); sync(); ms = (time.perf_counter() - t0) * 1e3
            ctx.profile_enable(True); ctx.profile_reset(); step(); sync()
            per_rank.append(round(ms, 2)); tables.append(kernel_ms(ctx))
    slow = int(np.argmax(per_rank))
    res["partition_shards"][str(G)] = {"step_ms_per_rank": per_rank, "bounding_rank": slow, "step_ms": per_rank[slow], "kernels_ms_of_bounding_rank": tables[slow],
                                       "allreduce_bytes": int(simka_amd.api.stats_layout(n, (1 if wl["simple"] else 0) | (2 if wl.get("complex") else 0))["head"]) * 8}
    print("partition shards G=%d: per rank %s" % (G, per_rank), flush=True)
    if G == 1:
        res["sample_shards"]["1"] = {"step_ms": per_rank[0]}
        continue
    # ---- (b) sample shards
    count_ms, gather_ms, sends, nparts = [], [], [], None
    for r in range(G):
        mine = sdist.samples_of(r, G, n)
        with simka_amd.SimkaContext(n, **kw) as ctx:
            def count():
                ctx.reset()
                for s in mine:
                    ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
                ctx.sync()
            count(); sync()
            t0 = time.perf_counter(); count(); sync(); count_ms.append(round((time.perf_counter() - t0) * 1e3, 2))
            local = {s: ctx.export_sample_device(s, dev) for s in mine}        # (first pass: the allocator's first touch of the send buffers)
            sync(); del local
            t0 = time.perf_counter()
            local = {s: ctx.export_sample_device(s, dev) for s in mine}
            sync(); gather_ms.append(round((time.perf_counter() - t0) * 1e3, 2))
        nparts = len(next(iter(local.values()))[1])
        sends.append(local)
    bounds = sdist.partition_bounds(nparts, G)
    # bytes rank r sends to every peer (12 per solid record), and what rank g imports
    send_bytes = []
    for r in range(G):
        row = []
        for g in range(G):
            row.append(int(sum(int(pc[bounds[g]:bounds[g + 1]].astype(np.int64).sum()) for (_, pc, _, _) in sends[r].values()) * 12))
        send_bytes.append(row)
    merge = {}
    for g in sorted({0, G // 2, G - 1}):
        lo, hi = bounds[g], bounds[g + 1]
        imports = []
        for r in range(G):
            for s, (tot, pc, keys, counts) in sends[r].items():
                pre = np.concatenate([[0], np.cumsum(pc.astype(np.int64))])
                pcm = np.zeros_like(pc); pcm[lo:hi] = pc[lo:hi]
                imports.append((s, tot, pcm, keys[int(pre[lo]): int(pre[hi])], counts[int(pre[lo]): int(pre[hi])]))
        with simka_amd.SimkaContext(n, **kw) as ctx:
            def imp():
                ctx.reset()
                for s, tot, pcm, kk, cc in imports:
                    ctx.import_sample_device(s, tot, pcm, kk, cc)
                ctx.sync()
            imp(); ctx.merge(); ctx.stats(); sync()           # (first pass: arena chunks mapped, merge buffers allocated)
            t0 = time.perf_counter(); imp(); sync(); t_imp = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter(); ctx.merge(); ctx.stats(); sync(); t_mrg = (time.perf_counter() - t0) * 1e3
            imp(); ctx.profile_enable(True); ctx.profile_reset(); ctx.merge(); ctx.stats(); sync()
            merge[str(g)] = {"import_ms": round(t_imp, 2), "merge_ms": round(t_mrg, 2), "kernels_ms": kernel_ms(ctx)}
    del sends
    torch.cuda.empty_cache()
    worst_merge = max(v["import_ms"] + v["merge_ms"] for v in merge.values())
    res["sample_shards"][str(G)] = {"count_ms_per_rank": count_ms, "export_ms_per_rank": gather_ms, "merge_of_partition_range": merge,
                                    "send_bytes_rank_to_rank": send_bytes, "bytes_sent_per_rank": [int(sum(row) - row[i]) for i, row in enumerate(send_bytes)],
                                    "step_ms_compute_only": round(max(c + e for c, e in zip(count_ms, gather_ms)) + worst_merge, 2)}
    print("sample shards G=%d: count %s export %s merge %s" % (G, count_ms, gather_ms, {g: (v["import_ms"], v["merge_ms"]) for g, v in merge.items()}), flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print("written", out_path)
