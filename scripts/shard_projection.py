#!/usr/bin/env python3
"""Compute time per RANK of the two multi-GPU decompositions, measured on ONE GPU for G = 1, 2, 4, 8 ranks (collectives unmeasured).

    python scripts/shard_projection.py [workload=c3] [out.json]

(a) partition shards (north_star's split: rank g keeps the minimizer partitions p % G == g, every rank scans every read, ONE all-reduce
    of the N x N partials, ref: src/SimkaPotara.hpp:974-1124 one merge job per partition): the whole step of shard g of G, for EVERY g --
    the slowest rank bounds the step; its kernel table shows the replicated scan.
(b) sample shards (one count job per sample, one merge job per partition range -- the reference's own job structure, ref:
    src/SimkaPotara.hpp:813-1124): rank r counts the samples s % G == r (timed), gathers their spectra into ONE destination-major send buffer (timed: dist.pack_batch, the entry point bench.py uses; the bytes
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
    # ---- (b) sample shards, through the BATCH entry points bench.py uses (simka_amd/dist.py::pack_batch / import_batch)
    count_ms, gather_ms, packs, P, kw_ = [], [], [], None, 1
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
            info = ctx.spectrum_info(mine[0])
            P, kw_ = info[1], info[2]
            pk = sdist.pack_batch(ctx, mine, P, kw_, G, n, dev); ctx.sync(); sync(); del pk       # (first pass: the allocator's first touch of the send buffers)
            t0 = time.perf_counter()
            meta, tot_send, ks, ks2, cs, splits = sdist.pack_batch(ctx, mine, P, kw_, G, n, dev)
            ctx.sync(); sync(); gather_ms.append(round((time.perf_counter() - t0) * 1e3, 2))
            packs.append((meta.clone() if isinstance(meta, torch.Tensor) else meta.copy(), tot_send.copy(), ks, ks2, cs, list(splits)))
    send_bytes = [[int(x) * (12 if kw_ == 1 else 20) for x in pk[5]] for pk in packs]
    tot_all = np.stack([pk[1] for pk in packs])
    merge = {}
    for g in sorted({0, G // 2, G - 1}):
        # the all-to-all, by slicing: rank g receives block g of every rank's send buffer
        meta_recv = torch.stack([packs[r][0][g] for r in range(G)]) if isinstance(packs[0][0], torch.Tensor) else np.stack([packs[r][0][g] for r in range(G)])
        kr, cr, kr2 = [], [], []
        for r in range(G):
            sp = packs[r][5]; lo_ = sum(sp[:g])
            kr.append(packs[r][2][lo_: lo_ + sp[g]]); cr.append(packs[r][4][lo_: lo_ + sp[g]])
            if kw_ == 2:
                kr2.append(packs[r][3][lo_: lo_ + sp[g]])
        kr, cr = torch.cat(kr), torch.cat(cr)
        kr2 = torch.cat(kr2) if kw_ == 2 else None
        with simka_amd.SimkaContext(n, **kw) as ctx:
            def imp():
                sdist.import_batch(ctx, g, G, n, P, kw_, meta_recv, tot_all, kr, kr2, cr, dev)
                ctx.sync()
            imp(); ctx.merge(); ctx.stats(); sync()           # (first pass: arena chunks mapped, merge buffers allocated)
            t0 = time.perf_counter(); imp(); sync(); t_imp = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter(); ctx.merge(); ctx.stats(); sync(); t_mrg = (time.perf_counter() - t0) * 1e3
            imp(); ctx.profile_enable(True); ctx.profile_reset(); ctx.merge(); ctx.stats(); sync()
            merge[str(g)] = {"import_ms": round(t_imp, 2), "merge_ms": round(t_mrg, 2), "kernels_ms": kernel_ms(ctx)}
        del kr, cr, kr2
    del packs
    torch.cuda.empty_cache()
    worst_merge = max(v["import_ms"] + v["merge_ms"] for v in merge.values())
    res["sample_shards"][str(G)] = {"count_ms_per_rank": count_ms, "pack_ms_per_rank": gather_ms, "merge_of_partition_range": merge,
                                    "send_bytes_rank_to_rank": send_bytes, "bytes_sent_per_rank": [int(sum(row) - row[i]) for i, row in enumerate(send_bytes)],
                                    "step_ms_compute_only": round(max(c + e for c, e in zip(count_ms, gather_ms)) + worst_merge, 2)}
    print("sample shards G=%d: count %s pack %s merge %s" % (G, count_ms, gather_ms, {g: (v["import_ms"], v["merge_ms"]) for g, v in merge.items()}), flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print("written", out_path)
