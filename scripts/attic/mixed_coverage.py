"""Per-sample count time when the run mixes high-coverage samples (few distinct k-mers per partition: 2048-slot tables are chosen
from the first sample) with a low-coverage one (every k-mer distinct: tables over-fill -> redo list).  usage: mixed_coverage.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench
wl = dict(bench.WORKLOADS["c2"]); wl["n"] = 2
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
R, L, k = wl["reads"], wl["L"], wl["k"]
rnd = torch.randint(-2**62, 2**62, (reads[0].numel(),), dtype=torch.int64, device=dev)     # random bases: every k-mer distinct
for order in ("high,low", "low,high"):
    ctx = simka_amd.SimkaContext(2, kmer_size=k, abundance_min=1, max_kmers_per_sample=R * (L - k + 1))
    seq = [reads[0], rnd] if order == "high,low" else [rnd, reads[0]]
    for it in range(2):
        ctx.reset()
        ts = []
        for i, t in enumerate(seq):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.count_sample(i, t.data_ptr(), R * L, R, fixed_len=L, on_device=True)
            ctx.sample_totals(i); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(order, "-> count_sample ms:", ["%.2f" % x for x in ts])
    ctx.close()
