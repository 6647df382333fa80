"""Cost of hot k-mers (adapter-like repeated reads): one C2-size sample where a share of the reads are copies of a few reads.
Their k-mers overflow the capacity-sized level-2 regions -> spill buffer -> general kernel (or, past the spill capacity, the
exact redo of the sample).  usage: hot_kmers.py [k]      (k = 33..51: two-word k-mers, where hot k-mers only repeat inside a
wave's table; k >= 52: the bucket count)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench
wl = dict(bench.WORKLOADS["c2"]); wl["n"] = 2
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
R, L, k = wl["reads"], wl["L"], (int(sys.argv[1]) if len(sys.argv) > 1 else wl["k"])
wpr = None
base = reads[0].clone()
def with_hot(nhot, copies):
    """overwrite reads: `nhot` source reads, each copied `copies` times (whole 100-base reads = 25 bytes... 200 bits: not word aligned) -> work on bytes"""
    t = base.clone()
    b = t.view(torch.uint8)                      # 4 bases per byte, 25 bytes per 100-base read
    for h in range(nhot):
        src = b[h * 25:(h + 1) * 25].clone()
        lo = (1000 + h * copies) * 25
        b[lo: lo + copies * 25] = src.repeat(copies)
    return t
for nhot, copies in ((0, 0), (1, 20000), (10, 20000), (40, 5000), (200, 1000), (1000, 200)):
    t = with_hot(nhot, copies) if nhot else base
    ctx = simka_amd.SimkaContext(2, kmer_size=k, abundance_min=2, max_kmers_per_sample=R * (L - k + 1))
    for it in range(3):
        ctx.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.count_sample(0, t.data_ptr(), R * L, R, fixed_len=L, on_device=True)
        tot = ctx.sample_totals(0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%4d hot reads x %5d copies (%4.1f %% of the reads): count_sample %.2f ms, D_all %d, paths %s" % (nhot, copies, 100.0 * nhot * copies / R, dt * 1e3, tot["D_all"], ctx.count_paths()))
    ctx.close()
