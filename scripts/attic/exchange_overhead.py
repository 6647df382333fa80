"""Host-side cost of the spectrum exchange without the collectives (1 GPU): info + gather of the local samples, import of a block.
C2 shape, emulating rank 0 of `world`.  usage: exchange_overhead.py [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench
from simka_amd import dist as sdist
from simka_amd.api import SampleTotals

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wl = dict(bench.WORKLOADS["c2"])
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=2, max_kmers_per_sample=R * (L - k + 1))
def T(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    ctx.reset()
    mine = sdist.samples_of(0, world, n)
    t0 = T()
    for s in mine: ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
    ctx.sync(); t1 = T()
    P = ctx.nb_partitions()
    pc, tot = ctx.samples_spectrum_info(mine)
    t2 = T()
    off = sdist._excl_cumsum_rows(pc.reshape(1, -1)).reshape(pc.shape).astype(np.uint64)      # sample-major is fine for timing
    total = int(pc.astype(np.int64).sum())
    ks = torch.empty(total, dtype=torch.int64, device=dev); cs = torch.empty(total, dtype=torch.int32, device=dev)
    ctx.gather_samples_device(mine, off, ks, cs)
    t3 = T()
    ctx.reset()
    t4 = T()
    tin = (SampleTotals * len(mine))(*[tot[j] for j in range(len(mine))])
    ctx.import_samples_device(np.arange(len(mine)), tin, 0, pc, off, P, ks, cs)      # as samples 0..len-1 of a fresh run
    t5 = T()
    print("world %d: count %d samples %.2f ms | info %.2f | offsets+gather %.2f | reset %.2f | import %.2f ms (%d records)" %
          (world, len(mine), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, total))
