"""Time ONE rank's share of the partition-sharded job on a single GPU: shard 0 of G for G = 1,2,4,8 (C2 shape).
Shows how much of the step is the replicated scan.  usage: shard_step_time.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simka_amd
import bench

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
lib = simka_amd.load_library()
dev = torch.device("cuda:0")
pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
for G in (1, 2, 4, 8):
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=wl["amin"], simple_dist=wl["simple"], shard_index=0, shard_count=G,
                                 max_kmers_per_sample=R * (L - k + 1))
    def step():
        ctx.reset()
        for s in range(n):
            ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
        ctx.merge()
        ctx.stats()
    step(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms = (time.time() - t) / 3 * 1e3
    print("G=%d shard 0: %.2f ms/step" % (G, ms))
    ctx.close()
