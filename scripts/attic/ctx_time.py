"""how long a context takes to create / destroy (the driver pays this once per run): usage ctx_time.py"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, simka_amd
torch.cuda.init(); torch.cuda.synchronize()
for n, kocc in ((100, 120_000_000), (100, 1_200_000_000), (100, 1_200_000_000), (10, 120_000_000)):
    t = time.time()
    ctx = simka_amd.SimkaContext(n, kmer_size=31, abundance_min=2, simple_dist=True, max_kmers_per_sample=kocc)
    t1 = time.time()
    ctx.close()
    t2 = time.time()
    print("n=%d kocc=%.1e create %.2f s destroy %.2f s" % (n, kocc, t1 - t, t2 - t1), flush=True)
