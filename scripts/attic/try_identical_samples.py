"""N identical samples: every k-mer is shared by all N (group size N).  usage: try_identical_samples.py N"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simka_amd
from simka_amd import synth
N = int(sys.argv[1]); R = 50; L = 100; k = 21
g = synth.genome_len_for(R * 4, L)
pool, gw = synth.genome_pool_cpu(g)
ids, cdf = synth.sample_profile(0)
pk = np.concatenate([synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(0)), np.zeros(2, dtype=np.uint64)])
ctx = simka_amd.SimkaContext(N, kmer_size=k, abundance_min=1, simple_dist=True)
for i in range(N):
    ctx.count_sample(i, pk, R * L, R, fixed_len=L)
t0 = ctx.sample_totals(0)
ctx.merge()
pr = ctx.stats().pairs()
ok = (pr["a"] == t0["D"]).all() and (pr["S_ij"] == t0["N"]).all() and (pr["bc"] == t0["N"]).all()
print("N", N, "D", t0["D"], "pairs", len(pr["a"]), "all pairs see every k-mer:", bool(ok))
