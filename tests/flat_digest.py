"""Helper for tests that need a fresh process (environment knobs are read once per process): count + merge a fixed synthetic
workload on the GPU and print the sha1 of the flat statistics.  usage: flat_digest.py [shard_index shard_count]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simka_amd
from simka_amd import synth

n, R, L, k = 4, 4000, 100, 21
g = synth.genome_len_for(R, L)
pool, gw = synth.genome_pool_cpu(g)
si, sc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True, log2_partitions=8, shard_index=si, shard_count=sc)
for s in range(n):
    ids, cdf = synth.sample_profile(s)
    pk = synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s))
    if s == 1:
        pk[: len(pk) // 3] = 0                       # a third of sample 1 is poly-A: level-1 overflow -> exact redo of a pass
    ctx.count_sample(s, np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)
ctx.merge()
flat = ctx.stats().flat
print("DIGEST", hashlib.sha1(flat.tobytes()).hexdigest(), int(flat[0]), int(flat[1]))
