"""after another GPU process: which part of SimkaContext.stats() is slow?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, simka_amd
from simka_amd import synth, api
n, R, L = 12, 20000, 150
g = synth.genome_len_for(R, L); pool, gw = synth.genome_pool_cpu(g)
packed = []
for s in range(n):
    ids, cdf = synth.sample_profile(s)
    packed.append(np.concatenate([synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s)), np.zeros(2, dtype=np.uint64)]))
ctx = simka_amd.SimkaContext(n, kmer_size=31, abundance_min=2, simple_dist=True)
for it in range(4):
    ctx.reset()
    for s in range(n):
        ctx.count_sample(s, packed[s], R * L, R, fixed_len=L)
    t0 = time.perf_counter(); ctx.merge(); t1 = time.perf_counter()
    nn = ctx.lib.simka_stats_nb_u64(ctx.nb_samples, ctx.dist_flags)
    flat = np.zeros(nn, dtype=np.uint64); t2 = time.perf_counter()
    ctx._check(ctx.lib.simka_stats_download(ctx.h, flat.ctypes.data, nn, None)); t3 = time.perf_counter()
    st = api.Stats(ctx.nb_samples, ctx.dist_flags, flat); t4 = time.perf_counter()
    ctx._check(ctx.lib.simka_stats_download(ctx.h, flat.ctypes.data, nn, None)); t5 = time.perf_counter()
    print("merge %.2f  alloc %.2f  download %.2f  Stats() %.2f  download again (same buffer) %.2f   [%d words]" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, nn))
