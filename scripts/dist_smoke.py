"""1-rank RCCL smoke of the N-GPU protocols on ONE GPU: (a) partition shards: totals all-reduce -> merge -> head all-reduce;
(b) sample shards: the whole exchange path (all_to_all_single / all_gather on the nccl backend, rank 0 sending to itself,
SIMKA_FORCE_EXCHANGE=1) -- both must reproduce the plain single-context statistics bit for bit."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simka_amd
from simka_amd import dist as sdist, synth
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
n, R, L, k = 4, 5000, 100, 21
g = synth.genome_len_for(R, L)
pool, gw = synth.genome_pool_cpu(g)
packed = []
for s in range(n):
    ids, cdf = synth.sample_profile(s)
    packed.append(np.concatenate([synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s)), np.zeros(2, dtype=np.uint64)]))
# sample 1: one read 3000 times -- counts beyond the -complex-dist histogram (SIMKA_HIST_MAX = 1024) travel on the list of large counts,
# filled by the count kernels in (a) and by k_import_hist from the imported spectra in (b)
hot = synth.unpack_ascii(packed[1][:-2], R * L).reshape(R, L).copy()
hot[100:3100] = hot[7]
hot_packed, hot_off, hot_nb, _ = simka_amd.pack_reads([row.tobytes() for row in hot])


def count_into(c, s):
    if s == 1:
        c.count_sample(s, hot_packed, hot_nb, R, offsets=hot_off, nb_input_reads=R)
    else:
        c.count_sample(s, packed[s], R * L, R, fixed_len=L)


kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True, device=local)
ctx = simka_amd.SimkaContext(n, shard_index=rank, shard_count=world, **kw)
for s in range(n):
    count_into(ctx, s)
sdist.allreduce_totals_device(ctx)
ctx.merge()
sdist.allreduce_stats_device(ctx, totals_already_reduced=True)
a = ctx.stats().flat.copy()
ctx.close()
os.environ["SIMKA_FORCE_EXCHANGE"] = "1"
ctx = simka_amd.SimkaContext(n, max_kmers_per_sample=R * (L - k + 1), **kw)
sdist.count_exchange_merge(ctx, lambda s: count_into(ctx, s), n, dev)
b = ctx.stats().flat.copy()
ctx.close()
# (c) the same exchange for k = 33 (two-word keys: high and low words in separate all-to-alls) against the plain wide-k run
kw33 = dict(kw, kmer_size=33)
os.environ.pop("SIMKA_FORCE_EXCHANGE")
ctx = simka_amd.SimkaContext(n, **kw33)
for s in range(n):
    count_into(ctx, s)
ctx.merge()
c = ctx.stats().flat.copy()
ctx.close()
os.environ["SIMKA_FORCE_EXCHANGE"] = "1"
ctx = simka_amd.SimkaContext(n, **kw33)
sdist.count_exchange_merge(ctx, lambda s: count_into(ctx, s), n, dev)
d = ctx.stats().flat.copy()
ctx.close()
if rank == 0:
    good = world > 1 or (np.array_equal(a, b) and np.array_equal(c, d))
    print("dist smoke", "ok" if good else "MISMATCH", world, int(a[0]), int(b[0]), int(a[1]), int(b[1]), int(c[0]), int(d[0]))
dist.barrier(); dist.destroy_process_group()
