"""1-rank nccl smoke of the sharded protocol used on N GPUs: totals all-reduce -> merge -> head all-reduce (complex-dist)."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simka_amd
from simka_amd import dist as sdist, synth
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
n, R, L, k = 4, 5000, 100, 21
g = synth.genome_len_for(R, L)
pool, gw = synth.genome_pool_cpu(g)
ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True, device=local, shard_index=rank, shard_count=world)
for s in range(n):
    ids, cdf = synth.sample_profile(s)
    pk = synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s))
    ctx.count_sample(s, np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)
sdist.allreduce_totals_device(ctx)
ctx.merge()
sdist.allreduce_stats_device(ctx, totals_already_reduced=True)
st = ctx.stats()
if rank == 0:
    print("dist smoke ok", world, int(st.flat[0]), int(st.flat[1]), float(st.matrices()["mat_abundance_jensenshannon"][0, 1]))
dist.barrier(); dist.destroy_process_group()
