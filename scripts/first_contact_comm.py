#!/usr/bin/env python3
"""One stage of scripts/first_contact.sh: a collective of the C ABI between the ranks of this job, checked against its known answer.

    python -m torch.distributed.run --nproc-per-node 2 ... scripts/first_contact_comm.py allreduce|alltoallv

Backend "nccl" (default): the RCCL communicator of the C ABI (simka_comm_*), one GPU per rank.  SIMKA_BENCH_BACKEND=gloo: the dry run of a
one-GPU box -- the ranks share GPU 0 and the same buffers travel through torch.distributed on gloo (launch, rendezvous, buffer layout and
checks are the ones of the real run; only the transport differs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

stage = sys.argv[1]
backend = os.environ.get("SIMKA_BENCH_BACKEND", "nccl")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
device = local if backend == "nccl" else 0
torch.cuda.set_device(device)
dev = torch.device("cuda", device)
dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=dev if backend == "nccl" else None)
import simka_amd
from simka_amd import api, dist as sdist

comm = None
if backend == "nccl":
    comm = sdist.create_comm(rank, world, device)
    if rank == 0:
        print("[first-contact] RCCL of the C ABI:", api.Comm.library(), flush=True)
stream = torch.cuda.current_stream().cuda_stream
if stage == "allreduce":
    n = 100_000
    t = (torch.arange(n, dtype=torch.int64, device=dev) + 1) * (rank + 1)
    if comm is not None:
        comm.allreduce_u64(t.data_ptr(), n, stream)
        torch.cuda.synchronize()
    else:
        h = t.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM); t.copy_(h)
    want = (torch.arange(n, dtype=torch.int64) + 1) * (world * (world + 1) // 2)
    assert torch.equal(t.cpu(), want), "all-reduce: wrong sum on rank %d" % rank
elif stage == "alltoallv":
    # rank r sends (r + 1) * (d + 1) * 1000 words to rank d, word j of that block = r * 1e9 + d * 1e6 + j
    send_counts = [(rank + 1) * (d + 1) * 1000 for d in range(world)]
    recv_counts = [(r + 1) * (rank + 1) * 1000 for r in range(world)]
    src = torch.cat([torch.arange(c, dtype=torch.int64) + rank * 10**9 + d * 10**6 for d, c in enumerate(send_counts)]).to(dev)
    dst = torch.zeros(sum(recv_counts), dtype=torch.int64, device=dev)
    if comm is not None:
        comm.alltoallv(src.data_ptr(), send_counts, dst.data_ptr(), recv_counts, 8, stream)
        torch.cuda.synchronize()
    else:
        hs, hd = src.cpu(), dst.cpu()
        dist.all_to_all_single(hd, hs, recv_counts, send_counts)
        dst.copy_(hd)
    want = torch.cat([torch.arange(c, dtype=torch.int64) + r * 10**9 + rank * 10**6 for r, c in enumerate(recv_counts)])
    assert torch.equal(dst.cpu(), want), "all-to-all: wrong blocks on rank %d" % rank
else:
    raise SystemExit("unknown stage " + stage)
dist.barrier()
if comm is not None:
    comm.close()
if rank == 0:
    print("[first-contact] %s ok on %d ranks (%s)" % (stage, world, backend), flush=True)
dist.destroy_process_group()
