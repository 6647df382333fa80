"""N>1 path on CPU: world_size-2 gloo.  Each rank owns the partitions p % 2 == rank (as GPU g owns its
minimizer-partition shard), holds that shard's accumulators in the product's flat u64 layout, and ONE
all-reduce(sum) (SimkaStatistics::operator+=, ref: src/core/SimkaDistance.cpp:156-213) must give the
single-process result -- checked against the golden CSVs through the product's host finalisation."""
import glob
import gzip
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib
    import simka_amd
    from simka_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_lib.Oracle()
    o.load_input(os.path.join(ROOT, "tests", "golden", "example", "simka_input.txt"))
    nparts = 8
    o.run(31, 2, simple=True, complex_=True, nparts=nparts, shard_index=rank, shard_count=world)
    flat = o.flat_stats(simple=True, complex_=True, nparts=nparts, shard_index=rank, shard_count=world)
    total = sdist.allreduce_stats_host(flat)          # the product's reduction helper (int64 view, SUM)
    if rank == 0:
        st = simka_amd.Stats(o.n, simka_amd.DIST_SIMPLE | simka_amd.DIST_COMPLEX, total)
        st.write_matrices(outdir, o.ids(), gz=True)
        np.save(os.path.join(outdir, "flat.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_allreduce_matches_goldens(oracle_mod, golden_dir, tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dist")
    os.makedirs(out)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20
    # and the reduced buffer equals the single-process accumulators bit for bit (KL fixed point: within 2 ulp of 2^-52 per shard)
    o = oracle_mod.Oracle()
    o.load_input(os.path.join(golden_dir, "example", "simka_input.txt"))
    o.run(31, 2, simple=True, complex_=True)
    ref = o.flat_stats(simple=True, complex_=True)
    got = np.load(os.path.join(out, "flat.npy"))
    import simka_amd
    lay = simka_amd.api.stats_layout(o.n, 3)
    P = lay["nb_pairs"]
    klo = lay["acc0"] + 7 * P
    assert np.array_equal(got[:klo], ref[:klo]) and np.array_equal(got[klo + P:lay["derived"]], ref[klo + P:lay["derived"]])
    assert np.max(np.abs(got[klo:klo + P].view(np.int64) - ref[klo:klo + P].view(np.int64))) <= 4


def test_shard_plan():
    from simka_amd import dist as sdist
    assert sdist.shard_of(rank=3, world=8) == (3, 8)
    with pytest.raises(ValueError):
        sdist.shard_of(rank=8, world=8)
