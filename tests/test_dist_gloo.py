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
    # and the reduced buffer equals the single-process accumulators bit for bit (KL fixed point 2^-60: the oracle hands its sum over as a double, i.e. to ~1e-17)
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
    assert np.max(np.abs(got[klo:klo + P].view(np.int64) - ref[klo:klo + P].view(np.int64))) <= 4096       # 3.5e-15 absolute


def test_shard_plan():
    from simka_amd import dist as sdist
    assert sdist.shard_of(rank=3, world=8) == (3, 8)
    with pytest.raises(ValueError):
        sdist.shard_of(rank=8, world=8)


# ---- sample shards + spectrum exchange (simka_amd/dist.py::exchange_spectra) on gloo ------------------------------------
_NPARTS = 16


def _partition_major(kmers, counts):
    part = ((kmers * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(60)).astype(np.int64)      # 16 partitions
    order = np.lexsort((kmers, part))
    return np.bincount(part, minlength=_NPARTS).astype(np.uint32), kmers[order], counts[order], part[order]


def _exchange_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib
    from simka_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_lib.Oracle()
    o.load_input(os.path.join(ROOT, "tests", "golden", "example", "simka_input.txt"))
    o.run(21, 2)
    n = o.n
    tot = o.totals()
    spectra = [_partition_major(*o.sample_solid(i)) for i in range(n)]
    tdict = lambda i: {key: int(tot[key][i]) for key in ("nb_reads", "D", "N", "Q", "K_occ", "D_all")}
    # this rank "counted" the samples s % world == rank
    local = {s: (tdict(s), spectra[s][0], torch.from_numpy(spectra[s][1].view(np.int64).copy()), torch.from_numpy(spectra[s][2].astype(np.int32)))
             for s in sdist.samples_of(rank, world, n)}
    incoming = sdist.exchange_spectra(local, _NPARTS, n, torch.device("cpu"))
    lo, hi = sdist.partition_bounds(_NPARTS, world)[rank: rank + 2]
    assert sorted(x[0] for x in incoming) == list(range(n))
    a = np.zeros((n, n), dtype=np.int64)
    sets = {}
    for s, t, pc, k, c in incoming:
        pcs, ks, cs, parts = spectra[s]
        sel = (parts >= lo) & (parts < hi)
        assert t == tdict(s)
        assert np.array_equal(pc[lo:hi], pcs[lo:hi]) and pc[:lo].sum() == 0 and pc[hi:].sum() == 0
        assert np.array_equal(k.numpy().view(np.uint64), ks[sel]) and np.array_equal(c.numpy().astype(np.uint32), cs[sel])
        sets[s] = k.numpy()
    for i in range(n):          # the rank-local "merge" of its partition range: distinct shared k-mers per pair
        for j in range(i + 1, n):
            a[i, j] = len(np.intersect1d(sets[i], sets[j], assume_unique=True))
    at = torch.from_numpy(a)
    dist.all_reduce(at)
    if rank == 0:
        np.save(os.path.join(outdir, "a.npy"), at.numpy())
        np.save(os.path.join(outdir, "a_ref.npy"), o.acc("a").astype(np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_spectrum_exchange_routes_partition_ranges(oracle_mod, tmp_path, world):
    """Samples counted on rank s % world, merged by partition range: after exchange_spectra every rank holds, for ALL samples,
    exactly the records of its partition range (and the global per-sample totals); the rank-local pair counts all-reduce to the
    oracle's matrix.  world 3: uneven partition ranges and sample counts (5 samples)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "xchg")
    os.makedirs(out)
    mp.spawn(_exchange_worker, args=(world, port, out), nprocs=world, join=True)
    a, ref = np.load(os.path.join(out, "a.npy")), np.load(os.path.join(out, "a_ref.npy"))
    iu = np.triu_indices(a.shape[0], 1)
    assert np.array_equal(a[iu], ref[iu])
