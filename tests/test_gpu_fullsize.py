"""-m gpu: BASELINE.json configs[1] at FULL size (10 samples x 1M x 100 bp, k=21) through size-independent properties --
the oracle needs minutes there, so parity is checked by invariants the domain offers:
  * conservation: with -abundance-min 1 every occurrence is counted: N_i = K_occ_i = R*(L-k+1), Q_i >= N_i, D_i = D_all_i
  * idempotence: a duplicated sample is at distance 0 (a = D, bc = N, S_ij = S_ji = N) and indistinguishable from its twin
  * shard additivity: the two halves of the partition space (shard 0/2, 1/2) sum to the unsharded accumulators, bit for bit
    (the reference checks the same by varying its job/partition counts, tests/simple_test.py:125-133)
  * geometry invariance: another partition count gives identical accumulators
  * symmetry / ranges of the final matrices
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, R, L, K = 10, 1_000_000, 100, 21


@pytest.fixture(scope="module")
def device_reads(gpu_required):
    torch = gpu_required
    import simka_amd
    from simka_amd import synth
    lib = simka_amd.load_library()
    dev = torch.device("cuda", 0)
    g = synth.genome_len_for(R, L)
    gw = (g + 31) // 32
    pool = torch.empty(synth.NB_GENOMES * gw, dtype=torch.int64, device=dev)
    assert lib.simka_synth_genomes(None, pool.data_ptr(), synth.NB_GENOMES, gw, synth.POOL_SEED) == 0
    reads = []
    for s in range(N - 1):
        ids, cdf = synth.sample_profile(s)
        d_ids = torch.from_numpy(ids.astype(np.int32)).to(dev)
        d_cdf = torch.from_numpy(cdf.view(np.int32)).to(dev)
        t = torch.zeros((R * L + 31) // 32 + 2, dtype=torch.int64, device=dev)
        assert lib.simka_synth_reads(None, t.data_ptr(), R, L, pool.data_ptr(), gw, g, d_ids.data_ptr(), d_cdf.data_ptr(), synth.NB_SEL,
                                     synth.sample_seed(s), synth.ERR_THRESHOLD16) == 0
        torch.cuda.synchronize()
        reads.append(t)
    reads.append(reads[0].clone())          # sample 9 is a copy of sample 0
    # the device generator equals the numpy one (spot check of the first words of sample 3)
    ids, cdf = synth.sample_profile(3)
    pool_cpu, gw_cpu = synth.genome_pool_cpu(g)
    assert gw_cpu == gw
    ref = synth.reads_cpu(2000, L, pool_cpu, gw, g, ids, cdf, synth.sample_seed(3))
    got = reads[3][: len(ref) - 1].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, ref[: len(ref) - 1])
    return reads


def _run(reads, amin, **kw):
    import simka_amd
    ctx = simka_amd.SimkaContext(N, kmer_size=K, abundance_min=amin, simple_dist=True, complex_dist=True,
                                 max_kmers_per_sample=R * (L - K + 1), **kw)
    for s in range(N):
        ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
    if kw.get("shard_count", 1) > 1 and "totals" in kw:
        pass
    return ctx


def test_conservation_and_duplicate_sample(device_reads):
    ctx = _run(device_reads, 1)
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    ps, pr = st.per_sample(), st.pairs()
    occ = R * (L - K + 1)
    assert np.all(ps["K_occ"] == occ) and np.all(ps["N"] == occ)          # every occurrence counted exactly once
    assert np.array_equal(ps["D"], ps["D_all"]) and np.all(ps["Q"] >= ps["N"])
    assert np.all(ps["D"] > 0.15 * occ) and np.all(ps["D"] < 0.5 * occ)   # 20x coverage + 1 % errors
    cell = N - 2                                                           # pair (0, 9) in i<j order
    assert pr["a"][cell] == ps["D"][0] and pr["bc"][cell] == ps["N"][0]
    assert pr["S_ij"][cell] == ps["N"][0] and pr["S_ji"][cell] == ps["N"][0]
    assert pr["chord"][cell] == ps["Q"][0] and pr["canb"][cell] == 0
    m = st.matrices()
    for name in ("mat_abundance_braycurtis", "mat_presenceAbsence_jaccard", "mat_abundance_chord", "mat_abundance_hellinger"):
        assert abs(float(m[name][0, 9])) < 1e-6, name
        assert np.allclose(m[name][0, 1:9], m[name][9, 1:9], atol=0)      # the twin sees everyone else identically
    for name, mat in m.items():
        assert np.all(np.isfinite(mat)) and np.all(np.diag(mat) == 0), name
        if "asym" not in name:
            assert np.array_equal(mat, mat.T), name
    assert np.all(m["mat_abundance_braycurtis"] <= 1.0) and np.all(m["mat_presenceAbsence_jaccard"] <= 1.0)
    # identities between accumulators: a <= min(D), bc <= S_ij, S_ij <= N_i
    iu = np.triu_indices(N, 1)
    D, Nn = ps["D"].astype(np.int64), ps["N"].astype(np.int64)
    assert np.all(pr["a"].astype(np.int64) <= np.minimum(D[iu[0]], D[iu[1]]))
    assert np.all(pr["bc"] <= pr["S_ij"]) and np.all(pr["bc"] <= pr["S_ji"])
    assert np.all(pr["S_ij"].astype(np.int64) <= Nn[iu[0]]) and np.all(pr["S_ji"].astype(np.int64) <= Nn[iu[1]])


def test_shard_additivity_and_geometry_invariance(device_reads):
    import simka_amd
    full = _run(device_reads, 2)
    full.merge()
    ref = full.stats()
    full.close()
    lay = ref.layout
    # two shards on the same GPU, totals made global before the merge (complex-dist protocol), heads summed after
    shards = [_run(device_reads, 2, shard_index=i, shard_count=2) for i in range(2)]
    tot = sum(c.totals_download().astype(np.uint64) for c in shards)
    for c in shards:
        c.totals_upload(tot)
        c.merge()
    flats = [c.stats().flat for c in shards]
    for c in shards:
        c.close()
    head = lay["head"]
    summed = flats[0].copy()
    summed[:head] = flats[0][:head] + flats[1][:head]
    st = simka_amd.Stats(N, 3, summed[: lay["derived"]])
    P = lay["nb_pairs"]
    klo = lay["acc0"] + 7 * P
    assert np.array_equal(st.flat[:klo], ref.flat[:klo])                   # every integer accumulator, bit for bit
    assert np.array_equal(st.flat[klo + P: lay["derived"]], ref.flat[klo + P: lay["derived"]])
    assert np.max(np.abs(st.flat[klo:klo + P].view(np.int64) - ref.flat[klo:klo + P].view(np.int64))) <= 64   # KL fixed point: rounding of 2 sums
    for name, m in ref.matrices().items():
        np.testing.assert_allclose(st.matrices()[name], m, rtol=1e-6, atol=1e-7)
    # another partition geometry
    other = _run(device_reads, 2, log2_partitions=13)
    other.merge()
    st2 = other.stats()
    other.close()
    assert np.array_equal(st2.flat[:klo], ref.flat[:klo])
    assert np.array_equal(st2.flat[klo + P: lay["derived"]], ref.flat[klo + P: lay["derived"]])


def _oracle_of(oracle_mod, reads, n, R_, L_):
    """the CPU oracle over the same reads, downloaded from the device generator (all host cores: sort-count per sample is
    parallel over the oracle's partitions)"""
    from simka_amd import synth
    import os
    orc = oracle_mod.Oracle()
    offs = np.arange(R_ + 1, dtype=np.uint64) * L_
    for s in range(n):
        pk = reads[s].cpu().numpy().view(np.uint64)
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk[: (R_ * L_ + 31) // 32], R_ * L_), offs)
    return orc, min(os.cpu_count() or 1, 64)


def _flat_equal_to_oracle(st, orc, n, simple, complex_):
    """every per-sample total and every integer accumulator bit for bit, KL to 1e-9, the matrices to 1e-6 relative"""
    ot, ps, pr = orc.totals(), st.per_sample(), st.pairs()
    for key in ("K_occ", "D_all", "D", "N", "Q"):
        assert np.array_equal(ps[key].astype(np.uint64), ot[key]), key
    iu = np.triu_indices(n, 1)
    S = orc.acc("S")
    assert np.array_equal(pr["S_ij"], S[iu]) and np.array_equal(pr["S_ji"], S.T[iu])
    assert np.array_equal(pr["a"], orc.acc("a")[iu]) and np.array_equal(pr["bc"], orc.acc("bc")[iu])
    if simple:
        assert np.array_equal(pr["chord"], orc.acc("chord")[iu]) and np.array_equal(pr["hell"], orc.acc("hell")[iu])
    if complex_:
        assert np.array_equal(pr["whit"], orc.acc("whit")[iu]) and np.array_equal(pr["canb"], orc.acc("canb")[iu])
        np.testing.assert_allclose(pr["kl"], orc.kl()[iu], rtol=1e-9, atol=1e-15)
    assert (int(st.view.nb_distinct_kmers), int(st.view.nb_shared_kmers)) == orc.global_counts()
    m = st.matrices()
    for w, name in enumerate(orc.matrix_names()):
        if name in m:
            np.testing.assert_allclose(m[name], orc.matrix(w), rtol=1e-6, atol=0, err_msg=name)


def test_c2_full_size_bit_exact_vs_oracle(device_reads, oracle_mod):
    """BASELINE configs[1] AT FULL SIZE (10 x 1M x 100 bp, k = 21, abundance-min 2; all three distance families, which contain the
    configuration's Bray-Curtis + Jaccard): 8e8 k-mer occurrences through the HIP path against the CPU oracle -- whole-path equality
    in the spirit of the reference's own test (ref: tests/simple_test.py:29-68)."""
    ctx = _run(device_reads, 2)
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    orc, threads = _oracle_of(oracle_mod, device_reads, N, R, L)
    orc.run(K, 2, simple=True, complex_=True, nparts=4 * threads, threads=threads)
    _flat_equal_to_oracle(st, orc, N, True, True)
    orc.close()


class _OracleResult:
    """What the parity checks need from an oracle run, kept after the oracle (and its copy of the reads) is gone."""

    def __init__(self, orc, simple, complex_):
        self.n = orc.n
        self.simple, self.complex_ = simple, complex_
        self.tot = {k: v.copy() for k, v in orc.totals().items()}
        self.accs = {name: orc.acc(name) for name in (["S", "a", "bc"] + (["chord", "hell"] if simple else []) + (["whit", "canb"] if complex_ else []))}
        self.kl_ = orc.kl() if complex_ else None
        self.glob = orc.global_counts()
        self.names = orc.matrix_names()
        self.mats = [orc.matrix(w) for w in range(len(self.names))]

    def totals(self):
        return self.tot

    def acc(self, name):
        return self.accs[name]

    def kl(self):
        return self.kl_

    def global_counts(self):
        return self.glob

    def matrix_names(self):
        return self.names

    def matrix(self, w):
        return self.mats[w]


def _device_workload(torch, name, **over):
    import sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    import simka_amd
    wl = dict(bench.WORKLOADS[name], **over)
    _, reads = bench.gen_device_samples(simka_amd.load_library(), torch, wl, torch.device("cuda", 0))
    return wl, reads


def _hip_stats(reads, wl, order=None, **kw):
    import simka_amd
    n, R_, L_, k_ = wl["n"], wl["reads"], wl["L"], wl["k"]
    order = list(range(n)) if order is None else order
    with simka_amd.SimkaContext(len(order), kmer_size=k_, abundance_min=wl["amin"], simple_dist=wl["simple"], complex_dist=bool(wl.get("complex")),
                                max_kmers_per_sample=R_ * (L_ - k_ + 1), **kw) as ctx:
        for slot, s in enumerate(order):
            ctx.count_sample(slot, reads[s].data_ptr(), R_ * L_, R_, fixed_len=L_, on_device=True)
        ctx.merge()
        return ctx.stats()


def _oracle_result(oracle_mod, reads, wl):
    orc, threads = _oracle_of(oracle_mod, reads, wl["n"], wl["reads"], wl["L"])
    orc.run(wl["k"], wl["amin"], simple=wl["simple"], complex_=bool(wl.get("complex")), nparts=4 * threads, threads=threads)
    res = _OracleResult(orc, wl["simple"], bool(wl.get("complex")))
    orc.close()
    return res


@pytest.fixture(scope="module")
def c3_100k(gpu_required, oracle_mod):
    """BASELINE configs[2] / configs[3]'s shape (100 samples, 150 bp, k = 31, -simple-dist, abundance-min 2) at 100 000 reads per sample:
    the reads on the device + ONE oracle run that the single-context test and the two 8-rank decompositions of configs[3] share."""
    wl, reads = _device_workload(gpu_required, "c3", reads=100_000)
    return wl, reads, _oracle_result(oracle_mod, reads, wl)


def test_c3_shape_at_100k_reads_bit_exact_vs_oracle(c3_100k):
    """BASELINE configs[2]'s shape at 100 000 reads per sample -- 1.2e9 k-mer occurrences, a hundredth of the full depth and 20x the
    depth of test_baseline_config_shapes_vs_oracle -- bit-exact vs the oracle.  The partition geometry is the one the full-depth run
    uses per k-mer occurrence (sized from the sample)."""
    wl, reads, ref = c3_100k
    _flat_equal_to_oracle(_hip_stats(reads, wl), ref, wl["n"], True, False)


def test_c4_eight_partition_shards_sum_to_the_oracle(c3_100k):
    """BASELINE configs[3] (C3's job with the minimizer space sharded over 8 GPUs + ONE all-reduce of the N x N partials,
    ref: src/SimkaPotara.hpp:974-1124 one merge job per partition, src/core/SimkaDistance.cpp:156-213 operator+=) emulated on one
    GPU: eight contexts with shard_index g of 8 scan every read and keep the partitions p % 8 == g; the SUM of their flat statistics
    -- what the all-reduce computes -- is compared with the ORACLE, not with the single-context run."""
    import simka_amd
    wl, reads, ref = c3_100k
    n = wl["n"]
    total = None
    for g in range(8):
        st = _hip_stats(reads, wl, shard_index=g, shard_count=8, solid_capacity=60_000_000)
        f = st.flat.copy()
        lay = st.layout
        total = f if total is None else total + f          # accumulators AND the per-sample totals rows (partial per shard) add up
    # D_all / K_occ rows: every shard counts its own share as well
    summed = simka_amd.Stats(n, st.dist_flags, total[: lay["derived"]])
    _flat_equal_to_oracle(summed, ref, n, True, False)


def test_c4_eight_sample_shards_exchange_and_merge_to_the_oracle(c3_100k):
    """BASELINE configs[3] the other way round (simka_amd/dist.py::count_exchange_merge: one count job per sample, one merge job per
    partition RANGE -- the reference's own job structure, ref: src/SimkaPotara.hpp:813-1124): rank r of 8 counts the samples
    s % 8 == r, the solid spectra are routed to the rank that owns their partition range (the all-to-all done by hand), every rank
    imports + merges its range, and the summed heads are compared with the ORACLE."""
    import torch
    import simka_amd
    from simka_amd import dist as sdist
    wl, reads, ref = c3_100k
    n, R_, L_, k_ = wl["n"], wl["reads"], wl["L"], wl["k"]
    world = 8
    dev = torch.device("cuda:0")
    kw = dict(kmer_size=k_, abundance_min=wl["amin"], simple_dist=True, complex_dist=False, max_kmers_per_sample=R_ * (L_ - k_ + 1), solid_capacity=120_000_000)
    sends, nparts = [], None
    for r in range(world):
        with simka_amd.SimkaContext(n, **kw) as c:
            mine = sdist.samples_of(r, world, n)
            for s in mine:
                c.count_sample(s, reads[s].data_ptr(), R_ * L_, R_, fixed_len=L_, on_device=True)
            local = {s: c.export_sample_device(s, dev) for s in mine}
        nparts = len(next(iter(local.values()))[1])
        sends.append(local)
    packs = [sdist.pack_spectra(sends[r], nparts, world, n, r, dev) for r in range(world)]
    tot_all = np.stack([p[1] for p in packs])
    lay = simka_amd.api.stats_layout(n, 1)
    head = lay["head"]
    total = np.zeros(head, dtype=np.uint64)
    last = None
    for g in range(world):
        meta_recv = np.stack([packs[r][0][g] for r in range(world)])
        kr, cr = [], []
        for r in range(world):
            splits = packs[r][4]
            lo = sum(splits[:g])
            kr.append(packs[r][2][lo: lo + splits[g]]); cr.append(packs[r][3][lo: lo + splits[g]])
        incoming = sdist.unpack_spectra(meta_recv, tot_all, torch.cat(kr), torch.cat(cr), nparts, world, n, g)
        with simka_amd.SimkaContext(n, **kw) as c:
            for s, t, pc, kk, cc in incoming:
                c.import_sample_device(s, t, pc, kk, cc)
            c.merge()
            last = c.stats()
        total += last.flat[:head]
    flat = last.flat.copy()
    flat[:head] = total
    _flat_equal_to_oracle(simka_amd.Stats(n, last.dist_flags, flat[: lay["derived"]]), ref, n, True, False)


def test_c3_at_a_tenth_of_its_depth_bit_exact_vs_oracle(gpu_required, oracle_mod):
    """bench.py's `c3_10` (BASELINE configs[2] at 1M reads per sample: 1.2e10 k-mer occurrences, 2^16 x 8 partitions per sample as a
    full-depth run has per occurrence) bit-exact vs the oracle -- the deepest oracle comparison of the suite (minutes of CPU)."""
    wl, reads = _device_workload(gpu_required, "c3_10")
    st = _hip_stats(reads, wl)
    ref = _oracle_result(oracle_mod, reads, wl)
    del reads
    _flat_equal_to_oracle(st, ref, wl["n"], True, False)


def test_c5_shape_at_10k_reads_bit_exact_vs_oracle(gpu_required, oracle_mod):
    """BASELINE configs[4]'s shape (500 samples, 150 bp, k = 31, -simple-dist -complex-dist: the tiled pair accumulator and the
    closed forms of updateDistanceComplex, ref: src/core/SimkaAlgorithm.hpp:404-516) at 10 000 reads per sample -- 25x the depth of
    test_baseline_config_shapes_vs_oracle[c5_shape]; the oracle's literal O(N^2) complex update is what bounds the depth."""
    wl, reads = _device_workload(gpu_required, "c5_5", reads=10_000)
    st = _hip_stats(reads, wl)
    ref = _oracle_result(oracle_mod, reads, wl)
    del reads
    _flat_equal_to_oracle(st, ref, wl["n"], True, True)
