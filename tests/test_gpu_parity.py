"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the reference's golden CSVs."""
import filecmp
import glob
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = [(21, 0), (21, 2), (31, 0), (31, 2)]


def _load_example(golden_dir):
    import simka_amd
    samples = simka_amd.parse_input_file(os.path.join(golden_dir, "example", "simka_input.txt"))
    packed = []
    for s in samples:
        seqs = []
        for f in s["files"]:
            seqs += list(simka_amd.read_sequences(f))
        packed.append(simka_amd.pack_reads(seqs))
    return samples, packed


def _run_gpu(packed, k, amin, simple=True, complex_=True, **kw):
    import simka_amd
    ctx = simka_amd.SimkaContext(len(packed), kmer_size=k, abundance_min=amin, simple_dist=simple, complex_dist=complex_, **kw)
    for i, (pk, off, nb, nin) in enumerate(packed):
        ctx.count_sample(i, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
    totals = [ctx.sample_totals(i) for i in range(len(packed))]
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    return totals, st


def _check_vs_oracle(totals, st, orc, simple=True, complex_=True):
    ot = orc.totals()
    for i, t in enumerate(totals):
        for key in ("K_occ", "D_all", "D", "N", "Q"):
            assert int(t[key]) == int(ot[key][i]), (i, key, t[key], ot[key][i])
    n = orc.n
    iu = np.triu_indices(n, 1)
    pr = st.pairs()
    S = orc.acc("S")
    assert np.array_equal(pr["S_ij"], S[iu])
    assert np.array_equal(pr["S_ji"], S.T[iu])
    assert np.array_equal(pr["a"], orc.acc("a")[iu])
    assert np.array_equal(pr["bc"], orc.acc("bc")[iu])
    if simple:
        assert np.array_equal(pr["chord"], orc.acc("chord")[iu])
        assert np.array_equal(pr["hell"], orc.acc("hell")[iu])
        assert np.array_equal(pr["bc"], orc.acc("kul")[iu])     # kul[i][j] == bc identity
    if complex_:
        assert np.array_equal(pr["whit"], orc.acc("whit")[iu])      # incl. the reference's int32-cast quirk
        assert np.array_equal(pr["canb"], orc.acc("canb")[iu])
        np.testing.assert_allclose(pr["kl"], orc.kl()[iu], rtol=1e-9, atol=1e-15)
    d, s = orc.global_counts()
    assert (int(st.view.nb_distinct_kmers), int(st.view.nb_shared_kmers)) == (d, s)


@pytest.mark.parametrize("k,amin", CONFIGS)
def test_example_accumulators_and_csv(gpu_required, oracle_mod, golden_dir, tmp_path, k, amin):
    """C1: example/simka_input.txt -- integer accumulators bit-exact vs the oracle, CSV bytes == tests/truth."""
    samples, packed = _load_example(golden_dir)
    totals, st = _run_gpu(packed, k, amin, simple=True)
    orc = oracle_mod.Oracle()
    orc.load_input(os.path.join(golden_dir, "example", "simka_input.txt"))
    orc.run(k, amin, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)
    out = str(tmp_path / "res")
    st.write_matrices(out, [s["id"] for s in samples], gz=True)
    truth = os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin))
    compared = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        name = os.path.basename(gzf)[:-3]
        ref = os.path.join(truth, name)
        if not os.path.exists(ref):
            continue            # mat_abundance_jaccard is not pinned by the reference either
        with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
            assert f.read() == g.read(), name
        compared += 1
    assert compared == 20       # every golden of the configuration, -simple-dist and -complex-dist included


def _synthetic(n_samples, nb_reads, read_len, seed_shift=0):
    from simka_amd import synth
    g = synth.genome_len_for(nb_reads, read_len)
    pool, gw = synth.genome_pool_cpu(g)
    out = []
    for s in range(n_samples):
        ids, cdf = synth.sample_profile(s + seed_shift)
        pk = synth.reads_cpu(nb_reads, read_len, pool, gw, g, ids, cdf, synth.sample_seed(s + seed_shift))
        out.append(pk)
    return out


@pytest.mark.parametrize("k,amin,n,R,L,kw", [
    (21, 2, 6, 3000, 100, {}),
    (31, 1, 4, 2000, 150, {}),
    (21, 2, 6, 3000, 100, {"log2_partitions": 6}),          # two-level partitioning (k_split path)
    (21, 2, 6, 3000, 100, {"log2_partitions": 4, "log2_subranges": 3}),
    (15, 2, 5, 2500, 80, {"log2_partitions": 3}),
    (21, 2, 6, 3000, 100, {"log2_partitions": 2}),          # partitions far above the LDS table: multi-round k_count
    (31, 1, 3, 3000, 120, {"log2_partitions": 1, "log2_subranges": 1}),   # ... and over-full k_group sub-ranges
    (1, 1, 3, 500, 60, {}), (2, 2, 3, 500, 60, {}), (5, 1, 4, 800, 60, {}),    # tiny k: 4 / 16 / 1024 possible k-mers, huge counts
    (9, 2, 4, 2000, 100, {"abundance_max": 40}),                          # -abundance-max cuts the frequent k-mers
    (21, 2, 1, 3000, 100, {}),                                            # one sample: no pairs, 1 x 1 matrices
    (31, 2, 3, 400, 30, {}),                                              # reads shorter than k: nothing to count
])
def test_synthetic_vs_oracle(gpu_required, oracle_mod, k, amin, n, R, L, kw):
    from simka_amd import synth
    packed = _synthetic(n, R, L)
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), np.arange(R + 1, dtype=np.uint64) * L, R * L, R) for pk in packed]
    totals, st = _run_gpu(inputs, k, amin, simple=True, **kw)
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), np.arange(R + 1, dtype=np.uint64) * L)
    orc.run(k, amin, amax=kw.get("abundance_max", 999999999), simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)
    # floating-point distances: <= 1e-6 relative (north_star tolerance) -- here they are float32-identical
    for w, name in enumerate(orc.matrix_names()):
        if name in st.matrices():
            np.testing.assert_allclose(st.matrices()[name], orc.matrix(w), rtol=1e-6, atol=0)


@pytest.mark.parametrize("name,k,amin,n,R,L,simple,complex_", [
    # BASELINE configs[2] / configs[3] shape: 100 samples, k = 31, full -simple-dist set, abundance-min 2 -- at a depth the oracle finishes in seconds
    ("c3_shape", 31, 2, 100, 5000, 150, True, False),
    # BASELINE configs[4] shape: 500 samples (tiled pair accumulator: N beyond one LDS tile), k = 31, -simple-dist -complex-dist
    ("c5_shape", 31, 2, 500, 400, 150, True, True),
])
def test_baseline_config_shapes_vs_oracle(gpu_required, oracle_mod, name, k, amin, n, R, L, simple, complex_):
    """The headline configurations' SHAPES (sample count, k, read length, distance families, abundance filter) against the oracle:
    every per-sample total and every integer accumulator bit for bit, KL to 1e-9, every matrix to 1e-6 relative."""
    from simka_amd import synth
    packed = _synthetic(n, R, L)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    totals, st = _run_gpu(inputs, k, amin, simple=simple, complex_=complex_)
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, amin, simple=simple, complex_=complex_, nparts=64, threads=min(32, os.cpu_count() or 1))
    _check_vs_oracle(totals, st, orc, simple=simple, complex_=complex_)
    mats = st.matrices()
    for w, mname in enumerate(orc.matrix_names()):
        if mname in mats:
            np.testing.assert_allclose(mats[mname], orc.matrix(w), rtol=1e-6, atol=0, err_msg=mname)


def test_complex_dist_list_of_large_counts_is_rebuilt_when_it_overflows(gpu_required, oracle_mod, monkeypatch):
    """-complex-dist keeps counts >= 1024 on a fixed-size device list (Whittaker's one-sided terms); when it overflows the list
    is rebuilt at its exact size from the resident spectra instead of failing after the merge (the reference has no such
    limit).  Tiny k gives every k-mer a huge count; SIMKA_OVF_CAP shrinks the list to 2 entries."""
    from simka_amd import synth
    monkeypatch.setenv("SIMKA_OVF_CAP", "2")
    k, amin, n, R, L = 3, 1, 4, 3000, 60
    packed = _synthetic(n, R, L, seed_shift=3)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    totals, st = _run_gpu(inputs, k, amin, simple=True, complex_=True)
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, amin, simple=True, complex_=True)
    assert int(min(orc.totals()["N"])) // 32 >= 1024           # (32 canonical 3-mers per sample: every count is far beyond the exact bins)
    _check_vs_oracle(totals, st, orc)


def test_fixed_len_equals_offsets(gpu_required):
    """fixed_len fast path == explicit offsets."""
    R, L = 4000, 100
    packed = _synthetic(3, R, L, seed_shift=10)
    a = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), np.arange(R + 1, dtype=np.uint64) * L, R * L, R) for pk in packed]
    _, st1 = _run_gpu(a, 21, 2)
    import simka_amd
    ctx = simka_amd.SimkaContext(3, kmer_size=21, abundance_min=2, simple_dist=True, complex_dist=True)
    for i, pk in enumerate(packed):
        ctx.count_sample(i, np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)
    ctx.merge()
    st2 = ctx.stats()
    ctx.close()
    assert np.array_equal(st1.flat, st2.flat)


@pytest.mark.parametrize("k,amin", CONFIGS)
def test_cli_drop_in_on_example(gpu_required, golden_dir, tmp_path, k, amin):
    """The `simka` host driver, invoked as tests/simple_test.py:94 invokes the reference binary; outputs compared as
    tests/simple_test.py:29-68 does (gunzip, string-equal with tests/truth, files present in both dirs)."""
    import subprocess
    from simka_amd import build as b
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    cmd = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out", out, "-out-tmp", tmp,
           "-simple-dist", "-complex-dist", "-kmer-size", str(k), "-abundance-min", str(amin), "-verbose", "0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    truth = os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin))
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20
    # resource-invariance of the reference's test (tests/simple_test.py:125-133): other core/memory settings, same bytes
    out2 = str(tmp_path / "out2")
    r = subprocess.run(cmd[:4] + [out2] + cmd[5:] + ["-nb-cores", "2", "-max-memory", "2000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        with gzip.open(gzf, "rb") as f, gzip.open(os.path.join(out2, os.path.basename(gzf)), "rb") as g:
            assert f.read() == g.read()


def test_heavy_repeat_sample_takes_exact_path(gpu_required, oracle_mod):
    """A poly-A sample puts every k-mer into ONE level-1 bucket: the capacity-sized scatter must flag the overflow and the
    sample be redone with the exact histogram path; low-complexity and ordinary samples mix in one run."""
    from simka_amd import synth
    R, L, k = 2000, 100, 21
    packed = _synthetic(2, R, L, seed_shift=20)
    polya = np.zeros((R * L + 31) // 32, dtype=np.uint64)                       # code 0 = 'A'
    rep = synth.unpack_ascii(packed[0], R * L).copy()
    rep[: R * L // 2] = np.frombuffer(b"ACGT" * (R * L // 8), dtype=np.uint8)   # half tandem repeat, half random
    import simka_amd
    rep_packed, rep_off, rep_nb, _ = simka_amd.pack_reads([rep[i * L:(i + 1) * L].tobytes() for i in range(R)])
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([polya, np.zeros(2, dtype=np.uint64)]), offs, R * L, R),
              (np.concatenate([packed[0], np.zeros(2, dtype=np.uint64)]), offs, R * L, R),
              (rep_packed, rep_off, rep_nb, R),
              (np.concatenate([packed[1], np.zeros(2, dtype=np.uint64)]), offs, R * L, R)]
    totals, st = _run_gpu(inputs, k, 2)
    orc = oracle_mod.Oracle()
    orc.add_sample_ascii("polyA", np.full(R * L, ord("A"), dtype=np.uint8), offs)
    orc.add_sample_ascii("s0", synth.unpack_ascii(packed[0], R * L), offs)
    orc.add_sample_ascii("rep", rep, offs)
    orc.add_sample_ascii("s1", synth.unpack_ascii(packed[1], R * L), offs)
    orc.run(k, 2, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)
    assert int(totals[0]["D"]) == 1 and int(totals[0]["N"]) == R * (L - k + 1)


def test_exact_sizing_env_matches_capacity_sizing(gpu_required, monkeypatch):
    R, L = 3000, 100
    packed = _synthetic(3, R, L, seed_shift=30)
    a = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), np.arange(R + 1, dtype=np.uint64) * L, R * L, R) for pk in packed]
    _, st1 = _run_gpu(a, 21, 2, log2_partitions=6)
    import subprocess, sys, json
    code = ("import os,sys,numpy as np;os.environ['SIMKA_EXACT_SIZING']='1';sys.path.insert(0,%r);sys.path.insert(0,%r);"
            "import test_gpu_parity as t;R,L=3000,100;p=t._synthetic(3,R,L,seed_shift=30);"
            "a=[(np.concatenate([pk,np.zeros(2,dtype=np.uint64)]),np.arange(R+1,dtype=np.uint64)*L,R*L,R) for pk in p];"
            "_,st=t._run_gpu(a,21,2,log2_partitions=6);np.save(sys.argv[1],st.flat)") % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"))
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "simka_exact_flat.npy")
    subprocess.run([sys.executable, "-c", code, out], check=True)
    assert np.array_equal(np.load(out), st1.flat)


def _random_reads_packed(R, L, seed):
    """i.i.d. uniform bases: essentially every k-mer distinct (the low-coverage extreme)."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=R * L, dtype=np.uint64)
    nw = (R * L + 31) // 32
    codes = np.concatenate([codes, np.zeros(nw * 32 - R * L, dtype=np.uint64)]).reshape(nw, 32)
    return np.bitwise_or.reduce(codes << (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :], axis=1)


@pytest.mark.parametrize("order", ["low_first", "high_first"])
def test_low_coverage_all_distinct_partitions(gpu_required, oracle_mod, order):
    """Partitions whose k-mers are (almost) all distinct need the big LDS table; when the table size picked from the first
    sample is too small for a later one, the over-full partitions must be finished by the general kernel."""
    from simka_amd import synth
    R, L, k = 6000, 100, 21
    hi = _synthetic(2, R, L, seed_shift=40)
    lo = [_random_reads_packed(R, L, 7), _random_reads_packed(R, L, 8)]
    lo[1][: len(lo[1]) // 2] = lo[0][: len(lo[0]) // 2]          # share half of the reads so that pairs exist
    packed = (lo + hi) if order == "low_first" else (hi + lo)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    totals, st = _run_gpu(inputs, k, 1)
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, 1, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)


@pytest.mark.parametrize("pb,subranges", [(1, 0), (2, 2), (1, 4)])
def test_segments_beyond_65535_records_take_32_bit_rows(gpu_required, oracle_mod, pb, subranges):
    """A (sample, partition) segment with more than 65535 solid k-mers does not fit the 16-bit rows the count kernels leave for the
    merge: simka_merge sees the largest segment and rebuilds every batch's rows in 32 bits (k_segment_rows<true>, k_group<.., true>).
    Three samples of ~280k distinct k-mers in 2 / 4 partitions: 70k .. 140k records per segment; exact vs the oracle."""
    from simka_amd import synth
    R, L, k = 4000, 100, 31
    packed = [_random_reads_packed(R, L, 21 + i) for i in range(3)]
    packed[1][: len(packed[1]) // 2] = packed[0][: len(packed[0]) // 2]
    packed[2][len(packed[2]) // 3:] = packed[0][len(packed[0]) // 3:]
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    totals, st = _run_gpu(inputs, k, 1, log2_partitions=pb, log2_subranges=subranges)
    assert max(int(t["D"]) for t in totals) >> pb > 65535
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, 1, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)


def test_cli_fastq_gz_inputs(gpu_required, golden_dir, tmp_path):
    """Same sequences as the example, delivered as FASTQ / FASTQ.gz / FASTA.gz: the matrices must not change."""
    import subprocess
    import simka_amd
    from simka_amd import build as b
    ex = os.path.join(golden_dir, "example")
    d = tmp_path / "in"
    d.mkdir()

    def seqs(name):
        return list(simka_amd.read_sequences(os.path.join(ex, name)))

    def fastq(path, ss, gz):
        txt = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(ss))
        (gzip.open if gz else open)(path, "wb").write(txt)

    fastq(str(d / "A.fastq"), seqs("A.fasta"), False)
    fastq(str(d / "B.fq.gz"), seqs("B.fasta"), True)
    with gzip.open(str(d / "C.fasta.gz"), "wb") as f:
        f.write(open(os.path.join(ex, "C.fasta"), "rb").read())
    for n in ("D_paired_1.fasta", "D_paired_2.fasta"):
        (d / n).write_bytes(open(os.path.join(ex, n), "rb").read())
    (d / "input.txt").write_text("A: A.fastq\nB: B.fq.gz\nC: C.fasta.gz\nD: D_paired_1.fasta ; D_paired_2.fasta\n"
                                 "E: A.fastq , A.fastq ; B.fq.gz , B.fq.gz\n")
    out = str(tmp_path / "out")
    r = subprocess.run([b.CLI_PATH, "-in", str(d / "input.txt"), "-out", out, "-out-tmp", str(tmp_path / "tmp"), "-simple-dist",
                        "-complex-dist", "-kmer-size", "31", "-abundance-min", "2", "-verbose", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20


@pytest.mark.parametrize("shards,k,sort_path,shape", [(2, 21, False, "small"), (3, 21, False, "small"), (3, 33, False, "small"), (2, 47, False, "small"),
                                                      (3, 21, True, "small"), (8, 31, False, "c4")])
def test_sharded_contexts_sum_to_unsharded(gpu_required, oracle_mod, monkeypatch, shards, k, sort_path, shape):
    """Partition shards (power-of-two and not) on one GPU: per-sample totals and every pair accumulator add up to the
    single-context result, which equals the oracle.  Hash pipeline (a shard keeps the partitions p % shard_count == shard_index) and sort-based pipeline
    (k >= 32, or forced: a shard keeps the canonical k-mers that hash to it)."""
    import simka_amd
    from simka_amd import synth
    if sort_path:
        monkeypatch.setenv("SIMKA_SORT_PATH", "1")
    # "c4": north_star's decomposition at BASELINE configs[3]'s shape (100 samples, k = 31, 8 partition shards + one sum of the heads)
    n, R, L = (100, 1500, 150) if shape == "c4" else (4, 5000, 100)
    packed = _synthetic(n, R, L, seed_shift=50)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    _, ref = _run_gpu(inputs, k, 2)
    ctxs = []
    for i in range(shards):
        c = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True, shard_index=i, shard_count=shards)
        for s, (pk, off, nb, nin) in enumerate(inputs):
            c.count_sample(s, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
        ctxs.append(c)
    tot = sum(c.totals_download().astype(np.uint64) for c in ctxs)
    flats = []
    for c in ctxs:
        c.totals_upload(tot)
        c.merge()
        flats.append(c.stats().flat)
        c.close()
    lay = ref.layout
    head = sum(f[: lay["head"]].astype(np.uint64) for f in flats)
    P = lay["nb_pairs"]
    klo = lay["acc0"] + 7 * P
    assert np.array_equal(head[:klo], ref.flat[:klo])
    assert np.max(np.abs(head[klo:klo + P].view(np.int64) - ref.flat[klo:klo + P].view(np.int64))) <= 8 * shards
    assert np.array_equal(flats[0][lay["tot0"]: lay["derived"]], ref.flat[lay["tot0"]: lay["derived"]])     # global totals
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, 2, simple=True, complex_=True, nparts=64, threads=min(32, os.cpu_count() or 1))
    iu = np.triu_indices(n, 1)
    assert np.array_equal(ref.pairs()["a"], orc.acc("a")[iu]) and np.array_equal(ref.pairs()["whit"], orc.acc("whit")[iu])


@pytest.mark.parametrize("n,simple,complex_", [(140, True, True), (110, True, False), (135, False, False),
                                               (1100, False, False), (2500, True, False)])   # groups larger than one hash round / one span
def test_many_samples_tiled_pair_accumulators(gpu_required, oracle_mod, n, simple, complex_):
    """More samples than one LDS tile of pair cells: k_pairs<false> walks (I,J) sample tiles.  Samples share genomes, so
    groups span many samples and tiles."""
    import simka_amd
    from simka_amd import synth
    R, L, k = (400 if n < 1000 else 60), 100, 21
    g = synth.genome_len_for(R * 4, L)
    pool, gw = synth.genome_pool_cpu(g)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    packed = []
    for s in range(n):
        ids, cdf = synth.sample_profile(s % 7)                    # 7 community profiles, different reads
        packed.append(synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s if n < 1000 else s % 40)))
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=1, simple_dist=simple, complex_dist=complex_)
    for s, pk in enumerate(packed):
        ctx.count_sample(s, np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)
    totals = [ctx.sample_totals(i) for i in range(n)]
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, 1, simple=simple, complex_=complex_, nparts=8, threads=8)
    _check_vs_oracle(totals, st, orc, simple=simple, complex_=complex_)


@pytest.mark.parametrize("n,complex_,fallback", [(330, True, False), (900, False, False), (330, True, True)])
def test_tiled_pairs_when_most_tile_pairs_share_nothing(gpu_required, oracle_mod, monkeypatch, n, complex_, fallback):
    """Samples in clusters of 10 that share reads only inside their cluster: almost every (span, sample-tile pair) has no pair
    at all, whole ranges of spans are empty for a tile pair (the batching of the tile-major pair kernel skips them), and the
    groups of a cluster sit inside one tile or straddle two.  fallback: more tiles than the tile-major path takes (forced
    here through SIMKA_TM_MAX_TILES): the scan-and-compact kernel runs with its own LDS geometry on the same spans."""
    import simka_amd
    from simka_amd import synth
    if fallback:
        monkeypatch.setenv("SIMKA_TM_MAX_TILES", "2")
    R, L, k = 40, 100, 21
    g = synth.genome_len_for(R * 4, L)
    pool, gw = synth.genome_pool_cpu(g)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    packed = []
    ids, cdf = synth.sample_profile(0)
    for s in range(n):
        c = s // 10
        base = synth.unpack_ascii(synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(1000 + c)), R * L).copy()
        own = synth.unpack_ascii(synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(5000 + s)), R * L)
        m = (s % 10) * 3 * L                                       # the first (s % 10) * 3 reads are the sample's own
        base[:m] = own[:m]
        packed.append(base)
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=1, simple_dist=True, complex_dist=complex_)
    for s, asc in enumerate(packed):
        pk, off, nb, nin = simka_amd.pack_reads([asc[i * L:(i + 1) * L].tobytes() for i in range(R)])
        ctx.count_sample(s, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
    totals = [ctx.sample_totals(i) for i in range(n)]
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    orc = oracle_mod.Oracle()
    for s, asc in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, asc, offs)
    orc.run(k, 1, simple=True, complex_=complex_, nparts=8, threads=8)
    _check_vs_oracle(totals, st, orc, simple=True, complex_=complex_)


def test_kmer_shared_by_more_samples_than_a_span(gpu_required, oracle_mod):
    """1100 samples drawn from 3 read sets that share genomes: groups of 360..1100 entries.  Groups above K3_CAP (1024)
    records cannot be hashed in one k_group round: they take the huge-group list and k_pairs_global."""
    import simka_amd
    from simka_amd import synth
    n, R, L, k = 1100, 20, 100, 21
    g = synth.genome_len_for(R * 4, L)
    pool, gw = synth.genome_pool_cpu(g)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    ids, cdf = synth.sample_profile(0)
    sets = [synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(v)) for v in range(3)]
    sets[1][: len(sets[0]) // 2] = sets[0][: len(sets[0]) // 2]      # sets 0 and 1 share half their reads, set 2 a quarter
    sets[2][: len(sets[0]) // 4] = sets[0][: len(sets[0]) // 4]
    packed = [sets[s % 3] for s in range(n)]
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=1, simple_dist=True, complex_dist=True)
    for s, pk in enumerate(packed):
        ctx.count_sample(s, np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)
    totals = [ctx.sample_totals(i) for i in range(n)]
    ctx.merge()
    st = ctx.stats()
    ctx.close()
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, 1, simple=True, complex_=True, nparts=8, threads=8)
    _check_vs_oracle(totals, st, orc, simple=True, complex_=True)


def _run_cli(args, out, log=None, nb_matrices=18):
    import subprocess
    from simka_amd import build as b
    r = subprocess.run([b.CLI_PATH] + args + ["-out", out, "-verbose", "0" if log is None else "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    if log is not None:
        log.append(r.stdout)
    res = {}
    for gzf in sorted(glob.glob(os.path.join(out, "*.csv.gz"))):
        with gzip.open(gzf, "rb") as f:
            res[os.path.basename(gzf)] = f.read()
    assert len(res) == nb_matrices
    return res


def _oracle_csv(oracle_mod, input_txt, out, k, amin, **policy):
    """the oracle's matrices for an input file under read policies (row a2), as {file name: bytes} like _run_cli"""
    o = oracle_mod.Oracle()
    o.load_input(input_txt)
    o.set_read_policy(**policy)
    o.run(k, amin, simple=True, complex_=False)
    o.write_matrices(out, gz=False)
    res = {}
    for f in sorted(glob.glob(os.path.join(out, "*.csv"))):
        with open(f, "rb") as h:
            res[os.path.basename(f) + ".gz"] = h.read()
    return res, o


@pytest.mark.parametrize("policy", [
    {"max_reads": 7}, {"max_reads": 25}, {"max_reads": 60, "min_read_size": 70},
    {"min_read_size": 90}, {"min_shannon": 1.9}, {"max_reads": 12, "min_shannon": 1.5, "min_read_size": 50}, {},
])
def test_cli_read_policies_vs_oracle(gpu_required, oracle_mod, tmp_path, policy):
    """-max-reads / -min-read-size / -min-shannon-index through the `simka` driver against the ORACLE's restatement of
    SimkaInputIterator / SimkaSequenceFilter (ref: src/core/SimkaCommons.hpp:159-436): several files per paired part (the counter
    runs across them), paired parts, UNEQUAL files per part (files-per-part = composition / nbPaired, :174), FASTQ, low-complexity
    and short reads.  Every CSV byte-identical."""
    rng = np.random.default_rng(11)
    genome = bytes(rng.choice(list(b"ACGT"), size=6000).tolist())

    def reads(seed, n):
        r = np.random.default_rng(seed)
        out = []
        for i in range(n):
            ln = int(r.integers(40, 140))
            st = int(r.integers(0, len(genome) - ln))
            s = genome[st:st + ln]
            if i % 9 == 0:
                s = b"A" * (ln - 8) + b"CGTACGTA"          # low complexity
            if i % 13 == 0:
                s = (b"AC" * ln)[:ln]                       # Shannon index 1.0
            out.append(s)
        return out

    def fasta(name, rs):
        (tmp_path / name).write_bytes(b"".join(b">r%d\n%s\n" % (i, s) for i, s in enumerate(rs)))

    def fastq(name, rs):
        (tmp_path / name).write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(rs)))

    fasta("a1.fa", reads(1, 20)); fasta("a2.fa", reads(2, 30)); fasta("a3.fa", reads(3, 40))
    fastq("b1.fq", reads(4, 35)); fastq("b2.fq", reads(5, 15))
    fasta("c1.fa", reads(6, 50)); fasta("c2.fa", reads(7, 50)); fasta("c3.fa", reads(1, 20) + reads(6, 30))
    (tmp_path / "in.txt").write_text(
        "P: a1.fa , a2.fa , a3.fa\n"              # one part, three files
        "Q: b1.fq ; b2.fq\n"                       # two paired parts (FASTQ)
        "R: c1.fa , c2.fa ; c3.fa , a3.fa\n"       # two parts, two files each
        "U: a1.fa , c1.fa ; b1.fq\n"               # unequal parts: 3 / 2 = 1 file per part -> a1 | c1, b1 never read
        "V: c3.fa\n")
    k, amin = 17, 1
    args = ["-in", str(tmp_path / "in.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", str(k), "-abundance-min", str(amin)]
    if policy.get("max_reads"):
        args += ["-max-reads", str(policy["max_reads"])]
    if policy.get("min_read_size"):
        args += ["-min-read-size", str(policy["min_read_size"])]
    if policy.get("min_shannon"):
        args += ["-min-shannon-index", str(policy["min_shannon"])]
    got = _run_cli(args, str(tmp_path / "o1"))
    ref, orc = _oracle_csv(oracle_mod, str(tmp_path / "in.txt"), str(tmp_path / "o2"), k, amin, **policy)
    assert int(orc.totals()["nb_reads"].sum()) > 20
    assert sorted(got) == sorted(ref)
    for name in ref:
        assert got[name] == ref[name], name


@pytest.mark.parametrize("k", [33, 63, 64, 90, 100, 127])
def test_cli_every_kmer_span_of_the_reference_vs_oracle(gpu_required, oracle_mod, golden_dir, tmp_path, k):
    """`simka -kmer-size k` for the spans the reference builds (32 / 64 / 96 / 128, ref: CMakeLists.txt:66-71, src/SimkaPotara.cpp:132-141)
    on the example data set (reads of 100 bp and more): the driver's CSV bytes equal the oracle's, -simple-dist and -complex-dist."""
    inp = os.path.join(golden_dir, "example", "simka_input.txt")
    got = _run_cli(["-in", inp, "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-complex-dist", "-kmer-size", str(k), "-abundance-min", "1", "-max-reads", "-1"], str(tmp_path / "o1"), nb_matrices=21)
    orc = oracle_mod.Oracle()
    orc.load_input(inp)
    orc.run(k, 1, simple=True, complex_=True)
    orc.write_matrices(str(tmp_path / "o2"), gz=False)
    assert (int(orc.totals()["K_occ"].sum()) > 0) == (k <= 100)       # (the example's reads hold 100 bases: k = 127 counts nothing, in both)
    ref = sorted(glob.glob(os.path.join(str(tmp_path / "o2"), "*.csv")))
    assert len(ref) >= 20
    for f in ref:
        with open(f, "rb") as h:
            assert got[os.path.basename(f) + ".gz"] == h.read(), (k, os.path.basename(f))


def test_cli_auto_max_reads_vs_oracle(gpu_required, oracle_mod, golden_dir, tmp_path):
    """-max-reads 0: the driver derives (min + mean) / 2 of the samples' read counts per paired part (computeMaxReads, ref:
    src/core/SimkaAlgorithm.cpp:377-445; exact counts where gatb estimates) -- same value and same matrices as the oracle."""
    inp = os.path.join(golden_dir, "example", "simka_input.txt")
    o = oracle_mod.Oracle()
    o.load_input(inp)
    m = o.auto_max_reads()
    assert 0 < m < 146
    got = _run_cli(["-in", inp, "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", "21", "-abundance-min", "1", "-max-reads", "0"], str(tmp_path / "o1"))
    ref, _ = _oracle_csv(oracle_mod, inp, str(tmp_path / "o2"), 21, 1, max_reads=m)
    for name in ref:
        assert got[name] == ref[name], name


def test_cli_max_reads_equals_truncated_inputs(gpu_required, golden_dir, tmp_path):
    """-max-reads m: the first m reads of every ';'-part (SimkaInputIterator, ref: src/core/SimkaCommons.hpp:239-289) --
    same matrices as running on files truncated to m reads.  (The example's parts reach m inside their first file.)"""
    import simka_amd
    ex = os.path.join(golden_dir, "example")
    m = 10
    d = tmp_path / "trunc"
    d.mkdir()
    for name in ("A", "B", "C", "D_paired_1", "D_paired_2"):
        seqs = list(simka_amd.read_sequences(os.path.join(ex, name + ".fasta")))[:m]
        (d / (name + ".fasta")).write_bytes(b"".join(b">%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    (d / "input.txt").write_text("A: A.fasta\nB: B.fasta\nC: C.fasta\nD: D_paired_1.fasta ; D_paired_2.fasta\nE: A.fasta ; B.fasta\n")
    common = ["-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", "21", "-abundance-min", "1"]
    a = _run_cli(["-in", os.path.join(ex, "simka_input.txt"), "-max-reads", str(m)] + common, str(tmp_path / "o1"))
    b = _run_cli(["-in", str(d / "input.txt")] + common, str(tmp_path / "o2"))
    assert a == b


def test_cli_read_filters_equal_prefiltered_inputs(gpu_required, tmp_path):
    """-min-read-size / -min-shannon-index (SimkaSequenceFilter, ref: src/core/SimkaCommons.hpp:317-436)."""
    import math
    rng = np.random.default_rng(5)

    def shannon(s):
        n = len(s)
        tot = 0.0
        for ch in b"ACGT":
            f = s.count(bytes([ch])) / n
            if f:
                tot += f * math.log(f) / math.log(2)
        return abs(tot)

    def reads(seed):
        r = np.random.default_rng(seed)
        base = bytes(r.choice(list(b"ACGT"), size=3000).tolist())
        out = []
        for i in range(300):
            ln = int(r.integers(40, 130))
            st = int(r.integers(0, 3000 - ln))
            s = base[st:st + ln]
            if i % 7 == 0:
                s = b"A" * (ln - 6) + b"CGTACG"          # low complexity
            if i % 11 == 0:
                s = (b"AC" * ln)[:ln]                     # Shannon index 1.0
            out.append(s)
        return out

    raw, flt = tmp_path / "raw", tmp_path / "flt"
    raw.mkdir(); flt.mkdir()
    for name, seed in (("X", 1), ("Y", 2), ("Z", 1)):
        rs = reads(seed) if name != "Z" else reads(1)[100:] + reads(2)[:100]
        keep = [s for s in rs if len(s) >= 80 and shannon(s) >= 1.5]
        assert 20 < len(keep) < len(rs)
        (raw / (name + ".fa")).write_bytes(b"".join(b">r\n%s\n" % s for s in rs))
        (flt / (name + ".fa")).write_bytes(b"".join(b">r\n%s\n" % s for s in keep))
    for dd in (raw, flt):
        (dd / "in.txt").write_text("X: X.fa\nY: Y.fa\nZ: Z.fa\n")
    common = ["-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", "15", "-abundance-min", "1"]
    a = _run_cli(["-in", str(raw / "in.txt"), "-min-read-size", "80", "-min-shannon-index", "1.5"] + common, str(tmp_path / "o1"))
    b = _run_cli(["-in", str(flt / "in.txt")] + common, str(tmp_path / "o2"))
    assert a == b


def test_export_import_sample_equals_recounting(gpu_required):
    """-keep-tmp at the C ABI: spectra exported from one context and imported into a fresh one (which then counts the
    remaining samples) give the same flat statistics, bit for bit, as counting everything -- including -complex-dist
    (the count histogram of an imported sample is rebuilt from its counts) and an empty sample."""
    import simka_amd
    R, L, k = 2500, 100, 21
    packed = _synthetic(5, R, L, seed_shift=3)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    inputs[2] = (np.zeros(2, dtype=np.uint64), np.zeros(1, dtype=np.uint64), 0, 0)          # an empty sample travels too
    kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True)

    def count(ctx, i):
        pk, off, nb, nin = inputs[i]
        ctx.count_sample(i, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)

    with simka_amd.SimkaContext(5, **kw) as a:
        for i in range(5):
            count(a, i)
        spectra = [a.export_sample(i) for i in range(4)]
        a.merge()
        ref = a.stats().flat.copy()
        geo = a.geometry()
    assert sum(int(s[1].sum()) for s in spectra) == sum(len(s[2]) for s in spectra) > 0
    # a later run: samples 0..3 come from their spectra (the first import fixes the partition count), sample 4 is counted
    with simka_amd.SimkaContext(5, **kw) as b:
        for i in range(4):
            b.import_sample(i, *spectra[i])
        count(b, 4)
        assert [b.sample_totals(i) for i in range(4)] == [s[0] for s in spectra]
        b.merge()
        got = b.stats().flat.copy()
        assert b.geometry()["log2_level1"] + b.geometry()["log2_level2"] == geo["log2_level1"] + geo["log2_level2"]
    assert np.array_equal(ref, got)
    # mismatching partition count is refused
    with simka_amd.SimkaContext(5, log2_partitions=geo["log2_level1"] + geo["log2_level2"] + 1, **kw) as c:
        count(c, 4)
        with pytest.raises(simka_amd.api.SimkaError):
            c.import_sample(0, *spectra[0])


def test_cli_keep_tmp_adds_samples_without_recounting(gpu_required, golden_dir, tmp_path):
    """README.md:205-206 of the reference: run with -keep-tmp, add samples to the input file, run again -- the old samples are
    not recounted and the result equals a run from scratch (here: the goldens)."""
    import subprocess
    from simka_amd import build as b
    ex = os.path.join(golden_dir, "example")
    lines = [l for l in open(os.path.join(ex, "simka_input.txt")).read().splitlines() if l.strip()]
    fix = lambda l: l.split(":")[0] + ": " + " ; ".join(",".join(os.path.join(ex, f.strip()) for f in part.split(",")) for part in l.split(":", 1)[1].split(";"))
    first, full = str(tmp_path / "in3.txt"), str(tmp_path / "in5.txt")
    open(first, "w").write("\n".join(fix(l) for l in lines[:3]) + "\n")
    open(full, "w").write("\n".join(fix(l) for l in lines) + "\n")
    tmp = str(tmp_path / "tmp")
    base = [b.CLI_PATH, "-out-tmp", tmp, "-keep-tmp", "-simple-dist", "-complex-dist", "-kmer-size", "21", "-abundance-min", "2"]
    r = subprocess.run(base + ["-in", first, "-out", str(tmp_path / "out3")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "reused" not in r.stdout
    specs = sorted(os.listdir(os.path.join(tmp, "simka_output_temp", "solid")))
    assert len(specs) == 3 and all(s.endswith(".g0of1.spec") for s in specs)
    r = subprocess.run(base + ["-in", full, "-out", str(tmp_path / "out5")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("k-mer spectrum reused") == 3, r.stdout
    truth = os.path.join(golden_dir, "truth", "results_k21_t2")
    n = 0
    for gzf in glob.glob(os.path.join(str(tmp_path / "out5"), "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20
    # other parameters: the kept spectra do not apply, everything is recounted (and still right)
    r = subprocess.run(base[:-1] + ["0", "-in", full, "-out", str(tmp_path / "out5b")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "reused" not in r.stdout
    with gzip.open(os.path.join(str(tmp_path / "out5b"), "mat_abundance_braycurtis.csv.gz"), "rb") as f, \
            open(os.path.join(golden_dir, "truth", "results_k21_t0", "mat_abundance_braycurtis.csv"), "rb") as g:
        assert f.read() == g.read()


@pytest.mark.parametrize("world,complex_,shape", [(2, True, "small"), (3, False, "small"), (8, False, "small"), (8, False, "c4"), (3, False, "staged")])
def test_sample_shards_then_partition_range_merge(gpu_required, monkeypatch, world, complex_, shape):
    """The N-GPU job of simka_amd/dist.py::count_exchange_merge, emulated on one GPU: `world` contexts each count the samples
    s % world == r and export them on the device; the pack / unpack phases route every sample's slice of partition range g to
    context g (the all-to-all is done by hand here, by torch.distributed on gloo in tests/test_dist_gloo.py); each context
    imports, merges, and the heads sum to the single-context result bit for bit; totals are global everywhere."""
    import torch
    import simka_amd
    from simka_amd import dist as sdist
    dev = torch.device("cuda:0")
    if shape == "staged":        # the partition tables of the exchange go through the context's pinned staging slots whatever their size (at scale: from 256 KB)
        monkeypatch.setenv("SIMKA_STAGE_MIN", "1")
    # "c4": BASELINE configs[3]'s shape -- 100 samples, k = 31, -simple-dist, abundance-min 2, 8 ranks -- at a depth of 2000 reads
    n, R, L, k = (100, 2000, 150, 31) if shape == "c4" else (7, 3000, 100, 21)
    packed = _synthetic(n, R, L, seed_shift=11)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=complex_, max_kmers_per_sample=R * (L - k + 1))

    def count(ctx, s):
        ctx.count_sample(s, np.concatenate([packed[s], np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)

    with simka_amd.SimkaContext(n, **kw) as c:
        for s in range(n):
            count(c, s)
        c.merge()
        ref = c.stats()
    lay = simka_amd.api.stats_layout(n, ref.dist_flags)
    head = lay["head"]
    # phase A on every rank
    sends = []
    nparts = None
    for r in range(world):
        with simka_amd.SimkaContext(n, **kw) as c:
            mine = sdist.samples_of(r, world, n)
            for s in mine:
                count(c, s)
            local = {s: c.export_sample_device(s, dev) for s in mine}
        if local:
            nparts = len(next(iter(local.values()))[1])
        sends.append(local)
    packs = [sdist.pack_spectra(sends[r], nparts, world, n, r, dev) for r in range(world)]
    tot_all = np.stack([p[1] for p in packs])
    total = np.zeros(head, dtype=np.uint64)
    for g in range(world):
        # the all-to-all, by hand: rank g receives block g of every rank's meta and the g-th split of its key / count buffers
        meta_recv = np.stack([packs[r][0][g] for r in range(world)])
        kr, cr = [], []
        for r in range(world):
            splits = packs[r][4]
            lo = sum(splits[:g])
            kr.append(packs[r][2][lo: lo + splits[g]]); cr.append(packs[r][3][lo: lo + splits[g]])
        assert sdist.recv_splits_of(meta_recv) == [int(x.numel()) for x in kr]
        incoming = sdist.unpack_spectra(meta_recv, tot_all, torch.cat(kr), torch.cat(cr), nparts, world, n, g)
        with simka_amd.SimkaContext(n, **kw) as c:
            for s, t, pc, kk, cc in incoming:
                c.import_sample_device(s, t, pc, kk, cc)
            c.merge()
            st = c.stats()
        total += st.flat[:head]
        tail = slice(head, lay["derived"])
        assert np.array_equal(st.flat[tail], ref.flat[tail])                # per-sample totals: global on every rank
    assert np.array_equal(total, ref.flat[:head])


@pytest.mark.parametrize("world,k,complex_", [(2, 21, True), (3, 31, False), (8, 21, False), (3, 33, False)])
def test_batch_exchange_with_its_tables_on_the_device(gpu_required, world, k, complex_):
    """The batch form of the spectrum exchange that bench.py --gpus N runs (simka_amd/dist.py::pack_batch / import_batch): for one-word
    k-mers the per-(sample, partition) tables never leave the device (simka_pack_plan / simka_pack_run / simka_import_block_device: range
    sums, destination offsets, run lengths, foff / fcnt of the imported runs).  `world` emulated ranks on one GPU -- uneven partition
    ranges for world = 3, empty sample slots for world = 8 --, the all-to-all by slicing; the summed heads equal the single-context run
    bit for bit, and the device tables equal the host-side ones of the per-sample path (pack_spectra).  k = 33: the two-word route."""
    import torch
    import simka_amd
    from simka_amd import dist as sdist
    dev = torch.device("cuda:0")
    n, R, L = 7, 3000, 100
    packed = _synthetic(n, R, L, seed_shift=23)
    kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=complex_, max_kmers_per_sample=R * (L - k + 1))

    def count(ctx, s):
        ctx.count_sample(s, np.concatenate([packed[s], np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)

    with simka_amd.SimkaContext(n, **kw) as c:
        for s in range(n):
            count(c, s)
        c.merge()
        ref = c.stats()
    lay = simka_amd.api.stats_layout(n, ref.dist_flags)
    head = lay["head"]
    packs, P, kw_ = [], None, 1
    for r in range(world):
        mine = sdist.samples_of(r, world, n)
        with simka_amd.SimkaContext(n, **kw) as c:
            for s in mine:
                count(c, s)
            if mine:
                info = c.spectrum_info(mine[0]); P, kw_ = info[1], info[2]
            P_r = P if P is not None else 1
            meta, tot_send, ks, ks2, cs, splits = sdist.pack_batch(c, mine, P_r, kw_, world, n, dev)
            if kw_ == 1 and mine:      # the device tables against the host-side ones of the per-sample path
                local = {s: c.export_sample_device(s, dev) for s in mine}
                m_host, _, k_host, c_host, sp_host = sdist.pack_spectra(local, P_r, world, n, r, dev)
                assert list(sp_host) == list(splits) and torch.equal(k_host, ks) and torch.equal(c_host, cs)
                assert np.array_equal(meta.cpu().numpy()[:, : len(mine), :], m_host[:, : len(mine), : meta.shape[2]])
            packs.append((meta.clone() if isinstance(meta, torch.Tensor) else meta.copy(), tot_send.copy(), ks, ks2, cs, list(splits)))
    assert P is not None
    tot_all = np.stack([pk[1] for pk in packs])
    total = np.zeros(head, dtype=np.uint64)
    for g in range(world):
        if isinstance(packs[0][0], torch.Tensor):
            meta_recv = torch.stack([packs[r][0][g] for r in range(world)])
        else:
            meta_recv = np.stack([packs[r][0][g] for r in range(world)])
        kr, cr, kr2 = [], [], []
        for r in range(world):
            sp = packs[r][5]; lo_ = sum(sp[:g])
            kr.append(packs[r][2][lo_: lo_ + sp[g]]); cr.append(packs[r][4][lo_: lo_ + sp[g]])
            if kw_ == 2:
                kr2.append(packs[r][3][lo_: lo_ + sp[g]])
        with simka_amd.SimkaContext(n, **kw) as c:
            sdist.import_batch(c, g, world, n, P, kw_, meta_recv, tot_all, torch.cat(kr), torch.cat(kr2) if kw_ == 2 else None, torch.cat(cr), dev)
            if complex_:      # the totals are global already (imported): nothing to all-reduce before the merge
                pass
            c.merge()
            st = c.stats()
        total += st.flat[:head]
        tail = slice(head, lay["derived"])
        assert np.array_equal(st.flat[tail], ref.flat[tail])
    if complex_:
        P_ = lay["nb_pairs"]; klo = lay["acc0"] + 7 * P_
        assert np.array_equal(total[:klo], ref.flat[:klo]) and np.array_equal(total[klo + P_: head], ref.flat[klo + P_: head])
        assert np.max(np.abs(total[klo:klo + P_].view(np.int64) - ref.flat[klo:klo + P_].view(np.int64))) <= 64 * world
    else:
        assert np.array_equal(total, ref.flat[:head])


def test_rejected_block_and_stale_plan_leave_the_context_usable(gpu_required):
    """Error behaviour of the device-table exchange: (1) a block that simka_import_block_device rejects AFTER its table kernel
    ran (the meta rows do not sum to nb_records) must leave no foff / fcnt rows behind -- the same context then imports the correct block
    and merges to the single-context result; (2) a pack plan dies with simka_reset / an import: simka_pack_run then fails with
    SIMKA_ERR_STATE instead of gathering with stale starts."""
    import torch
    import simka_amd
    from simka_amd import dist as sdist
    dev = torch.device("cuda:0")
    n, R, L, k = 4, 2000, 100, 21
    packed = _synthetic(n, R, L, seed_shift=61)
    kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=False, max_kmers_per_sample=R * (L - k + 1))

    def count(ctx, s):
        ctx.count_sample(s, np.concatenate([packed[s], np.zeros(2, dtype=np.uint64)]), R * L, R, fixed_len=L)

    with simka_amd.SimkaContext(n, **kw) as c:
        for s in range(n):
            count(c, s)
        P = c.spectrum_info(0)[1]
        mine = list(range(n))
        meta, tot_send, ks, _, cs, splits = sdist.pack_batch(c, mine, P, 1, 1, n, dev)
        c.merge()
        ref = c.stats().flat.copy()
        # (2) the plan is dropped by a reset
        splits2 = c.pack_plan(mine, 1)
        assert list(splits2) == list(splits)
        c.reset()
        with pytest.raises(simka_amd.SimkaError) as ei:
            c.pack_run(ks, cs, meta)
        assert "plan" in str(ei.value)
    tot_all = tot_send[None]
    with simka_amd.SimkaContext(n, **kw) as c:
        bad = meta.clone()
        bad[0, 0, 0] += 1                       # one record more than the block holds
        with pytest.raises(simka_amd.SimkaError) as ei:
            sdist.import_batch(c, 0, 1, n, P, 1, bad, tot_all, ks, None, cs, dev)
        assert "sum to" in str(ei.value)
        with pytest.raises(simka_amd.SimkaError):            # nothing was imported: no sample is counted
            c.merge()
        sdist.import_batch(c, 0, 1, n, P, 1, meta, tot_all, ks, None, cs, dev)
        c.merge()
        assert np.array_equal(c.stats().flat, ref)


@pytest.mark.parametrize("gpus", [2, 3])
def test_cli_multi_gpu_sample_shards_on_one_device(gpu_required, golden_dir, tmp_path, gpus):
    """`simka -nb-gpus G`: samples counted by one-sample contexts (GPU i % G), spectra exported and imported by partition range
    into G merge contexts, heads summed on the host.  -gpu-shared puts all contexts on one device so the path runs here;
    the goldens must come out byte for byte, -complex-dist included, and a -keep-tmp rerun with another G reuses the spectra."""
    import subprocess
    from simka_amd import build as b
    tmp = str(tmp_path / "tmp")
    base = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out-tmp", tmp, "-simple-dist", "-complex-dist",
            "-kmer-size", "31", "-abundance-min", "2", "-keep-tmp", "-gpu-shared"]
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")

    def run(out, g):
        r = subprocess.run(base + ["-out", out, "-nb-gpus", str(g)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        n = 0
        for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
            ref = os.path.join(truth, os.path.basename(gzf)[:-3])
            if os.path.exists(ref):
                with gzip.open(gzf, "rb") as f, open(ref, "rb") as h:
                    assert f.read() == h.read(), os.path.basename(gzf)
                n += 1
        assert n == 20
        return r.stdout

    assert "reused" not in run(str(tmp_path / "o1"), gpus)
    assert run(str(tmp_path / "o2"), 5 - gpus).count("k-mer spectrum reused") == 5        # other GPU count, same spectra


@pytest.mark.parametrize("gpus,k,amin,extra", [(2, 31, 2, []), (3, 21, 0, []), (5, 31, 0, ["-host-parse"]), (8, 21, 2, []), (2, 31, 2, ["-max-reads", "300"])])
def test_cli_partition_shards_on_one_device(gpu_required, golden_dir, tmp_path, gpus, k, amin, extra):
    """`simka -nb-gpus G -gpu-shards partition`: BASELINE.json north_star's decomposition in the C++ driver -- G contexts with shard (g, G), each
    scanning every sample and keeping the minimizer partitions p % G == g, the per-sample totals made global before the -complex-dist merges,
    ONE sum of the N x N accumulators at the end (here on the host: -gpu-shared puts the G contexts on one device; on G distinct devices
    the same code takes the RCCL all-reduce).  The goldens byte for byte, all 20 matrices; a read policy against the one-GPU run."""
    import subprocess
    from simka_amd import build as b
    out = str(tmp_path / "o")
    base = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-complex-dist",
            "-kmer-size", str(k), "-abundance-min", str(amin), "-verbose", "2"] + extra
    r = subprocess.run(base + ["-out", out, "-gpu-shared", "-nb-gpus", str(gpus), "-gpu-shards", "partition"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "%d partition shards, one all-reduce on the host" % gpus in r.stdout, r.stdout
    if "-max-reads" in extra:
        ref = str(tmp_path / "ref")
        r1 = subprocess.run(base + ["-out", ref], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r1.returncode == 0, r1.stdout
        names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ref, "*.csv.gz")))
        assert len(names) == 21           # (the reference's tests/truth holds 20 of the 21 matrices)
        for nme in names:
            with gzip.open(os.path.join(out, nme), "rb") as f, gzip.open(os.path.join(ref, nme), "rb") as h:
                assert f.read() == h.read(), nme
    else:
        truth = os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin))
        n = 0
        for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
            ref = os.path.join(truth, os.path.basename(gzf)[:-3])
            if os.path.exists(ref):
                with gzip.open(gzf, "rb") as f, open(ref, "rb") as h:
                    assert f.read() == h.read(), os.path.basename(gzf)
                n += 1
        assert n == 20
    # the per-sample lines of the driver's report are the whole samples' totals, not a shard's
    r1 = subprocess.run(base + ["-out", str(tmp_path / "one")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r1.returncode == 0, r1.stdout
    pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith("\t") and " / " in ln]
    assert pick(r.stdout) == pick(r1.stdout) and len(pick(r.stdout)) == 5


@pytest.mark.parametrize("gpus,extra", [(2, []), (3, []), (5, []), (2, ["-host-spectra"]), (3, ["-host-parse"]), (2, ["-max-reads", "300"])])
def test_cli_multi_gpu_spectra_stay_on_the_devices(gpu_required, golden_dir, tmp_path, gpus, extra):
    """`simka -nb-gpus G` without -keep-tmp: GPU g counts the samples i = g, g + G, ... in one context (text parsed on the GPU), the
    spectra go from the send buffer of the GPU that counted them to the GPU that merges their partition range (simka_device_copy:
    a peer copy between distinct devices, here -gpu-shared) and never touch the host.  The goldens byte for byte, -complex-dist
    included; -host-spectra takes the older route; a read policy (host parser) goes through the same exchange."""
    import subprocess
    from simka_amd import build as b
    out = str(tmp_path / "o")
    base = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-complex-dist",
            "-kmer-size", "31", "-abundance-min", "2", "-verbose", "2"] + extra
    r = subprocess.run(base + ["-out", out, "-gpu-shared", "-nb-gpus", str(gpus)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert ("spectra exchanged between the GPUs" in r.stdout) == ("-host-spectra" not in extra), r.stdout
    if "-max-reads" in extra:          # no goldens for the policy: against the one-GPU run
        ref = str(tmp_path / "ref")
        r1 = subprocess.run(base + ["-out", ref], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r1.returncode == 0, r1.stdout
        names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ref, "*.csv.gz")))
        assert len(names) >= 20
        for nme in names:
            with gzip.open(os.path.join(ref, nme), "rb") as f, gzip.open(os.path.join(out, nme), "rb") as h:
                assert f.read() == h.read(), nme
        return
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as h:
                assert f.read() == h.read(), os.path.basename(gzf)
            n += 1
    assert n == 20


def test_rccl_communicator_of_the_c_abi_with_one_rank(gpu_required, golden_dir):
    """simka_comm_* on the GPU box: librccl.so is dlopen'ed, a one-rank communicator is created from its own unique id, and the three
    collectives of the C ABI run on it -- all-reduce of u64 words, the uneven all-to-all (rank 0 sends to itself), the all-reduce of a
    context's statistics (a sum over one rank: unchanged).  What a second rank would add is only the peer."""
    import torch
    import simka_amd
    from simka_amd import api
    try:
        uid = api.Comm.unique_id()
    except api.SimkaError as e:
        pytest.skip("no RCCL on this machine: %s" % e)
    # ONE RCCL per process: torch has loaded its own librccl.so (it is linked against it), and that copy serves the C ABI
    which = api.Comm.library()
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
    assert loaded and all(os.path.realpath(p_) == os.path.realpath(loaded[0]) for p_ in loaded), "more than one librccl in the process: %s" % sorted(set(loaded))
    assert os.path.realpath(which.split(" (")[0]) == os.path.realpath(loaded[0]), (which, loaded[0])
    comm = api.Comm(uid, 1, 0, 0)
    dev = torch.device("cuda", 0)
    t = torch.arange(1000, dtype=torch.int64, device=dev)
    comm.allreduce_u64(t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.int64))
    src = torch.arange(5000, dtype=torch.int64, device=dev) * 3 + 1
    dst = torch.zeros(5000, dtype=torch.int64, device=dev)
    comm.alltoallv(src.data_ptr(), [5000], dst.data_ptr(), [5000], 8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    samples, packed = _load_example(golden_dir)
    with simka_amd.SimkaContext(len(packed), kmer_size=21, abundance_min=2, simple_dist=True) as ctx:
        for i, (pk, off, nb, nin) in enumerate(packed):
            ctx.count_sample(i, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
        ctx.merge()
        before = ctx.stats().flat.copy()
        ctx.allreduce_stats(comm, "all")
        ctx.sync()
        assert np.array_equal(ctx.stats().flat, before)
    comm.close()


def test_rccl_single_rank_runs_both_multi_gpu_protocols(gpu_required):
    """scripts/dist_smoke.py: torch.distributed on the real "nccl" (= RCCL) backend with one rank -- the partition-shard protocol
    and the whole sample-shard exchange (all_to_all_single with uneven splits, all_gather, head all-reduce; rank 0 sends to
    itself) reproduce the single-context statistics bit for bit, -complex-dist included."""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "scripts", "dist_smoke.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0 and "dist smoke ok" in r.stdout, r.stdout[-2000:]


def test_arena_mode_is_fixed_at_creation_and_reported(gpu_required):
    """simka_arena_info (advisor, round 4): a lone context maps its arena lazily; a context created while another one is alive on the
    device takes a plain allocation (decided in simka_create, not at the lazy geometry setup); the address ranges of destroyed contexts
    are retired and the total is visible."""
    import simka_amd
    kw = dict(kmer_size=21, abundance_min=2, max_kmers_per_sample=200_000)
    with simka_amd.SimkaContext(2, **kw) as a:
        ia = a.arena_info()
        with simka_amd.SimkaContext(2, **kw) as b:
            ib = b.arena_info()
        assert ib["mapped_range"] is False and ib["mapped_records"] == ib["reserved_records"] > 0
    retired0 = ia["retired_va_bytes"]
    if os.environ.get("SIMKA_ARENA_MALLOC"):
        return
    assert ia["mapped_range"] is True and ia["reserved_records"] > 0
    with simka_amd.SimkaContext(2, **kw) as c:
        # (round 5) an arena of at most four chunks is backed when the geometry is set up -- before the context has launched a kernel:
        # chunks mapped while kernels run are the suspected pattern behind a rare GPU memory access fault (docs/rounds/r05.md)
        pk, off, nb, nin = simka_amd.pack_reads([b"ACGTTGCAAGGCTTAACCGGTTAAGCGCGATATCGGCTAAGCTT", b"TTGCAAGGCTTAACCGGTTAAGCGCGATATCGGCTAAGCTTACG"])
        c.count_sample(0, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
        ic = c.arena_info()
        if ic["reserved_records"] <= 4 * (1 << 27):
            assert ic["mapped_records"] >= min(ic["reserved_records"], 1 << 27), ic
    assert ic["mapped_range"] is True and ic["retired_va_bytes"] >= retired0 + ia["reserved_records"] * 12


def test_capacity_errors_are_reported_not_silent(gpu_required):
    """A solid-spectrum arena / merge buffer that is too small is an error code with a message (the reference would fill the
    disk), never a silently truncated result; the context stays usable after simka_reset."""
    import simka_amd
    R, L, k = 3000, 100, 21
    packed = _synthetic(3, R, L, seed_shift=40)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]

    def run(ctx):
        for i, (pk, off, nb, nin) in enumerate(inputs):
            ctx.count_sample(i, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
        ctx.merge()
        return ctx.stats().flat.copy()

    with simka_amd.SimkaContext(3, kmer_size=k, abundance_min=1) as good:
        ref = run(good)
    with simka_amd.SimkaContext(3, kmer_size=k, abundance_min=1, solid_capacity=20000) as c:     # ~240k solid k-mers per sample
        with pytest.raises(simka_amd.api.SimkaError) as e:
            run(c)
        assert "arena" in str(e.value)
    with simka_amd.SimkaContext(3, kmer_size=k, abundance_min=1) as c:
        assert np.array_equal(run(c), ref)
        c.reset()
        assert np.array_equal(run(c), ref)          # a context is reusable run after run
        with pytest.raises(simka_amd.api.SimkaError):
            c.count_sample(0, *inputs[0][:1], inputs[0][2], R, offsets=offs)     # after the merge: state error, not a crash


@pytest.mark.parametrize("copies,pb", [(3000, 8), (3000, 11), (5500, 6)])
def test_hot_reads_overflow_regions_into_spill_runs(gpu_required, oracle_mod, copies, pb):
    """Adapter-like input: half (or nearly all) of a sample's reads are copies of ONE read, so ~80 k-mers carry thousands of
    occurrences each and overflow their capacity-sized level-2 regions.  The excess goes to the spill buffer as runs; k_count
    finds the runs of its partition through the run list.  Exact vs the oracle, next to an ordinary sample."""
    import simka_amd
    from simka_amd import synth
    R, L, k = 6000, 100, 21
    packed = _synthetic(2, R, L, seed_shift=60)
    a = synth.unpack_ascii(packed[0], R * L).reshape(R, L).copy()
    a[100:100 + copies] = a[7]
    hot, off, nb, _ = simka_amd.pack_reads([row.tobytes() for row in a])
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(hot, off, nb, R), (np.concatenate([packed[1], np.zeros(2, dtype=np.uint64)]), offs, R * L, R)]
    totals, st = _run_gpu(inputs, k, 2, log2_partitions=pb)
    orc = oracle_mod.Oracle()
    orc.add_sample_ascii("hot", a.reshape(-1), offs)
    orc.add_sample_ascii("plain", synth.unpack_ascii(packed[1], R * L), offs)
    orc.run(k, 2, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)


def test_deep_samples_counted_in_passes(gpu_required):
    """A sample too deep for the scratch buffers is counted in several passes over its reads, each keeping a subset of the
    level-1 buckets (SIMKA_FORCE_PASSES forces that here).  Same statistics bit for bit as one pass -- including a sample whose
    poly-A third overflows a level-1 bucket, so one of the passes is redone with exact sizing."""
    import subprocess, sys

    def digest(passes):
        env = dict(os.environ)
        if passes:
            env["SIMKA_FORCE_PASSES"] = str(passes)
        r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "tests", "flat_digest.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
        assert r.returncode == 0 and lines, r.stdout[-2000:]
        return lines[0]

    one = digest(0)
    for p in (2, 3, 8):
        assert digest(p) == one, p


@pytest.mark.parametrize("passes", [0, 3])
def test_fill_cursors_per_bucket_are_result_neutral(gpu_required, golden_dir, tmp_path, passes):
    """Round 6: a long launch of k_skm_scan gives every level-1 bucket 2^n fill cursors (a returning atomic on one word is served every
    ~11 ns, and every tile bumps every bucket's cursor); the cursors share the bucket's region chunk by chunk, the chunk sort and the
    count kernels' gather see the bucket as before.  SIMKA_SCAN_SUB forces n on small inputs: the statistics of tests/flat_digest.py
    (synthetic samples, one with a poly-A third that overflows a bucket -> exact redo with one cursor) are identical for n = 0..3, alone and
    under SIMKA_FORCE_PASSES; and the driver reproduces the goldens with eight cursors per bucket."""
    import subprocess, sys

    def digest(n):
        env = dict(os.environ, SIMKA_SCAN_SUB=str(n))
        if passes:
            env["SIMKA_FORCE_PASSES"] = str(passes)
        r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "tests", "flat_digest.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
        assert r.returncode == 0 and lines, r.stdout[-2000:]
        return lines[0]

    one = digest(0)
    for n in (1, 2, 3):
        assert digest(n) == one, n
    if passes:
        return
    from simka_amd import build as b
    out = str(tmp_path / "o")
    r = subprocess.run([b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out", out, "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-complex-dist",
                        "-kmer-size", "31", "-abundance-min", "2"], env=dict(os.environ, SIMKA_SCAN_SUB="3"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as h:
                assert f.read() == h.read(), os.path.basename(gzf)
            n += 1
    assert n == 20


def test_cli_spectra_larger_than_the_arena_merge_by_partition_ranges(gpu_required, golden_dir, tmp_path):
    """When the solid spectra of all samples do not fit the HBM arena (here: -solid-capacity far too small) the driver notices the
    NOMEM, recounts with one-sample contexts, keeps the spectra in host memory and merges the partition space range by range;
    -merge-ranges asks for that directly.  Byte-identical CSVs either way."""
    import subprocess
    from simka_amd import build as b
    base = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist",
            "-complex-dist", "-kmer-size", "21", "-abundance-min", "0"]
    truth = os.path.join(golden_dir, "truth", "results_k21_t0")

    def run(out, extra):
        r = subprocess.run(base + ["-out", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        n = 0
        for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
            ref = os.path.join(truth, os.path.basename(gzf)[:-3])
            if os.path.exists(ref):
                with gzip.open(gzf, "rb") as f, open(ref, "rb") as h:
                    assert f.read() == h.read(), os.path.basename(gzf)
                n += 1
        assert n == 20
        return r.stdout

    assert "partition ranges" in run(str(tmp_path / "o1"), ["-merge-ranges", "3"])
    log = run(str(tmp_path / "o2"), ["-solid-capacity", "20000"])          # the 5 samples hold ~45000 solid k-mers
    assert "do not fit the GPU memory" in log and "partition ranges" in log


@pytest.mark.parametrize("k,amin,n,R,L,how", [(21, 2, 5, 3000, 100, "buckets"), (31, 1, 4, 2000, 150, "buckets"), (9, 2, 3, 1500, 80, "buckets"), (32, 2, 4, 2500, 120, "buckets"),
                                              (33, 1, 4, 2500, 120, "buckets"), (47, 2, 5, 2500, 150, "buckets"), (63, 1, 3, 2000, 150, "buckets"), (55, 2, 4, 2500, 150, "buckets"),
                                              (33, 1, 150, 120, 100, "buckets"),     # more samples than one LDS tile: the tile-major pair kernel on the sorted CSR
                                              (31, 2, 4, 2000, 150, "sort"), (47, 2, 4, 2500, 150, "sort"), (63, 1, 3, 2000, 150, "sort"), (63, 2, 3, 2000, 150, "failed")])
def test_sort_based_path_for_wide_kmers(gpu_required, oracle_mod, monkeypatch, k, amin, n, R, L, how):
    """k >= 52 (k-mers of up to 126 bits, the reference's span-64 build) takes the path of simka_wide.hip that looks at every k-mer
    occurrence on its own -- scattered into buckets by a hash of the k-mer and counted per bucket in an LDS table ("buckets"), or sorted
    by k-mer ("sort": what a bucket with too many distinct k-mers falls back to; "failed": that fallback after a bucket count declared
    failed, with the abundance histogram of -complex-dist restored); the same path is forced for smaller k (SIMKA_SORT_PATH) as a
    cross-check of the partitioned pipeline.  Totals and every accumulator vs the oracle (128-bit k-mers), -simple-dist and
    -complex-dist.  k >= 32 has no golden vectors in the reference (parity unpinned): the oracle is the same code that
    reproduces the k = 21 / 31 goldens."""
    monkeypatch.setenv("SIMKA_SORT_PATH", "1")
    if how == "sort":
        monkeypatch.setenv("SIMKA_WIDE_COUNT_SORT", "1")
    if how == "failed":
        monkeypatch.setenv("SIMKA_WIDE_COUNT_FAIL", "1")
    _wide_case(oracle_mod, k, amin, n, R, L, expect="sorted", full_sorts=0 if how == "buckets" else n)


@pytest.mark.parametrize("k,amin,n,R,L,how,fixed", [(64, 2, 4, 2500, 150, "buckets", True), (65, 1, 3, 2000, 150, "buckets", False), (96, 2, 4, 2500, 200, "buckets", True),
                                                    (127, 2, 4, 2500, 250, "buckets", True), (127, 1, 3, 1500, 150, "sort", False), (100, 2, 3, 1500, 90, "buckets", True)])
def test_kmers_of_64_to_127_bases(gpu_required, oracle_mod, monkeypatch, k, amin, n, R, L, how, fixed):
    """k = 64..127 (the reference's Kmer<span=96/128> builds, ref: CMakeLists.txt:66-71, src/SimkaPotara.cpp:132-141): the scan rolls
    four-word k-mers and hands their 126-bit fingerprints to the two-word path (bucket count or sort, bucket merge).  Totals and every
    accumulator, -simple-dist and -complex-dist, against the oracle, which keeps the k-mers whole (dictionary ranks, pinned against a
    pure-Python restatement on strings: tests/test_oracle_golden.py).  (100, ..., L = 90): reads shorter than k -- nothing to count."""
    if how == "sort":
        monkeypatch.setenv("SIMKA_WIDE_COUNT_SORT", "1")
    _wide_case(oracle_mod, k, amin, n, R, L, expect="sorted", fixed=fixed, full_sorts=0 if how == "buckets" else n)


def _wide_case(oracle_mod, k, amin, n, R, L, expect, fixed=True, full_sorts=0, **ctx_kw):
    from simka_amd import synth
    packed = _synthetic(n, R, L, seed_shift=70)
    offs = np.arange(R + 1, dtype=np.uint64) * L
    inputs = [(np.concatenate([pk, np.zeros(2, dtype=np.uint64)]), offs, R * L, R) for pk in packed]
    import simka_amd
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, simple_dist=True, complex_dist=True, **ctx_kw)
    for s, (pk, of, nb, nr) in enumerate(inputs):
        ctx.count_sample(s, pk, nb, nr, fixed_len=L if fixed else 0, offsets=None if fixed else of, nb_input_reads=nr)
    totals = [ctx.sample_totals(i) for i in range(n)]
    ctx.merge()
    st = ctx.stats()
    paths = ctx.count_paths()
    ctx.close()
    assert paths[expect] == n and paths["partitioned"] + paths["sorted"] == n and paths["full_sorts"] == full_sorts, paths
    orc = oracle_mod.Oracle()
    for s, pk in enumerate(packed):
        orc.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk, R * L), offs)
    orc.run(k, amin, simple=True, complex_=True)
    _check_vs_oracle(totals, st, orc)
    for w, name in enumerate(orc.matrix_names()):
        if name in st.matrices():
            np.testing.assert_allclose(st.matrices()[name], orc.matrix(w), rtol=1e-6, atol=0)


@pytest.mark.parametrize("k,amin,n,R,L,fixed", [(32, 2, 4, 2500, 120, True), (33, 1, 4, 2500, 120, True), (34, 1, 3, 2000, 100, False), (35, 2, 3, 2000, 100, True),
                                                (36, 1, 3, 2000, 100, True), (41, 2, 4, 2500, 150, False), (47, 2, 5, 2500, 150, True), (50, 1, 3, 2000, 150, True),
                                                (51, 1, 3, 2000, 150, False), (51, 2, 3, 3000, 51, True), (33, 1, 150, 120, 100, True)])
def test_wide_kmers_on_the_partitioned_pipeline(gpu_required, oracle_mod, k, amin, n, R, L, fixed):
    """32 <= k <= 51: a super-k-mer record still holds a k-mer, so the samples are counted per minimizer partition in LDS tables of
    two-word keys (k_skm_count_wide; the minimizer window sits in the middle of the k-mer, every parity of k - m) and only their
    solid records are sorted; the merge is the sorted one.  Same oracle, same checks as the sort-based path."""
    _wide_case(oracle_mod, k, amin, n, R, L, expect="partitioned", fixed=fixed)


@pytest.mark.parametrize("k", [33, 57, 70])
def test_wide_kmers_on_a_non_blocking_caller_stream(gpu_required, oracle_mod, k):
    """simka_config.stream: a caller-supplied NON-BLOCKING stream (torch's pool streams are created with hipStreamNonBlocking) does not
    synchronise with the null stream, so every kernel and copy of the k >= 32 state must run on that very stream -- a round-4 slip had
    them on the null stream, which only the blocking private stream of the other tests forgave (advisor, round 4)."""
    torch = gpu_required
    stream = torch.cuda.Stream()
    for rep in range(3):
        _wide_case(oracle_mod, k, 2, 4, 3000 + 500 * rep, 150, expect="partitioned" if k <= 51 else "sorted", stream=stream.cuda_stream)


@pytest.mark.parametrize("per_part,general,expect", [(3000, False, "partitioned"), (192, True, "partitioned"), (200000, False, "sorted")])
def test_wide_kmers_partitions_beyond_the_tables(gpu_required, oracle_mod, monkeypatch, per_part, general, expect):
    """The three ways a partition of two-word k-mers gets counted: in a wave's table (the other tests), in the block's table when it
    outgrows that (here: partitions made 16 times too large, or every partition sent there), and -- when even that one fills --
    the whole sample on the sort-based path.  Same results."""
    monkeypatch.setenv("SIMKA_WIDE_PER_PART", str(per_part))
    if general:
        monkeypatch.setenv("SIMKA_SKM_GENERAL", "1")
    _wide_case(oracle_mod, 37, 2, 3, 4000, 120, expect=expect)       # ("sorted": occurrence by occurrence, in hash buckets)


@pytest.mark.parametrize("k,n,env", [(33, 6, "SIMKA_WIDE_MERGE_SORT"), (47, 4, "SIMKA_WIDE_MERGE_SORT"), (63, 3, "SIMKA_WIDE_MERGE_SORT"), (33, 5, "SIMKA_WIDE_MERGE_FAIL")])
def test_wide_merge_by_full_sort_equals_the_grouped_merge(gpu_required, oracle_mod, monkeypatch, k, n, env):
    """The merge of two-word k-mers groups the records by hash bucket + an LDS table (k_wlocal_group); the full sort by k-mer stays as
    the route for a bucket whose table fills and is forced here (directly, and after a grouping declared failed): same oracle, same checks."""
    monkeypatch.setenv(env, "1")
    _wide_case(oracle_mod, k, 1, n, 3000, 130, expect="partitioned" if k <= 51 else "sorted", full_sorts=1)


def test_wide_merge_groups_kmers_shared_by_thousands_of_samples(gpu_required):
    """2500 samples with the same few reads: every k-mer's group holds 2500 records -- far beyond what a thread block keeps in
    registers between the two passes of k_wlocal_group, and more than a bucket's mean -- and the merge still groups them (no fallback
    to the sort): all pairs at distance zero, the shared k-mers = one sample's distinct k-mers."""
    import simka_amd
    n, R, L, k = 2500, 12, 90, 35
    pk = _synthetic(1, R, L, seed_shift=91)[0]
    packed = np.concatenate([pk, np.zeros(2, dtype=np.uint64)])
    with simka_amd.SimkaContext(n, kmer_size=k, abundance_min=1, simple_dist=False, complex_dist=False) as ctx:
        for s in range(n):
            ctx.count_sample(s, packed, R * L, R, fixed_len=L)
        t0 = ctx.sample_totals(0)
        ctx.merge()
        st = ctx.stats()
        assert ctx.count_paths()["full_sorts"] == 0
    m = st.matrices()
    assert t0["D"] > 500
    for name in ("mat_abundance_braycurtis", "mat_presenceAbsence_jaccard"):
        assert name in m and np.all(m[name] == 0.0), name
    assert int(st.view.nb_shared_kmers) == int(t0["D"]) == int(st.view.nb_distinct_kmers)


def test_example_goldens_through_the_sort_path(gpu_required, golden_dir, tmp_path, monkeypatch):
    """The reference's own example (k = 31, abundance-min 2, all 20 matrices) through the sort-based path: the CSV bytes of
    tests/truth must come out of BOTH pipelines."""
    monkeypatch.setenv("SIMKA_SORT_PATH", "1")
    samples, packed = _load_example(golden_dir)
    totals, st = _run_gpu(packed, 31, 2, simple=True)
    out = str(tmp_path / "res")
    st.write_matrices(out, [s["id"] for s in samples], gz=True)
    truth = os.path.join(golden_dir, "truth", "results_k31_t2")
    n = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20


@pytest.mark.parametrize("seconds,seed,env", [(20, 12345, {}), (40, 60606, {"SIMKA_SCAN_SUB": "1"}), (30, 777, {"SIMKA_LANES": "1", "SIMKA_SCAN_SUB": "3"})])
def test_randomised_inputs_against_the_oracle(gpu_required, oracle_mod, seconds, seed, env):
    """scripts/fuzz_vs_oracle.py for 20 + 40 + 30 s (three seeds; ~1500 random cases): random k (1..127), abundance windows, sample counts, read
    lengths (0..180, shorter than k included), N / IUPAC / lowercase letters, empty samples, partition geometries, distance families --
    totals and every accumulator bit-exact, matrices within 1e-6 (NaN where the reference's arithmetic gives NaN: an empty sample with
    -complex-dist).  The second and third run force two / eight fill cursors per level-1 bucket (round 6) on these small inputs."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "scripts", "fuzz_vs_oracle.py"), str(seconds), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-3000:]
    # the script opens with the KL edge block (identical and near-identical samples, k = 21 and 31: the both-present sum cancels to exactly 0
    # and the Jensen-Shannon cell is 1 as in the reference, a near-identical pair never goes negative) -- part of this test, not a side script
    assert r.stdout.count("jensen-shannon reference") == 12, r.stdout[:2000]


def test_wide_kmers_through_spectrum_export_paths(gpu_required, golden_dir, tmp_path):
    """k = 33 (two-word keys, sort-based path): -keep-tmp reuse, -nb-gpus 2 / 3 (spectra exported, imported by key-prefix range into
    per-GPU merge contexts) and -merge-ranges give the same CSV bytes as the plain single-context run, and the API round trip
    export -> import reproduces the flat statistics."""
    import subprocess
    import simka_amd
    from simka_amd import build as b
    base = [b.CLI_PATH, "-in", os.path.join(golden_dir, "example", "simka_input.txt"), "-simple-dist", "-complex-dist", "-kmer-size", "33",
            "-abundance-min", "1"]

    def run(name, extra):
        out = str(tmp_path / name)
        r = subprocess.run(base + ["-out", out, "-out-tmp", str(tmp_path / ("tmp_" + name.split("_")[0]))] + extra, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        res = {os.path.basename(f): gzip.open(f, "rb").read() for f in glob.glob(os.path.join(out, "*.csv.gz"))}
        assert len(res) == 21
        return res, r.stdout

    ref, _ = run("plain", [])
    res2, log_g2 = run("g2", ["-nb-gpus", "2", "-gpu-shared"])            # spectra from GPU to GPU (high and low words in separate buffers)
    assert res2 == ref and "spectra exchanged between the GPUs" in log_g2
    assert run("g3", ["-nb-gpus", "3", "-gpu-shared"])[0] == ref
    assert run("g2h", ["-nb-gpus", "2", "-gpu-shared", "-host-spectra"])[0] == ref      # ... and through host memory
    assert run("ranges", ["-merge-ranges", "4"])[0] == ref
    first, log1 = run("keep_a", ["-keep-tmp"])
    again, log2 = run("keep_b", ["-keep-tmp"])
    assert first == ref and again == ref and "reused" not in log1 and log2.count("k-mer spectrum reused") == 5
    # API round trip
    samples, packed = _load_example(golden_dir)
    kw = dict(kmer_size=33, abundance_min=1, simple_dist=True, complex_dist=True)
    with simka_amd.SimkaContext(len(packed), **kw) as a:
        for i, (pk, off, nb, nin) in enumerate(packed):
            a.count_sample(i, pk, nb, len(off) - 1, offsets=off, nb_input_reads=nin)
        spectra = [a.export_sample(i) for i in range(len(packed))]
        a.merge()
        flat = a.stats().flat.copy()
    assert all(len(s[2]) == 2 * len(s[3]) for s in spectra)
    with simka_amd.SimkaContext(len(packed), **kw) as c:
        for i in reversed(range(len(packed))):
            c.import_sample(i, *spectra[i])
        c.merge()
        assert np.array_equal(c.stats().flat, flat)


def test_hash_and_sort_pipelines_agree_at_scale(gpu_required):
    """scripts/cross_check.py, 3 rounds: device-generated samples of 50k..300k reads with adapter-like hot reads and poly-A stretches
    (spill runs, exact redo), random k / abundance-min / distance families -- the flat statistics of the two pipelines are identical."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "scripts", "cross_check.py"), "3", "11"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0 and "cross-check ok" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("workload,alt_pb", [("c2", 13), ("c3_10", 14), ("c3", 20), ("c5_5", 14)])
def test_full_size_size_independent_properties(gpu_required, workload, alt_pb):
    """BASELINE configs[1] at FULL size (c2: 10 samples x 1M x 100 bp, k = 21, 8e8 k-mer occurrences -- far beyond what the
    oracle finishes in seconds) and configs[2]'s shape at a tenth of its depth (c3_10: 100 samples x 1M x 150 bp, k = 31,
    -simple-dist, 1.2e10 occurrences; the full 37 GB of reads are bench.py --workload c3), checked through properties that hold
    and configs[2] ITSELF (c3: 100 samples x 10M x 150 bp, k = 31, 1.2e11 occurrences, ~190 GB of HBM)
    and configs[4]'s shape at the largest depth one GPU holds (c5_5: 500 samples x 1M x 150 bp, k = 31, -simple-dist -complex-dist,
    6e10 occurrences, the tiled pair accumulator; Whittaker / Canberra / KL join the permutation check)
    for any input:
      * the partition geometry is result-neutral: another number of partitions gives bit-identical statistics;
      * the path is equivariant under a permutation of the samples (S_ij <-> S_ji where the order of a pair flips);
      * a sample fed twice is at distance zero from itself: a = D, S_ij = S_ji = bc = N, chord = Q;
      * bounds every (i, j) obeys: a <= min(D_i, D_j), bc <= min(S_ij, S_ji), S_ij <= N_i, S_ji <= N_j."""
    import sys
    import torch
    import simka_amd
    if ROOT_DIR not in sys.path:
        sys.path.insert(0, ROOT_DIR)
    import bench
    lib = simka_amd.load_library()
    wl = dict(bench.WORKLOADS[workload])
    n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
    dev = torch.device("cuda:0")
    cplx = bool(wl.get("complex"))
    _, reads = bench.gen_device_samples(lib, torch, wl, dev)

    def run(order, **kw):
        with simka_amd.SimkaContext(len(order), kmer_size=k, abundance_min=wl["amin"], simple_dist=True, complex_dist=cplx, max_kmers_per_sample=R * (L - k + 1), **kw) as c:
            for slot, s in enumerate(order):
                c.count_sample(slot, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
            c.merge()
            return c.stats()

    ident = list(range(n))
    base = run(ident)
    assert int(base.per_sample()["K_occ"].sum()) == n * R * (L - k + 1)
    # geometry
    other = run(ident, log2_partitions=alt_pb)
    if cplx:      # the both-present KL sums are 2^-60 fixed point: integer adds, so the geometry does not change them either
        assert np.array_equal(base.flat[: base.layout["derived"]], other.flat[: other.layout["derived"]])
        np.testing.assert_allclose(other.pairs()["kl"], base.pairs()["kl"], rtol=1e-12, atol=1e-15)
    else:
        assert np.array_equal(base.flat, other.flat)
    del other
    # permutation of the samples
    perm = [int(x) for x in np.random.default_rng(5).permutation(n)]
    pst = run(perm)
    bp, pp = base.per_sample(), pst.per_sample()
    for name in ("D", "N", "Q", "D_all", "K_occ"):
        assert np.array_equal(pp[name], bp[name][perm]), name
    bS, pS = base.dense("S"), pst.dense("S")
    full = lambda m: m + m.T
    assert np.array_equal(pS, bS[np.ix_(perm, perm)])
    for name in ("a", "bc", "chord", "hell") + (("whit", "canb") if cplx else ()):
        assert np.array_equal(full(pst.dense(name)), full(base.dense(name))[np.ix_(perm, perm)]), name
    if cplx:      # KL: the pair term is evaluated in the order of the pair, so a flipped pair may round differently
        iu_ = np.triu_indices(n, 1)
        kb, kp = np.zeros((n, n)), np.zeros((n, n))
        kb[iu_], kp[iu_] = base.pairs()["kl"], pst.pairs()["kl"]
        np.testing.assert_allclose(full(kp), full(kb)[np.ix_(perm, perm)], rtol=1e-9, atol=1e-15)
    # bounds
    pr = base.pairs()
    iu = np.triu_indices(n, 1)
    D, N = bp["D"], bp["N"]
    assert np.all(pr["a"] <= np.minimum(D[iu[0]], D[iu[1]]))
    assert np.all(pr["bc"] <= np.minimum(pr["S_ij"], pr["S_ji"]))
    assert np.all(pr["S_ij"] <= N[iu[0]]) and np.all(pr["S_ji"] <= N[iu[1]])
    # a duplicated sample
    dup = run([0, 0, 1])
    dp, dr = dup.per_sample(), dup.pairs()            # pairs in order (0,1) (0,2) (1,2)
    assert dp["D"][0] == dp["D"][1] == bp["D"][0] and dp["N"][0] == dp["N"][1]
    assert dr["a"][0] == dp["D"][0] and dr["S_ij"][0] == dr["S_ji"][0] == dr["bc"][0] == dp["N"][0] and dr["chord"][0] == dp["Q"][0]
    m = dup.matrices()
    for name, mat in m.items():
        if name.startswith("mat_"):
            # (the reference never writes _kulczynski_minNiNj[j][i], ref: src/core/SimkaDistance.cpp:1024-1038, so its abundance
            # Kulczynski distance of identical samples is 1 - (1 + 0) / 2; the goldens pin that quirk and the host mirrors it)
            want = 0.5 if name == "mat_abundance_kulczynski" else 0.0
            if name == "mat_abundance_jensenshannon":       # KL == 0 -> 1 in the reference (ref: src/core/SimkaDistance.cpp:1001-1007)
                want = 1.0
            v = float(mat[0, 1])
            # chord / Hellinger are sqrt(2 - 2 x / (sqrt(Q) sqrt(Q))) in the reference (ref: src/core/SimkaDistance.cpp:942-972):
            # for identical samples the denominator may round a hair below x and the reference's own formula yields NaN
            if np.isnan(v) and name in ("mat_abundance_chord", "mat_abundance_hellinger"):
                continue
            assert abs(v - want) < 1e-6, (name, mat[0, 1])
    i01 = int(np.flatnonzero((iu[0] == 0) & (iu[1] == 1))[0])
    assert dr["S_ij"][1] == dr["S_ij"][2] == pr["S_ij"][i01] and dr["bc"][1] == dr["bc"][2] == pr["bc"][i01]


def test_tile_major_pair_kernels_agree_with_the_scan_and_compact_kernel(gpu_required):
    """scripts/cross_check_tiled.py, 5 rounds: 130..800 samples (copies of D distinct device-generated ones, so groups of a few to
    hundreds of samples), random k (hash and sort pipelines) / abundance-min / distance families -- the flat statistics of the
    tile-major path, of k_pairs<TILED> and of the sort pipeline are identical."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "scripts", "cross_check_tiled.py"), "5", "21"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0 and "tiled cross-check ok" in r.stdout, r.stdout[-3000:]


def test_bench_on_two_gpus_over_rccl(gpu_required):
    """`bench.py --gpus 2` as the driver launches it (no launcher: it spawns its own ranks, one per GPU, RCCL through the C ABI):
    both decompositions run, report n_gpus = 2 and the distance matrices of the 1-GPU run.  Needs two devices."""
    import json
    import subprocess
    import sys
    torch = gpu_required
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    def run(n):
        r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", str(n), "--workload", "c2", "--reads", "200000", "--steps", "1",
                            "--warmup", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one, two = run(1), run(2)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["matrix_checksum"] == one["config"]["matrix_checksum"]
    assert set(two["decompositions"]) == {"sample", "partition"}
    assert all(d["matrix_checksum"] == one["config"]["matrix_checksum"] for d in two["decompositions"].values())
    assert "RCCL" in two["config"]["collectives"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_n_ranks_on_one_gpu_over_gloo(gpu_required, ranks):
    """The N > 1 path of bench.py end to end on a ONE-GPU box (what scripts/mgpu_on_one_gpu.sh does by hand): `bench.py --gpus N`
    spawns its own N ranks, they share GPU 0, torch.distributed runs on gloo instead of RCCL (SIMKA_BENCH_BACKEND).  Both
    decompositions (partition shards + one all-reduce; sample shards + spectrum exchange + one all-reduce) must report the job
    totals and the distance-matrix checksum of the 1-rank run."""
    import json
    import subprocess
    import sys

    def run(n):
        env = dict(os.environ, SIMKA_BENCH_BACKEND="gloo")
        r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", str(n), "--workload", "c2", "--reads", "200000", "--steps", "1",
                            "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-two-streams"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one, many = run(1), run(ranks)
    assert one["n_gpus"] == 1 and many["n_gpus"] == ranks
    assert set(many["decompositions"]) == {"sample", "partition"}
    for d in many["decompositions"].values():
        assert d["matrix_checksum"] == one["config"]["matrix_checksum"]
    for key in ("distinct_kmers", "solid_kmers", "kmer_occurrences"):
        assert many["config"][key] == one["config"][key]


def test_first_contact_script_dry_run_over_gloo(gpu_required):
    """scripts/first_contact.sh -- the staged first run on a multi-GPU node (2-rank all-reduce and all-to-all of the C ABI, then bench.py
    --gpus 2 with either decomposition, each under a timeout, the first failing stage named) -- as a DRY RUN on this one-GPU box: the two
    ranks share GPU 0 and gloo carries the collectives; launch, rendezvous, buffers and checks are those of the real run."""
    import subprocess
    env = dict(os.environ, SIMKA_BENCH_BACKEND="gloo", FIRST_CONTACT_TIMEOUT="400")
    r = subprocess.run(["bash", os.path.join(ROOT_DIR, "scripts", "first_contact.sh"), "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "all stages passed" in r.stdout, r.stdout[-3000:]
    assert "allreduce ok on 2 ranks" in r.stdout and "alltoallv ok on 2 ranks" in r.stdout


@pytest.mark.parametrize("how", ["raise", "hang"])
def test_bench_keeps_the_line_of_one_decomposition_when_the_other_fails(gpu_required, how):
    """First-contact safety of `bench.py --gpus N` (no multi-GPU box has run the exchange yet): the second decomposition failing --
    an exception, or a hang cut by the watchdog -- still leaves ONE bench line, from the decomposition that ran, with the failure named."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SIMKA_BENCH_BACKEND="gloo", SIMKA_BENCH_FAIL_MODE="sample", SIMKA_BENCH_FAIL_HOW=how, SIMKA_BENCH_MODE_TIMEOUT="20")
    r = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--workload", "c2", "--reads", "100000", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-two-streams"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and set(d["decompositions"]) == {"partition"} and d["value"] > 0
    assert "sample" in json.dumps(d["decomposition_failures"])


def _stats_via_host_and_device(files_per_sample, k, amin=1):
    """the same samples once through the host parser (read_sequences + simka_pack_read) and once through simka_ingest_*"""
    import simka_amd
    from simka_amd import api
    n = len(files_per_sample)
    out = []
    for mode in ("host", "device"):
        with simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, simple_dist=True) as ctx:
            for s, texts in enumerate(files_per_sample):
                if mode == "device":
                    r = ctx.ingest_text(s, texts)
                    assert r is not None, "sample %d was flagged irregular" % s
                else:
                    seqs = []
                    for t in texts:
                        import tempfile
                        with tempfile.NamedTemporaryFile(suffix=".txt") as f:
                            f.write(t); f.flush()
                            got = list(api.read_sequences(f.name))
                        if not got:
                            break
                        seqs += got
                    packed, offsets, nb, nin = api.pack_reads(seqs)
                    ctx.count_sample(s, packed, nb, len(offsets) - 1, offsets=offsets, nb_input_reads=nin)
            totals = [ctx.sample_totals(s) for s in range(n)]
            ctx.merge()
            out.append((totals, ctx.stats().flat.copy()))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 33, 51, 55])
def test_device_ingest_equals_host_parse(gpu_required, k):
    """simka_ingest_* (FASTA / FASTQ text parsed on the GPU, simka_ingest.hip) against the host parser, bit for bit: multi-line FASTA
    sequences (a fragment continues over the line break), N and other letters (end a fragment), lower case, CR LF, a header without
    a sequence, no newline at the end of the file, several files per sample (the second one appended at an odd base offset), FASTQ.
    k = 21: one-word k-mers; 33 / 51: two words on the partitioned pipeline; 55: the sort-based path."""
    rng = np.random.default_rng(5)

    def seq(n, with_n=False):
        s = bytearray(rng.choice(list(b"ACGT"), size=n).tolist())
        if with_n:
            for p in rng.integers(0, n, size=max(1, n // 40)):
                s[int(p)] = rng.choice(list(b"NnRYx-"))
        return bytes(s)

    genome = seq(4000)

    def read(ln, with_n=False):
        st = int(rng.integers(0, len(genome) - ln))
        s = bytearray(genome[st:st + ln])
        if with_n:
            for p in rng.integers(0, ln, size=2):
                s[int(p)] = ord("N")
        return bytes(s)

    def fasta(reads, width=0, eol=b"\n", final_eol=True):
        out = bytearray()
        for i, r in enumerate(reads):
            out += b">read%d some text ACGT" % i + eol
            if width:
                for p in range(0, len(r), width):
                    out += r[p:p + width] + eol
            elif r:
                out += r + eol
        if not final_eol and out.endswith(eol):
            out = out[:-len(eol)]
        return bytes(out)

    def fastq(reads, eol=b"\n"):
        return b"".join(b"@r%d\n" % i + r + eol + b"+" + eol + b"@" * len(r) + eol for i, r in enumerate(reads))       # ('@' in the qualities)

    a = [read(int(rng.integers(60, 160)), with_n=(i % 5 == 0)) for i in range(400)]
    b = [read(int(rng.integers(30, 200)), with_n=(i % 7 == 0)) for i in range(300)]
    files = [
        [fasta(a)],                                                     # two-line FASTA
        [fasta(a, width=37), fasta(b, width=61)],                       # multi-line, two files: the second starts at an odd base offset
        [fasta([x.lower() for x in b], eol=b"\r\n")],                   # lower case, CR LF
        [fasta(a[:50] + [b""] + a[50:120], final_eol=False)],           # a header without sequence; no newline at the end
        [fastq(b), fastq(a[:77], eol=b"\r\n")],                         # FASTQ, two files
        [fasta(a, width=1000) + b"\n\n"],                               # blank lines at the end only
        [fasta(b, width=53, eol=b"\r\r\n")],                             # two CRs before the LF, multi-line: every trailing CR goes, the fragment continues
    ]
    (th, fh), (td, fd) = _stats_via_host_and_device(files, k)
    assert th == td
    assert np.array_equal(fh, fd)
    assert all(t["K_occ"] > 0 and t["nb_reads"] > 0 for t in th)


@pytest.mark.gpu
def test_device_ingest_flags_what_it_does_not_parse(gpu_required):
    import simka_amd
    with simka_amd.SimkaContext(4, kmer_size=15, abundance_min=1) as ctx:
        assert ctx.ingest_text(0, [b">a\nACGTACGTACGTACGTACGT\n\n>b\nACGTTTGACCAGTAGCAT\n"]) is None          # a blank line inside the file
        assert ctx.ingest_text(0, [b"@r\nACGTACGTACGTACGTACGT\nACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIII\n"]) is None     # multi-line FASTQ
        assert ctx.ingest_text(0, [b"ACGTACGTAGCTAGCATGCAT\n>x\nACGT\n"]) is None                               # sequence before any header
        assert ctx.ingest_text(0, [b">ok\nACGTACGTACGTACGTACGTAAA\n"]) == (23, 1)                              # ... and the sample can still be counted


@pytest.mark.gpu
def test_device_ingest_reproduces_the_goldens(gpu_required, golden_dir, tmp_path):
    """C1 through the device-side parser: the example's FASTA files as raw text -> simka_ingest_* -> the reference's golden CSVs."""
    import simka_amd
    from simka_amd import api
    samples = api.parse_input_file(os.path.join(golden_dir, "example", "simka_input.txt"))
    k, amin = 21, 2
    with simka_amd.SimkaContext(len(samples), kmer_size=k, abundance_min=amin, simple_dist=True, complex_dist=True) as ctx:
        for s, smp in enumerate(samples):
            assert ctx.ingest_text(s, [open(f, "rb").read() for f in smp["files"]]) is not None
        ctx.merge()
        st = ctx.stats()
    out = str(tmp_path / "res")
    st.write_matrices(out, [s["id"] for s in samples], gz=True)
    truth = os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin))
    compared = 0
    for gzf in glob.glob(os.path.join(out, "*.csv.gz")):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), ref
            compared += 1
    assert compared == 20


@pytest.mark.gpu
def test_cli_device_parse_equals_host_parse(gpu_required, tmp_path):
    """The `simka` driver parses plain-text inputs on the GPU by default (simka_ingest_*) and on the host with -host-parse; an input
    the device parser flags (a blank line inside a file) silently takes the host parser.  Same CSV bytes either way."""
    rng = np.random.default_rng(3)
    genome = bytes(rng.choice(list(b"ACGT"), size=5000).tolist())

    def reads(n, seed):
        r = np.random.default_rng(seed)
        out = []
        for i in range(n):
            ln = int(r.integers(50, 150)); st = int(r.integers(0, len(genome) - ln))
            s = bytearray(genome[st:st + ln])
            if i % 6 == 0:
                s[int(r.integers(0, ln))] = ord("N")
            out.append(bytes(s))
        return out

    def fasta(name, rs, width=0, blank_at=None):
        with open(str(tmp_path / name), "wb") as f:
            for i, s in enumerate(rs):
                f.write(b">r%d\n" % i)
                if width:
                    for p in range(0, len(s), width):
                        f.write(s[p:p + width] + b"\n")
                else:
                    f.write(s + b"\n")
                if blank_at == i:
                    f.write(b"\n")
    fasta("a.fa", reads(300, 1)); fasta("b.fa", reads(200, 2), width=40); fasta("c.fa", reads(250, 3), blank_at=17)
    with open(str(tmp_path / "d.fq"), "wb") as f:
        for i, s in enumerate(reads(220, 4)):
            f.write(b"@q%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))
    (tmp_path / "in.txt").write_text("A: a.fa\nB: a.fa , b.fa\nC: c.fa\nD: d.fq ; b.fa\n")
    args = ["-in", str(tmp_path / "in.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", "19", "-abundance-min", "1"]
    dev = _run_cli(args, str(tmp_path / "o1"))
    host = _run_cli(args + ["-host-parse"], str(tmp_path / "o2"))
    assert sorted(dev) == sorted(host) and len(dev) >= 15
    for name in host:
        assert dev[name] == host[name], name


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,gz", [(700, False), (5000, False), (900, True), (100000, True)])
def test_cli_device_parse_in_pieces(gpu_required, tmp_path, chunk, gz):
    """Large inputs reach the device-side parser in pieces cut at record boundaries (-ingest-chunk, 1 GiB by default): with pieces of a
    few hundred bytes every cut rule is exercised -- multi-line FASTA (cut before a header), FASTQ (cut on a line count that is a
    multiple of four, '@' in the qualities), several files per sample.  Same CSV bytes as the host parser."""
    rng = np.random.default_rng(8)
    genome = bytes(rng.choice(list(b"ACGT"), size=4000).tolist())

    def reads(n, seed):
        r = np.random.default_rng(seed)
        return [genome[st:st + ln] for st, ln in ((int(r.integers(0, 3700)), int(r.integers(40, 260))) for _ in range(n))]
    with open(str(tmp_path / "a.fa"), "wb") as f:
        for i, s in enumerate(reads(120, 1)):
            f.write(b">r%d\n" % i + b"".join(s[p:p + 60] + b"\n" for p in range(0, len(s), 60)))
    with open(str(tmp_path / "b.fq"), "wb") as f:
        for i, s in enumerate(reads(150, 2)):
            f.write(b"@q%d\n%s\n+\n%s\n" % (i, s, b"@>" * (len(s) // 2) + b"I" * (len(s) % 2)))
    with open(str(tmp_path / "c.fa"), "wb") as f:
        for i, s in enumerate(reads(90, 3)):
            f.write(b">c%d\n%s\n" % (i, s))
    ext = ""
    if gz:      # the same files gzipped (b.fq as two concatenated members): the loader threads inflate them, the GPU parses the text -- in pieces as well
        ext = ".gz"
        for name in ("a.fa", "b.fq", "c.fa"):
            data = open(str(tmp_path / name), "rb").read()
            with open(str(tmp_path / (name + ".gz")), "wb") as f:
                if name == "b.fq":
                    half = data.index(b"\n@q75\n") + 1
                    f.write(gzip.compress(data[:half]) + gzip.compress(data[half:]))
                else:
                    f.write(gzip.compress(data))
    (tmp_path / "in.txt").write_text("A: a.fa%s\nB: b.fq%s\nC: c.fa%s , a.fa%s\nD: b.fq%s ; c.fa%s\n" % ((ext,) * 6))
    args = ["-in", str(tmp_path / "in.txt"), "-out-tmp", str(tmp_path / "tmp"), "-simple-dist", "-kmer-size", "17", "-abundance-min", "1"]
    log = []
    dev = _run_cli(args + ["-ingest-chunk", str(chunk)], str(tmp_path / "o1"), log)
    host = _run_cli(args + ["-host-parse"], str(tmp_path / "o2"))
    import re
    m = re.search(r"ingest: (\d+) samples parsed on the GPU \((\d+) pieces of text\), (\d+) on the host", log[0])
    assert m and int(m.group(1)) == 4 and int(m.group(3)) == 0 and int(m.group(2)) > (12 if chunk < 10000 else 5), log[0][-600:]      # really in pieces, really on the GPU
    assert sorted(dev) == sorted(host) and len(dev) >= 15
    for name in host:
        assert dev[name] == host[name], name
