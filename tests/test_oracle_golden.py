"""The oracle is only trusted because it reproduces the reference's own golden vectors:
tests/golden/truth == /root/reference/tests/truth (80 CSVs, byte-for-byte as tests/simple_test.py:29-68),
plus the integer known answers of SURVEY.md Appendix A.6 and the partition invariance that
tests/simple_test.py:125-133 checks."""
import glob
import os

import numpy as np
import pytest

CONFIGS = [(21, 0), (21, 2), (31, 0), (31, 2)]


def _oracle(oracle_mod, golden_dir, k, amin, **kw):
    o = oracle_mod.Oracle()
    o.load_input(os.path.join(golden_dir, "example", "simka_input.txt"))
    o.run(k, amin, simple=True, complex_=True, **kw)
    return o


@pytest.mark.parametrize("k,amin", CONFIGS)
def test_oracle_reproduces_golden_csv(oracle_mod, golden_dir, tmp_path, k, amin):
    o = _oracle(oracle_mod, golden_dir, k, amin)
    out = str(tmp_path / "o")
    o.write_matrices(out, gz=False)
    truth = sorted(glob.glob(os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin), "*.csv")))
    assert len(truth) == 20
    for ref in truth:
        with open(ref, "rb") as f, open(os.path.join(out, os.path.basename(ref)), "rb") as g:
            assert f.read() == g.read(), os.path.basename(ref)


def test_input_grammar(oracle_mod, golden_dir):
    """ref: src/core/SimkaAlgorithm.cpp:245-351 -- 5 samples, E = A,A ; B,B (2 paired parts, 4 files)."""
    o = oracle_mod.Oracle()
    o.load_input(os.path.join(golden_dir, "example", "simka_input.txt"))
    assert o.ids() == ["A", "B", "C", "D", "E"]
    assert [len(o.files(i)) for i in range(5)] == [1, 1, 1, 2, 4]
    assert [o.nb_paired(i) for i in range(5)] == [1, 1, 1, 2, 2]
    assert all(os.path.isabs(f) and os.path.exists(f) for i in range(5) for f in o.files(i))


# per-sample (K_occ, D_all) and (D, N, Q)
A6 = {
    (21, 0): ([(8910, 4320), (9180, 8100), (8820, 8460), (13140, 11790), (36180, 8100)],
              [(4320, 8910, 18630), (8100, 9180, 11520), (8460, 8820, 9720), (11790, 13140, 17640), (8100, 36180, 212040)], (12600, 12420)),
    (21, 2): ([(8910, 4320), (9180, 8100), (8820, 8460), (13140, 11790), (36180, 8100)],
              [(4320, 8910, 18630), (990, 2070, 4410), (270, 630, 1530), (990, 2340, 6840), (8100, 36180, 212040)], (8100, 4320)),
    (31, 0): ([(7920, 3840), (8160, 7200), (7840, 7520), (11680, 10480), (32160, 7200)],
              [(3840, 7920, 16560), (7200, 8160, 10240), (7520, 7840, 8640), (10480, 11680, 15680), (7200, 32160, 188480)], (11200, 11040)),
    (31, 2): ([(7920, 3840), (8160, 7200), (7840, 7520), (11680, 10480), (32160, 7200)],
              [(3840, 7920, 16560), (880, 1840, 3920), (240, 560, 1360), (880, 2080, 6080), (7200, 32160, 188480)], (7200, 3840)),
}
# pair AB: a ; bc ; S_ij/S_ji ; chord ; hell
A6_AB = {(21, 0): (4320, 5400, 8910, 5400, 11430, 5400), (21, 2): (990, 2070, 2250, 2070, 4770, 2070),
         (31, 0): (3840, 4800, 7920, 4800, 10160, 4800), (31, 2): (880, 1840, 2000, 1840, 4240, 1840)}


@pytest.mark.parametrize("k,amin", CONFIGS)
def test_oracle_integer_known_answers(oracle_mod, golden_dir, k, amin):
    o = _oracle(oracle_mod, golden_dir, k, amin)
    t = o.totals()
    pre, post, glob_ = A6[(k, amin)]
    assert [(int(a), int(b)) for a, b in zip(t["K_occ"], t["D_all"])] == pre
    assert [(int(a), int(b), int(c)) for a, b, c in zip(t["D"], t["N"], t["Q"])] == post
    assert o.global_counts() == glob_
    S = o.acc("S")
    ab = (int(o.acc("a")[0, 1]), int(o.acc("bc")[0, 1]), int(S[0, 1]), int(S[1, 0]), int(o.acc("chord")[0, 1]), int(o.acc("hell")[0, 1]))
    assert ab == A6_AB[(k, amin)]
    # identities of Appendix A.6
    D = t["D"].astype(np.int64)
    a = o.acc("a").astype(np.int64)
    iu = np.triu_indices(5, 1)
    assert np.array_equal(o.acc("kul")[iu], o.acc("bc")[iu])
    assert np.array_equal(o.acc("canb")[iu].astype(np.int64), (D[:, None] + D[None, :] - 2 * a)[iu])


@pytest.mark.parametrize("nparts,threads", [(7, 1), (16, 4)])
def test_oracle_partition_invariance(oracle_mod, golden_dir, nparts, threads):
    """ref: tests/simple_test.py:125-133 -- a different partition / job count must give the same matrices."""
    a = _oracle(oracle_mod, golden_dir, 31, 2)
    b = _oracle(oracle_mod, golden_dir, 31, 2, nparts=nparts, threads=threads)
    for name in ("S", "a", "bc", "chord", "hell", "kul", "whit", "canb"):
        assert np.array_equal(a.acc(name), b.acc(name)), name
    for w in range(len(a.matrix_names())):
        assert np.array_equal(a.matrix(w), b.matrix(w))


def test_oracle_edge_cases(oracle_mod):
    """Unpinned by the goldens; gatb-conventional behaviour: non-ACGT windows skipped,
    case-insensitive, reads shorter than k and empty samples contribute nothing."""
    o = oracle_mod.Oracle()
    reads = [b"ACGTNACGTACGTA", b"acgtacgtacgta", b"ACG", b""]
    cat = np.frombuffer(b"".join(reads), dtype=np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    o.add_sample_ascii("x", cat, off)
    o.add_sample_ascii("empty", np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    o.run(5, 1, simple=True, complex_=True)
    t = o.totals()
    # read 1: windows of 5 without N: "ACGTA CGTAC GTACG TACGT ACGTA" (5) ; read 2: 13-5+1 = 9 ; read 3/4: none
    assert int(t["K_occ"][0]) == 5 + 9
    assert int(t["K_occ"][1]) == 0 and int(t["D"][1]) == 0
    assert np.all(o.matrix(4)[0, 1] == 1.0)      # jaccard with an empty sample: guard -> 1


# ---- row a2: SimkaInputIterator / SimkaSequenceFilter, traced by hand from ref: src/core/SimkaCommons.hpp:159-436 -------------
def _write_reads(path, lens, letters=b"ACGT", seed=0):
    """FASTA with reads of the given lengths (distinct pseudo-random content); returns the lengths"""
    r = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i, ln in enumerate(lens):
            f.write(b">r%d\n%s\n" % (i, bytes(r.choice(list(letters), size=ln).tolist())))
    return list(lens)


def _kocc(lens, k):
    return sum(max(0, ln - k + 1) for ln in lens)


def test_input_iterator_known_answers(oracle_mod, tmp_path):
    k = 11
    f1 = _write_reads(str(tmp_path / "f1.fa"), [30, 31], seed=1)
    f2 = _write_reads(str(tmp_path / "f2.fa"), [40, 41, 42, 43, 44], seed=2)
    g1 = _write_reads(str(tmp_path / "g1.fa"), [50, 51, 52, 53, 54], seed=3)
    open(str(tmp_path / "empty.fa"), "wb").close()
    (tmp_path / "in.txt").write_text(
        "S1: f2.fa\n"                      # one part, one file
        "S2: f1.fa , f2.fa\n"              # one part, two files: the counter runs across them
        "S3: f2.fa ; g1.fa\n"              # two paired parts
        "S4: f1.fa , f2.fa ; g1.fa\n"      # unequal parts: files-per-part = 3 / 2 = 1 -> part 0 = f1, part 1 = f2, g1 never read
        "S5: empty.fa , f2.fa\n")          # an empty first file ends the sample (first(): _isDone = _ref->isDone())

    def run(m, **pol):
        o = oracle_mod.Oracle()
        o.load_input(str(tmp_path / "in.txt"))
        o.set_read_policy(max_reads=m, **pol)
        o.run(k, 1)
        t = o.totals()
        return [int(x) for x in t["nb_reads"]], [int(x) for x in t["K_occ"]]

    # all reads
    nr, ko = run(0)
    assert nr == [5, 7, 10, 7, 0]
    assert ko == [_kocc(f2, k), _kocc(f1 + f2, k), _kocc(f2 + g1, k), _kocc(f1 + f2, k), 0]
    # -max-reads 3: m reads per part; a file switch delivers one read the counter does not see (S2: f1r1 f1r2 | f2r1 f2r2 -> c=2, f2r3 -> c=3 stop)
    nr, ko = run(3)
    assert nr == [3, 4, 6, 5, 0]
    assert ko[0] == _kocc(f2[:3], k)
    assert ko[1] == _kocc(f1 + f2[:2], k)
    assert ko[2] == _kocc(f2[:3] + g1[:3], k)
    assert ko[3] == _kocc(f1 + f2[:3], k)        # part 0 = f1 (2 reads < m), part 1 = f2 (3 reads)
    # -min-read-size 42 with -max-reads 2: only passing reads are delivered and counted
    nr, ko = run(2, min_read_size=42)
    assert nr[0] == 2 and ko[0] == _kocc([42, 43], k)
    assert nr[2] == 4 and ko[2] == _kocc([42, 43, 50, 51], k)
    # what -max-reads 0 resolves to: (min + mean) / 2 of the reads in ALL listed files / paired parts: per sample 5, 7, 5, 6, 5
    o = oracle_mod.Oracle()
    o.load_input(str(tmp_path / "in.txt"))
    assert o.auto_max_reads() == (5 + (5 + 7 + 5 + 6 + 5) // 5) // 2


def test_shannon_filter_known_answers(oracle_mod, tmp_path):
    """getShannonIndex (ref: src/core/SimkaCommons.hpp:388-432): float frequencies of A / C / T / G / N (every other letter counts
    as A), |sum f log2 f|; a read passes when index >= -min-shannon-index."""
    k = 5
    reads = [b"ACGT" * 10,           # 2.0
             b"AC" * 20,             # 1.0
             b"A" * 40,              # 0.0
             b"AACC" * 5 + b"GGTT" * 5,   # 2.0
             b"A" * 30 + b"C" * 10]  # 0.811
    with open(str(tmp_path / "s.fa"), "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">%d\n%s\n" % (i, s))
    (tmp_path / "in.txt").write_text("S: s.fa\n")

    def nreads(th):
        o = oracle_mod.Oracle()
        o.load_input(str(tmp_path / "in.txt"))
        o.set_read_policy(min_shannon=th)
        o.run(k, 1)
        return int(o.totals()["nb_reads"][0])

    assert nreads(0) == 5
    assert nreads(0.5) == 4          # the homopolymer goes
    assert nreads(0.9) == 3
    assert nreads(1.0) == 3          # index 1.0 >= 1.0 passes
    assert nreads(1.5) == 2
    assert nreads(2.0) == 2


def test_driver_read_policy_vs_literal_iterator(oracle_mod, tmp_path):
    """Row f2 as a real cross-check, on CPU: the `simka` driver states the read policies as derived loops (for_counted_reads in
    simka_cli.cpp: first read of a file free, counter across the files of a part, read m + 1 dropped, a file that delivers nothing
    ends the sample, files-per-part = files / parts), the oracle keeps the literal restatement of SimkaInputIterator
    (ref: src/core/SimkaCommons.hpp:159-314).  `simka -parse-only` (no GPU needed) reports the reads / bases it would count per
    sample; the oracle counts them at k = 1 (K_occ = ACGT bases).  Edge cases: empty first file, a file whose every read is filtered,
    -max-reads reached exactly at a file boundary, more paired parts than files, unequal parts, FASTQ."""
    import subprocess
    from simka_amd import build as b
    b.build()
    a1 = _write_reads(str(tmp_path / "a1.fa"), [60 + i for i in range(20)], seed=1)
    a2 = _write_reads(str(tmp_path / "a2.fa"), [90 + i for i in range(12)], seed=2)
    _write_reads(str(tmp_path / "short.fa"), [20, 21, 22, 23], seed=3)          # every read below -min-read-size 50
    _write_reads(str(tmp_path / "c1.fa"), [70, 25, 71, 26, 72, 73, 27, 74], seed=4)    # passing and filtered reads interleaved
    open(str(tmp_path / "empty.fa"), "wb").close()
    with open(str(tmp_path / "q1.fq"), "wb") as f:
        rq = np.random.default_rng(9)
        for i, ln in enumerate(a2[:7]):
            sq = bytes(rq.choice(list(b"ACGT"), size=ln).tolist())
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, sq, b"I" * ln))
    (tmp_path / "in.txt").write_text(
        "E1: empty.fa , a1.fa\n"                # an empty first file ends the sample
        "E2: a1.fa , short.fa , a2.fa\n"        # with -min-read-size 50 the middle file delivers nothing: the sample ends after a1
        "E3: a1.fa , a2.fa\n"                   # -max-reads 19 / 20 / 21: the limit falls on the boundary between the files
        "E4: a1.fa ; a2.fa ; c1.fa\n"           # three parts of one file
        "E5: a1.fa , a2.fa ; c1.fa\n"           # unequal parts: 3 / 2 = 1 file per part -> a1 | a2
        "E6: c1.fa , q1.fq ; a2.fa , a1.fa\n"   # two files per part, FASTQ inside
        "E7: c1.fa\n")
    exe = b.CLI_PATH

    def driver(m, mrs):
        r = subprocess.run([exe, "-parse-only", "-in", str(tmp_path / "in.txt"), "-out-tmp", str(tmp_path / "tmp"), "-max-reads", str(m if m else -1),
                            "-min-read-size", str(mrs), "-verbose", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-1500:]
        out = {}
        for line in r.stdout.splitlines():
            if line.startswith("sample "):
                sid, rest = line[len("sample "):].split(": ")
                out[sid] = (int(rest.split()[0]), int(rest.split()[2]))
        return out

    def literal(m, mrs):
        o = oracle_mod.Oracle()
        o.load_input(str(tmp_path / "in.txt"))
        o.set_read_policy(max_reads=m, min_read_size=mrs)
        o.run(1, 1)
        t = o.totals()
        return {sid: (int(t["nb_reads"][i]), int(t["K_occ"][i])) for i, sid in enumerate(o.ids())}

    seen = set()
    for m in (0, 1, 2, 5, 19, 20, 21, 31, 32, 33, 1000):
        for mrs in (0, 50):
            got, ref = driver(m, mrs), literal(m, mrs)
            assert got == ref, (m, mrs, got, ref)
            seen.add((m, mrs, tuple(sorted(got.items()))))
    ref = literal(0, 50)
    assert ref["E1"] == (0, 0) and ref["E2"][0] == 20 and ref["E7"][0] == 5       # the edge cases are exercised, not vacuous
    assert literal(20, 0)["E3"][0] == 21 and literal(19, 0)["E3"][0] == 19
    assert len(seen) > 12


@pytest.mark.parametrize("k", [32, 33, 47])
def test_oracle_wide_kmers_partition_invariance(oracle_mod, golden_dir, k):
    """k >= 32 (no golden vectors exist): at least the oracle's own partitioning / threading must not change its answer (its ranges are
    sorted on the low bits of 128-bit k-mers; the 64-bit sort path of k <= 31 must not be taken for them)."""
    a = oracle_mod.Oracle(); a.load_input(os.path.join(golden_dir, "example", "simka_input.txt")); a.run(k, 2, simple=True, complex_=True)
    b = oracle_mod.Oracle(); b.load_input(os.path.join(golden_dir, "example", "simka_input.txt")); b.run(k, 2, simple=True, complex_=True, nparts=64, threads=4)
    for name in ("S", "a", "bc", "chord", "hell", "whit"):
        assert np.array_equal(a.acc(name), b.acc(name)), name
    assert int(a.acc("a").sum()) > 0


@pytest.mark.parametrize("k", [31, 63, 64, 65, 96, 127])
def test_oracle_kmers_of_any_width_vs_python_strings(oracle_mod, k):
    """The oracle's k-mer extraction and counting at every word width (one word, __int128, and -- k >= 64 -- whole four-word k-mers
    ranked in a dictionary) against a pure-Python restatement on STRINGS: canonical = min(s, reverse complement) as text, counts in a
    dict.  Three small samples, reads of varying length with N letters, abundance-min 2: per-sample totals and the default
    accumulators (shared abundance, shared distinct k-mers, Bray-Curtis numerator) must agree exactly.
    (k = 64..127 is the reference's Kmer<span=96/128>, ref: CMakeLists.txt:66-71.)"""
    from collections import Counter
    rng = np.random.default_rng(100 + k)
    genome = "".join(rng.choice(list("ACGT"), size=1500))
    comp = str.maketrans("ACGT", "TGCA")
    samples = []
    for s in range(3):
        reads = []
        for _ in range(int(rng.integers(25, 40))):
            ln = int(rng.integers(k - 5, 3 * k))
            st = int(rng.integers(0, len(genome) - ln))
            r = list(genome[st:st + ln])
            if rng.random() < 0.3 and ln:
                r[int(rng.integers(0, ln))] = "N"
            r = "".join(r)
            if rng.random() < 0.5:
                r = r.translate(comp)[::-1]
            reads.append(r)
        samples.append(reads)
    amin = 2
    spectra, occ, dall = [], [], []
    for reads in samples:
        c = Counter()
        for r in reads:
            for i in range(len(r) - k + 1):
                w = r[i:i + k]
                if "N" in w:
                    continue
                rc = w.translate(comp)[::-1]
                c[min(w, rc)] += 1
        occ.append(sum(c.values())); dall.append(len(c))
        spectra.append({km: n for km, n in c.items() if n >= amin})
    o = oracle_mod.Oracle()
    for s, reads in enumerate(samples):
        o.add_sample_ascii("S%d" % s, np.frombuffer("".join(reads).encode(), dtype=np.uint8), np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64))
    o.run(k, amin, simple=True, complex_=False, nparts=3, threads=2)
    t = o.totals()
    assert [int(x) for x in t["K_occ"]] == occ and [int(x) for x in t["D_all"]] == dall
    assert [int(x) for x in t["D"]] == [len(sp) for sp in spectra]
    assert [int(x) for x in t["N"]] == [sum(sp.values()) for sp in spectra]
    assert [int(x) for x in t["Q"]] == [sum(v * v for v in sp.values()) for sp in spectra]
    S, a, bc = o.acc("S"), o.acc("a"), o.acc("bc")
    for i in range(3):
        for j in range(i + 1, 3):
            both = set(spectra[i]) & set(spectra[j])
            assert int(a[i, j]) == len(both)
            assert int(S[i, j]) == sum(spectra[i][x] for x in both) and int(S[j, i]) == sum(spectra[j][x] for x in both)
            assert int(bc[i, j]) == sum(min(spectra[i][x], spectra[j][x]) for x in both)
    assert sum(occ) > 0 and any(len(sp) for sp in spectra)
