"""Randomised parity check: GPU path (hash pipeline, or the sort path for k >= 32) against the oracle on random small inputs --
random k, abundance window, sample count, variable read lengths, N letters, lowercase, empty reads, partition geometry.
usage: fuzz_vs_oracle.py [seconds] [seed]      (FUZZ_KS=33,36,51 restricts the k-mer sizes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simka_amd, oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
np.set_printoptions(threshold=100000, linewidth=220)
rng = np.random.default_rng(seed)
t_end = time.time() + budget
KS = [int(x) for x in os.environ.get("FUZZ_KS", "").split(",") if x] or [1, 2, 3, 5, 8, 11, 15, 16, 17, 21, 25, 31, 32, 33, 34, 35, 36, 37, 40, 45, 50, 51, 52, 63, 64, 65, 80, 96, 127]


def kl_edge_cases():
    """The both-present KL sum of (near-)identical samples.  Reference (ref: src/core/SimkaAlgorithm.hpp:437-446,
    src/core/SimkaDistance.cpp:1001-1007): every term p ln(2p/(p+q)) + q ln(2q/(p+q)) is computed in double and added to a long double;
    IDENTICAL samples give terms that are exactly 0, the sum is exactly 0, and the Jensen-Shannon cell is then 1 (`if kl == 0 return 1`,
    a quirk); NEAR-identical samples give terms of either sign around 1e-17, and a negative sum makes sqrt(kl / 2) NaN.
    Here: every term is clamped at 0 and accumulated in 2^-60 fixed point -- identical samples cancel exactly as well (the pair loop and
    the per-entry p ln p use the same logarithm), so that cell is 1 as in the reference; a near-identical pair can never go negative:
    where the reference prints NaN or a value below 1e-8 out of rounding noise, this path prints a value in [0, 1e-8]."""
    rs = np.random.default_rng(7)
    genome = rs.integers(0, 4, size=3000)
    base = [np.frombuffer(b"ACTG", dtype=np.uint8)[genome[st:st + 100]].tobytes() for st in rs.integers(0, 2900, size=600)]
    extra = np.frombuffer(b"ACTG", dtype=np.uint8)[rs.integers(0, 4, size=100)].tobytes()
    for k in (21, 31):
        samples = [base, list(base), base + [base[0]], base + [extra]]       # 1 = 0; 2: one read twice; 3: one foreign read
        ctx = simka_amd.SimkaContext(len(samples), kmer_size=k, abundance_min=1, simple_dist=True, complex_dist=True)
        orc = oracle_lib.Oracle()
        for s_, reads in enumerate(samples):
            packed, off, nb, nfrag = simka_amd.pack_reads(reads)
            ctx.count_sample(s_, packed, nb, len(off) - 1, offsets=off, nb_input_reads=len(reads))
            orc.add_sample_ascii("S%d" % s_, np.frombuffer(b"".join(reads), dtype=np.uint8), np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64))
        ctx.merge(); st = ctx.stats(); ctx.close()
        orc.run(k, 1, simple=True, complex_=True)
        iu = np.triu_indices(len(samples), 1)
        js_ref = orc.matrix(orc.matrix_names().index("mat_abundance_jensenshannon")); js = st.matrices()["mat_abundance_jensenshannon"]
        kl_ref, kl = orc.kl()[iu], st.pairs()["kl"]
        for c, (i, j) in enumerate(zip(*iu)):
            print("k=%d pair (%d,%d): kl reference % .3e  here % .3e | jensen-shannon reference %s  here %s" % (k, i, j, kl_ref[c], kl[c], js_ref[i, j], js[i, j]))
        assert kl[0] == 0.0 and kl_ref[0] == 0.0 and js[0, 1] == 1.0 and js_ref[0, 1] == 1.0, "identical samples: KL exactly 0, Jensen-Shannon 1 (the reference's quirk)"
        assert np.all(kl >= 0.0) and np.all(np.abs(kl - kl_ref) <= 1e-9 * np.abs(kl_ref) + 1e-13)
        assert np.all(np.isfinite(js))


kl_edge_cases()
ncase = 0
while time.time() < t_end:
    k = int(rng.choice(KS))
    n = int(rng.integers(1, 9))
    amin = int(rng.choice([0, 1, 2, 3])); amax = int(rng.choice([999999999, 999999999, 50, 5]))
    simple = bool(rng.integers(0, 2)); cplx = bool(rng.integers(0, 2))
    pb = int(rng.choice([0, 0, 1, 3, 6, 9]))
    genome = rng.integers(0, 4, size=int(rng.integers(200, 4000)))
    samples = []
    for s in range(n):
        reads = []
        for r in range(int(rng.integers(0, 400))):
            L = int(rng.integers(0, 180))
            st = int(rng.integers(0, max(1, len(genome) - L)))
            seq = np.frombuffer(b"ACTG", dtype=np.uint8)[genome[st:st + L]].copy()
            if L and rng.random() < 0.3:            # substitutions, N, lowercase
                for _ in range(int(rng.integers(1, 4))):
                    seq[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"ACGTNnacgtRY", dtype=np.uint8))
            if rng.random() < 0.5:
                comp = {65: 84, 67: 71, 71: 67, 84: 65}
                seq = np.array([comp.get(int(c), int(c)) for c in seq[::-1]], dtype=np.uint8)
            reads.append(seq.tobytes())
        samples.append(reads)
    kw = {}
    if pb and k <= 31:
        kw["log2_partitions"] = min(pb, 2 * k)
    shards = int(rng.choice([1, 1, 2, 3])) if n >= 2 else 1
    if shards > 1:
        # partition shards: every shard sees every read and keeps its level-1 buckets; integer accumulators and totals add up
        cplx = False
        packs = [simka_amd.pack_reads(reads) for reads in samples]
        acc = None; tot_sum = None
        for si in range(shards):
            c = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, abundance_max=amax, simple_dist=simple, shard_index=si, shard_count=shards, **kw)
            for s, (packed, off, nb, nfrag) in enumerate(packs):
                c.count_sample(s, packed, nb, len(off) - 1, offsets=off, nb_input_reads=len(samples[s]))
            tt = [c.sample_totals(i) for i in range(n)]
            c.merge(); stx = c.stats(); c.close()
            pr_ = stx.pairs()
            if acc is None:
                acc = {key: np.array(v, dtype=np.uint64) for key, v in pr_.items() if key in ("S_ij", "S_ji", "a", "bc", "chord", "hell")}
                tot_sum = [dict(t) for t in tt]
            else:
                for key in acc: acc[key] += np.array(pr_[key], dtype=np.uint64)
                for i in range(n):
                    for key in ("K_occ", "D_all", "D", "N", "Q"): tot_sum[i][key] += tt[i][key]
        orc = oracle_lib.Oracle()
        for s, reads in enumerate(samples):
            ascii_ = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, dtype=np.uint8)
            orc.add_sample_ascii("S%d" % s, ascii_, np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64))
        orc.run(k, amin, amax=amax, simple=simple, complex_=False)
        ot = orc.totals(); iu = np.triu_indices(n, 1); S = orc.acc("S")
        tag = "case %d (shards %d): k=%d n=%d amin=%d amax=%d simple=%d pb=%d" % (ncase, shards, k, n, amin, amax, simple, pb)
        for i in range(n):
            for key in ("K_occ", "D_all", "D", "N", "Q"):
                assert int(tot_sum[i][key]) == int(ot[key][i]), (tag, i, key)
        assert np.array_equal(acc["S_ij"], S[iu]) and np.array_equal(acc["S_ji"], S.T[iu]) and np.array_equal(acc["a"], orc.acc("a")[iu]) and np.array_equal(acc["bc"], orc.acc("bc")[iu]), tag
        if simple:
            assert np.array_equal(acc["chord"], orc.acc("chord")[iu]) and np.array_equal(acc["hell"], orc.acc("hell")[iu]), tag
        ncase += 1
        continue
    ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, abundance_max=amax, simple_dist=simple, complex_dist=cplx, **kw)
    orc = oracle_lib.Oracle()
    for s, reads in enumerate(samples):
        packed, off, nb, nfrag = simka_amd.pack_reads(reads)
        ctx.count_sample(s, packed, nb, len(off) - 1, offsets=off, nb_input_reads=len(reads))
        ascii_ = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, dtype=np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
        orc.add_sample_ascii("S%d" % s, ascii_, offs)
    totals = [ctx.sample_totals(i) for i in range(n)]
    ctx.merge(); st = ctx.stats(); ctx.close()
    orc.run(k, amin, amax=amax, simple=simple, complex_=cplx)
    ot = orc.totals()
    tag = "case %d: k=%d n=%d amin=%d amax=%d simple=%d complex=%d pb=%d" % (ncase, k, n, amin, amax, simple, cplx, pb)
    for i, t in enumerate(totals):
        for key in ("nb_reads", "K_occ", "D_all", "D", "N", "Q"):
            assert int(t[key]) == int(ot[key][i]), (tag, i, key, t[key], ot[key][i])
    if n >= 2:
        iu = np.triu_indices(n, 1); pr = st.pairs(); S = orc.acc("S")
        assert np.array_equal(pr["S_ij"], S[iu]) and np.array_equal(pr["S_ji"], S.T[iu]), tag
        assert np.array_equal(pr["a"], orc.acc("a")[iu]) and np.array_equal(pr["bc"], orc.acc("bc")[iu]), tag
        if simple:
            assert np.array_equal(pr["chord"], orc.acc("chord")[iu]) and np.array_equal(pr["hell"], orc.acc("hell")[iu]), tag
        if cplx:
            assert np.array_equal(pr["whit"], orc.acc("whit")[iu]) and np.array_equal(pr["canb"], orc.acc("canb")[iu]), tag
            np.testing.assert_allclose(pr["kl"], orc.kl()[iu], rtol=1e-9, atol=1e-15, err_msg=tag)      # NaN (empty sample) must be NaN in both
        mats = st.matrices()
        # Jensen-Shannon is sqrt(kl / 2), and 1 when kl == 0 exactly (ref quirk): where the KL sum of two near-identical samples is
        # rounding noise (|kl| < 1e-13) the reference's own result is an accident of its operation order -- those cells are skipped
        noisy = np.zeros((n, n), dtype=bool)
        if cplx:
            klm = np.zeros((n, n)); klm[iu] = np.abs(orc.kl()[iu]); noisy = (klm + klm.T) < 1e-13
            np.fill_diagonal(noisy, False)
        for w, name in enumerate(orc.matrix_names()):
            if name in mats and "jensenshannon" in name:
                np.testing.assert_allclose(np.where(noisy, 0, mats[name]), np.where(noisy, 0, orc.matrix(w)), rtol=1e-6, atol=2e-9, err_msg=tag + " " + name)
            elif name in mats:
                # atol: near-identical samples (k = 1..3: a handful of k-mers) give Jensen-Shannon ~1e-7 out of terms that cancel to 1e-13 of
                # their size; in double -- the reference's own d1/d2 are doubles -- that is ill-conditioned (1e-4 relative, 1e-10 absolute)
                np.testing.assert_allclose(mats[name], orc.matrix(w), rtol=1e-6, atol=2e-9, err_msg=tag + " " + name)
    d, sh = orc.global_counts()
    assert (int(st.view.nb_distinct_kmers), int(st.view.nb_shared_kmers)) == (d, sh), (tag, int(st.view.nb_distinct_kmers), int(st.view.nb_shared_kmers), d, sh)
    ncase += 1
print("fuzz ok: %d random cases (seed %d)" % (ncase, seed))
