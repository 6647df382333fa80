"""BASELINE.json configs[4] at its STATED depth on ONE MI355X: 500 samples x 5M 150 bp reads, k = 31, -simple-dist -complex-dist.

94 GB of packed reads + ~235 GB of solid spectra do not fit 288 GB at once, so the job runs the way the library shards it across GPUs --
as S partition shards (shard s keeps the minimizer partitions p % S == s), here one after the other on the same device:
  * every sample is GENERATED on the device into one reused buffer (simka_synth_reads) right before it is counted: no read ever touches
    the host, no more than one sample of reads is resident;
  * -complex-dist needs the samples' totals N_i before any merge, and a shard's spectra cannot stay resident while the next
    shard counts: shards 0 .. S-2 are counted once for their totals and once more for their merge, the last shard once (2 S - 1 passes
    over the samples);
  * each shard merges with the global totals uploaded (simka_totals_upload) and the heads are added on the host -- the ONE all-reduce of
    the multi-GPU decomposition.
Size-independent properties checked on the result (tests/test_gpu_parity.py::test_full_size_size_independent_properties' set, minus the
ones that need a second full run unless --verify-shards is given): every occurrence counted (sum K_occ = n R (L - k + 1)); the bounds
a <= min(D_i, D_j), bc <= min(S_ij, S_ji), S_ij <= N_i; a sample fed twice (the last slot repeats sample 0) at distance zero from its twin and
seen identically by everyone else; finite, symmetric matrices with a zero diagonal; --verify-shards S2: the whole job again as S2 shards
with another partition count, bit-identical integer accumulators.
usage: python scripts/c5_full_depth.py [--reads 5000000] [--samples 500] [--shards 3] [--verify-shards 0] [--out profiles/r06_c5_full_depth.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer-size", type=int, default=31)
    ap.add_argument("--shards", type=int, default=3)
    ap.add_argument("--verify-shards", type=int, default=0, help="run the job a second time with this many shards (and one partition bit less) and compare")
    ap.add_argument("--verify-log2-partitions", type=int, default=17)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import simka_amd
    from simka_amd import synth
    lib = simka_amd.load_library()
    dev = torch.device("cuda:0")
    n, R, L, k = args.samples, args.reads, args.read_len, args.kmer_size
    g = synth.genome_len_for(R, L)
    gw = (g + 31) // 32
    pool = torch.empty(synth.NB_GENOMES * gw, dtype=torch.int64, device=dev)
    assert lib.simka_synth_genomes(None, pool.data_ptr(), synth.NB_GENOMES, gw, synth.POOL_SEED) == 0
    nw = (R * L + 31) // 32
    buf = [torch.zeros(nw + 2, dtype=torch.int64, device=dev) for _ in range(2)]       # (two: the next sample is generated while the kernels of this one run)
    prof = [synth.sample_profile(s if s < n - 1 else 0) for s in range(n)]            # the last slot repeats sample 0
    d_prof = [(torch.from_numpy(ids.astype(np.int32)).to(dev), torch.from_numpy(cdf.view(np.int32)).to(dev)) for ids, cdf in prof]
    seeds = [synth.sample_seed(s if s < n - 1 else 0) for s in range(n)]
    torch.cuda.synchronize()
    fr, tot = torch.cuda.mem_get_info()
    occ = R * (L - k + 1)
    timing = {}

    def count_all(ctx, tag):
        t0 = time.perf_counter()
        for s in range(n):
            b = buf[s & 1]
            assert lib.simka_synth_reads(None, b.data_ptr(), R, L, pool.data_ptr(), gw, g, d_prof[s][0].data_ptr(), d_prof[s][1].data_ptr(),
                                         synth.NB_SEL, seeds[s], synth.ERR_THRESHOLD16) == 0
            torch.cuda.synchronize()                       # (the generator runs on the null stream; the context's lanes do not wait for it)
            ctx.count_sample(s, b.data_ptr(), R * L, R, fixed_len=L, on_device=True)
            if s & 1:
                ctx.sync()                                 # both buffers are free again
        ctx.sync()
        timing[tag] = time.perf_counter() - t0

    def run(S, log2_partitions=0):
        """-> (flat head summed over the shards with the global totals rows, per-phase seconds)"""
        lay = simka_amd.api.stats_layout(n, simka_amd.DIST_SIMPLE | simka_amd.DIST_COMPLEX)
        head = lay["head"]
        kw = dict(kmer_size=k, abundance_min=2, simple_dist=True, complex_dist=True, max_kmers_per_sample=occ, shard_count=S, log2_partitions=log2_partitions)
        tsum = np.zeros(5 * n, dtype=np.uint64)
        tparts = {}
        total = None
        arena = {}
        order = list(range(S - 1)) + [S - 1] + list(range(S - 1))      # totals of shards 0..S-2, then the last shard (totals + merge), then 0..S-2 again
        for step, s_ in enumerate(order):
            final = step >= S - 1
            t_ctx = time.perf_counter()
            with simka_amd.SimkaContext(n, shard_index=s_, **kw) as ctx:
                count_all(ctx, "S%d shard %d %s count" % (S, s_, "merge-pass" if final else "totals-pass"))
                if not final or s_ == S - 1:
                    tparts[s_] = ctx.totals_download()
                if s_ == S - 1:
                    for v in tparts.values():
                        tsum += v
                if final:
                    ctx.totals_upload(tsum)
                    t0 = time.perf_counter()
                    ctx.merge()
                    st = ctx.stats()
                    timing["S%d shard %d merge + download" % (S, s_)] = time.perf_counter() - t0
                    total = st.flat[:head].copy() if total is None else total + st.flat[:head]
                    try:
                        arena[s_] = ctx.arena_info()
                    except Exception:
                        pass
            timing["S%d shard %d context (create .. destroy)" % (S, s_)] = timing.get("S%d shard %d context (create .. destroy)" % (S, s_), 0.0) + time.perf_counter() - t_ctx
        flat = np.zeros(lay["total"], dtype=np.uint64)
        flat[:head] = total
        flat[lay["tot0"]: lay["tot0"] + 5 * n] = tsum
        return simka_amd.Stats(n, simka_amd.DIST_SIMPLE | simka_amd.DIST_COMPLEX, flat), arena

    t_all = time.perf_counter()
    st, arena = run(args.shards)
    wall = time.perf_counter() - t_all
    ps, pr = st.per_sample(), st.pairs()
    checks = {}
    checks["every occurrence counted (sum K_occ == n R (L - k + 1))"] = bool(int(ps["K_occ"].sum()) == n * occ)
    iu = np.triu_indices(n, 1)
    D, N = ps["D"], ps["N"]
    checks["a <= min(D_i, D_j)"] = bool(np.all(pr["a"] <= np.minimum(D[iu[0]], D[iu[1]])))
    checks["bc <= min(S_ij, S_ji)"] = bool(np.all(pr["bc"] <= np.minimum(pr["S_ij"], pr["S_ji"])))
    checks["S_ij <= N_i and S_ji <= N_j"] = bool(np.all(pr["S_ij"] <= N[iu[0]]) and np.all(pr["S_ji"] <= N[iu[1]]))
    cell = n - 2                                               # pair (0, n - 1) in i < j order
    checks["the twin of sample 0: a = D, bc = S_ij = S_ji = N, chord = Q, canberra = 0"] = bool(
        pr["a"][cell] == D[0] and pr["bc"][cell] == N[0] and pr["S_ij"][cell] == N[0] and pr["S_ji"][cell] == N[0] and pr["chord"][cell] == ps["Q"][0] and pr["canb"][cell] == 0)
    m = st.matrices()
    # (two cells follow the REFERENCE's formulas into their corner cases for a sample fed twice: chord = sqrt(2 - 2 Q / (sqrt(Q) sqrt(Q))) may round
    #  below zero -> NaN, and Jensen-Shannon returns 1 when the KL sum is exactly 0 -- ref: src/core/SimkaDistance.cpp:1001-1007; the twin
    #  cell is therefore judged on its accumulators above and excluded from the finiteness check)
    def clean(x):
        y = x.copy(); y[0, n - 1] = 0; y[n - 1, 0] = 0
        return y
    bad = [nm for nm, x in m.items() if not (np.all(np.isfinite(clean(x))) and np.all(np.diag(x) == 0) and (nm.endswith("_asym") or np.array_equal(clean(x), clean(x).T)))]
    checks["21 matrices finite, symmetric, zero diagonal"] = bool(len(m) == 21 and not bad)
    if bad:
        print("matrices failing finite / symmetric / zero diagonal:", bad, file=sys.stderr)
    zero_or_nan = lambda v: bool(np.isnan(v) or abs(float(v)) < 1e-6)
    checks["twin at distance 0 (bray-curtis, jaccard, hellinger; chord 0 or NaN; KL sum exactly 0) and seen identically by the other samples"] = bool(
        all(abs(float(m[nm][0, n - 1])) < 1e-6 for nm in ("mat_abundance_braycurtis", "mat_presenceAbsence_jaccard", "mat_abundance_hellinger"))
        and zero_or_nan(m["mat_abundance_chord"][0, n - 1]) and float(pr["kl"][cell]) == 0.0
        and all(np.array_equal(m[nm][0, 1:n - 1], m[nm][n - 1, 1:n - 1]) for nm in ("mat_abundance_braycurtis", "mat_presenceAbsence_jaccard", "mat_abundance_chord", "mat_abundance_jensenshannon")))
    if args.verify_shards:
        st2, _ = run(args.verify_shards, log2_partitions=args.verify_log2_partitions)
        lay = st.layout
        kl0 = lay["acc0"] + 7 * lay["nb_pairs"]
        same = np.array_equal(st.flat[:kl0], st2.flat[:kl0]) and np.array_equal(st.flat[kl0 + lay["nb_pairs"]: lay["derived"]], st2.flat[kl0 + lay["nb_pairs"]: lay["derived"]])
        klf = np.max(np.abs(st.flat[kl0: kl0 + lay["nb_pairs"]].view(np.int64) - st2.flat[kl0: kl0 + lay["nb_pairs"]].view(np.int64)))
        checks["%d shards with another partition count give the same integer accumulators as %d shards (KL fixed point within %d ulp of 2^-60)" % (args.verify_shards, args.shards, int(klf))] = bool(same and klf <= 64 * max(args.shards, args.verify_shards))
    K_dist = float(ps["D_all"].sum())
    count_s = sum(v for kname, v in timing.items() if kname.endswith("count") and kname.startswith("S%d " % args.shards))
    merge_s = sum(v for kname, v in timing.items() if "merge + download" in kname and kname.startswith("S%d " % args.shards))
    out = {
        "workload": "BASELINE configs[4] at its stated depth on one MI355X: %d samples x %d x %d bp reads, k = %d, -simple-dist -complex-dist, abundance-min 2" % (n, R, L, k),
        "decomposition": "%d partition shards one after the other on one device, samples generated on the device one at a time, %d passes over the samples" % (args.shards, 2 * args.shards - 1),
        "wall_clock_s": wall, "count_passes_s": count_s, "merges_s": merge_s,
        "kmer_occurrences": float(n) * occ, "distinct_kmers": K_dist, "solid_kmers": float(D.sum()), "pair_updates": float(pr["a"].sum()),
        "distinct_kmers_per_s": K_dist / wall, "hbm_free_at_start_GB": fr / 1e9, "hbm_total_GB": tot / 1e9,
        "arena_records_per_shard": {str(k_): v for k_, v in arena.items()},
        "phases_s": timing, "checks": checks, "all_checks_pass": bool(all(checks.values())),
        "matrix_checksum": __import__("hashlib").sha1(b"".join(np.ascontiguousarray(m[x]).tobytes() for x in sorted(m))).hexdigest()[:16],
    }
    print(json.dumps(out, indent=1))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)
    return 0 if out["all_checks_pass"] else 1


if __name__ == "__main__":
    sys.exit(main())
