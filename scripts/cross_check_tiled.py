"""Tiled pair accumulation (more samples than one LDS tile of pair cells): the tile-major kernels (k_tile_major + k_pairs_tm)
against the scan-and-compact kernel k_pairs<TILED> (SIMKA_PAIRS_LEGACY=1, the one the oracle tests pinned first) and against the
sort-based pipeline, on device-generated samples at sizes and sample counts the CPU oracle would take far too long for.  The flat
statistics must be identical.  Sample i is a copy of distinct sample i % D, so groups range from a few samples (D = n) to
hundreds (small D), with and without -complex-dist.
usage: cross_check_tiled.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
for rd in range(rounds):
    n = int(rng.integers(130, 800)); D = int(rng.choice([n, n, max(2, n // 3), 40, 7]))
    R = int(rng.choice([1500, 5000, 20000])); L = int(rng.choice([75, 100, 150]))
    k = int(rng.choice([15, 21, 31, 33])); amin = int(rng.choice([1, 2])); cplx = bool(rng.integers(0, 2)); simple = bool(rng.integers(0, 2))
    wl = dict(n=min(n, D), reads=R, L=L, k=k, amin=amin, simple=simple)
    pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
    flats = []
    variants = [("tile-major", {}), ("legacy", {"SIMKA_PAIRS_LEGACY": "1"})] + ([("sort+tile-major", {"SIMKA_SORT_PATH": "1"})] if k <= 31 and R <= 5000 else [])
    for name, env in variants:
        for key in ("SIMKA_PAIRS_LEGACY", "SIMKA_SORT_PATH"): os.environ.pop(key, None)
        os.environ.update(env)
        with simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, simple_dist=simple, complex_dist=cplx, max_kmers_per_sample=R * (L - k + 1)) as ctx:
            for s in range(n):
                ctx.count_sample(s, reads[s % D].data_ptr(), R * L, R, fixed_len=L, on_device=True)
            ctx.merge()
            flats.append(ctx.stats().flat.copy())
    for key in ("SIMKA_PAIRS_LEGACY", "SIMKA_SORT_PATH"): os.environ.pop(key, None)
    same = all(np.array_equal(flats[0], f) for f in flats[1:])
    print("round %d: n=%d distinct=%d R=%d L=%d k=%d amin=%d simple=%d complex=%d [%s] -> %s (distinct k-mers %d, shared %d)" % (
        rd, n, D, R, L, k, amin, simple, cplx, ", ".join(v[0] for v in variants), "identical" if same else "DIFFERENT", int(flats[0][0]), int(flats[0][1])))
    assert same
print("tiled cross-check ok")
