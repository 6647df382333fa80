"""Hash pipeline vs sort pipeline on the same device-resident samples, at sizes the CPU oracle would take minutes for: the flat
statistics must be identical.  Samples get adapter-like hot reads and a poly-A stretch so that the spill / exact-redo paths run.
usage: cross_check.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
for rd in range(rounds):
    n = int(rng.integers(2, 9)); R = int(rng.choice([50_000, 120_000, 300_000])); L = int(rng.choice([75, 100, 150]))
    k = int(rng.choice([11, 21, 27, 31])); amin = int(rng.choice([1, 2, 3])); cplx = bool(rng.integers(0, 2)); simple = bool(rng.integers(0, 2))
    wl = dict(n=n, reads=R, L=L, k=k, amin=amin, simple=simple)
    pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
    bytes_per_read = L // 4 if L % 4 == 0 else None
    for s in range(n):
        b = reads[s].view(torch.uint8)
        if bytes_per_read and rng.random() < 0.6:          # hot reads: `copies` copies of one read
            copies = int(rng.integers(2000, R // 4)); src = b[:bytes_per_read].clone()
            b[1000 * bytes_per_read:(1000 + copies) * bytes_per_read] = src.repeat(copies)
        if rng.random() < 0.4:                             # poly-A stretch (code 0): one level-1 bucket overflows
            b[: int(b.numel() * rng.uniform(0.02, 0.2))] = 0
    flats = []
    for sort_path in (False, True):
        if sort_path: os.environ["SIMKA_SORT_PATH"] = "1"
        else: os.environ.pop("SIMKA_SORT_PATH", None)
        t0 = time.time()
        with simka_amd.SimkaContext(n, kmer_size=k, abundance_min=amin, simple_dist=simple, complex_dist=cplx, max_kmers_per_sample=R * (L - k + 1)) as ctx:
            for s in range(n):
                ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
            ctx.merge()
            flats.append(ctx.stats().flat.copy())
        dt = time.time() - t0
    os.environ.pop("SIMKA_SORT_PATH", None)
    same = np.array_equal(flats[0], flats[1])
    print("round %d: n=%d R=%d L=%d k=%d amin=%d simple=%d complex=%d -> %s (distinct %d, shared %d)" % (rd, n, R, L, k, amin, simple, cplx,
          "identical" if same else "DIFFERENT", int(flats[0][0]), int(flats[0][1])))
    assert same
print("cross-check ok")
