#!/usr/bin/env python3
"""CPU simulation (numpy) of minimizer partitioning on the synthetic generator: partition load balance and
super-k-mer run lengths for candidate (m, hash) choices.  Design aid for the count pipeline, not a test."""
import sys, time
import numpy as np
from scipy.ndimage import minimum_filter1d
sys.path.insert(0, "/root/repo")
from simka_amd import synth

def codes_of(R, L, sample):
    g = synth.genome_len_for(R_FULL, L)      # genome size of the FULL workload's coverage model scaled below
    return None

def run(R, L, k, m, P_log2, nmax, sample=0, cov_reads=None, seed_mul=0x9E3779B1):
    g = synth.genome_len_for(cov_reads or R, L)
    pool, gw = synth.genome_pool_cpu(g)
    ids, cdf = synth.sample_profile(sample)
    pk = synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(sample))
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((pk[:, None] >> shifts) & np.uint64(3)).reshape(-1)[: R * L].astype(np.uint32).reshape(R, L)
    # forward / reverse-complement m-mers at every position of every read
    nm = L - m + 1
    f = np.zeros((R, nm), dtype=np.uint32)
    r = np.zeros((R, nm), dtype=np.uint32)
    for i in range(m):
        f |= codes[:, i:i + nm] << np.uint32(2 * i)
        r |= (codes[:, i:i + nm] ^ np.uint32(2)) << np.uint32(2 * (m - 1 - i))
    c = np.minimum(f, r)
    mask = np.uint32((1 << (2 * m)) - 1)
    with np.errstate(over="ignore"):
        h = (c * np.uint32(seed_mul)) & mask
        h ^= h >> np.uint32(m)                      # fold the well-mixed top bits down
        h = (h * np.uint32(0x85EBCA6B)) & mask
    w = k - m + 1
    nk = L - k + 1
    mn = minimum_filter1d(h, size=w, axis=1, origin=-(w // 2), mode="nearest")[:, :nk]   # min over h[q .. q+w-1]
    # check on a few
    q = 5
    assert np.array_equal(mn[:, q], h[:, q:q + w].min(axis=1))
    with np.errstate(over="ignore"):
        pid = ((mn.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - P_log2)).astype(np.uint32)
    loads = np.bincount(pid.reshape(-1), minlength=1 << P_log2)
    # runs: consecutive k-mers of a read with the same pid, capped at nmax
    start = np.ones((R, nk), dtype=bool)
    start[:, 1:] = pid[:, 1:] != pid[:, :-1]
    nruns_nat = int(start.sum())
    # cap: run of length len -> ceil(len/nmax) records
    idx = np.flatnonzero(start.reshape(-1))
    lens = np.diff(np.append(idx, R * nk))
    # runs do not cross reads because start[:,0] = True
    nrec = int(np.ceil(lens / nmax).sum())
    K = R * nk
    print("k=%d m=%d w=%d P=2^%d mean load %.0f | CV %.3f  max/mean %.2f  p99.9/mean %.2f  p0.1/mean %.2f | k-mers per natural run %.2f, per record (cap %d) %.2f, bytes/kmer %.2f"
          % (k, m, w, P_log2, loads.mean(), loads.std() / loads.mean(), loads.max() / loads.mean(),
             np.percentile(loads, 99.9) / loads.mean(), np.percentile(loads, 0.1) / loads.mean(), K / nruns_nat, nmax, K / nrec, 16.0 * nrec / K))
    return loads

if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    L = 150
    k = 31
    # C3 has 10M reads per sample at ~3072 k-mers per partition (P = 2^19): scale P with R so the mean load stays ~3000,
    # and keep C3's genome size (coverage model of the full workload would make genomes tiny at small R: use cov_reads=R*10?)
    for m in (9, 11, 12, 13):
        for mean_target in (3072, 12288):
            P_log2 = int(round(np.log2(R * (L - k + 1) / mean_target)))
            t = time.time()
            run(R, L, k, m, P_log2, 22)
