"""Seeded synthetic metagenomes: a pool of random genomes; each sample draws
reads from 16 of them with log-normal abundances, both strands, 1% substitutions, fixed length.

The read generator is integer-only (SplitMix64 counters), so the numpy code below and the HIP
kernels k_synth_genomes / k_synth_reads (simka_amd/csrc/simka_kernels.hip) produce identical bits.
"""
import math

import numpy as np

POOL_SEED = 0x51A4A
NB_GENOMES = 64
NB_SEL = 16
ERR_THRESHOLD16 = 655          # 655/65536 = 1.0 % substitutions
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_ERRKEY = np.uint64(0xA5A5A5A5A5A5A5A5)
_GKEY = np.uint64(0xD1B54A32D192ED03)


def rng(key, ctr):
    """simka_rng of simka_device.h on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        key = np.asarray(key, dtype=np.uint64)
        ctr = np.asarray(ctr, dtype=np.uint64)
        z = key + (ctr + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def genome_len_for(nb_reads, read_len):
    """g = R*L/(16*20): ~20x mean coverage of a sample's 16 genomes."""
    return max(int(nb_reads) * int(read_len) // (NB_SEL * 20), 4 * int(read_len))


def sample_seed(sample_index):
    return 1000 + int(sample_index)


def sample_profile(sample_index, nb_genomes=NB_GENOMES, nb_sel=NB_SEL):
    """(genome ids uint32[nb_sel], cdf uint32[nb_sel]): random subset + log-normal(sigma=1) weights."""
    seed = np.uint64(sample_seed(sample_index) * 7919 + 17)
    ids = list(range(nb_genomes))
    for i in range(nb_sel):                       # partial Fisher-Yates
        j = i + int(rng(seed, i)) % (nb_genomes - i)
        ids[i], ids[j] = ids[j], ids[i]
    w = []
    for i in range(nb_sel):                       # Box-Muller on two uniform 53-bit draws
        u1 = (int(rng(seed, 1000 + 2 * i)) >> 11) / float(1 << 53)
        u2 = (int(rng(seed, 1001 + 2 * i)) >> 11) / float(1 << 53)
        z = math.sqrt(-2.0 * math.log(max(u1, 1e-300))) * math.cos(2.0 * math.pi * u2)
        w.append(math.exp(z))
    tot = sum(w)
    acc = 0.0
    cdf = []
    for x in w:
        acc += x
        cdf.append(min(int(acc / tot * 4294967296.0), 0xFFFFFFFF))
    cdf[-1] = 0xFFFFFFFF
    return np.array(ids[:nb_sel], dtype=np.uint32), np.array(cdf, dtype=np.uint32)


def genome_pool_cpu(genome_len, nb_genomes=NB_GENOMES, seed=POOL_SEED):
    """uint64 words [nb_genomes * genome_words], 32 bases per word."""
    gw = (genome_len + 31) // 32
    g = np.repeat(np.arange(nb_genomes, dtype=np.uint64), gw)
    w = np.tile(np.arange(gw, dtype=np.uint64), nb_genomes)
    with np.errstate(over="ignore"):
        return rng(np.uint64(seed) ^ (g * _GKEY), w), gw


def reads_cpu(nb_reads, read_len, pool, genome_words, genome_len, genome_ids, cdf, seed, err_thr=ERR_THRESHOLD16):
    """2-bit packed reads, bit-identical to k_synth_reads."""
    L = int(read_len)
    nb_bases = int(nb_reads) * L
    b = np.arange(nb_bases, dtype=np.uint64)
    r = b // np.uint64(L)
    i = b - r * np.uint64(L)
    seed = np.uint64(seed)
    rr = np.arange(nb_reads, dtype=np.uint64)
    h0 = rng(seed, np.uint64(2) * rr)
    h1 = rng(seed, np.uint64(2) * rr + np.uint64(1))
    u = (h0 >> np.uint64(32)).astype(np.uint64)
    sel = np.minimum(np.searchsorted(cdf.astype(np.uint64), u, side="right"), len(cdf) - 1)
    gbase = genome_ids.astype(np.uint64)[sel] * np.uint64(genome_words)
    with np.errstate(over="ignore"):
        start = ((h0 & np.uint64(0xFFFFFFFF)) * np.uint64(genome_len - L + 1)) >> np.uint64(32)
    strand = (h1 & np.uint64(1)).astype(np.uint64)
    st, sb, ss = start[r.astype(np.int64)], gbase[r.astype(np.int64)], strand[r.astype(np.int64)]
    gp = np.where(ss == 1, st + (np.uint64(L - 1) - i), st + i)
    c = (pool[(sb + (gp >> np.uint64(5))).astype(np.int64)] >> ((gp & np.uint64(31)) * np.uint64(2))) & np.uint64(3)
    c = np.where(ss == 1, c ^ np.uint64(2), c)
    he = rng(seed ^ _ERRKEY, b)
    err = (he & np.uint64(0xFFFF)) < np.uint64(err_thr)
    sub = (c + np.uint64(1) + ((he >> np.uint64(16)) % np.uint64(3))) & np.uint64(3)
    c = np.where(err, sub, c)
    nw = (nb_bases + 31) // 32
    pad = nw * 32 - nb_bases
    c = np.concatenate([c, np.zeros(pad, dtype=np.uint64)]).reshape(nw, 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    return np.bitwise_or.reduce(c << shifts, axis=1)


_ASCII = np.frombuffer(b"ACTG", dtype=np.uint8)


def unpack_ascii(packed, nb_bases):
    """packed 2-bit words -> uint8 ASCII array (code A0 C1 T2 G3)."""
    packed = np.asarray(packed, dtype=np.uint64)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((packed[:, None] >> shifts) & np.uint64(3)).reshape(-1)[:nb_bases]
    return _ASCII[codes.astype(np.int64)]
