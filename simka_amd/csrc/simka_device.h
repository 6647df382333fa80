// simka_device.h -- key arithmetic shared by the HIP kernels and the host driver.
//
// Everything here is result-neutral with respect to the reference: distances
// depend only on each sample's canonical k-mer multiset, so the 2-bit code, the key
// permutation and the partition function are free design choices.  The choices:
//   * code A=0 C=1 T=2 G=3, complement = code^2 (ref: src/core/SimkaCommons.hpp:400-411)
//   * canonical k-mer = min(forward, reverse-complement) as 2k-bit integers
//     (gatb Kmer<span>::ModelCanonical, used at ref: src/minikc/MiniKC.hpp:152-158)
//   * key = bijective mix of the canonical k-mer on W=2k bits; its TOP bits select the
//     partition (where the reference's Repartitor maps a minimizer to a partition,
//     ref: src/minikc/MiniKC.hpp:252-253), the next bits the sub-range used by the merge.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(SIMKA_EMU)
#define SIMKA_HD __host__ __device__ __forceinline__
#else
#define SIMKA_HD inline
#endif

struct SimkaKeyCfg {
    uint32_t k;            // k-mer size (1..31)
    uint32_t W;            // 2k
    uint64_t mask;         // 2^W - 1
    uint32_t xs;           // xor-shift distance of the mix
    uint32_t l1, l2;       // log2 #level-1 / #level-2 partitions
    uint32_t pb;           // l1 + l2: log2 #partitions
    uint32_t t;            // log2 #sub-ranges per partition (merge granularity)
    uint32_t shard_index, shard_count;
};

#define SIMKA_MIX_M1 0xff51afd7ed558ccdULL
#define SIMKA_EMPTY_KEY 0xffffffffffffffffULL   // never a key: keys are < 2^62

// Bijection on W-bit integers: an odd multiply mod 2^W and a xor-shift-right are both invertible.
// Only the TOP bits of the key route it (partition, sub-range), and the top bits of a product depend on every input bit
// (multiplicative hashing); the xor-shift folds the high half into the low bits, which the LDS tables hash again
// (simka_slot_hash / the 32-bit table hash).  One 64-bit multiply: the scan kernel is instruction-bound, a second
// multiply round costs 7 % of k_scan and buys no measurable balance (measured on C2 / C3).
SIMKA_HD uint64_t simka_mix(uint64_t x, uint64_t mask, uint32_t xs) {
    x = (x * SIMKA_MIX_M1) & mask;
    x ^= x >> xs;
    return x;
}

SIMKA_HD uint32_t simka_key_l1(uint64_t key, const SimkaKeyCfg &c) { return (uint32_t)(key >> (c.W - c.l1)); }
SIMKA_HD uint32_t simka_key_l2(uint64_t key, const SimkaKeyCfg &c) {
    return (uint32_t)(key >> (c.W - c.pb)) & ((1u << c.l2) - 1u);
}
SIMKA_HD uint32_t simka_key_part(uint64_t key, const SimkaKeyCfg &c) { return (uint32_t)(key >> (c.W - c.pb)); }
SIMKA_HD uint32_t simka_key_sub(uint64_t key, const SimkaKeyCfg &c) {
    return (uint32_t)(key >> (c.W - c.pb - c.t)) & ((1u << c.t) - 1u);
}
// slot hash for the LDS tables: top bits of a 64-bit multiply see every key bit
SIMKA_HD uint32_t simka_slot_hash(uint64_t key) { return (uint32_t)((key * 0xd6e8feb86659fd93ULL) >> 40); }

// shard ownership of a level-1 bucket: b1 % shard_count == shard_index (mask when the count is a power of two --
// an integer division per k-mer costs ~30 instructions in the scan kernels)
SIMKA_HD bool simka_owns_l1(uint32_t b1, const SimkaKeyCfg &c) {
    if (c.shard_count == 1u) return true;
    if ((c.shard_count & (c.shard_count - 1u)) == 0u) return (b1 & (c.shard_count - 1u)) == c.shard_index;
    return (b1 % c.shard_count) == c.shard_index;
}

// rank of an OWNED level-1 bucket among the buckets of its shard (b1 = shard_index + rank * shard_count), and of an owned
// partition among the shard's partitions: level-1 buckets and level-2 regions are laid out by these, so a shard (or one pass
// over a sample that is counted in several passes) allocates only its share
SIMKA_HD uint32_t simka_bucket_rank(uint32_t b1, const SimkaKeyCfg &c) {
    if (c.shard_count == 1u) return b1;
    if ((c.shard_count & (c.shard_count - 1u)) == 0u) return b1 >> (31u - __builtin_clz(c.shard_count));
    return b1 / c.shard_count;
}
SIMKA_HD uint64_t simka_region_index(uint32_t part, const SimkaKeyCfg &c) {
    if (c.shard_count == 1u) return part;
    return ((uint64_t)simka_bucket_rank(part >> c.l2, c) << c.l2) | (part & ((1u << c.l2) - 1u));
}

// floor(sqrt(x)) exactly, x < 2^64.  The reference adds sqrt((double)(ci*cj)) to a u64, i.e.
// floor of the correctly-rounded double sqrt (ref: src/core/SimkaAlgorithm.hpp:397), which equals
// this for x < 2^52.
SIMKA_HD uint64_t simka_isqrt(uint64_t x) {
    uint64_t r = (uint64_t)sqrt((double)x);
    while (r * r > x) r--;
    while ((r + 1) * (r + 1) <= x) r++;
    return r;
}

// index of the unordered pair (i<j) among N samples
SIMKA_HD uint64_t simka_pair_index(uint64_t i, uint64_t j, uint64_t n) { return i * n - i * (i + 1) / 2 + (j - i - 1); }

// counter-based generator of the synthetic data set (SplitMix64 finaliser)
SIMKA_HD uint64_t simka_rng(uint64_t key, uint64_t ctr) {
    uint64_t z = key + (ctr + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
