// simka_device.h -- key arithmetic shared by the HIP kernels and the host driver.
//
// Everything here is result-neutral with respect to the reference: distances
// depend only on each sample's canonical k-mer multiset, so the 2-bit code, the key
// permutation and the partition function are free design choices.  The choices:
//   * code A=0 C=1 T=2 G=3, complement = code^2 (ref: src/core/SimkaCommons.hpp:400-411)
//   * canonical k-mer = min(forward, reverse-complement) as 2k-bit integers
//     (gatb Kmer<span>::ModelCanonical, used at ref: src/minikc/MiniKC.hpp:152-158)
//   * partition = function of the k-mer's minimizer (simka_skm.hip), as the reference's Repartitor maps a minimizer to a
//     partition (ref: src/minikc/MiniKC.hpp:252-253)
//   * key (what the solid spectra store) = bijective mix of the canonical k-mer on W=2k bits; bits below its top pb bits
//     select the sub-range used by the merge.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SIMKA_HD __host__ __device__ __forceinline__
#else
#define SIMKA_HD inline
#endif

struct SimkaKeyCfg {
    uint32_t k;            // k-mer size (1..31)
    uint32_t W;            // 2k
    uint64_t mask;         // 2^W - 1
    uint32_t xs;           // xor-shift distance of the mix
    uint32_t l1, l2;       // (sort-based wide-k path only)
    uint32_t pb;           // log2 #partitions
    uint32_t t;            // log2 #sub-ranges per partition (merge granularity)
    uint32_t shard_index, shard_count;
};

#define SIMKA_MIX_M1 0xff51afd7ed558ccdULL
#define SIMKA_EMPTY_KEY 0xffffffffffffffffULL   // never a key: keys are < 2^62

// Bijection on W-bit integers: an odd multiply mod 2^W and a xor-shift-right are both invertible.
// Only the TOP bits of the key route it (partition, sub-range), and the top bits of a product depend on every input bit
// (multiplicative hashing); the xor-shift folds the high half into the low bits, which the LDS tables hash again
// (simka_slot_hash / the 32-bit table hash).  One 64-bit multiply: the scan kernel is instruction-bound, a second
// multiply round buys no measurable balance (measured on C2 / C3 in round 1).
SIMKA_HD uint64_t simka_mix(uint64_t x, uint64_t mask, uint32_t xs) {
    x = (x * SIMKA_MIX_M1) & mask;
    x ^= x >> xs;
    return x;
}

// 32-bit hash of a key (= canonical k-mer): slot in the count kernels' LDS tables, and -- its top SIMKA_SEG_BITS bits -- the order of a
// (sample, partition) segment of the arena, hence the merge's sub-range of the key
SIMKA_HD uint32_t simka_key_hash32(uint64_t key) { return (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA6Bu; }
// slot hash for the LDS tables: top bits of a 64-bit multiply see every key bit
SIMKA_HD uint32_t simka_slot_hash(uint64_t key) { return (uint32_t)((key * 0xd6e8feb86659fd93ULL) >> 40); }

// ---- environment knobs -----------------------------------------------------------------------------------------------------
// Two classes.  simka_test_knob(): switches the TEST SUITE needs to force the rarely taken routes of the
// shipped library (fallback kernels, sort paths, passes, table overflows) plus SIMKA_LANES / SIMKA_ARENA_MALLOC / SIMKA_DEBUG_SYNC --
// always compiled in.  simka_exp_knob(): tuning experiments (block counts, fan-outs, span sizes, debug statistics) -- compiled in
// only with -DSIMKA_DEBUG_KNOBS (scripts/build_variant.sh NAME -DSIMKA_DEBUG_KNOBS); the shipped library ignores them.
#include <stdlib.h>
static inline const char *simka_test_knob(const char *name) { return getenv(name); }
static inline const char *simka_exp_knob(const char *name) {
#ifdef SIMKA_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name; return (const char *)0;
#endif
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

// partition shards for 32 <= k <= 63: a canonical k-mer (hi, lo) belongs to shard  mix(hi, lo) * G >> 32  (every context of a sharded
// run scans all reads and keeps its own k-mers)
SIMKA_HD bool simka_wide_owns(unsigned long long hi, unsigned long long lo, uint32_t shard_index, uint32_t shard_count) {
    unsigned long long x = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return (uint32_t)(((x >> 32) * (unsigned long long)shard_count) >> 32) == shard_index;
}
