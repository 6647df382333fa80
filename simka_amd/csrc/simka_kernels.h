// simka_kernels.h -- launch geometry and argument blocks shared by the kernels and the host driver.
#pragma once
#include <stdint.h>
#include "simka_device.h"

// scalars of every dynamic-LDS kernel live in the first SIMKA_LDS_HEAD bytes of the region
#define SIMKA_LDS_HEAD 512

// count side: see simka_skm.hip (SKM_* geometry)
#define K2_SLAB 4096          // arena records reserved per global atomic by a count block (a partition's solid records stay contiguous:
                              // what is left of a slab when the next partition does not fit is lost, so slabs are several partitions long)
#ifndef SIMKA_TARGET_PER_PART
#define SIMKA_TARGET_PER_PART 3072   // sizing: k-mer occurrences per partition (minimizer partitions vary ~3x around it; the fast count
                                     // kernel's 2048-slot table takes ~1500 distinct k-mers, larger partitions go through k_skm_count)
#endif
// K3  k_segment_rows / k_group
#define SIMKA_SEG_BITS 4       // a (sample, partition) segment of the arena is ordered by the top 4 bits of the key (SKM_SORT_BITS of the count kernels)
#define SIMKA_SEG_BLOCKS (1 << SIMKA_SEG_BITS)
#define K3_BLOCK 256          // k_group<256>; k_group<512> (twice the records per round, twice the table) for more than 256 samples
#define K3_CAP 1024           // records hashed per round = entries per span (x2 with 512 threads)
#define K3_TARGET 940         // mean records per sub-range the merge aims for (sub-range bits t; x2 with 512 threads)
#define K3_PRESPLIT 1024      // a sub-range above this is split on one more key bit before hashing (x2 with 512 threads)
#define K3_STACK 72           // refinement stack of k_group: deeper than the 62 key bits
#define K3_HEAD 384           // bytes of block scalars in front of k_group's tables
// dynamic LDS of k_group<GB>: table keys / packed prefix, record counts, group sizes, record samples, record slots, the sample tile, the stack
#define K3_LDS_BYTES(GB_) ((size_t)K3_HEAD + (size_t)(GB_) * K3_UNROLL * 2 * 8 + (size_t)(GB_) * K3_UNROLL * 4 + (size_t)(GB_) * K3_UNROLL * 2 * 2 + (size_t)(GB_) * K3_UNROLL * 2 + (size_t)(GB_) * K3_UNROLL * ((GB_) == K3_BLOCK ? 1 : 2) + (size_t)(GB_) * 8 + ((size_t)(GB_) + 2) * 4 + (size_t)K3_STACK * 8)
#define K3_TABLE 2048         // = 2*K3_CAP slots
#define K3_UNROLL 4           // K3_BLOCK*K3_UNROLL = K3_CAP: a whole sub-range in one batch of independent loads
#define K3_SLAB_ENT 32768     // CSR entries / groups / span slots reserved per global atomic by a k_group block
#define K3_SLAB_GRP 16384
#define K3_SLAB_SPAN 8
// K4  k_pairs
#define K4_BLOCK_BIG 1024
#define K4_BLOCK_SMALL 256

// per-sample totals row (device), additive over shards
#define SIMKA_NB_TOTALS 5
#define SIMKA_TOT_D 0
#define SIMKA_TOT_N 1
#define SIMKA_TOT_Q 2
#define SIMKA_TOT_DALL 3
#define SIMKA_TOT_KOCC 4

// pair accumulators, order in the flat statistics buffer
#define SIMKA_ACC_SIJ 0
#define SIMKA_ACC_SJI 1
#define SIMKA_ACC_A 2
#define SIMKA_ACC_BC 3
#define SIMKA_ACC_CHORD 4      // simple (present iff SIMKA_DIST_SIMPLE)
#define SIMKA_ACC_HELL 5
// complex: two more arrays after the simple ones (index = nacc32 + {0,1}); 64-bit LDS cells
//   WHIT : sum over both-present pairs of  w(ci,cj) - g(ci,Nj) - g(cj,Ni)   (two's complement), + host bias
//   KLFIX: sum over both-present pairs of the KL term, fixed point 2^-52 (two's complement)
#define SIMKA_KL_SCALE 1152921504606846976.0   // 2^60: a cell's both-present KL sum is < 2 ln 2, so the i64 sum stays below 2^61
#define SIMKA_HIST_MAX 1024    // per-sample histogram of solid counts (complex): exact bins below, list above

// device error word bits
#define SIMKA_DEVERR_TABLE_OVERFLOW 1u
#define SIMKA_DEVERR_ARENA_FULL 2u
#define SIMKA_DEVERR_SAMPLE_TOO_BIG 4u
#define SIMKA_DEVERR_GROUP_OVERFLOW 8u
#define SIMKA_DEVERR_CSR_FULL 16u
#define SIMKA_DEVERR_UNORDERED 32u
#define SIMKA_DEVERR_SEGMENT_TOO_BIG 64u   // a (sample, partition) segment holds more than 65535 solid k-mers: the merge index has 16-bit rows

struct SimkaScanArgs {
    const uint64_t *packed;
    uint64_t nb_bases, nb_words;
    const uint64_t *offsets;
    uint64_t nb_reads;
    uint32_t fixed_len;
    const uint32_t *tile_r0;                 // variable-length reads: index of the read holding the first base of every scan tile (k_tile_reads)
};

struct SimkaCountOut {
    unsigned long long *arena_cursor;
    const unsigned long long *sample_base;   // &sample_base[sample]
    unsigned long long arena_cap;
    unsigned long long *solid_keys;
    uint32_t *solid_counts;
    uint32_t *foff, *fcnt;                   // this sample's rows [nparts]
    uint16_t *seg_rows;                      // the merge's index of the arena, [nparts][N][SIMKA_SEG_BLOCKS] (see k_segment_rows), at this sample; NULL: built at merge time
    unsigned long long *seg_abs;             // [nparts][N], at this sample
    unsigned long long *totals;              // [SIMKA_NB_TOTALS][N]
    uint32_t sample, nb_samples;
    uint32_t slab, pad_;                     // arena records a block reserves at a time
    uint32_t *err;
    unsigned long long *phase;               // debug phase timers (SIMKA_PHASE_PROF builds), else NULL
    unsigned long long *hist;                // [N][SIMKA_HIST_MAX] histogram of solid counts, NULL unless complex
    uint32_t *ovf_list;                      // (sample,count) pairs for counts >= SIMKA_HIST_MAX
    unsigned long long *ovf_cursor;
    unsigned long long ovf_cap;
};

struct SimkaMergeIn {
    const unsigned long long *solid_keys;
    const uint32_t *solid_counts;
    const unsigned long long *sample_base;   // [N]
    const uint32_t *foff, *fcnt;             // [N][nparts]
    uint32_t nb_samples;
    uint64_t nparts;
};

struct SimkaSpan {
    unsigned long long ebase, gbase;
    uint32_t nent, ngrp;
    uint32_t maxc, pad;                      // largest count among the span's records (overflow bound of k_pairs)
};

struct SimkaCsrOut {
    unsigned long long *entries;             // (sample<<32 | count)
    uint32_t *groups;                        // (entry offset within span << 16 | size)
    SimkaSpan *spans;
    SimkaSpan *huge;                         // groups shared by more than K3_CAP samples, one span each (k_pairs_global)
    unsigned long long *cursors;             // [0] entries, [1] groups, [2] spans, [3] huge spans
    unsigned long long cap_entries, cap_groups, cap_spans, cap_huge;
    uint32_t span_cap;                       // a span grows up to this many entries (SimkaPairCfg::span_cap)
    unsigned long long *glob;                // [0] nb distinct k-mers, [1] nb shared k-mers
    uint32_t *err;
};

#define SIMKA_SPAN_MAX 4096        // largest span (entries) k_group may build / k_pairs can stage
#define SIMKA_PAIR_TN 256          // k_pairs, complex: LDS table of per-sample N (one tile, or tile I + tile J)

struct SimkaPairCfg {
    uint32_t nb_samples;
    uint32_t tile, ntiles;       // sample tile edge, #tiles
    uint32_t nacc;               // all accumulator arrays: nacc32 + nacc64
    uint32_t nacc32;             // 4 (default) or 6 (simple): u32 LDS cells
    uint32_t nacc64;             // 0 or 2 (complex): u64 LDS cells
    uint32_t simple;             // chord/hell present
    uint32_t ncell, ncell_pad;   // LDS cells per accumulator
    uint32_t span_cap;           // entries per span (multiple of K3_CAP, <= SIMKA_SPAN_MAX)
    uint64_t nb_pairs;           // N(N-1)/2
    const unsigned long long *tot_n;   // [N] per-sample N_i (GLOBAL totals), complex only
};
