// simka_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Simka hot path.
//
// The count side (replaces gatb SortingCountAlgorithm + the SimkaCompressedProcessor plugin, ref: src/SimkaCount.cpp:291-297,
// src/minikc/MiniKC.hpp:54-79) lives in simka_skm.hip (super-k-mer pipeline); this file holds the shared block primitives,
// the arena helpers of the count kernels, and the merge side.
//
// Merge side over all samples (replaces SimkaMergeAlgorithm::execute's heap merge and
// SimkaCountProcessorSimple::updateDistance*, ref: src/SimkaMerge.cpp:1164-1326,
// src/core/SimkaAlgorithm.hpp:341-402):
//
//   k_regroup : gather one partition's records from the N samples, order them by sub-range
//   k_group   : per sub-range LDS hash grouping -> CSR groups (k-mer -> [(sample,count)...])
//   k_pairs   : persistent blocks, LDS-privatised pair accumulators, all s(s-1)/2 pairs per group
//   k_pairs_global : the same for k-mers shared by more samples than one span holds (global u64 atomics)
//
// Everything is integer work on HBM / LDS; no MFMA.  Block sizes are multiples of the 64-lane
// wavefront; every global access pattern that carries real traffic is a contiguous run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "simka_device.h"
#include "simka_kernels.h"
#ifndef KG_PIVOTS
#define KG_PIVOTS 1              // k_group: coarse pivots of the sample prefix table in registers (round 6)
#endif

typedef unsigned long long ull;

// --------------------------------------------------------------------------------------------
// block-wide exclusive scan of a u32 array living in LDS (n items, in place); returns the total.
// Caller must have synchronised after the last write to a[].  tmp has BLOCK entries.
// --------------------------------------------------------------------------------------------
// inclusive scan over the 64 lanes of a wave with DPP row shifts / broadcasts (VALU only: no LDS crossbar round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// one value per thread: exclusive prefix over the block in `excl`, block total returned; ONE barrier (tmp: BLOCK/64 words,
// not reused before the caller's next barrier)
template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan1(uint32_t v, uint32_t &excl, uint32_t *tmp) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t NW = BLOCK / 64;
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63u) tmp[wave] = inc;
    __syncthreads();
    uint32_t wpre = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) { const uint32_t t = tmp[w]; if (w < wave) wpre += t; total += t; }
    excl = wpre + inc - v;
    return total;
}

template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t *a, uint32_t n, uint32_t *tmp) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t NW = BLOCK / 64;
    const uint32_t ipt = (n + BLOCK - 1) / BLOCK;
    const uint32_t b = tid * ipt;
    const uint32_t e = (b + ipt < n) ? b + ipt : n;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += a[i];
    const uint32_t v = wave_incl_scan(s);             // inclusive scan across the 64 lanes of the wave (DPP: no LDS permutes)
    if (lane == 63u) tmp[wave] = v;
    __syncthreads();
    uint32_t wpre = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) { const uint32_t t = tmp[w]; if (w < wave) wpre += t; total += t; }
    uint32_t run = wpre + v - s;
    for (uint32_t i = b; i < e; i++) { const uint32_t x = a[i]; a[i] = run; run += x; }
    __syncthreads();
    return total;
}

// k_tile_reads: variable-length reads.  tile_r0[b] = index of the read that holds base b * tile - back (largest r with
// offsets[r] <= that), one thread per tile; k_skm_scan stages the read starts of its tile in LDS from it.  back: how far before
// its first window a tile's first k-mer starts (k >= 36: the minimizer window sits in the middle of the k-mer).
__global__ void __launch_bounds__(256)
k_tile_reads(const uint64_t *offsets, uint64_t nb_reads, uint64_t nb_bases, uint32_t ntiles, uint64_t tile, uint64_t back, uint32_t *tile_r0) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > ntiles) return;
    const uint64_t pos = (uint64_t)b * tile > back ? (uint64_t)b * tile - back : 0ull;
    uint64_t lo = 0, hi = nb_reads;
    if (pos >= nb_bases) { tile_r0[b] = (uint32_t)(nb_reads ? nb_reads - 1 : 0); return; }
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= pos) lo = mid; else hi = mid; }
    tile_r0[b] = (uint32_t)lo;
}

// reserve `ns` arena records for one partition out of the block's private slab (thread 0 only)
// A block reserves arena space in slabs of o.slab records.  What is left of a slab when a partition does not fit is lost, so the
// loss must be bounded for the host's upper bound of the arena cursor (run_count_kernels, SIMKA_SLAB_WASTE_NUM/DEN): a partition of
// at least a quarter of a slab that does not fit takes EXACTLY its records from the global cursor and leaves the open slab alone;
// only a smaller one opens a new slab, losing less than a quarter of the old one -- at most 1/3 on top of the records placed.
__device__ __forceinline__ ull slab_take(ull &slab_pos, ull &slab_end, uint32_t ns, const SimkaCountOut &o, ull sample_base, uint32_t &ok) {
    ull b;
    if (slab_pos + ns > slab_end) {
        if ((ull)ns * 4ull >= (ull)o.slab) {
            b = atomicAdd(o.arena_cursor, (ull)ns);
            if (b + ns > o.arena_cap) { atomicOr(o.err, SIMKA_DEVERR_ARENA_FULL); ok = 0; return 0; }
        } else {
            slab_pos = atomicAdd(o.arena_cursor, (ull)o.slab);
            slab_end = slab_pos + o.slab;
            if (slab_end > o.arena_cap) { atomicOr(o.err, SIMKA_DEVERR_ARENA_FULL); ok = 0; slab_end = slab_pos; return 0; }
            b = slab_pos;
            slab_pos += ns;
        }
    } else {
        b = slab_pos;
        slab_pos += ns;
    }
    if (b - sample_base + ns > 0xffffffffull) { atomicOr(o.err, SIMKA_DEVERR_SAMPLE_TOO_BIG); ok = 0; }
    return b;
}

// (lbins: how many of the SIMKA_HIST_MAX exact bins the kernel keeps in LDS; the rarer counts above go straight to the global histogram)
__device__ __forceinline__ void count_hist(const SimkaCountOut &o, uint32_t *lhist, uint32_t c, uint32_t lbins = SIMKA_HIST_MAX) {
    if (c < lbins) atomicAdd(&lhist[c], 1u);
    else if (c < SIMKA_HIST_MAX) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + c], 1ull);
    else { const ull w = atomicAdd(o.ovf_cursor, 1ull); if (w < o.ovf_cap) { o.ovf_list[2 * w] = o.sample; o.ovf_list[2 * w + 1] = c; } }
}

#ifdef SIMKA_PHASE_PROF
#define PH_DECL ull ph_t = wall_clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i) { const ull n_ = wall_clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; }
#define PH_WAITVM __builtin_amdgcn_s_waitcnt(0x0070);   /* vmcnt(0): separate the load wait from the insert phase */
#define PH_FLUSH if (threadIdx.x == 0 && o.phase) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&o.phase[i_], ph_acc[i_]); }
#else
#define PH_DECL
#define PH(i)
#define PH_WAITVM
#define PH_FLUSH
#endif

// per-partition record totals over all samples (input of the host-side partition scan)
// (part_total[nparts], zeroed by the caller: the largest (sample, partition) segment -- beyond 65535 records the merge index takes 32-bit rows)
__global__ void __launch_bounds__(256)
k_part_totals(const uint32_t *fcnt, uint32_t nb_samples, uint64_t nparts, ull *part_total) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    ull s = 0; uint32_t mx = 0;
    if (p < nparts) {
        for (uint32_t i = 0; i < nb_samples; i++) { const uint32_t c = fcnt[(size_t)i * nparts + p]; s += c; mx = c > mx ? c : mx; }
        part_total[p] = s;
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) { const uint32_t t_ = __shfl_xor(mx, o_, 64); mx = t_ > mx ? t_ : mx; }
    if ((threadIdx.x & 63u) == 0 && mx > 0xffffu) atomicMax(&part_total[nparts], (ull)mx);
}

// --------------------------------------------------------------------------------------------
// K3  k_segment_rows: the merge's view of the arena.  The count kernels leave every (sample, partition) segment ordered by the top
// SIMKA_SEG_BITS bits of the key (their tables are walked in slot order), so a sub-range of a segment is a SLICE of it and the
// N-way merge can read its inputs in place (the reference opens the N sorted files solid/part_p/__p__*.gz side by side,
// ref: src/SimkaMerge.cpp:1082-1103).  One wave per segment: row[p][s] = the 16 exclusive ends of the key-prefix blocks inside the
// segment (u16: a segment has < 65536 records ... else the last entries saturate and the partition is flagged), seg_abs[p][s] = its
// first record in the arena.  Partition-major, so that a merge block reads the N rows of its partition as one run.
// A segment that is not ordered (a spectrum imported from a foreign source) flags SIMKA_DEVERR_UNORDERED.
// --------------------------------------------------------------------------------------------
// ROW32: 32-bit block ends (four uint4 per row) for contexts in which a segment holds more than 65535 records -- a user-set small
// log2_partitions, or a hot low-complexity partition counted by k_skm_count in rounds: the merge then rebuilds the rows of every batch
// here and k_group reads the wide form.
template <bool ROW32>
__global__ void __launch_bounds__(256)
k_segment_rows(SimkaMergeIn in, SimkaKeyCfg cfg, uint64_t part_begin, uint32_t np, ull *seg_abs, uint4 *rows, uint32_t *err) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nseg = (uint64_t)np * in.nb_samples;
    const uint64_t nw = (uint64_t)gridDim.x * 4u;
    for (uint64_t g = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); g < nseg; g += nw) {
        const uint64_t pi = g / in.nb_samples;
        const uint32_t s = (uint32_t)(g - pi * in.nb_samples);
        const uint64_t p = part_begin + pi;
        const uint32_t n = in.fcnt[(size_t)s * in.nparts + p];
        const ull b = in.sample_base[s] + in.foff[(size_t)s * in.nparts + p];
        uint32_t e[SIMKA_SEG_BLOCKS];
#pragma unroll
        for (uint32_t q = 0; q < SIMKA_SEG_BLOCKS; q++) e[q] = 0;
        uint32_t last = 0; bool bad = false;
        if (!ROW32 && n > 0xffffu && lane == 0) atomicOr(err, SIMKA_DEVERR_SEGMENT_TOO_BIG);
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const bool v = i0 + lane < n;
            const uint32_t blk = v ? simka_key_hash32(in.solid_keys[b + i0 + lane]) >> (32u - SIMKA_SEG_BITS) : SIMKA_SEG_BLOCKS;     // (beyond the end: above every block)
            const uint32_t prev = lane ? (uint32_t)__shfl_up((int)blk, 1, 64) : last;
            if (v && prev > blk) bad = true;
            last = (uint32_t)__builtin_amdgcn_readlane((int)blk, 63);
#pragma unroll
            for (uint32_t q = 0; q < SIMKA_SEG_BLOCKS; q++) e[q] += (uint32_t)__popcll(__ballot(blk <= q));
        }
        if (__ballot(bad)) { if (lane == 0) atomicOr(err, SIMKA_DEVERR_UNORDERED); }
        if (lane == 0) {
            seg_abs[g] = b;
            if (ROW32) {
                rows[4 * g] = make_uint4(e[0], e[1], e[2], e[3]); rows[4 * g + 1] = make_uint4(e[4], e[5], e[6], e[7]);
                rows[4 * g + 2] = make_uint4(e[8], e[9], e[10], e[11]); rows[4 * g + 3] = make_uint4(e[12], e[13], e[14], e[15]);
            } else {
                uint4 r0, r1;
                r0.x = e[0] | (e[1] << 16); r0.y = e[2] | (e[3] << 16); r0.z = e[4] | (e[5] << 16); r0.w = e[6] | (e[7] << 16);
                r1.x = e[8] | (e[9] << 16); r1.y = e[10] | (e[11] << 16); r1.z = e[12] | (e[13] << 16); r1.w = e[14] | (e[15] << 16);
                rows[2 * g] = r0; rows[2 * g + 1] = r1;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// K3a  k_group: the N-way merge.  Where the reference pops a min-heap to collect the abundance
// vector of one k-mer (ref: src/SimkaMerge.cpp:1198-1263), a block hashes the records of one
// sub-range into LDS, which groups equal k-mers; groups with >= min_share samples are emitted
// as CSR (the gate of SimkaMergeAlgorithm::insert, ref: src/SimkaMerge.cpp:1307-1326).
// Persistent blocks (3 per CU) walk the sub-ranges; output space (entries / groups / span slots) is
// reserved in slabs, one global atomic per slab instead of three per sub-range.
// --------------------------------------------------------------------------------------------
// GB: threads per block.  256 (records per round GCAP = 1024, table 2048 slots, four blocks per CU) up to 256 samples; 512 (2048 / 4096, two
// blocks per CU) beyond: all samples of C5's 500 are then ONE tile -- the rows stay in registers, a sub-range is gathered once instead
// of once per sample tile and is not split again on key bits (k_group on c5_50: 28.6 -> 15.5 ms)
#ifdef SIMKA_PHASE_PROF
__device__ ull g_group_phase[8];
#define PG_DECL ull pg_t = wall_clock64(), pg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PG(i) { const ull n_ = wall_clock64(); pg_acc[i] += n_ - pg_t; pg_t = n_; }
#define PG_FLUSH if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_group_phase[i_], pg_acc[i_]); }
#else
#define PG_DECL
#define PG(i)
#define PG_FLUSH
#endif
// ROW32: the rows of the batch are 32-bit (k_segment_rows<true>: a segment beyond 65535 records somewhere): read from memory where they are
// needed instead of living in registers -- the rare, slower form.
// the kernel arguments as the kernel-argument segment holds them
struct KGroupArgs { SimkaMergeIn in; const ull *seg_abs; const uint16_t *rows; uint32_t np; SimkaKeyCfg cfg; uint32_t min_share; SimkaCsrOut o; uint32_t pstride; };
template <int GB, bool ROW32>
__global__ void __launch_bounds__(GB)
k_group(SimkaMergeIn in, const ull *seg_abs, const uint16_t *rows, uint32_t np,
        SimkaKeyCfg cfg, uint32_t min_share, SimkaCsrOut o_unused, uint32_t pstride) {
    // The output descriptor (15 pointers and capacities) and the mix constants are used once per round or less.  As ordinary arguments
    // they sit in scalar registers for the whole kernel, the compiler runs out of those (106) and parks values in vector lanes:
    // ~200 v_readlane per wave and round in a kernel that is bound by instruction issue.  They are read from the kernel-argument
    // segment where they are used instead (scalar loads through the constant cache); KG_FRESH keeps the loads inside the round.
    typedef const __attribute__((address_space(4))) KGroupArgs *KGroupArgsP;
    KGroupArgsP ka = (KGroupArgsP)__builtin_amdgcn_kernarg_segment_ptr();
#define KG_FRESH() asm volatile("" : "+s"(ka))
    (void)o_unused;
    // pstride: work item pi of the batch is partition pi * pstride of the index arrays (1: consecutive partitions; G: the partitions
    // p = g + G i a partition shard owns -- the others are empty, and walking them cost as much as grouping the owned ones)
    constexpr int GCAP = GB * K3_UNROLL, GTAB = 2 * GCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_slab = (ull *)smem;                              // [6] ent pos/end, grp pos/end, span pos/end
    ull &s_ebase = *(ull *)(smem + 48);
    ull &s_gbase = *(ull *)(smem + 56);
    uint32_t &s_nrec = *(uint32_t *)(smem + 64);
    uint32_t &s_ovf = *(uint32_t *)(smem + 68);
    uint32_t &s_ndist = *(uint32_t *)(smem + 72);
    uint32_t &s_nshared = *(uint32_t *)(smem + 76);
    int &s_sp = *(int *)(smem + 80);
    uint32_t &s_maxc = *(uint32_t *)(smem + 84);
    uint32_t *tmp = (uint32_t *)(smem + 96);               // [GB/64]
    ull &s_open = *(ull *)(smem + 336);                    // slot of the block's open span (~0: none)
    uint32_t &s_open_nent = *(uint32_t *)(smem + 344);     // its entries / groups / largest count so far
    uint32_t &s_open_ngrp = *(uint32_t *)(smem + 348);
    uint32_t &s_open_maxc = *(uint32_t *)(smem + 352);
    uint32_t &s_soff = *(uint32_t *)(smem + 356);          // entry offset of this round inside its span
    // 31 688 bytes with GB = 256 (K3_LDS_BYTES): FIVE blocks per CU -- a round is a chain of dependent steps, the blocks of a CU are what
    // overlaps it (1 / 2 / 3 / 4 blocks: 31.4 / 17.1 / 12.4 / 10.1 ms on c3_10).  The dispatcher hands out LDS in granules of 1280 bytes
    // (scripts/ubench/lds_occupancy.hip: five blocks are co-resident up to 32 000 bytes each, not the 32 768 the occupancy API computes).
    // What made room: the packed prefix lives where the table keys were (dead once the records are hashed), a record's value is 4 + 1
    // bytes (GB = 256 serves at most 256 samples; 4 + 2 beyond), a stack entry 8 bytes.
    typedef typename std::conditional<GB == K3_BLOCK, uint8_t, uint16_t>::type smp_t;
    ull *tkeys = (ull *)(smem + K3_HEAD);                  // [GTAB] the keys of the round's hash table ...
    uint32_t *gpk = (uint32_t *)tkeys;                     // [GTAB] ... then the packed prefix: entries | groups << 20; later the fill cursor
    uint32_t *rcnt = (uint32_t *)(tkeys + GTAB);           // [GCAP] count of the record ...
    uint16_t *scnt = (uint16_t *)(rcnt + GCAP);            // [GTAB] group size (<= GCAP records per round: 16 bits, added through the 32-bit word)
    uint16_t *rslot = scnt + GTAB;                         // [GCAP]
    smp_t *rsmp = (smp_t *)(rslot + GCAP);                 // [GCAP] ... and its sample (GB = 256: N <= 256; else nb_samples <= 65535)
    // a tile of samples while the records are gathered: first record of the sample's slice in the arena, the slices' exclusive prefix
    ull *sbeg = (ull *)(rsmp + GCAP);                      // [GB]
    uint32_t *spre = (uint32_t *)(sbeg + GB);              // [GB + 2]
    ull *s_stack = (ull *)(spre + GB + 2);                 // [K3_STACK] refinement DFS, one level per key bit: (1 << selector bits) | value

    const uint32_t tid = threadIdx.x;
    const uint32_t free_bits = cfg.W;                      // an over-full sub-range is split on the bits of simka_mix(key), a bijection on W bits: every bit fixed = one k-mer
    if (tid < 6) s_slab[tid] = 0;
    if (tid == 0) { s_ndist = 0; s_nshared = 0; s_open = ~0ull; s_open_nent = 0; s_open_ngrp = 0; s_open_maxc = 0; }
    __syncthreads();

    // Work item = a partition of the batch: its sub-ranges one after the other, so the lines of the N segments that two neighbouring
    // slices share are touched by ONE block back to back.  The rows of the first GB samples live in registers for the whole
    // partition (thread s: sample s), the rows of the NEXT partition are loaded while this one is grouped.
    const uint32_t nsub = 1u << cfg.t, N = in.nb_samples;
    const uint32_t bw = SIMKA_SEG_BLOCKS >> cfg.t;         // key-prefix blocks per sub-range (t <= SIMKA_SEG_BITS)
    const uint4 *rows4 = (const uint4 *)rows;
    // end of key-prefix block i (0 .. 15) of a row held as two uint4; i == ~0u: 0
    auto row_end = [](const uint4 &r0, const uint4 &r1, uint32_t i) -> uint32_t {
        if (i == ~0u) return 0u;
        // (i is wave-uniform: a switch, i.e. scalar branches -- as a select chain the compiler turned the row into a 32-byte scratch array
        //  indexed through a scalar register: 48 bytes of private memory per lane and a scratch load per use)
        uint32_t a;
        switch (i >> 1) {
            case 0: a = r0.x; break; case 1: a = r0.y; break; case 2: a = r0.z; break; case 3: a = r0.w; break;
            case 4: a = r1.x; break; case 5: a = r1.y; break; case 6: a = r1.z; break; default: a = r1.w; break;
        }
        return (i & 1u) ? a >> 16 : a & 0xffffu;
    };
    uint4 rr0 = make_uint4(0, 0, 0, 0), rr1 = rr0, nr0 = rr0, nr1 = rr0; ull rab = 0, nab = 0;       // this partition's row / the next one's (sample tid)
    const uint32_t *rows32 = (const uint32_t *)rows;
    // end of key-prefix block i of the segment (partition pi of the batch, sample s_), whatever the row width
    auto seg_end = [&](uint32_t pi_, uint32_t s_, uint32_t i) -> uint32_t {
        if (i == ~0u) return 0u;
        const size_t g = (size_t)pi_ * pstride * in.nb_samples + s_;
        return ROW32 ? rows32[g * SIMKA_SEG_BLOCKS + i] : (uint32_t)rows[g * SIMKA_SEG_BLOCKS + i];
    };
    // Which partitions a block takes: Q = min(4, #sub-ranges) neighbouring blocks of ONE XCD (blockIdx % 8: its own L2) share a partition,
    // a quarter of its sub-ranges each -- the lines of the N segments that neighbouring slices share are then fetched into that L2 once
    // (with a partition per block, 128 partitions are open per XCD: 23 MB of segments against 4 MB of L2, every line fetched 2-3 times).
    const uint32_t Q = nsub < 4u ? nsub : 4u, jq = nsub / Q;
    const uint32_t xcd = blockIdx.x & 7u, bx = blockIdx.x >> 3, gx = (gridDim.x + 7u - xcd) >> 3;       // this block's rank among the blocks of its XCD
    const uint32_t quarter = bx % Q, nsets = gx / Q, pstep = 8u * nsets;      // (the host launches a multiple of 32 blocks: nsets >= 1, no block left over)
    uint32_t cur_pi = (bx / Q) * 8u + xcd, cur_j = 0;
    if (bx / Q >= nsets) cur_pi = np;
    if (cur_pi < np && tid < N) { const size_t g = (size_t)cur_pi * pstride * N + tid; if (!ROW32) { nr0 = rows4[2 * g]; nr1 = rows4[2 * g + 1]; } nab = seg_abs[g]; }
    // f(key, sample << 32 | count) for every record of the sub-range: a tile of GB samples at a time -- their slices from the
    // rows, an exclusive scan, then one record per thread (the thread finds its sample in the prefix table)
    bool tile_ready = false;       // (uniform) the tables of sample tile 0 are already in LDS (the scan that gave R): the first gather reuses them
    auto for_records = [&](auto &&f) {
        for (uint32_t s0 = 0; s0 < N; s0 += GB) {
            const uint32_t s = s0 + tid;
            uint32_t c = 0; ull b = 0;
            uint32_t tot;
            if (s0 == 0 && tile_ready) { tile_ready = false; tot = spre[GB]; }
            else {
            if (s0 == 0) {
                const uint32_t lo = ROW32 ? (tid < N ? seg_end(cur_pi, tid, cur_j * bw - 1u) : 0u) : row_end(rr0, rr1, cur_j * bw - 1u);      // (cur_j == 0: ~0u)
                const uint32_t hi = ROW32 ? (tid < N ? seg_end(cur_pi, tid, (cur_j + 1u) * bw - 1u) : 0u) : row_end(rr0, rr1, (cur_j + 1u) * bw - 1u);
                c = hi - lo; b = rab + lo;
            } else if (s < N) {
                const size_t g = (size_t)cur_pi * pstride * N + s;
                const uint32_t lo = seg_end(cur_pi, s, cur_j * bw - 1u), hi = seg_end(cur_pi, s, (cur_j + 1u) * bw - 1u);
                c = hi - lo; b = seg_abs[g] + lo;
            }
            __syncthreads();           // (the tables of the tile before are done with)
            sbeg[tid] = b;
            uint32_t excl;
            tot = block_excl_scan1<GB>(c, excl, tmp + 8);
            spre[tid] = excl;
            if (tid == 0) spre[GB] = tot;
            __syncthreads();
            }
            const uint32_t ns = N - s0 < (uint32_t)GB ? N - s0 : (uint32_t)GB;
#if KG_PIVOTS
            // (round 6) up to 128 samples in the tile: the prefixes of samples 16, 32, .. 112 in registers -- a record's group of 16 samples is a
            // count of seven independent compares, the search over the prefix table in LDS four dependent reads instead of seven
            uint32_t pv[7];
            const bool piv = ns <= 128u;
            if (piv) {
#pragma unroll
                for (int q = 0; q < 7; q++) pv[q] = (uint32_t)(q + 1) * 16u < ns ? spre[(q + 1) * 16] : 0xffffffffu;
            }
#endif
            for (uint32_t i0 = tid; i0 < tot; i0 += GB * K3_UNROLL) {
                ull kk[K3_UNROLL], vv[K3_UNROLL];
#pragma unroll
                for (int u = 0; u < K3_UNROLL; u++) {
                    const uint32_t i = i0 + (uint32_t)u * GB;
                    kk[u] = SIMKA_EMPTY_KEY; vv[u] = 0;
                    if (i < tot) {
                        uint32_t lo = 0, hi = ns;          // largest x with spre[x] <= i
#if KG_PIVOTS
                        if (piv) {
                            uint32_t grp = 0;
#pragma unroll
                            for (int q = 0; q < 7; q++) grp += pv[q] <= i ? 1u : 0u;
                            lo = grp * 16u; hi = lo + 16u < ns ? lo + 16u : ns;
                        }
#endif
                        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (spre[mid] <= i) lo = mid; else hi = mid; }
                        const ull at = sbeg[lo] + (i - spre[lo]);
                        kk[u] = in.solid_keys[at]; vv[u] = ((ull)(s0 + lo) << 32) | (ull)in.solid_counts[at];
                    }
                }
#pragma unroll
                for (int u = 0; u < K3_UNROLL; u++) if (kk[u] != SIMKA_EMPTY_KEY) f(kk[u], vv[u]);
            }
            __syncthreads();           // (every record of the tile is in the table before the next tile's rows replace this one's; the sample tile has 3 KB of its own since round 5)
        }
    };
    PG_DECL
    for (; cur_pi < np; cur_pi += pstep) {
        {      // a new partition: its rows arrived while the one before was grouped; the next one's set off
            rr0 = nr0; rr1 = nr1; rab = nab;
            const uint32_t npi = cur_pi + pstep;
            nr0 = make_uint4(0, 0, 0, 0); nr1 = nr0; nab = 0;
            if (npi < np && tid < N) { const size_t g = (size_t)npi * pstride * N + tid; if (!ROW32) { nr0 = rows4[2 * g]; nr1 = rows4[2 * g + 1]; } nab = seg_abs[g]; }
        }
      for (cur_j = quarter * jq; cur_j < (quarter + 1u) * jq; cur_j++) {
        const uint32_t this_j = cur_j;
        KG_FRESH();
        // records of the sub-range over all samples
        uint32_t R = 0;
        if (N <= (uint32_t)GB) {      // one tile of samples: its scan is the gather's scan too
            const uint32_t lo = ROW32 ? (tid < N ? seg_end(cur_pi, tid, this_j * bw - 1u) : 0u) : row_end(rr0, rr1, this_j * bw - 1u);
            const uint32_t c = (ROW32 ? (tid < N ? seg_end(cur_pi, tid, (this_j + 1u) * bw - 1u) : 0u) : row_end(rr0, rr1, (this_j + 1u) * bw - 1u)) - lo;
            __syncthreads();
            sbeg[tid] = rab + lo;
            uint32_t excl;
            R = block_excl_scan1<GB>(c, excl, tmp + 8);
            spre[tid] = excl;
            if (tid == 0) spre[GB] = R;
            __syncthreads();
            tile_ready = true;
        } else {
            uint32_t c = ROW32 ? (tid < N ? seg_end(cur_pi, tid, (this_j + 1u) * bw - 1u) - seg_end(cur_pi, tid, this_j * bw - 1u) : 0u)
                               : row_end(rr0, rr1, (this_j + 1u) * bw - 1u) - row_end(rr0, rr1, this_j * bw - 1u);
            for (uint32_t s = tid + GB; s < N; s += GB) c += seg_end(cur_pi, s, (this_j + 1u) * bw - 1u) - seg_end(cur_pi, s, this_j * bw - 1u);
            __syncthreads();
            uint32_t excl;
            R = block_excl_scan1<GB>(c, excl, tmp + 8);
        }
        PG(0)
        if (R == 0) continue;
        uint32_t e0 = 0;
        while (((R >> e0) > GCAP) && e0 < free_bits) e0++;
        const uint32_t nvals0 = 1u << e0;
        for (uint32_t v0 = 0; v0 < nvals0; v0++) {
            __syncthreads();
            if (tid == 0) { s_sp = 1; s_stack[0] = (1ull << e0) | (ull)v0; }
            while (true) {
                __syncthreads();
                if (s_sp == 0) break;
                const ull top_ = s_stack[s_sp - 1];
                const uint32_t e = 63u - (uint32_t)__clzll((long long)top_);
                const ull val_ = top_ ^ (1ull << e);                  // up to free_bits (<= 62) selector bits
                __syncthreads();
                if (tid == 0) { s_sp--; s_nrec = 0; s_ovf = 0; s_maxc = 0; }
                {
                    ulonglong2 *k2 = (ulonglong2 *)tkeys; uint4 *c4 = (uint4 *)scnt;
                    const ulonglong2 ek = make_ulonglong2(SIMKA_EMPTY_KEY, SIMKA_EMPTY_KEY);
                    const uint4 z = make_uint4(0, 0, 0, 0);
                    for (uint32_t i = tid; i < GTAB / 2; i += GB) k2[i] = ek;
                    for (uint32_t i = tid; i < GTAB / 8; i += GB) c4[i] = z;
                }
                __syncthreads();
                PG(1)
                // ---- hash the records of this (sub-)range; K3_UNROLL independent loads per thread
                const uint32_t selshift = free_bits - e;
                uint32_t mymax = 0;
                for_records([&](ull key, ull val) {
                    if (e && ((simka_mix(key, ka->cfg.mask, ka->cfg.xs) >> selshift) & ((1ull << e) - 1ull)) != val_) return;
                    const uint32_t idx = atomicAdd(&s_nrec, 1u);
                    if (idx >= GCAP) { s_ovf = 1; return; }
                    uint32_t slot = simka_slot_hash(key) & (GTAB - 1u);
                    for (;;) {   // 2*GCAP slots, at most GCAP records: always terminates
                        const ull prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY, key);
                        if (prev == SIMKA_EMPTY_KEY || prev == key) break;
                        slot = (slot + 1u) & (GTAB - 1u);
                    }
                    atomicAdd((uint32_t *)scnt + (slot >> 1), 1u << ((slot & 1u) * 16u));
                    rslot[idx] = (uint16_t)slot;
                    rcnt[idx] = (uint32_t)val; rsmp[idx] = (smp_t)(val >> 32);
                    if ((uint32_t)val > mymax) mymax = (uint32_t)val;
                });
#pragma unroll
                for (int o_ = 32; o_ > 0; o_ >>= 1) { const uint32_t t_ = __shfl_xor(mymax, o_, 64); mymax = t_ > mymax ? t_ : mymax; }
                if ((tid & 63u) == 0 && mymax) atomicMax(&s_maxc, mymax);
                __syncthreads();
                PG(2)
                if (s_ovf) {
                    if (e >= free_bits) {
                        // Every key bit is fixed: the selected records are ONE k-mer shared by more than GCAP samples.  Such a
                        // group cannot be staged in LDS; it becomes a span of its own on the huge list (k_pairs_global).
                        const uint32_t s_ = s_nrec;
                        __syncthreads();
                        if (tid == 0) {
                            const ull eb_ = atomicAdd(&ka->o.cursors[0], (ull)s_), hs = atomicAdd(&ka->o.cursors[3], 1ull);
                            if (eb_ + s_ > ka->o.cap_entries || hs >= ka->o.cap_huge) { atomicOr(ka->o.err, SIMKA_DEVERR_CSR_FULL); s_ovf = 2; }
                            else {
                                SimkaSpan sp; sp.ebase = eb_; sp.gbase = 0; sp.nent = s_; sp.ngrp = 1; sp.maxc = 0; sp.pad = 0;
                                ka->o.huge[hs] = sp;
                                s_ebase = eb_; s_ndist++; s_nshared++;
                            }
                            s_nrec = 0;      // now the fill cursor
                        }
                        __syncthreads();
                        if (s_ovf != 2) {
                            const ull eb_ = s_ebase;
                            for_records([&](ull key, ull val) {
                                if (e && ((simka_mix(key, ka->cfg.mask, ka->cfg.xs) >> selshift) & ((1ull << e) - 1ull)) != val_) return;
                                ka->o.entries[eb_ + atomicAdd(&s_nrec, 1u)] = val;
                            });
                        }
                        continue;
                    }
                    if (tid == 0) {   // refine: two children with one more selector bit
                        if (s_sp + 2 > K3_STACK) { atomicOr(ka->o.err, SIMKA_DEVERR_GROUP_OVERFLOW); s_sp = 0; }
                        else {
                            s_stack[s_sp] = (2ull << e) | (val_ * 2ull + 1ull); s_sp++;
                            s_stack[s_sp] = (2ull << e) | (val_ * 2ull); s_sp++;
                        }
                    }
                    continue;
                }
                const uint32_t nrec = s_nrec;
                if (nrec == 0) continue;
                // ---- group geometry: one packed prefix over the table slots (entries | groups << 20)
                // (a thread owns the GTAB / GB = 8 slots [8 tid, 8 tid + 8): one 16-byte read of their counts, the packed values and
                //  their prefix in registers, ONE block scan of the threads' sums, two 16-byte stores -- the slot-strided loop + the
                //  generic in-LDS scan behind it took a quarter of the kernel)
                static_assert(GTAB == 8 * GB, "eight slots per thread");
                uint32_t ndist = 0, nshared = 0;
                uint32_t pv[8];
                {
                    const uint4 c4 = ((const uint4 *)scnt)[tid];
                    const uint32_t cw[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t c = (cw[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
                        ndist += c ? 1u : 0u;
                        nshared += c > 1u ? 1u : 0u;
                        pv[q] = (c >= min_share) ? (c | (1u << 20)) : 0u;
                    }
                }
                {   // one LDS atomic per wave for the two statistics (packed: both stay below 2^16 per wave and round)
                    const uint32_t both = wave_incl_scan(ndist | (nshared << 16));
                    if ((tid & 63u) == 63u && both) { atomicAdd(&s_ndist, both & 0xffffu); atomicAdd(&s_nshared, both >> 16); }
                }
                PG(3)
                uint32_t tsum = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) tsum += pv[q];
                uint32_t run;
                const uint32_t tot = block_excl_scan1<GB>(tsum, run, tmp);
                {
                    uint32_t ex[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) { ex[q] = run; run += pv[q]; }
                    ((uint4 *)gpk)[2 * tid] = make_uint4(ex[0], ex[1], ex[2], ex[3]);
                    ((uint4 *)gpk)[2 * tid + 1] = make_uint4(ex[4], ex[5], ex[6], ex[7]);
                }
                PG(4)
                const uint32_t nent = tot & 0xfffffu, ngrp = tot >> 20;
                if (ngrp == 0) continue;
                if (tid == 0) {
                    // slab reservations: one global atomic per K3_SLAB_* items.  A round whose output lands right behind the
                    // block's open span (same slabs) EXTENDS that span up to ka->o.span_cap entries: k_pairs pays its per-span
                    // overhead (scans, barriers, pair-range search) once per span, so longer spans are cheaper.
                    uint32_t ok = 1, fresh = 0;
                    if (s_slab[0] + nent > s_slab[1]) { s_slab[0] = atomicAdd(&ka->o.cursors[0], (ull)K3_SLAB_ENT); s_slab[1] = s_slab[0] + K3_SLAB_ENT; fresh = 1; if (s_slab[1] > ka->o.cap_entries) ok = 0; }
                    if (s_slab[2] + ngrp > s_slab[3]) { s_slab[2] = atomicAdd(&ka->o.cursors[1], (ull)K3_SLAB_GRP); s_slab[3] = s_slab[2] + K3_SLAB_GRP; fresh = 1; if (s_slab[3] > ka->o.cap_groups) ok = 0; }
                    const bool extend = !fresh && s_open != ~0ull && s_open_nent + nent <= ka->o.span_cap && s_open_ngrp + ngrp <= ka->o.span_cap / 2u;
                    if (!extend && s_slab[4] + 1 > s_slab[5]) { s_slab[4] = atomicAdd(&ka->o.cursors[2], (ull)K3_SLAB_SPAN); s_slab[5] = s_slab[4] + K3_SLAB_SPAN; if (s_slab[5] > ka->o.cap_spans) ok = 0; }
                    if (!ok) { atomicOr(ka->o.err, SIMKA_DEVERR_CSR_FULL); s_ovf = 2; }
                    else {
                        if (!extend) { s_open = s_slab[4]; s_slab[4] += 1; s_open_nent = 0; s_open_ngrp = 0; s_open_maxc = 0; }
                        s_soff = s_open_nent;
                        s_ebase = s_slab[0]; s_gbase = s_slab[2];
                        s_open_nent += nent; s_open_ngrp += ngrp; s_open_maxc = s_maxc > s_open_maxc ? s_maxc : s_open_maxc;
                        SimkaSpan sp; sp.ebase = s_slab[0] - s_soff; sp.gbase = s_slab[2] - (s_open_ngrp - ngrp); sp.nent = s_open_nent; sp.ngrp = s_open_ngrp; sp.maxc = s_open_maxc; sp.pad = 0;
                        ka->o.spans[s_open] = sp;
                        s_slab[0] += nent; s_slab[2] += ngrp;
                    }
                }
                __syncthreads();
                PG(5)
                if (s_ovf == 2) continue;
                const ull eb = s_ebase, gb = s_gbase;
                const uint32_t soff = s_soff;
                {   // the thread's eight slots again: group descriptors out, gpk becomes the entry offset (advanced as fill cursor)
                    const uint4 c4 = ((const uint4 *)scnt)[tid];
                    const uint4 g0 = ((const uint4 *)gpk)[2 * tid], g1 = ((const uint4 *)gpk)[2 * tid + 1];
                    const uint32_t cw[4] = { c4.x, c4.y, c4.z, c4.w };
                    uint32_t gv[8] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w };
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t c = (cw[q >> 1] >> ((q & 1) * 16)) & 0xffffu, g_ = gv[q];
                        if (c >= min_share) ka->o.groups[gb + (g_ >> 20)] = (((g_ & 0xfffffu) + soff) << 16) | c;    // start is relative to the span
                        gv[q] = g_ & 0xfffffu;
                    }
                    ((uint4 *)gpk)[2 * tid] = make_uint4(gv[0], gv[1], gv[2], gv[3]);
                    ((uint4 *)gpk)[2 * tid + 1] = make_uint4(gv[4], gv[5], gv[6], gv[7]);
                }
                __syncthreads();
                for (uint32_t i = tid; i < nrec; i += GB) {
                    const uint32_t slot = rslot[i];
                    if (scnt[slot] >= min_share) ka->o.entries[eb + atomicAdd(&gpk[slot], 1u)] = ((ull)rsmp[i] << 32) | (ull)rcnt[i];
                }
                PG(6)
            }
        }
      }
    }
    PG_FLUSH
    __syncthreads();
    if (tid == 0) {
        if (s_ndist) atomicAdd(&ka->o.glob[0], (ull)s_ndist);      // _nbDistinctKmers  (:1315)
        if (s_nshared) atomicAdd(&ka->o.glob[1], (ull)s_nshared);  // _nbSharedKmers    (:1319-1321)
    }
#undef KG_FRESH
    // unused span slots of this block's last slab: mark empty
    for (ull i = s_slab[4] + tid; i < s_slab[5]; i += GB) { SimkaSpan sp; sp.ebase = 0; sp.gbase = 0; sp.nent = 0; sp.ngrp = 0; sp.maxc = 0; sp.pad = 0; ka->o.spans[i] = sp; }
}

// --------------------------------------------------------------------------------------------
// K3b  k_pairs: SimkaCountProcessorSimple::updateDistanceDefault / updateDistanceSimple
// (ref: src/core/SimkaAlgorithm.hpp:356-402).  For every group (one k-mer, s samples) all
// s(s-1)/2 pairs i<j update
//     S[i][j]+=ci  S[j][i]+=cj  a[ij]+=1  bc[ij]+=min(ci,cj)        (default)
//     chord[ij]+=ci*cj  hell[ij]+=floor(sqrt(ci*cj))                 (simple; kul == bc)
// into u32 LDS accumulators private to the block; a wrap of the low word carries 2^32 straight
// into the global u64 cell, so the sums are exact.  Blocks are persistent over the span list.
// --------------------------------------------------------------------------------------------
// |(int)(u64)| as the reference computes it: abs((int)(...)) then u64 += int  (ref: src/core/SimkaAlgorithm.hpp:481)
__device__ __forceinline__ ull simka_whit_abs(ull d) {
    const int t = (int)d;
    const int r = (t == (int)0x80000000) ? t : (t < 0 ? -t : t);
    return (ull)(long long)r;
}

__device__ __forceinline__ void tri_unrank(uint32_t idx, uint32_t s, uint32_t &x, uint32_t &y) {
    // pairs (x<y) of s items, row-major; row x starts at x*s - x*(x+1)/2
    if (s <= 2048u) {      // (2 s - 1)^2 < 2^24: single precision holds the discriminant exactly, 32-bit corrections by at most one row
        const uint32_t fsu = 2u * s - 1u;
        const uint32_t disc = fsu * fsu - 8u * idx;                  // >= 1 for idx < s (s - 1) / 2
        int xi = (int)(((float)fsu - __fsqrt_rn((float)disc)) * 0.5f);
        const int S = (int)s, id = (int)idx;
        xi = xi < 0 ? 0 : (xi > S - 2 ? S - 2 : xi);
        while (xi > 0 && (xi * S - xi * (xi + 1) / 2) > id) xi--;
        while (xi + 1 <= S - 2 && ((xi + 1) * S - (xi + 1) * (xi + 2) / 2) <= id) xi++;
        x = (uint32_t)xi;
        y = (uint32_t)(id - (xi * S - xi * (xi + 1) / 2) + xi + 1);
        return;
    }
    const double fs = 2.0 * (double)s - 1.0;
    double disc = fs * fs - 8.0 * (double)idx;
    if (disc < 0) disc = 0;
    long long xi = (long long)((fs - sqrt(disc)) * 0.5);
    const long long S = (long long)s, id = (long long)idx;
    if (xi < 0) xi = 0;
    if (xi > S - 2) xi = S - 2;
    while (xi > 0 && (xi * S - xi * (xi + 1) / 2) > id) xi--;
    while (xi + 1 <= S - 2 && ((xi + 1) * S - (xi + 1) * (xi + 2) / 2) <= id) xi++;
    x = (uint32_t)xi;
    y = (uint32_t)(id - (xi * S - xi * (xi + 1) / 2) + xi + 1);
}

// floor(sqrt(x)) for the Hellinger term: 32-bit fast path, exact
// x < 2^32 (every count of the span below 65536, known per span): no 64-bit product, no normalisation, no branch
__device__ __forceinline__ uint32_t pair_isqrt32(uint32_t v) {
    uint32_t r = (uint32_t)__fsqrt_rn((float)v);
    r = r > 65535u ? 65535u : r;
    r -= (r * r > v) ? 1u : 0u;
    r += (2u * r < v - r * r) ? 1u : 0u;
    return r;
}
__device__ __forceinline__ uint32_t pair_isqrt(ull x) {
    if (x >> 32) return (uint32_t)simka_isqrt(x);
    // float sqrt of a 32-bit value is within 1 of the floor: two branch-free corrections (r <= 65535, so r * r fits 32 bits;
    // (r + 1)^2 is compared in 64 bits only through its carry-free form r * r + 2 r + 1 <= v  <=>  2 r < v - r * r)
    const uint32_t v = (uint32_t)x;
    uint32_t r = (uint32_t)__fsqrt_rn((float)v);
    r = r > 65535u ? 65535u : r;
    r -= (r * r > v) ? 1u : 0u;
    r += (2u * r < v - r * r) ? 1u : 0u;
    return r;
}

// ---- -complex-dist, both-present pairs: the two terms that cost the pair loop most, priced down (C5's shape, 500 samples: the
// pair kernel is 72 percent of the step and the complex terms were 54 percent of it)
// natural logarithm of a positive normal double from a 64-entry table of (1 / c, ln c), c = 1 + (i + 1/2) / 64, and a degree-6
// polynomial in r = m / c - 1, |r| < 2^-7: absolute error below 1e-15 (the remainder r^7 / 7 is 2.5e-16) -- the library log is
// correctly rounded with three times the instructions; the KL sums are kept in 2^-60 fixed point and compared to 1e-9 relative.
#define SIMKA_LNTAB 64
__device__ const double2 g_simka_lntab[SIMKA_LNTAB] = {
    { 0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106789p-8 },
    { 0x1.f44659e4a4271p-1, 0x1.7b91b07d5b11bp-6 },
    { 0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd60p-5 },
    { 0x1.e573ac901e574p-1, 0x1.b42dd711971bfp-5 },
    { 0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae1p-4 },
    { 0x1.d77b654b82c34p-1, 0x1.51b073f06183fp-4 },
    { 0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b21p-4 },
    { 0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4 },
    { 0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabbp-4 },
    { 0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a0p-3 },
    { 0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f74p-3 },
    { 0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3 },
    { 0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21dp-3 },
    { 0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3 },
    { 0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b563p-3 },
    { 0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd6p-3 },
    { 0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb91p-3 },
    { 0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5936p-3 },
    { 0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2 },
    { 0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad5p-2 },
    { 0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2 },
    { 0x1.7f405fd017f40p-1, 0x1.2895a13de86a3p-2 },
    { 0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d55p-2 },
    { 0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2 },
    { 0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2 },
    { 0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2 },
    { 0x1.6a13cd1537290p-1, 0x1.62c82f2b9c795p-2 },
    { 0x1.661ec6a5122f9p-1, 0x1.6e08eaa2ba1e4p-2 },
    { 0x1.623fa77016240p-1, 0x1.792a55fdd47a2p-2 },
    { 0x1.5e75bb8d015e7p-1, 0x1.842d1da1e8b17p-2 },
    { 0x1.5ac056b015ac0p-1, 0x1.8f11e873662c7p-2 },
    { 0x1.571ed3c506b3ap-1, 0x1.99d958117e08bp-2 },
    { 0x1.5390948f40febp-1, 0x1.a484090e5bb0ap-2 },
    { 0x1.5015015015015p-1, 0x1.af1293247786bp-2 },
    { 0x1.4cab88725af6ep-1, 0x1.b9858969310fbp-2 },
    { 0x1.49539e3b2d067p-1, 0x1.c3dd7a7cdad4dp-2 },
    { 0x1.460cbc7f5cf9ap-1, 0x1.ce1af0b85f3ebp-2 },
    { 0x1.42d6625d51f87p-1, 0x1.d83e7258a2f3ep-2 },
    { 0x1.3fb013fb013fbp-1, 0x1.e24881a7c6c26p-2 },
    { 0x1.3c995a47babe7p-1, 0x1.ec399d2468cc0p-2 },
    { 0x1.3991c2c187f63p-1, 0x1.f6123fa7028acp-2 },
    { 0x1.3698df3de0748p-1, 0x1.ffd2e0857f498p-2 },
    { 0x1.33ae45b57bcb2p-1, 0x1.04bdf9da926d2p-1 },
    { 0x1.30d190130d190p-1, 0x1.0986f4f573521p-1 },
    { 0x1.2e025c04b8097p-1, 0x1.0e44985d1cc8cp-1 },
    { 0x1.2b404ad012b40p-1, 0x1.12f719593efbcp-1 },
    { 0x1.288b01288b013p-1, 0x1.179eabbd899a1p-1 },
    { 0x1.25e22708092f1p-1, 0x1.1c3b81f713c25p-1 },
    { 0x1.23456789abcdfp-1, 0x1.20cdcd192ab6ep-1 },
    { 0x1.20b470c67c0d9p-1, 0x1.2555bce98f7cbp-1 },
    { 0x1.1e2ef3b3fb874p-1, 0x1.29d37fec2b08bp-1 },
    { 0x1.1bb4a4046ed29p-1, 0x1.2e47436e40268p-1 },
    { 0x1.19453808ca29cp-1, 0x1.32b1339121d71p-1 },
    { 0x1.16e0689427379p-1, 0x1.37117b54747b6p-1 },
    { 0x1.1485f0e0acd3bp-1, 0x1.3b68449fffc23p-1 },
    { 0x1.12358e75d3033p-1, 0x1.3fb5b84d16f42p-1 },
    { 0x1.0fef010fef011p-1, 0x1.43f9fe2f9ce67p-1 },
    { 0x1.0db20a88f4696p-1, 0x1.48353d1ea88dfp-1 },
    { 0x1.0b7e6ec259dc8p-1, 0x1.4c679afccee3ap-1 },
    { 0x1.0953f39010954p-1, 0x1.50913cc01686bp-1 },
    { 0x1.073260a47f7c6p-1, 0x1.54b2467999498p-1 },
    { 0x1.05197f7d73404p-1, 0x1.58cadb5cd7989p-1 },
    { 0x1.03091b51f5e1ap-1, 0x1.5cdb1dc6c1765p-1 },
    { 0x1.0101010101010p-1, 0x1.60e32f44788d9p-1 }
};
// a product / sum rounded on its own (HIP's __dmul_rn is a plain `*`, which the compiler may still fuse into a later add)
__device__ __forceinline__ double simka_mul_rn(double a, double b) { double r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double simka_add_rn(double a, double b) { double r = a + b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double simka_fast_ln(double h, const double2 *tab) {
    const long long b = __double_as_longlong(h);
    const int e = (int)((b >> 52) & 0x7ff) - 1023;
    const double m = __longlong_as_double((b & 0x000fffffffffffffll) | 0x3ff0000000000000ll);
    const double2 t = tab[(uint32_t)(b >> 46) & (SIMKA_LNTAB - 1u)];
    const double r = fma(m, t.x, -1.0);
    double q = -1.0 / 6.0;
    q = fma(q, r, 0.2); q = fma(q, r, -0.25); q = fma(q, r, 1.0 / 3.0); q = fma(q, r, -0.5); q = fma(q, r, 1.0);
    return fma((double)e, 0.69314718055994530942, fma(q, r, t.y));
}
// Whittaker, both-present part: |(int)(u64)(ci Nj) - (u64)(cj Ni)| - |..(ci Nj)| - |..(cj Ni)| with the reference's products in double
// (ref: src/core/SimkaAlgorithm.hpp:477-481).  A product below 2^53 is exact in double, so it IS the integer product (two 32 x 32
// multiplies instead of two conversions, a double multiply and the emulated double -> u64 conversion); beyond, the double path.
__device__ __forceinline__ ull simka_whit_term(uint32_t ci, uint32_t cj, ull ni, ull nj, double dni, double dnj) {
    ull uX = (ull)ci * nj, uY = (ull)cj * ni;
    if (((ni | nj) >> 32) || ((uX | uY) >> 53)) { uX = (ull)((double)ci * dnj); uY = (ull)((double)cj * dni); }
    return simka_whit_abs(uX - uY) - simka_whit_abs(uX) - simka_whit_abs(uY);
}

// LDS cells of the 32-bit accumulators are PACKED two per u64 -- (S_ij | S_ji<<32), (a | bc<<32), (chord | hell<<32) --
// so one non-returning ds_add_u64 feeds two accumulators.  Every half stays < 2^32 between flushes (`bound`), so the
// low half never carries into the high one.
// fold the block's private LDS accumulators into the global u64 accumulators acc[a][pair] (atomics) and zero them.
// Blocks start at different cells so that concurrent flushes do not queue on the same addresses.
template <bool TILED, int K4_BLOCK>
__device__ __forceinline__ void pairs_flush(ull *pk, ull *c64, ull *acc, const SimkaPairCfg &pc, bool rect, uint32_t baseI, uint32_t baseJ) {
    __syncthreads();
    const uint32_t CP = pc.ncell_pad, npk = pc.nacc32 >> 1, N = pc.nb_samples, T = pc.tile;
    const ull NP = pc.nb_pairs;
    const uint32_t rot = (uint32_t)(((ull)blockIdx.x * 2654435761ull) % CP);
    for (uint32_t c0 = threadIdx.x; c0 < CP; c0 += K4_BLOCK) {
        uint32_t c = c0 + rot; if (c >= CP) c -= CP;
        const ull v1 = pk[1 * CP + c];            // (a | bc): a counts every update of the cell
        const ull w0 = pc.nacc64 ? c64[c] : 0ull, w1 = pc.nacc64 ? c64[CP + c] : 0ull;
        if (!(v1 | w0 | w1)) continue;
        ull pg = c;                               // untiled: the cell index IS the pair index
        if (TILED) {
            uint32_t li, lj;
            if (rect) { li = c / T; lj = c - li * T; } else tri_unrank(c, T, li, lj);
            pg = simka_pair_index(baseI + li, baseJ + lj, N);
        }
        const ull v0 = pk[c];
        atomicAdd(&acc[SIMKA_ACC_SIJ * NP + pg], (ull)(uint32_t)v0); atomicAdd(&acc[SIMKA_ACC_SJI * NP + pg], v0 >> 32);
        atomicAdd(&acc[SIMKA_ACC_A * NP + pg], (ull)(uint32_t)v1); atomicAdd(&acc[SIMKA_ACC_BC * NP + pg], v1 >> 32);
        pk[c] = 0; pk[CP + c] = 0;
        if (npk > 2) {
            const ull v2 = pk[2 * CP + c];
            atomicAdd(&acc[SIMKA_ACC_CHORD * NP + pg], (ull)(uint32_t)v2); atomicAdd(&acc[SIMKA_ACC_HELL * NP + pg], v2 >> 32);
            pk[2 * CP + c] = 0;
        }
        if (pc.nacc64) {
            atomicAdd(&acc[(ull)pc.nacc32 * NP + pg], w0); atomicAdd(&acc[((ull)pc.nacc32 + 1) * NP + pg], w1);
            c64[c] = 0; c64[CP + c] = 0;
        }
    }
    __syncthreads();
}

#ifdef SIMKA_PHASE_PROF
__device__ ull g_pairs_phase[8];
#define PP_DECL ull pp_t = wall_clock64(), pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PP(i) { const ull n_ = wall_clock64(); pp_acc[i] += n_ - pp_t; pp_t = n_; }
#define PP_FLUSH if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_pairs_phase[i_], pp_acc[i_]); }
#else
#define PP_DECL
#define PP(i)
#define PP_FLUSH
#endif

// TILED=false: all N(N-1)/2 cells live in LDS (one "tile" = every sample).  TILED=true: block row blockIdx.y owns the
// sample-tile pair (I<=J); each span's entries are COMPACTED to the members of tile I (list A) and tile J (list B) with
// one packed block scan, so a block enumerates exactly the pairs it owns (A x A triangle on the diagonal, A x B
// rectangle off it) instead of filtering all of them.
// K4_BLOCK: 1024 threads when the (i,j) space is large (many pairs per span), 256 for few samples
template <bool TILED, int K4_BLOCK>
__global__ void __launch_bounds__(K4_BLOCK)
k_pairs(const SimkaSpan *spans, const ull *cursors, const ull *entries, const uint32_t *groups, SimkaPairCfg pc,
        ull *acc, ull *work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t CP = pc.ncell_pad, npk = pc.nacc32 >> 1;
    const bool cplx = pc.nacc64 != 0;
    ull *pk = (ull *)(smem + SIMKA_LDS_HEAD);                    // [npk][CP]     packed u32 pairs
    ull *c64 = pk + (size_t)npk * CP;                            // [nacc64][CP]  (whit, klfix)
    // a span holds <= EC = pc.span_cap entries in <= GC = EC/2 groups (every group has >= 2 entries)
    const uint32_t EC = pc.span_cap, GC = EC / 2u;
    ull *ent = c64 + (size_t)pc.nacc64 * CP;                     // [EC]          (sample<<32 | count)
    double2 *epp = (double2 *)(ent + EC);                        // [EC]          complex: (p = c / N_sample, p ln p): one 16-byte read per entry
    double *eplp = (double *)epp + (cplx ? EC : 0);              //               (second half of that array: the layout below counts 2 x EC doubles)
    double *tn = eplp + (cplx ? EC : 0);                         // [SIMKA_PAIR_TN] complex: N of the samples of tile I, then tile J
    ull *tnu = (ull *)(tn + (cplx ? SIMKA_PAIR_TN : 0));         // [SIMKA_PAIR_TN] complex: the same N as integers
    double2 *lntab = (double2 *)(tnu + (cplx ? SIMKA_PAIR_TN : 0));    // [SIMKA_LNTAB] complex: simka_fast_ln
    uint32_t *gdesc = (uint32_t *)(lntab + (cplx ? SIMKA_LNTAB : 0));   // [GC]      list A of the group: (start<<16 | size)
    uint32_t *gpref = gdesc + GC;                                // [GC+2]        pair prefix
    uint32_t *tmp = gpref + GC + 2;                              // [32]
    uint32_t *gdescB = tmp + 32;                                 // tiled: [GC]   list B of the group
    uint32_t *epre = gdescB + GC;                                // tiled: [EC+2] packed scan of the tile-membership flags
    uint16_t *idxA = (uint16_t *)(epre + EC + 2);                // tiled: [EC]   entry indices of tile I members
    uint16_t *idxB = idxA + EC;                                  // tiled: [EC]   entry indices of tile J members

    const uint32_t tid = threadIdx.x;
    const uint32_t N = pc.nb_samples, T = pc.tile;
    uint32_t I = 0, J = 0;                                       // tile pair (I<=J) of this block row
    if (TILED) {
        uint32_t r = blockIdx.y;
        for (I = 0; I < pc.ntiles; I++) { const uint32_t row = pc.ntiles - I; if (r < row) { J = I + r; break; } r -= row; }
    }
    const bool rect = TILED && I != J;
    const uint32_t baseI = I * T, baseJ = J * T;
    const uint32_t TD = TILED ? T : N;                           // edge of the triangular cell index
    for (uint32_t i = tid; i < npk * CP; i += K4_BLOCK) pk[i] = 0;
    for (uint32_t i = tid; i < pc.nacc64 * CP; i += K4_BLOCK) c64[i] = 0;
    if (cplx) {
        for (uint32_t i = tid; i < SIMKA_LNTAB; i += K4_BLOCK) lntab[i] = g_simka_lntab[i];
        for (uint32_t i = tid; i < TD; i += K4_BLOCK) {
            const ull ni_ = (baseI + i < N) ? pc.tot_n[baseI + i] : 1ull;
            tn[i] = (double)ni_; tnu[i] = ni_;
            if (rect) { const ull nj_ = (baseJ + i < N) ? pc.tot_n[baseJ + i] : 1ull; tn[T + i] = (double)nj_; tnu[T + i] = nj_; }
        }
    }
    // every packed half is fed by NON-returning atomics and receives at most one add per group, so
    // `bound` (sum over spans of #groups x largest count) < 2^32 guarantees no wrap; flush before it could.
    ull bound = 0, bound_q = 0;      // bound_q: same for the chord products (#groups x maxcount^2)

    const ull nspans = cursors[2];
    // software pipeline over this block's spans: descriptor two iterations ahead, entries/groups one iteration ahead
    // (registers), so the global-load latency of span i+1 hides behind the pair loop of span i.
    // A span has <= SIMKA_SPAN_MAX entries: up to 4 per thread for 1024-thread blocks, 16 for 256-thread ones.
    constexpr int EPT = (SIMKA_SPAN_MAX + K4_BLOCK - 1) / K4_BLOCK;     // entries (and group descriptors) per thread
    SimkaSpan span;
    span.ngrp = 0; span.nent = 0;
    // The descriptor of the span AFTER the next is loaded two iterations ahead -- and has to stay in VECTOR registers until it is needed: a
    // value the compiler knows to be wave-uniform is moved to scalar registers (v_readfirstlane) right behind its load, i.e. every block
    // sat out one trip to memory per span in the staging phase.  The address gets a lane-dependent zero the compiler cannot see through; the
    // fields become scalars an iteration later (to_uniform).
    uint32_t lane_zero; asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    struct SpanV { uint32_t eb_lo, eb_hi, gb_lo, gb_hi, nent, ngrp, maxc; } nspan_v;
    // Which span slots a block takes.  work != NULL (one tile: grid.y == 1): DYNAMIC -- chunks of KP_WCHUNK consecutive slots, chunk
    // blockIdx.x first, then whatever the global counter says (thread 0 grabs the chunk after next at the first slot of a chunk and
    // publishes it in LDS at the next loop-top barrier), so a block that drew expensive spans does not hold the launch back.
    // work == NULL (tile pairs: every block row walks all spans): static rows.  Span slots come in slabs of K3_SLAB_SPAN (one
    // k_group block each; the unused tail of a block's last slab is empty) and the grid is a multiple of the slab: with
    // slot = row * grid + block every block would always see the SAME position inside the slabs -- the blocks that land on slab
    // heads would get up to twice the real spans of the others.  Row r is rotated by 13 r instead.
    constexpr uint32_t KP_WCHUNK = 4;
    const bool dyn = work != nullptr;
    uint32_t *s_nextc = (uint32_t *)smem;            // [2] (dyn) the chunk after next, double-buffered
    const ull nrows = (nspans + gridDim.x - 1) / gridDim.x;
    ull row = 0;                                     // static: the row; dynamic: unused
    ull chunk = blockIdx.x, chunk_n = 0, chunk_n2 = 0, grab = 0;
    uint32_t kpos = 0, tog = 0;
    if (dyn) {
        if (tid == 0) s_nextc[0] = gridDim.x + (uint32_t)atomicAdd(work, 1ull);
        __syncthreads();
        chunk_n = s_nextc[0];
    }
    // slot of the span d iterations ahead (d <= 2 < KP_WCHUNK), ~0: none
    auto slot_ahead = [&](uint32_t d) -> ull {
        ull s_;
        if (dyn) { const uint32_t pos = kpos + d; s_ = (pos < KP_WCHUNK ? chunk : chunk_n) * KP_WCHUNK + (pos % KP_WCHUNK); }
        else { if (row + d >= nrows) return ~0ull; s_ = (row + d) * gridDim.x + (blockIdx.x + (row + d) * 13ull) % gridDim.x; }
        return s_ < nspans ? s_ : ~0ull;
    };
    auto load_span_v = [&](ull s_, SpanV &out) {
        out.eb_lo = 0; out.eb_hi = 0; out.gb_lo = 0; out.gb_hi = 0; out.nent = 0; out.ngrp = 0; out.maxc = 0;
        if (s_ != ~0ull) {
            const uint4 *q_ = (const uint4 *)(spans + s_) + lane_zero;
            const uint4 a_ = q_[0], b_ = q_[1];
            out.eb_lo = a_.x; out.eb_hi = a_.y; out.gb_lo = a_.z; out.gb_hi = a_.w; out.nent = b_.x; out.ngrp = b_.y; out.maxc = b_.z;
        }
    };
    auto to_uniform = [&](const SpanV &v, SimkaSpan &out) {
        out.ebase = ((ull)(uint32_t)__builtin_amdgcn_readfirstlane((int)v.eb_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v.eb_lo);
        out.gbase = ((ull)(uint32_t)__builtin_amdgcn_readfirstlane((int)v.gb_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v.gb_lo);
        out.nent = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.nent); out.ngrp = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.ngrp);
        out.maxc = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.maxc); out.pad = 0;
    };
    auto load_span = [&](uint32_t d, SimkaSpan &out) { out.ngrp = 0; out.nent = 0; const ull s_ = slot_ahead(d); if (s_ != ~0ull) out = spans[s_]; };
    auto more = [&]() -> bool { return dyn ? chunk * KP_WCHUNK < nspans : row < nrows; };
    load_span(0, span);
    load_span_v(slot_ahead(1), nspan_v);
    ull pre_e[EPT]; uint32_t pre_g[EPT];
#pragma unroll
    for (int q = 0; q < EPT; q++) {
        const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
        pre_e[q] = (i < span.nent) ? entries[span.ebase + i] : 0ull;
        pre_g[q] = (i < span.ngrp) ? groups[span.gbase + i] : 0u;
    }
    PP_DECL
    for (; more(); ) {
        // ---- current span: registers -> LDS
        PP(0)
        if (dyn && kpos == 1u && tid == 0) s_nextc[tog ^ 1u] = gridDim.x + (uint32_t)grab;
        __syncthreads();
        if (dyn && kpos == 1u) { tog ^= 1u; chunk_n2 = s_nextc[tog]; }
        if (dyn && kpos == 0u && tid == 0) grab = atomicAdd(work, 1ull);
        const SimkaSpan cur = span;
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            if (i < cur.nent) {
                const ull e = pre_e[q];
                const uint32_t si = (uint32_t)(e >> 32);
                // one tile: the entry carries the start of its sample's row of pair cells, li N - li (li + 1) / 2 - li (< 2^16: the
                // cells of all pairs fit LDS), so that a pair's cell is one add instead of two multiplies
                if (!TILED) ent[i] = e | ((ull)(si * TD - ((si * (si + 1u)) >> 1) - si) << 48);
                else ent[i] = e;
                uint32_t fl = 0, loc = si;                       // membership flags (A | B<<16), index into tn
                if (TILED) {
                    const uint32_t li = si - baseI, lj = si - baseJ;
                    if (li < T) { fl = 1u; loc = li; }
                    else if (rect && lj < T) { fl = 1u << 16; loc = T + lj; }
                    epre[i] = fl;
                }
                if (cplx && (!TILED || fl)) {
                    const double p = (double)(uint32_t)e / tn[loc];
                    epp[i] = make_double2(p, simka_mul_rn(p, simka_fast_ln(p, lntab)));      // (the SAME logarithm as in the pair loop: two identical samples cancel to exactly 0, as in the reference)
                }
            }
            if (i < cur.ngrp) { gdesc[i] = pre_g[q]; if (!TILED) { const uint32_t s_ = pre_g[q] & 0xffffu; gpref[i] = s_ * (s_ - 1u) / 2u; } }
        }
        // ---- issue the loads of the next span, fetch the descriptor after it
        to_uniform(nspan_v, span);
        const ull slot2 = slot_ahead(2);
        // (the position moves on here: every `continue` below goes straight to the next span)
        if (dyn) { if (++kpos == KP_WCHUNK) { chunk = chunk_n; chunk_n = chunk_n2; kpos = 0; } } else row++;
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            pre_e[q] = (i < span.nent) ? entries[span.ebase + i] : 0ull;
            pre_g[q] = (i < span.ngrp) ? groups[span.gbase + i] : 0u;
        }
        load_span_v(slot2, nspan_v);      // (behind the entry loads: the compiler waits for every older load before it overwrites the prefetch registers)
        PP(1)
        if (cur.ngrp == 0) continue;     // unused slot of a k_group span slab (uniform)
        const ull add = (ull)cur.ngrp * (ull)cur.maxc;
        const ull addq = (ull)cur.ngrp * (ull)cur.maxc * (ull)cur.maxc;
        // chord: packed non-returning adds as long as the span's products cannot wrap a half cell; else straight into the global u64 cell
        const bool chord_fast = cur.maxc < 46341u && addq < 0xffffffffull;
        if (bound + add >= 0xffffffffull || (chord_fast && bound_q + addq >= 0xffffffffull)) { pairs_flush<TILED, K4_BLOCK>(pk, c64, acc, pc, rect, baseI, baseJ); bound = 0; bound_q = 0; }
        bound += add;
        if (chord_fast) bound_q += addq;
        const bool smallc = chord_fast && cur.maxc < 32768u;       // every product of the span is a 32-bit value: the cheap chord / Hellinger update
        __syncthreads();
        PP(2)
        if (TILED) {
            // compact the tile members: one scan of the packed flags gives every entry its slot in list A / list B
            const uint32_t tot = block_excl_scan<K4_BLOCK>(epre, cur.nent, tmp);
            if (tid == 0) epre[cur.nent] = tot;
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
                if (i < cur.nent) {
                    const uint32_t si = (uint32_t)(ent[i] >> 32);
                    const uint32_t pos = epre[i];
                    if (si - baseI < T) idxA[pos & 0xffffu] = (uint16_t)i;
                    else if (rect && si - baseJ < T) idxB[pos >> 16] = (uint16_t)i;
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                const uint32_t g = tid + (uint32_t)q * K4_BLOCK;
                if (g < cur.ngrp) {
                    const uint32_t d = gdesc[g];
                    const uint32_t p0 = epre[d >> 16], p1 = epre[(d >> 16) + (d & 0xffffu)];
                    const uint32_t a0 = p0 & 0xffffu, nA = (p1 & 0xffffu) - a0, b0 = p0 >> 16, nB = (p1 >> 16) - b0;
                    gdesc[g] = (a0 << 16) | nA;
                    gdescB[g] = (b0 << 16) | nB;
                    gpref[g] = rect ? nA * nB : nA * (nA - 1u) / 2u;      // nA = 0 gives 0 either way (0 * 0xffffffff / 2 = 0)
                }
            }
            __syncthreads();
        }
        PP(3)
        const uint32_t P = block_excl_scan<K4_BLOCK>(gpref, cur.ngrp, tmp);
        if (tid == 0) gpref[cur.ngrp] = P;
        __syncthreads();
        PP(4)
        const uint32_t chunk = (P + K4_BLOCK - 1) / K4_BLOCK;
        uint32_t p = tid * chunk;
        const uint32_t pend = (p + chunk < P) ? p + chunk : P;
        if (p < pend) {
            uint32_t lo = 0, hi = cur.ngrp;    // largest g with gpref[g] <= p
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (gpref[mid] <= p) lo = mid; else hi = mid; }
            uint32_t g = lo;
            while (g + 1 < cur.ngrp && gpref[g + 1] <= p) g++;   // skip groups without pairs
            uint32_t d = gdesc[g];
            uint32_t a0 = d >> 16, nA = d & 0xffffu, b0 = 0, nB = 0, x, y;
            if (rect) {
                const uint32_t dB = gdescB[g]; b0 = dB >> 16; nB = dB & 0xffffu;
                const uint32_t r = p - gpref[g];
                x = r / nB; y = r - x * nB;
            } else tri_unrank(p - gpref[g], nA, x, y);
            uint32_t ix = TILED ? (uint32_t)idxA[a0 + x] : a0 + x;
            ull ex = ent[ix];
            PP(5)
            // The pair loop in two instances (round 6): FAST = what C3 runs -- -simple-dist, every product of the span a 32-bit value, no
            // -complex-dist -- with the three wave-uniform flags compile-time constants (the generic instance tests them for every pair: five
            // scalar tests and branches in a ~35-instruction iteration).
            auto pair_loop = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                const bool simple_ = FAST ? true : (pc.simple != 0u), smallc_ = FAST ? true : smallc, cplx_ = FAST ? false : cplx;
                for (; p < pend; p++) {
                    const uint32_t iy = rect ? (uint32_t)idxB[b0 + y] : (TILED ? (uint32_t)idxA[a0 + y] : a0 + y);
                    const ull ey = ent[iy];
                    uint32_t si = (uint32_t)(ex >> 32), sj = (uint32_t)(ey >> 32);
                    uint32_t ci = (uint32_t)ex, cj = (uint32_t)ey;
                    uint32_t li, lj, cell;
                    if (!TILED) {      // branch-free: order the pair with selects, the cell from the smaller sample's row start
                        const uint32_t rx_ = si >> 16, ry_ = sj >> 16;
                        si &= 0xffffu; sj &= 0xffffu;
                        const bool lo = si < sj;
                        const uint32_t a_ = lo ? ci : cj, b_ = lo ? cj : ci, smin = lo ? si : sj, smax = lo ? sj : si;
                        cell = (lo ? rx_ : ry_) + smax - 1u;
                        ci = a_; cj = b_; si = smin; sj = smax; li = si; lj = sj;
                    } else {
                    // off-diagonal tiles: every member of A precedes every member of B.  Elsewhere order the pair.
                    if (!rect && si > sj) { uint32_t t_ = si; si = sj; sj = t_; t_ = ci; ci = cj; cj = t_; }
                    li = si - baseI; lj = sj - baseJ;
                    cell = rect ? li * T + lj : li * TD - ((li * (li + 1u)) >> 1) + (lj - li - 1u);   // TD <= 65535: fits 32 bits
                    }
    #ifdef SIMKA_EXP_PAIRS_NO_BILINEAR     // experiment (garbage S / a / chord): what the pair loop costs when the bilinear accumulators are taken off it (upper bound of an MFMA offload)
                    if (simple_ && smallc_) { atomicAdd(&pk[1 * CP + cell], (ull)(ci < cj ? ci : cj) | ((ull)pair_isqrt32(ci * cj) << 32)); }
                    else
    #endif
                    {
                    atomicAdd(&pk[0 * CP + cell], (ull)ci | ((ull)cj << 32));                       // S_ij | S_ji
                    atomicAdd(&pk[1 * CP + cell], 1ull | ((ull)(ci < cj ? ci : cj) << 32));         // a | bc
                    }
    #ifdef SIMKA_EXP_PAIRS_NO_BILINEAR
                    if (simple_ && !smallc_) {
    #else
                    if (simple_) {
    #endif
                        if (smallc_) {       // (uniform per span) counts below 2^15: the product is a 32-bit value below 2^30
                            const uint32_t prod32 = ci * cj;
                            atomicAdd(&pk[2 * CP + cell], (ull)prod32 | ((ull)pair_isqrt32(prod32) << 32));  // chord | hell
                        } else {
                            const ull prod = (ull)ci * (ull)cj;
                            const ull hell = (ull)pair_isqrt(prod) << 32;
                            if (chord_fast) atomicAdd(&pk[2 * CP + cell], (ull)(uint32_t)prod | hell);  // chord | hell
                            else {   // huge counts: the product goes straight to the global u64 cell
                                atomicAdd(&pk[2 * CP + cell], hell);
                                atomicAdd(&acc[SIMKA_ACC_CHORD * pc.nb_pairs + simka_pair_index(si, sj, N)], prod);
                            }
                        }
                    }
                    if (cplx_) {
                        // updateDistanceComplex restricted to both-present pairs (ref: src/core/SimkaAlgorithm.hpp:437-446,477-481);
                        // the one-sided terms are closed forms of S / totals / count histograms, added on the host.
                        // KL: with p = ci/Ni, q = cj/Nj the reference's  p ln(2p/(p+q)) + q ln(2q/(p+q))  equals
                        // p ln p + q ln q - (p+q) ln((p+q)/2): one logarithm per pair, the p ln p terms are per entry.
                        const double2 px = epp[ix], py = epp[iy];
                        const double h = px.x + py.x;
                        double dd = simka_add_rn(px.y, py.y) - simka_mul_rn(h, simka_fast_ln(h * 0.5, lntab));      // (products rounded on their own, never fused into the subtraction: identical samples give 2 a - 2 a = 0 exactly)
                        dd = dd < 0.0 ? 0.0 : dd;            // >= 0 mathematically (Jensen); rounding noise must not drive a sum of near-identical samples negative
                        atomicAdd(&c64[1 * CP + cell], (ull)(long long)llrint(dd * SIMKA_KL_SCALE));
                        const uint32_t jn = (rect ? T : 0u) + lj;
                        atomicAdd(&c64[0 * CP + cell], simka_whit_term(ci, cj, tnu[li], tnu[jn], tn[li], tn[jn]));
                    }
                    // next pair of the span
                    y++;
                    if (rect ? (y == nB) : (y == nA)) {
                        x++; y = rect ? 0u : x + 1u;
                        if (rect ? (x == nA) : (y >= nA)) {   // group exhausted
                            g++;
                            while (g < cur.ngrp && gpref[g + 1] == gpref[g]) g++;   // groups without pairs for this tile pair
                            if (g >= cur.ngrp) break;
                            d = gdesc[g]; a0 = d >> 16; nA = d & 0xffffu; x = 0; y = rect ? 0u : 1u;
                            if (rect) { const uint32_t dB = gdescB[g]; b0 = dB >> 16; nB = dB & 0xffffu; }
                        }
                        ix = TILED ? (uint32_t)idxA[a0 + x] : a0 + x;
                        ex = ent[ix];
                    }
                }
            };
            if (pc.simple && smallc && !cplx) pair_loop(std::true_type()); else pair_loop(std::false_type());
        }
    }
    PP(6)
    PP_FLUSH
    __syncthreads();
    pairs_flush<TILED, K4_BLOCK>(pk, c64, acc, pc, rect, baseI, baseJ);
}

// --------------------------------------------------------------------------------------------
// Tiled pair accumulation over TILE-MAJOR spans.  k_pairs<true> re-stages every span for every sample-tile pair and
// compacts it to the members of the two tiles each time.  Here a pre-pass reorders each span ONCE:
//   k_tile_major: one wave per span, a stable counting sort of the span's entries by sample tile.  Entries keep their group
//   order inside a tile segment, so the members a group has in one tile are contiguous; each entry carries its span-local
//   group index (g << 48 | sample << 32 | count).  tm_off[span][t] = start of tile t's segment; complex: (p, p ln p) per entry,
//   computed once instead of once per tile pair (recomputing p ln p when a tile pair stages the entry saves eight prefetch registers
//   and costs the pair kernel 8 % on c5_5: measured in round 5).
//   k_pairs_tm: block row (I, J) stages only the two segments it owns; group runs are found from heads and tails (two 16-bit
//   LDS stores per run, no scan over the entries, no index lists) and pairs are enumerated exactly as in k_pairs.
// --------------------------------------------------------------------------------------------
#define KTM_WAVES 4                 // k_tile_major: waves (= spans in flight) per block
// What k_tile_major leaves per entry for -complex-dist: (p, p ln p) -- 16 prefetch registers per thread in k_pairs_tm<true>, which then
// keeps 8 VGPRs in scratch (stored / reloaded once per BATCH of spans, outside the pair loop) -- or, with -DKTM_PLNP_AT_STAGING=1, p alone and
// p ln p recomputed by every tile pair that stages the entry: no spill, but k_pairs_tm 1654 -> 1718 ms on c5_5 (round 5): not the default.
#ifndef KTM_PLNP_AT_STAGING
#define KTM_PLNP_AT_STAGING 0
#endif
#if KTM_PLNP_AT_STAGING
typedef double ktm_p_t;
#else
typedef double2 ktm_p_t;
#endif
#define KTM_NT_MAX 256              // largest number of sample tiles the tile-major path handles

__global__ void __launch_bounds__(64 * KTM_WAVES)
k_tile_major(const SimkaSpan *spans, const ull *cursors, const ull *entries, const uint32_t *groups, SimkaPairCfg pc,
             ull *tm_ent, ktm_p_t *tm_p, uint32_t *tm_off) {
    __shared__ uint16_t s_gid[KTM_WAVES][SIMKA_SPAN_MAX];
    __shared__ uint32_t s_tb[KTM_WAVES][KTM_NT_MAX + 4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint16_t *gid = s_gid[wave];
    uint32_t *tb = s_tb[wave];
    const uint32_t T = pc.tile, nt = pc.ntiles;
    const ull nspans = cursors[2];
    const ull lt_mask = (1ull << lane) - 1ull;
    const bool cplx = pc.nacc64 != 0;
    for (ull sp = (ull)blockIdx.x * KTM_WAVES + wave; sp < nspans; sp += (ull)gridDim.x * KTM_WAVES) {
        const SimkaSpan span = spans[sp];
        uint32_t *off = tm_off + sp * (nt + 1u);
        if (span.ngrp == 0) { for (uint32_t t = lane; t <= nt; t += 64u) off[t] = 0u; continue; }
        const ull *ent = entries + span.ebase;
        // the group of every entry, the size of every tile segment
        for (uint32_t t = lane; t <= nt; t += 64u) tb[t] = 0u;
        for (uint32_t g = lane; g < span.ngrp; g += 64u) {
            const uint32_t d = groups[span.gbase + g];
            const uint32_t st = d >> 16, sz = d & 0xffffu;
            for (uint32_t e = 0; e < sz; e++) gid[st + e] = (uint16_t)g;
        }
        for (uint32_t i = lane; i < span.nent; i += 64u) atomicAdd(&tb[(uint32_t)(ent[i] >> 32) / T], 1u);
        // exclusive scan over the tiles (nt <= 256: four per lane)
        {
            const uint32_t t0 = lane * 4u;
            uint32_t c[4], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) { c[q] = (t0 + q < nt) ? tb[t0 + q] : 0u; sum += c[q]; }
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += v; }
            uint32_t run = incl - sum;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) { if (t0 + q < nt) { tb[t0 + q] = run; off[t0 + q] = run; } run += c[q]; }
            if (lane == 0) off[nt] = span.nent;
        }
        // stable placement: entries in index order, 64 at a time; lanes of the same tile take consecutive slots
        for (uint32_t i0 = 0; i0 < span.nent; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const bool valid = i < span.nent;
            const ull e = valid ? ent[i] : 0ull;
            const uint32_t smp = (uint32_t)(e >> 32);
            const uint32_t tile = valid ? smp / T : 0xffffffffu;
            uint32_t pos = 0;
            ull remaining = __ballot(valid);
            while (remaining) {
                const uint32_t l0 = (uint32_t)__ffsll((long long)remaining) - 1u;
                const uint32_t t0 = (uint32_t)__shfl((int)tile, (int)l0, 64);
                const ull m = __ballot(valid && tile == t0);
                const uint32_t base = tb[t0];
                if (tile == t0 && valid) pos = base + (uint32_t)__popcll(m & lt_mask);
                if (lane == l0) tb[t0] = base + (uint32_t)__popcll(m);
                remaining &= ~m;
            }
            if (valid) {
                tm_ent[span.ebase + pos] = ((ull)gid[i] << 48) | ((ull)(smp & 0xffffu) << 32) | (ull)(uint32_t)e;
                if (cplx) {
                    const double p = (double)(uint32_t)e / (double)pc.tot_n[smp];
#if KTM_PLNP_AT_STAGING
                    tm_p[span.ebase + pos] = p;
#else
                    tm_p[span.ebase + pos] = make_double2(p, simka_mul_rn(p, simka_fast_ln(p, g_simka_lntab)));      // (the logarithm of the pair loop: identical samples cancel exactly)
#endif
                }
            }
        }
    }
}

// One iteration of a block takes a BATCH of consecutive spans -- as many as fit the staging arrays (members of its two tiles
// <= span_cap, groups <= span_cap / 2) out of a range of KTM_RANGE span slots -- so the fixed costs of an iteration (barriers,
// the scan over the groups, the per-thread pair search) are shared by several spans' pairs.  Group ids are made batch-wide by
// adding the number of groups of the spans staged before.
#define KTM_RANGE 32
struct KtmSpan { ull ebase; uint32_t ngrp, maxc, a0, na, b0, nb; };      // a span as tile pair (I, J) sees it
struct KtmSlot { uint32_t st, mid, gbase, pad; };                        // staging start, start of the J members, first group id
// a range of span slots cut into batches by thread 0: batch b = spans [bstart[b], bstart[b+1]), nm / ng / maxc per batch
struct KtmRange {
    KtmSpan d[KTM_RANGE];
    KtmSlot sl[KTM_RANGE + 1];           // per span; sl[last of a batch + 1].st closes the batch (sentinel written per batch end)
    uint32_t bstart[KTM_RANGE + 1], bnm[KTM_RANGE], bng[KTM_RANGE], bmaxc[KTM_RANGE];
    uint32_t nbatch, cnt;
};

// CPLX: the -complex-dist accumulators (whit, klfix) and the (p, p ln p) staging are compiled in -- the simple-only instance keeps no
// prefetch registers for them.  The span a STAGED entry belongs to (index in the range table, < KTM_RANGE = 32) rides in the five top
// bits of the entry (the batch-wide group id below it needs 11 of its 16 bits).
template <bool CPLX>
__global__ void __launch_bounds__(K4_BLOCK_BIG)
k_pairs_tm(const SimkaSpan *spans, const ull *cursors, const ull *tm_ent, const ktm_p_t *tm_p, const uint32_t *tm_off, SimkaPairCfg pc,
           ull *acc) {
    constexpr int K4_BLOCK = K4_BLOCK_BIG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t CP = pc.ncell_pad, npk = pc.nacc32 >> 1;
    // (CPLX = true: the test stays a run-time one on purpose.  With a compile-time constant the complex part of the pair loop is
    // scheduled into the simple part and every s_waitcnt of the loop becomes lgkmcnt(0): k_pairs_tm 166 -> 174 ms on c5_50, 1652 -> 1740 ms
    // on c5_5; sched barriers between the parts, or all LDS reads of a pair ahead of its atomics, did not bring it back -- round 5)
    const bool cplx = CPLX && pc.nacc64 != 0;
    ull *pk = (ull *)(smem + SIMKA_LDS_HEAD);                    // [npk][CP]     packed u32 pairs
    ull *c64 = pk + (size_t)npk * CP;                            // [nacc64][CP]  (whit, klfix)
    const uint32_t EC = pc.span_cap, GC = EC / 2u;
    ull *ent = c64 + (size_t)pc.nacc64 * CP;                     // [EC]          per staged span: segment I, then segment J: (g<<48 | sample<<32 | count)
    double2 *epp = (double2 *)(ent + EC);                        // [EC]          complex: (p = c / N_sample, p ln p): one 16-byte read per entry
    double *eplp = (double *)epp + (cplx ? EC : 0);              //               (second half of that array: the layout below counts 2 x EC doubles)
    double *tn = eplp + (cplx ? EC : 0);                         // [SIMKA_PAIR_TN] complex: N of the samples of tile I, then tile J
    ull *tnu = (ull *)(tn + (cplx ? SIMKA_PAIR_TN : 0));         // [SIMKA_PAIR_TN] complex: the same N as integers
    double2 *lntab = (double2 *)(tnu + (cplx ? SIMKA_PAIR_TN : 0));    // [SIMKA_LNTAB] complex: simka_fast_ln
    uint32_t *gdesc = (uint32_t *)(lntab + (cplx ? SIMKA_LNTAB : 0));   // [GC]      run of the group in segment I: (start<<16 | size)
    uint32_t *gpref = gdesc + GC;                                // [GC+2]        pair prefix
    uint32_t *tmp = gpref + GC + 2;                              // [32]
    uint32_t *gdescB = tmp + 32;                                 // [GC]          run of the group in segment J
    uint32_t *runA = gdescB + GC;                                // [GC]          (head | tail<<16) written by the run's first / last entry
    uint32_t *runB = runA + GC;                                  // [GC]
    // range tables, double-buffered (the last batch of a range is still being paired while the next range is laid out)
    KtmRange *s_rng = (KtmRange *)(runB + GC);                   // [2]

    const uint32_t tid = threadIdx.x;
    const uint32_t N = pc.nb_samples, T = pc.tile, nt = pc.ntiles;
    uint32_t I = 0, J = 0;                                       // tile pair (I<=J) of this block row
    {
        uint32_t r = blockIdx.y;
        for (I = 0; I < nt; I++) { const uint32_t row = nt - I; if (r < row) { J = I + r; break; } r -= row; }
    }
    const bool rect = I != J;
    const uint32_t baseI = I * T, baseJ = J * T;
    for (uint32_t i = tid; i < npk * CP; i += K4_BLOCK) pk[i] = 0;
    for (uint32_t i = tid; i < pc.nacc64 * CP; i += K4_BLOCK) c64[i] = 0;
    if (cplx) {
        for (uint32_t i = tid; i < SIMKA_LNTAB; i += K4_BLOCK) lntab[i] = g_simka_lntab[i];
        for (uint32_t i = tid; i < T; i += K4_BLOCK) {
            const ull ni_ = (baseI + i < N) ? pc.tot_n[baseI + i] : 1ull;
            tn[i] = (double)ni_; tnu[i] = ni_;
            if (rect) { const ull nj_ = (baseJ + i < N) ? pc.tot_n[baseJ + i] : 1ull; tn[T + i] = (double)nj_; tnu[T + i] = nj_; }
        }
    }
    ull bound = 0, bound_q = 0;

    const ull nspans = cursors[2];
    const ull nranges = (nspans + KTM_RANGE - 1) / KTM_RANGE;
    constexpr int EPT = (SIMKA_SPAN_MAX + K4_BLOCK - 1) / K4_BLOCK;
    ull rq = blockIdx.x;                 // current range of span slots
    uint32_t rbuf = 1, rnext = 0;        // table of the current range, next batch of it
    bool started = false;
    // lay out range rq in table `t`: the spans as this tile pair sees them (spans without a pair for it count as empty), cut
    // into batches that fit the staging arrays.  Three barriers.
    auto load_range = [&](uint32_t t) {
        KtmRange *R = s_rng + t;
        __syncthreads();
        uint32_t cnt = 0;
        if (rq < nranges) {
            const ull s0 = rq * KTM_RANGE;
            cnt = (uint32_t)((nspans - s0 < (ull)KTM_RANGE) ? (nspans - s0) : (ull)KTM_RANGE);
            if (tid < cnt) {
                const SimkaSpan sp = spans[s0 + tid];
                const uint32_t *o = tm_off + (s0 + tid) * (nt + 1u);
                KtmSpan d; d.ebase = sp.ebase; d.ngrp = sp.ngrp; d.maxc = sp.maxc; d.a0 = 0; d.na = 0; d.b0 = 0; d.nb = 0;
                if (sp.ngrp) {
                    d.a0 = o[I]; d.na = o[I + 1u] - d.a0;
                    if (rect) { d.b0 = o[J]; d.nb = o[J + 1u] - d.b0; }
                    if (rect ? (d.na == 0u || d.nb == 0u) : d.na < 2u) { d.ngrp = 0; d.na = 0; d.nb = 0; }      // nothing to pair up here
                }
                R->d[tid] = d;
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t nb = 0, j = 0;
            while (j < cnt) {
                while (j < cnt && R->d[j].ngrp == 0u) { KtmSlot sl; sl.st = 0; sl.mid = 0; sl.gbase = 0; sl.pad = 0; R->sl[j] = sl; j++; }   // leading empty spans
                if (j >= cnt) break;
                uint32_t nm = 0, ng = 0, mc = 0;
                R->bstart[nb] = j;
                const uint32_t j0 = j;
                while (j < cnt) {
                    const KtmSpan d = R->d[j];
                    const uint32_t m = d.na + d.nb;
                    if (j > j0 && (nm + m > EC || ng + d.ngrp > GC)) break;
                    KtmSlot sl; sl.st = nm; sl.mid = nm + d.na; sl.gbase = ng; sl.pad = 0; R->sl[j] = sl;
                    nm += m; ng += d.ngrp; mc = d.maxc > mc ? d.maxc : mc;
                    j++;
                }
                R->bnm[nb] = nm; R->bng[nb] = ng; R->bmaxc[nb] = mc;
                nb++;
            }
            R->bstart[nb] = cnt;
            R->nbatch = nb; R->cnt = cnt;
        }
        __syncthreads();
    };
    // the next batch (uniform): spans [bf, bl) of table rt
    struct Batch { uint32_t rt, bf, bl, nm, ng, maxc; bool done; };
    auto pick = [&]() -> Batch {
        Batch b; b.rt = 0; b.bf = 0; b.bl = 0; b.nm = 0; b.ng = 0; b.maxc = 0; b.done = false;
        bool flipped = false;             // the table of the batch in flight must survive: switch tables once per call, then
        for (;;) {                        // ranges without a pair for this tile pair are laid out over each other
            if (!started) { started = true; rbuf ^= 1u; flipped = true; load_range(rbuf); rnext = 0; }
            else if (rnext >= s_rng[rbuf].nbatch) {
                if (rq >= nranges) { b.done = true; return b; }
                rq += gridDim.x;
                if (!flipped) { rbuf ^= 1u; flipped = true; }
                load_range(rbuf); rnext = 0;
            }
            const KtmRange *R = s_rng + rbuf;
            if (R->cnt == 0u) { b.done = true; return b; }
            if (rnext >= R->nbatch) continue;
            b.rt = rbuf; b.bf = R->bstart[rnext]; b.bl = R->bstart[rnext + 1u];
            // (a batch ends where the next one starts, or before the trailing empty spans: either way its members end at bnm)
            b.nm = R->bnm[rnext]; b.ng = R->bng[rnext]; b.maxc = R->bmaxc[rnext];
            rnext++;
            return b;
        }
    };
    ull pre_e[EPT]; ktm_p_t pre_p[CPLX ? EPT : 1]; uint32_t pre_j = 0;       // pre_j: five bits per prefetched entry
    static_assert(EPT * 5 <= 32, "span indices of the prefetched entries share one register");
    static_assert(KTM_RANGE <= 32 && SIMKA_SPAN_MAX / 2 <= (1 << 11), "span index and group id share the top 16 bits of a prefetched entry");
    // issue the loads of batch B; pre_j = the spans (index in the range table) the staged entries belong to
#define KTM_FETCH(B) {                                                                        \
    const KtmRange *R_ = s_rng + (B).rt;                                                      \
    pre_j = 0u;                                                                               \
    _Pragma("unroll") for (int q = 0; q < EPT; q++) {                                         \
        const uint32_t i = tid + (uint32_t)q * K4_BLOCK;                                      \
        pre_e[q] = 0ull; if (CPLX) pre_p[CPLX ? q : 0] = ktm_p_t();                           \
        if (i < (B).nm) {                                                                     \
            /* the span of staged entry i: the LAST span of the batch that starts at or before i (the staging starts are         \
               non-decreasing; a span without members shares its start with its successor, never with its predecessor's members). \
               Binary search: the linear walk over the spans of a batch cost the slowest thread 30 LDS round trips per entry. */     \
            uint32_t j = (B).bf, hi_ = (B).bl;                                                \
            while (hi_ - j > 1u) { const uint32_t mid_ = (j + hi_) >> 1; if (R_->sl[mid_].st <= i) j = mid_; else hi_ = mid_; }   \
            const KtmSlot sl = R_->sl[j];                                                     \
            const KtmSpan d = R_->d[j];                                                       \
            const ull src = d.ebase + (i < sl.mid ? d.a0 + (i - sl.st) : d.b0 + (i - sl.mid)); \
            pre_e[q] = tm_ent[src]; pre_j |= j << (5 * q);    /* (not OR-ed into the entry here: that would wait for the load) */ \
            if (CPLX) pre_p[CPLX ? q : 0] = tm_p[src];                                        \
        }                                                                                     \
    } }
    uint32_t it = 0;
    Batch nxt = pick();
    if (!nxt.done) KTM_FETCH(nxt)
    PP_DECL
    for (; !nxt.done; it++) {
        PP(6)
        __syncthreads();
        PP(0)
        const Batch cur = nxt;
        const KtmRange *CR = s_rng + cur.rt;
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            // (the span index stays in bits 59..63 of the staged entry: batch-wide group ids are below GC <= 2048)
            const uint32_t j_ = (pre_j >> (5 * q)) & 31u;
            if (i < cur.nm) { ent[i] = pre_e[q] + ((ull)(CR->sl[j_].gbase | (j_ << 11)) << 48); 
#if KTM_PLNP_AT_STAGING
                              if (CPLX) { const double p_ = pre_p[CPLX ? q : 0]; epp[i] = make_double2(p_, simka_mul_rn(p_, simka_fast_ln(p_, lntab))); } }
#else
                              if (CPLX) epp[i] = pre_p[CPLX ? q : 0]; }
#endif
            if (i < cur.ng) { runA[i] = 0u; runB[i] = 0u; }
        }
        // the batch after this one (may lay out the next range into the other table: barriers), then issue its loads
        nxt = pick();
        if (!nxt.done) KTM_FETCH(nxt)
        __syncthreads();                  // staging of `cur` complete
        PP(1)
        const ull add = (ull)cur.ng * (ull)cur.maxc;
        const ull addq = (ull)cur.ng * (ull)cur.maxc * (ull)cur.maxc;
        const bool chord_fast = cur.maxc < 46341u && addq < 0xffffffffull;
        if (bound + add >= 0xffffffffull || (chord_fast && bound_q + addq >= 0xffffffffull)) { pairs_flush<true, K4_BLOCK>(pk, c64, acc, pc, rect, baseI, baseJ); bound = 0; bound_q = 0; }
        bound += add;
        if (chord_fast) bound_q += addq;
        const bool smallc = chord_fast && cur.maxc < 32768u;       // every product of the span is a 32-bit value: the cheap chord / Hellinger update
        // group runs: the first entry of a run stores its index, the last one the index behind it.  Neighbouring segments of
        // different spans never share a group id; the I and J segments of one span may, hence the explicit boundary.
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            if (i < cur.nm) {
                const uint32_t gj = (uint32_t)(ent[i] >> 48), g = gj & 0x7ffu, j = gj >> 11;       // (neighbours are compared on group AND span)
                const KtmSlot sl = CR->sl[j];
                const bool inA = i < sl.mid;
                uint16_t *run = (uint16_t *)(inA ? runA : runB) + 2u * g;
                const uint32_t lo = inA ? sl.st : sl.mid, hi = inA ? sl.mid : sl.mid + CR->d[j].nb;
                if (i == lo || (uint32_t)(ent[i - 1u] >> 48) != gj) run[0] = (uint16_t)i;
                if (i + 1u == hi || (uint32_t)(ent[i + 1u] >> 48) != gj) run[1] = (uint16_t)(i + 1u);
            }
        }
        __syncthreads();
        PP(2)
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t g = tid + (uint32_t)q * K4_BLOCK;
            if (g < cur.ng) {
                const uint32_t ra = runA[g], rb = runB[g];
                const uint32_t a0 = ra & 0xffffu, nA = (ra >> 16) - a0, b0 = rb & 0xffffu, nB = (rb >> 16) - b0;
                gdesc[g] = (a0 << 16) | nA;
                gdescB[g] = (b0 << 16) | nB;
                gpref[g] = rect ? nA * nB : nA * (nA - 1u) / 2u;      // nA = 0 gives 0 either way
            }
        }
        __syncthreads();
        PP(3)
        const uint32_t P = block_excl_scan<K4_BLOCK>(gpref, cur.ng, tmp);
        if (tid == 0) gpref[cur.ng] = P;
        __syncthreads();
        PP(4)
        const uint32_t chunk = (P + K4_BLOCK - 1) / K4_BLOCK;
        uint32_t p = tid * chunk;
        const uint32_t pend = (p + chunk < P) ? p + chunk : P;
        if (p < pend) {
            uint32_t lo = 0, hi = cur.ng;    // largest g with gpref[g] <= p
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (gpref[mid] <= p) lo = mid; else hi = mid; }
            uint32_t g = lo;
            while (g + 1 < cur.ng && gpref[g + 1] <= p) g++;   // skip groups without pairs
            uint32_t d = gdesc[g];
            uint32_t a0 = d >> 16, nA = d & 0xffffu, b0 = 0, nB = 0, x, y;
            if (rect) {
                const uint32_t dB = gdescB[g]; b0 = dB >> 16; nB = dB & 0xffffu;
                const uint32_t r = p - gpref[g];
                x = r / nB; y = r - x * nB;
            } else tri_unrank(p - gpref[g], nA, x, y);
            uint32_t ix = a0 + x;
            ull ex = ent[ix];
            PP(5)
            // two instances of the pair loop (as in k_pairs, round 6): FAST = -simple-dist with every product of the span a 32-bit value, the two
            // wave-uniform flags compile-time constants; the -complex-dist test stays a run-time one (a compile-time `cplx` reschedules the loop
            // for the worse: round 5)
            auto pair_loop = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                const bool simple_ = FAST ? true : (pc.simple != 0u), smallc_ = FAST ? true : smallc;
                for (; p < pend; p++) {
                    const uint32_t iy = rect ? b0 + y : a0 + y;
                    const ull ey = ent[iy];
                    uint32_t si = (uint32_t)(ex >> 32) & 0xffffu, sj = (uint32_t)(ey >> 32) & 0xffffu;
                    uint32_t ci = (uint32_t)ex, cj = (uint32_t)ey;
                    if (!rect && si > sj) { uint32_t t_ = si; si = sj; sj = t_; t_ = ci; ci = cj; cj = t_; }
                    const uint32_t li = si - baseI, lj = sj - baseJ;
                    const uint32_t cell = rect ? li * T + lj : li * T - ((li * (li + 1u)) >> 1) + (lj - li - 1u);
                    atomicAdd(&pk[0 * CP + cell], (ull)ci | ((ull)cj << 32));                       // S_ij | S_ji
                    atomicAdd(&pk[1 * CP + cell], 1ull | ((ull)(ci < cj ? ci : cj) << 32));         // a | bc
                    if (simple_) {
                        if (smallc_) {       // (uniform per span) counts below 2^15: the product is a 32-bit value below 2^30
                            const uint32_t prod32 = ci * cj;
                            atomicAdd(&pk[2 * CP + cell], (ull)prod32 | ((ull)pair_isqrt32(prod32) << 32));  // chord | hell
                        } else {
                            const ull prod = (ull)ci * (ull)cj;
                            const ull hell = (ull)pair_isqrt(prod) << 32;
                            if (chord_fast) atomicAdd(&pk[2 * CP + cell], (ull)(uint32_t)prod | hell);  // chord | hell
                            else {   // huge counts: the product goes straight to the global u64 cell
                                atomicAdd(&pk[2 * CP + cell], hell);
                                atomicAdd(&acc[SIMKA_ACC_CHORD * pc.nb_pairs + simka_pair_index(si, sj, N)], prod);
                            }
                        }
                    }
                    if (cplx) {
                        // same arithmetic as k_pairs (ref: src/core/SimkaAlgorithm.hpp:437-446,477-481)
                        const double2 px = epp[ix], py = epp[iy];
                        const double h = px.x + py.x;
                        double dd = simka_add_rn(px.y, py.y) - simka_mul_rn(h, simka_fast_ln(h * 0.5, lntab));      // (products rounded on their own, never fused into the subtraction: identical samples give 2 a - 2 a = 0 exactly)
                        dd = dd < 0.0 ? 0.0 : dd;
                        atomicAdd(&c64[1 * CP + cell], (ull)(long long)llrint(dd * SIMKA_KL_SCALE));
                        const uint32_t jn = (rect ? T : 0u) + lj;
                        atomicAdd(&c64[0 * CP + cell], simka_whit_term(ci, cj, tnu[li], tnu[jn], tn[li], tn[jn]));
                    }
                    y++;
                    if (rect ? (y == nB) : (y == nA)) {
                        x++; y = rect ? 0u : x + 1u;
                        if (rect ? (x == nA) : (y >= nA)) {   // group exhausted
                            g++;
                            while (g < cur.ng && gpref[g + 1] == gpref[g]) g++;
                            if (g >= cur.ng) break;
                            d = gdesc[g]; a0 = d >> 16; nA = d & 0xffffu; x = 0; y = rect ? 0u : 1u;
                            if (rect) { const uint32_t dB = gdescB[g]; b0 = dB >> 16; nB = dB & 0xffffu; }
                        }
                        ix = a0 + x;
                        ex = ent[ix];
                    }
                }
            };
            if (pc.simple && smallc) pair_loop(std::true_type()); else pair_loop(std::false_type());
        }
    }
#undef KTM_FETCH
    PP(6)
    PP_FLUSH
    __syncthreads();
    pairs_flush<true, K4_BLOCK>(pk, c64, acc, pc, rect, baseI, baseJ);
}

// K3c  k_pairs_global: groups shared by more than K3_CAP samples (k_group's huge list).  Pairs are enumerated from the
// CSR entries in global memory and every update is a global u64 atomic; same arithmetic as k_pairs.
// grid: x = slices of one group's pair space, y = groups (strided).
__global__ void __launch_bounds__(256)
k_pairs_global(const SimkaSpan *huge, const ull *cursors, const ull *entries, SimkaPairCfg pc, ull *acc) {
    const ull nh = cursors[3];
    const uint32_t N = pc.nb_samples;
    const ull NP = pc.nb_pairs;
    for (ull h = blockIdx.y; h < nh; h += gridDim.y) {
        const SimkaSpan sp = huge[h];
        const ull *ent = entries + sp.ebase;
        const uint32_t s_ = sp.nent;                               // <= 65535 samples: s(s-1)/2 < 2^32
        const uint32_t P = (uint32_t)(((ull)s_ * (s_ - 1u)) >> 1);
        const uint32_t nthr = gridDim.x * 256u;
        const uint32_t chunk = (P + nthr - 1u) / nthr;
        const ull p0 = (ull)(blockIdx.x * 256u + threadIdx.x) * chunk;
        if (p0 >= P) continue;
        const uint32_t pend = (p0 + chunk < P) ? (uint32_t)p0 + chunk : P;
        uint32_t x, y;
        tri_unrank((uint32_t)p0, s_, x, y);
        ull ex = ent[x];
        for (uint32_t p = (uint32_t)p0; p < pend; p++) {
            const ull ey = ent[y];
            uint32_t si = (uint32_t)(ex >> 32), sj = (uint32_t)(ey >> 32);
            uint32_t ci = (uint32_t)ex, cj = (uint32_t)ey;
            if (si > sj) { uint32_t t_ = si; si = sj; sj = t_; t_ = ci; ci = cj; cj = t_; }
            const ull pg = simka_pair_index(si, sj, N);
            atomicAdd(&acc[SIMKA_ACC_SIJ * NP + pg], (ull)ci);
            atomicAdd(&acc[SIMKA_ACC_SJI * NP + pg], (ull)cj);
            atomicAdd(&acc[SIMKA_ACC_A * NP + pg], 1ull);
            atomicAdd(&acc[SIMKA_ACC_BC * NP + pg], (ull)(ci < cj ? ci : cj));
            if (pc.simple) {
                const ull prod = (ull)ci * (ull)cj;
                atomicAdd(&acc[SIMKA_ACC_CHORD * NP + pg], prod);
                atomicAdd(&acc[SIMKA_ACC_HELL * NP + pg], (ull)pair_isqrt(prod));
            }
            if (pc.nacc64) {
                const double Ni = (double)pc.tot_n[si], Nj = (double)pc.tot_n[sj];
                const double pi_ = (double)ci / Ni, pj_ = (double)cj / Nj, hh = pi_ + pj_;
                double dd = simka_add_rn(simka_mul_rn(pi_, simka_fast_ln(pi_, g_simka_lntab)), simka_mul_rn(pj_, simka_fast_ln(pj_, g_simka_lntab))) - simka_mul_rn(hh, simka_fast_ln(hh * 0.5, g_simka_lntab));       // same form, logarithm and roundings as k_pairs
                dd = dd < 0.0 ? 0.0 : dd;
                atomicAdd(&acc[((ull)pc.nacc32 + 1) * NP + pg], (ull)(long long)llrint(dd * SIMKA_KL_SCALE));
                const ull uX = (ull)((double)ci * Nj), uY = (ull)((double)cj * Ni);
                atomicAdd(&acc[(ull)pc.nacc32 * NP + pg], simka_whit_abs(uX - uY) - simka_whit_abs(uX) - simka_whit_abs(uY));
            }
            y++;
            if (y == s_) { x++; y = x + 1u; ex = ent[x < s_ ? x : s_ - 1u]; }
        }
    }
}

// --------------------------------------------------------------------------------------------
// k_big_counts: (sample, count) of every solid record with count >= SIMKA_HIST_MAX, collected from the resident spectra.
// -complex-dist keeps such counts on a list next to the per-sample histogram; the count kernels fill a fixed-size list, and
// when it overflows (deep samples with high-copy sequences) the list is rebuilt here at its exact size.
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_big_counts(const uint32_t *solid_counts, const ull *sample_base, const uint32_t *foff, const uint32_t *fcnt, uint32_t nparts, uint32_t *list,
             ull *cursor, ull cap) {
    const uint32_t s = blockIdx.y;
    const ull base = sample_base[s];
    const uint32_t *fo = foff + (size_t)s * nparts, *fc = fcnt + (size_t)s * nparts;
    for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        const uint32_t n = fc[p];
        const ull src = base + fo[p];
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint32_t c = solid_counts[src + i];
            if (c >= SIMKA_HIST_MAX) { const ull w = atomicAdd(cursor, 1ull); if (w < cap) { list[2 * w] = s; list[2 * w + 1] = c; } }
        }
    }
}

// --------------------------------------------------------------------------------------------
// k_import_hist: -complex-dist histogram of the solid counts of IMPORTED spectra (simka_import_samples_device): sample s0 + blockIdx.y,
// partitions [p_lo, p_lo + p_w) of the arena.  What the count kernels do record by record (count_hist) for the samples they count.
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_import_hist(const uint32_t *solid_counts, const ull *sample_base, const uint32_t *foff, const uint32_t *fcnt, uint32_t nparts, uint32_t p_lo, uint32_t p_w,
              uint32_t s0, ull *hist, uint32_t *ovf_list, ull *ovf_cursor, ull ovf_cap) {
    __shared__ uint32_t lh[SIMKA_HIST_MAX];
    const uint32_t s = s0 + blockIdx.y;
    for (uint32_t i = threadIdx.x; i < SIMKA_HIST_MAX; i += 256) lh[i] = 0;
    __syncthreads();
    const ull base = sample_base[s];
    const uint32_t *fo = foff + (size_t)s * nparts, *fc = fcnt + (size_t)s * nparts;
    for (uint32_t p = p_lo + blockIdx.x; p < p_lo + p_w; p += gridDim.x) {
        const uint32_t n = fc[p];
        const ull src = base + fo[p];
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint32_t c = solid_counts[src + i];
            if (c < SIMKA_HIST_MAX) atomicAdd(&lh[c], 1u);
            else { const ull w = atomicAdd(ovf_cursor, 1ull); if (w < ovf_cap) { ovf_list[2 * w] = s; ovf_list[2 * w + 1] = c; } }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SIMKA_HIST_MAX; i += 256) if (lh[i]) atomicAdd(&hist[(size_t)s * SIMKA_HIST_MAX + i], (ull)lh[i]);
}

// --------------------------------------------------------------------------------------------
// k_gather_sample: the arena records of one sample, partition-major and gap-free (simka_export_sample)
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_sample(const ull *solid_keys, const uint32_t *solid_counts, const ull *sample_base, const uint32_t *foff, const uint32_t *fcnt,
                const ull *out_off, uint32_t nparts, ull *out_keys, uint32_t *out_counts) {
    const ull base = *sample_base;
    for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        const uint32_t n = fcnt[p];
        const ull src = base + foff[p], dst = out_off[p];
        for (uint32_t i = threadIdx.x; i < n; i += 256) { out_keys[dst + i] = solid_keys[src + i]; out_counts[dst + i] = solid_counts[src + i]; }
    }
}

// many samples at once (blockIdx.y = sample slot), destination offsets chosen by the caller per (sample, partition)
__global__ void __launch_bounds__(256)
k_gather_samples(const ull *solid_keys, const uint32_t *solid_counts, const ull *sample_base, const uint32_t *foff, const uint32_t *fcnt,
                 const uint32_t *samples, const ull *out_off, uint32_t nparts, ull *out_keys, uint32_t *out_counts) {
    const uint32_t j = blockIdx.y, s = samples[j];
    const ull base = sample_base[s];
    const uint32_t *fo = foff + (size_t)s * nparts, *fc = fcnt + (size_t)s * nparts;
    const ull *oo = out_off + (size_t)j * nparts;
    for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        const uint32_t n = fc[p];
        const ull src = base + fo[p], dst = oo[p];
        for (uint32_t i = threadIdx.x; i < n; i += 256) { out_keys[dst + i] = solid_keys[src + i]; out_counts[dst + i] = solid_counts[src + i]; }
    }
}

// --------------------------------------------------------------------------------------------
// The tables of the spectrum exchange (sample shards), computed on the device: with 2^19 partitions the per-(sample, partition) tables
// are tens of megabytes per rank, and the prefix sums over them on the host (numpy / C++ loops) cost ten times what the device
// needs to move the records themselves.
//   k_range_rowsum   records of sample slot j bound for rank g (the partitions [bounds[g], bounds[g + 1]))
//   k_range_offsets  destination offset of every run (slot j, partition p) in the destination-major send buffer
//                    [g][slot j][partitions of g] from the starts of the (j, g) rows; the run lengths go to meta[g][j][p - lo_g]
//   k_import_tables  receive side: slot (r, j) of the received block = sample slot_samples[r * nb_slots + j]: foff / fcnt of the
//                    sample's runs in my partition range (the runs of a slot are contiguous), the slot's total
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_range_rowsum(const uint32_t *fcnt, const uint32_t *samples, uint32_t nparts, uint32_t nb_ranges, ull *rows) {
    const uint32_t g = blockIdx.x, j = blockIdx.y;
    const uint32_t lo = (uint32_t)(((ull)nparts * g) / nb_ranges), hi = (uint32_t)(((ull)nparts * (g + 1u)) / nb_ranges);
    const uint32_t *fc = fcnt + (size_t)samples[j] * nparts;
    ull acc = 0;
    for (uint32_t p = lo + threadIdx.x; p < hi; p += 256) acc += fc[p];
    __shared__ ull s_part[256];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t st = 128; st; st >>= 1) { if (threadIdx.x < st) s_part[threadIdx.x] += s_part[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) rows[(size_t)j * nb_ranges + g] = s_part[0];
}

__global__ void __launch_bounds__(1024)
k_range_offsets(const uint32_t *fcnt, const uint32_t *samples, uint32_t nparts, uint32_t nb_ranges, const ull *starts, ull *xoff,
                int32_t *meta, uint32_t nb_slots, uint32_t width) {
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const uint32_t *fc = fcnt + (size_t)samples[j] * nparts;
    ull *xo = xoff + (size_t)j * nparts;
    __shared__ uint32_t tmp[16];
    __shared__ ull s_carry;
    for (uint32_t g = 0; g < nb_ranges; g++) {
        const uint32_t lo = (uint32_t)(((ull)nparts * g) / nb_ranges), hi = (uint32_t)(((ull)nparts * (g + 1u)) / nb_ranges);
        if (tid == 0) s_carry = starts[(size_t)j * nb_ranges + g];
        __syncthreads();
        int32_t *mrow = meta ? meta + ((size_t)g * nb_slots + j) * width : nullptr;
        for (uint32_t p0 = lo; p0 < hi; p0 += 1024) {
            const uint32_t p = p0 + tid;
            const uint32_t c = p < hi ? fc[p] : 0u;
            uint32_t excl;
            const uint32_t tot = block_excl_scan1<1024>(c, excl, tmp);
            const ull carry = s_carry;
            if (p < hi) { xo[p] = carry + excl; if (mrow) mrow[p - lo] = (int32_t)c; }
            __syncthreads();
            if (tid == 0) s_carry = carry + tot;
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(1024)
k_import_tables(const int32_t *meta, const uint32_t *slot_samples, uint32_t width, uint32_t w, uint32_t nparts, uint32_t p_lo,
                uint32_t *foff, uint32_t *fcnt, ull *slot_total) {
    const uint32_t slot = blockIdx.x, tid = threadIdx.x;
    const uint32_t s = slot_samples[slot];
    if (s == 0xffffffffu) { if (tid == 0) slot_total[slot] = 0; return; }
    const int32_t *mrow = meta + (size_t)slot * width;
    uint32_t *fo = foff + (size_t)s * nparts + p_lo, *fc = fcnt + (size_t)s * nparts + p_lo;
    __shared__ uint32_t tmp[16];
    __shared__ ull s_carry;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < w; p0 += 1024) {
        const uint32_t p = p0 + tid;
        const uint32_t c = p < w ? (uint32_t)mrow[p] : 0u;
        uint32_t excl;
        const uint32_t tot = block_excl_scan1<1024>(c, excl, tmp);
        const ull carry = s_carry;
        if (p < w) { fo[p] = c ? (uint32_t)(carry + excl) : 0u; fc[p] = c; }      // (a slot holds < 2^32 records: checked on the host from its total)
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) slot_total[slot] = s_carry;
}

// --------------------------------------------------------------------------------------------
// synthetic data (bench / test utility): genome pool and error-bearing reads, 2-bit packed
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_synth_genomes(uint64_t *pool, uint32_t nb_genomes, uint64_t genome_words, uint64_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nb_genomes * genome_words) return;
    const uint64_t gidx = i / genome_words, w = i % genome_words;
    pool[i] = simka_rng(seed ^ (gidx * 0xD1B54A32D192ED03ULL), w);   // 32 i.i.d. uniform bases per word
}

// one thread per OUTPUT word (32 bases), which may straddle two reads
__global__ void __launch_bounds__(256)
k_synth_reads(uint64_t *packed, uint64_t nb_reads, uint32_t L, const uint64_t *pool, uint64_t genome_words,
              uint64_t genome_len, const uint32_t *genome_ids, const uint32_t *cdf, uint32_t nb_sel, uint64_t seed,
              uint32_t err_thr) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nb_bases = nb_reads * (uint64_t)L;
    if (w * 32 >= nb_bases) return;
    uint64_t out = 0;
    uint64_t cur_read = ~0ull, gbase = 0, start = 0;
    uint32_t strand = 0;
    for (uint32_t bi = 0; bi < 32; bi++) {
        const uint64_t b = w * 32 + bi;
        if (b >= nb_bases) break;
        const uint64_t r = b / L;
        const uint32_t i = (uint32_t)(b - r * L);
        if (r != cur_read) {
            cur_read = r;
            const uint64_t h0 = simka_rng(seed, 2 * r);
            const uint64_t h1 = simka_rng(seed, 2 * r + 1);
            const uint32_t u = (uint32_t)(h0 >> 32);
            uint32_t sel = 0;
            while (sel + 1 < nb_sel && u >= cdf[sel]) sel++;     // cdf[] = cumulative weights * 2^32
            gbase = (uint64_t)genome_ids[sel] * genome_words;
            start = ((h0 & 0xffffffffull) * (genome_len - L + 1)) >> 32;
            strand = (uint32_t)(h1 & 1u);
        }
        const uint64_t gp = strand ? (start + (L - 1u - i)) : (start + i);
        uint32_t c = (uint32_t)(pool[gbase + (gp >> 5)] >> ((gp & 31u) * 2u)) & 3u;
        if (strand) c ^= 2u;
        const uint64_t he = simka_rng(seed ^ 0xA5A5A5A5A5A5A5A5ULL, b);
        if ((uint32_t)(he & 0xffffu) < err_thr) c = (c + 1u + (uint32_t)((he >> 16) % 3u)) & 3u;   // substitution
        out |= (uint64_t)c << (bi * 2u);
    }
    packed[w] = out;
}
