// simka_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Simka hot path.
//
// Data flow per sample (count side, replaces gatb SortingCountAlgorithm + the
// SimkaCompressedProcessor plugin, ref: src/SimkaCount.cpp:291-297, src/minikc/MiniKC.hpp:54-79):
//
//   packed reads --k_scan<false>--> level-1 histogram --k_layout--> bucket offsets
//                --k_scan<true>---> level-1 buckets of keys (LDS-staged multisplit, coalesced runs)
//                --k_split--------> every 8192-key chunk partitioned IN PLACE by level-2 bits
//                --k_count--------> per partition: LDS hash table (CAS insert) -> abundance filter
//                                   -> solid (key,count) records in the HBM arena + D/N/Q totals
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
    uint32_t v = s;                                   // inclusive scan across the 64 lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
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

// --------------------------------------------------------------------------------------------
// K1  k_scan: packed reads -> canonical k-mer keys -> level-1 histogram / level-1 buckets
// One thread owns K1_SEG consecutive k-mer START positions of the concatenated base array and
// rolls forward/reverse-complement words over its 64-base register window.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t window_base(uint64_t A, uint64_t B, uint32_t i) {
    const uint64_t w = (i < 32u) ? A : B;
    return (uint32_t)(w >> ((i & 31u) * 2u)) & 3u;
}

// k_tile_reads: variable-length reads.  tile_r0[b] = index of the read that holds base b * tile (largest r with
// offsets[r] <= b * tile), one thread per tile; k_scan stages the read starts of its tile in LDS from it.
__global__ void __launch_bounds__(256)
k_tile_reads(const uint64_t *offsets, uint64_t nb_reads, uint64_t nb_bases, uint32_t ntiles, uint64_t tile, uint32_t *tile_r0) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > ntiles) return;
    const uint64_t pos = (uint64_t)b * tile;
    uint64_t lo = 0, hi = nb_reads;
    if (pos >= nb_bases) { tile_r0[b] = (uint32_t)(nb_reads ? nb_reads - 1 : 0); return; }
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid] <= pos) lo = mid; else hi = mid; }
    tile_r0[b] = (uint32_t)lo;
}

// SHARDED: the context owns a subset of the level-1 buckets (partition shards); otherwise every k-mer is kept and the
// ownership test (with its integer modulo for non power-of-two shard counts) is not even compiled in.
template <bool SCATTER, bool FIXED, bool SHARDED>
__global__ void __launch_bounds__(K1_BLOCK)
k_scan(SimkaScanArgs a, SimkaKeyCfg cfg, ull *b1_count, ull *b1_cursor, uint64_t *l1_keys, ull *kocc, const ull *b1_limit,
       uint32_t *ovf_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t B1 = 1u << cfg.l1;
    // all LDS lives in the dynamic region (16-B aligned base, guide G17)
    uint32_t *hist = (uint32_t *)(smem + SIMKA_LDS_HEAD);   // [B1+32]: slots B1.. swallow the atomics of invalid positions
    uint32_t *loff = hist + B1 + 32;                   // [B1]
    uint32_t *tmp = loff + B1;                         // [K1_BLOCK]
    ull *gbase = (ull *)(tmp + K1_BLOCK);              // [B1]
    uint64_t *stage = (uint64_t *)(gbase + B1);        // [K1_BLOCK*K1_SEG]   (SCATTER only)
    uint32_t *rtab = (uint32_t *)(stage + (SCATTER ? K1_BLOCK * K1_SEG : 0));   // [K1_RTAB] (!FIXED) read starts after the tile's first base, relative to it

    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < B1 + 32; i += K1_BLOCK) hist[i] = 0;
    // variable-length reads: the starts of the reads that begin inside this tile (+ look-ahead), relative to the tile start
    const uint64_t T0 = (uint64_t)blockIdx.x * (K1_BLOCK * K1_SEG);
    uint32_t ntab = 0;              // 0: table not usable (too many reads in the tile) -> global binary search per thread
    if (!FIXED && a.tile_r0) {
        const uint64_t r0 = a.tile_r0[blockIdx.x], r1 = a.tile_r0[blockIdx.x + 1];
        // entries offsets[r0+1 .. r1+64] (clamped to offsets[nb_reads] = nb_bases): every read start in (T0, T0 + tile + look-ahead]
        uint64_t last = r1 + 64;
        if (last > a.nb_reads) last = a.nb_reads;
        const uint64_t cnt = last > r0 ? last - r0 : 0;
        // (zero-length reads could put more than 64 starts into the look-ahead: then the last staged start is too close)
        const bool covers = cnt > 0 && (last == a.nb_reads || a.offsets[last] - T0 > (uint64_t)(K1_BLOCK * K1_SEG + K1_SEG + 32u));
        if (covers && cnt <= K1_RTAB) {
            ntab = (uint32_t)cnt;
            for (uint32_t i = tid; i < ntab; i += K1_BLOCK) {
                const uint64_t d = a.offsets[r0 + 1 + i] - T0;
                rtab[i] = d < 0xffffffffull ? (uint32_t)d : 0xffffffffu;
            }
        }
    }
    __syncthreads();

    const uint64_t w0 = ((uint64_t)blockIdx.x * K1_BLOCK + tid) * K1_SEG;
    uint64_t keys[K1_SEG];
    uint32_t ranks[K1_SEG / 2];      // rank inside the tile's bucket run (< 8192): two u16 per register
#pragma unroll
    for (int q = 0; q < K1_SEG; q++) keys[q] = SIMKA_EMPTY_KEY;
#pragma unroll
    for (int q = 0; q < K1_SEG / 2; q++) ranks[q] = 0;

    if (w0 < a.nb_bases) {
        const uint64_t wi = w0 >> 5;
        const uint32_t sh = (uint32_t)(w0 & 31u) * 2u;
        const uint64_t lastw = a.nb_words - 1;
        const uint64_t W0 = a.packed[wi < lastw ? wi : lastw];
        const uint64_t W1 = a.packed[wi + 1 < lastw ? wi + 1 : lastw];
        const uint64_t W2 = a.packed[wi + 2 < lastw ? wi + 2 : lastw];
        const uint64_t A = sh ? ((W0 >> sh) | (W1 << (64u - sh))) : W0;
        const uint64_t B = sh ? ((W1 >> sh) | (W2 << (64u - sh))) : W1;

        // the read (fragment) that contains base w0, and where the next one starts
        uint64_t rd, next;
        uint32_t ti = 0;                 // (!FIXED, staged table) index of the next read start
        if (FIXED) {
            rd = w0 / a.fixed_len;
            next = (rd + 1) * (uint64_t)a.fixed_len;
        } else if (ntab) {
            // first staged read start beyond w0 (the table holds every start in (T0, w0 + look-ahead])
            const uint32_t w0rel = (uint32_t)(w0 - T0);
            uint32_t lo = 0, hi = ntab;          // smallest i with rtab[i] > w0rel; rtab[ntab-1] > w0rel unless the data end there
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rtab[mid] > w0rel) hi = mid; else lo = mid + 1; }
            ti = lo < ntab ? lo : ntab - 1u;
            rd = 0;
            next = T0 + rtab[ti];
        } else {
            uint64_t lo = 0, hi = a.nb_reads;
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (a.offsets[mid] <= w0) lo = mid; else hi = mid;
            }
            rd = lo;
            next = a.offsets[rd + 1];
        }
        // positions are handled relative to w0 in 32 bits: nrel = first position of the next read, erel = end of data
        const uint32_t SPAN = K1_SEG + 32u;                                  // > SEG + k - 1
        uint32_t nrel = (next - w0 < (uint64_t)SPAN) ? (uint32_t)(next - w0) : SPAN;
        const uint32_t erel = (a.nb_bases - w0 < (uint64_t)SPAN) ? (uint32_t)(a.nb_bases - w0) : SPAN;

        // No rolling update: the forward k-mer starting at window base q is a static funnel shift of (A,B); its reverse
        // complement is a static funnel shift of the reverse-complemented window (R0,R1), built ONCE per thread.
        //   R = revcomp(window bases 0 .. Wn-1), Wn = SEG + k - 1: base p of R = complement(window base Wn-1-p), so the
        //   reverse complement of the k-mer at q starts at base SEG-1-q of R -- independent of k.
        const uint32_t k = cfg.k;
        const uint64_t M5 = 0x5555555555555555ull, MA = 0xAAAAAAAAAAAAAAAAull;
        uint64_t ra = __brevll(A), rb = __brevll(B);
        ra = (((ra >> 1) & M5) | ((ra & M5) << 1)) ^ MA;                     // bases 31..0 complemented (code ^ 2)
        rb = (((rb >> 1) & M5) | ((rb & M5) << 1)) ^ MA;                     // bases 63..32
        const uint32_t rs = 2u * (64u - (K1_SEG + k - 1u));                  // 36 .. 96 bits: drop the bases beyond Wn
        const uint64_t R0 = (rs >= 64u) ? (ra >> (rs - 64u)) : ((rb >> rs) | (ra << (64u - rs)));
        const uint64_t R1 = (rs >= 64u) ? 0ull : (ra >> rs);
        const uint32_t aw[3] = { (uint32_t)A, (uint32_t)(A >> 32), (uint32_t)B };            // bases 0..47 of the window: positions q < 16 need 94 bits
        const uint32_t rw[3] = { (uint32_t)R0, (uint32_t)(R0 >> 32), (uint32_t)R1 };
#pragma unroll
        for (int q = 0; q < K1_SEG; q++) {
            if (FIXED) { const bool nb_ = (uint32_t)q >= nrel; nrel = nb_ ? nrel + a.fixed_len : nrel; }
            else if (ntab) while ((uint32_t)q >= nrel && ti + 1u < ntab) { ti++; const uint64_t nx_ = T0 + rtab[ti] - w0; nrel = nx_ < (uint64_t)SPAN ? (uint32_t)nx_ : SPAN; if (nx_ >= (uint64_t)SPAN) break; }
            else while ((uint32_t)q >= nrel && rd + 1 < a.nb_reads) { rd++; const uint64_t nx_ = a.offsets[rd + 1] - w0; nrel = nx_ < (uint64_t)SPAN ? (uint32_t)nx_ : SPAN; if (nx_ >= (uint64_t)SPAN) break; }
            // compile-time shifts < 32 bits after unrolling: each 64-bit extract is two v_alignbit_b32 over the 32-bit window words
            const uint32_t s1 = 2u * (uint32_t)q, s2 = 2u * (uint32_t)(K1_SEG - 1 - q);
            static_assert(2 * (K1_SEG - 1) < 32, "funnel shifts stay inside one 32-bit word");
            const uint64_t fwd = (((uint64_t)__builtin_amdgcn_alignbit(aw[2], aw[1], s1) << 32) | __builtin_amdgcn_alignbit(aw[1], aw[0], s1)) & cfg.mask;
            const uint64_t rev = (((uint64_t)__builtin_amdgcn_alignbit(rw[2], rw[1], s2) << 32) | __builtin_amdgcn_alignbit(rw[1], rw[0], s2)) & cfg.mask;
            const uint64_t canon = fwd < rev ? fwd : rev;
            const uint64_t key = simka_mix(canon, cfg.mask, cfg.xs);
            const uint32_t b1 = simka_key_l1(key, cfg);
            // the k-mer [q, q+k) must end inside its read (and inside the data)
            const bool ok = ((uint32_t)q + k <= nrel) & ((uint32_t)q < erel) & (SHARDED ? simka_owns_l1(b1, cfg) : true);
            const uint32_t rk = atomicAdd(&hist[ok ? b1 : B1 + (tid & 31u)], 1u);   // invalid positions hit a trash slot: no branch
            if (SCATTER) {
                keys[q] = ok ? key : SIMKA_EMPTY_KEY;
                ranks[q >> 1] |= (ok ? rk : 0u) << ((q & 1) * 16);
            }
        }
    }

    if (!SCATTER) {
        __syncthreads();
        for (uint32_t b = tid; b < B1; b += K1_BLOCK) {
            const uint32_t h = hist[b];
            if (h) atomicAdd(&b1_count[b], (ull)h);
        }
        return;
    }

    __syncthreads();
    // reserve this tile's run in every bucket: one returning global atomic per bucket (B1 <= K1_BLOCK: one per thread).  Its
    // result is only needed by the copy-out, so it stays in a register while the block scans and stages (latency hidden).
    constexpr int NBK = (1024 + K1_BLOCK - 1) / K1_BLOCK;       // level-1 buckets per thread (B1 <= 1024)
    ull gb[NBK]; uint32_t myh[NBK];
#pragma unroll
    for (int u = 0; u < NBK; u++) {
        const uint32_t b = tid + (uint32_t)u * K1_BLOCK;
        gb[u] = 0; myh[u] = 0;
        if (b < B1) {
            myh[u] = hist[b];
            loff[b] = myh[u];
            if (myh[u]) gb[u] = atomicAdd(&b1_cursor[b], (ull)myh[u]);
        }
    }
    __syncthreads();
    const uint32_t total = block_excl_scan<K1_BLOCK>(loff, B1, tmp);
#pragma unroll
    for (int q = 0; q < K1_SEG; q++) {
        if (keys[q] != SIMKA_EMPTY_KEY) stage[loff[simka_key_l1(keys[q], cfg)] + ((ranks[q >> 1] >> ((q & 1) * 16)) & 0xffffu)] = keys[q];
    }
#pragma unroll
    for (int u = 0; u < NBK; u++) {
        const uint32_t b = tid + (uint32_t)u * K1_BLOCK;
        if (b < B1) {
            // capacity-sized buckets (no histogram pass): a run that does not fit flags the sample for the exact path
            if (b1_limit && myh[u] && gb[u] + myh[u] > b1_limit[b]) { *ovf_flag = 1u; gb[u] = ~0ull; }
            gbase[b] = gb[u];
        }
    }
    __syncthreads();
    // coalesced copy-out: consecutive staged slots of one bucket go to consecutive HBM addresses
    for (uint32_t t = tid; t < total; t += K1_BLOCK) {
        const uint64_t key = stage[t];
        const uint32_t b = simka_key_l1(key, cfg);
        const ull gb = gbase[b];
#ifdef SIMKA_DEBUG_BOUNDS
        if (gb != ~0ull && (b >= B1 || gb + (t - loff[b]) >= a.nb_words * 64)) {
            printf("k_scan OOB: blk %u t %u total %u b %u gb %llu loff %u key %llx hist %u\n", blockIdx.x, t, total, b, gb, loff[b], (ull)key, hist[b < B1 ? b : 0]);
            continue;
        }
#endif
        if (gb != ~0ull) l1_keys[gb + (t - loff[b])] = key;
    }
}

// --------------------------------------------------------------------------------------------
// k_layout: level-1 bucket geometry (block 0; further blocks only clear the lane's partition counters).
//   mode 0 (exact, after the histogram pass): starts/ends/cursors from the counts, chunk table
//   mode 1 (capacity, before the scatter): bucket b owns [b*cap, (b+1)*cap), cursor at its start
//   mode 2 (capacity, after the scatter): ends from the cursors, chunk table
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_layout(const ull *b1_count, ull *b1_start, ull *b1_end, ull *b1_cursor, uint32_t *chunk_first, uint32_t B1, ull *arena_cursor,
         ull *sample_base, uint32_t mode_flags, ull cap, ull *kocc, const uint32_t *skip_flag, SimkaKeyCfg cfg, SimkaLaneClear clr) {
    // mode_flags bit 2: a later pass over the same sample (its occurrences add up, its arena base stays)
    const uint32_t mode = mode_flags & 3u;
    const bool later_pass = (mode_flags & 4u) != 0u;
    // the launch before the scatter (modes 0, 1) also resets the lane's level-2 state, instead of four memsets per sample:
    // blocks 1.. clear the partition counters, block 0 the cursors
    if (blockIdx.x > 0) {
        const uint32_t per = (clr.nparts + gridDim.x - 2u) / (gridDim.x - 1u);
        const uint32_t lo = (blockIdx.x - 1u) * per, hi = (lo + per < clr.nparts) ? lo + per : clr.nparts;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { clr.p_count[i] = 0u; clr.p_valid[i] = 0xffffffffu; }
        return;
    }
    if (mode != 2u && clr.p_count && threadIdx.x < 2u) { clr.spill_cursor[threadIdx.x] = 0ull; clr.redo_count[threadIdx.x] = 0ull; }
    if (mode == 2 && skip_flag && *skip_flag) return;      // the capacity-mode scatter overflowed: the sample is redone exactly
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *cnt = (ull *)(smem + SIMKA_LDS_HEAD);   // [B1]
    if (mode == 1) {
        // owned buckets get `cap` keys each, packed by their rank inside the shard; the others stay empty
        for (uint32_t b = threadIdx.x; b < B1; b += blockDim.x) {
            const bool own = simka_owns_l1(b, cfg);
            const ull st = (ull)simka_bucket_rank(b, cfg) * cap;
            b1_start[b] = own ? st : 0ull; b1_cursor[b] = own ? st : 0ull; b1_end[b] = own ? st + cap : 0ull;
        }
        if (threadIdx.x == 0 && !later_pass) *sample_base = *arena_cursor;
        return;
    }
    // exclusive scans of the bucket sizes (mode 0: starts) and of the chunk counts, 256 threads
    uint32_t *csz = (uint32_t *)(cnt + B1);          // [B1] chunks per bucket -> first chunk
    uint32_t *tmp = csz + B1;                        // scan scratch
    ull *wsum = (ull *)(tmp + 8);                    // [4] per-wave sums of the 64-bit scan
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ipt = (B1 + 255u) / 256u;
    const uint32_t b0 = tid * ipt, e0 = (b0 + ipt < B1) ? b0 + ipt : B1;
    ull mine = 0;
    for (uint32_t b = b0; b < e0; b++) {
        const ull c = (mode == 0) ? b1_count[b] : (b1_cursor[b] - b1_start[b]);
        cnt[b] = c; csz[b] = (uint32_t)((c + K2_CHUNK - 1) / K2_CHUNK);
        mine += c;
    }
    ull v = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const ull t = __shfl_up(v, o, 64); if (lane >= (uint32_t)o) v += t; }
    if (lane == 63u) wsum[wave] = v;
    __syncthreads();
    ull wpre = 0, total = 0;
    for (uint32_t w = 0; w < 4; w++) { const ull t = wsum[w]; if (w < wave) wpre += t; total += t; }
    ull run = wpre + v - mine;
    for (uint32_t b = b0; b < e0; b++) {
        const ull c = cnt[b];
        if (mode == 0) { b1_start[b] = run; b1_cursor[b] = run; b1_end[b] = run + c; }
        else b1_end[b] = b1_start[b] + c;
        run += c;
    }
    const uint32_t nchunks = block_excl_scan<256>(csz, B1, tmp);
    for (uint32_t b = tid; b < B1; b += 256) chunk_first[b] = csz[b];
    if (tid == 0) {
        chunk_first[B1] = nchunks;
        *kocc = (later_pass ? *kocc : 0ull) + total;   // k-mer occurrences of this shard = sum of its bucket sizes
        if (mode == 0 && !later_pass) *sample_base = *arena_cursor;   // where this sample's solid records start in the arena
    }
}

// --------------------------------------------------------------------------------------------
// K2a  k_split: level-2 scatter.  One block takes a K2_CHUNK-key chunk of a level-1 bucket, orders it
// by level-2 bits in LDS (keys stay in registers, LDS rank by returning ds_add) and appends each of the
// B2 runs to its partition's region of l2_keys with ONE global atomic per run, so every partition ends
// up contiguous in HBM and k_count streams it with perfectly coalesced loads.
// Regions are capacity-sized (cap2 = 1.5 x mean + slack; the hash spreads keys evenly).  A run that does
// not fit goes to the spill buffer with its partition id and the partition is finished by the general
// kernel; if even the spill buffer overflows the sample is flagged and redone with a full-size one.
// --------------------------------------------------------------------------------------------
template <bool NARROW>
__global__ void __launch_bounds__(K2_BLOCK)
k_split(const uint64_t *l1_keys, const ull *b1_start, const ull *b1_end, const uint32_t *chunk_first, SimkaKeyCfg cfg,
        SimkaL2 l2, uint32_t *flag) {
    if (*flag) return;                               // the level-1 scatter overflowed: the sample is redone exactly
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t B1 = 1u << cfg.l1, B2 = 1u << cfg.l2;
    ull *s_chunk = (ull *)smem;                      // [2][3] (start, n, b1) of the current / next chunk
    uint32_t *hist = (uint32_t *)(smem + SIMKA_LDS_HEAD);   // [B2] counts, then exclusive offsets
    uint32_t *tmp = hist + B2;                      // [16] scan scratch
    ull *gpos = (ull *)(tmp + 16);                  // [B2] destination of each run (bit 63: spill buffer)
    uint64_t *stage = (uint64_t *)(gpos + B2);      // [K2_CHUNK]

    const uint32_t tid = threadIdx.x;
    const uint32_t nchunks = chunk_first[B1];
    const uint32_t rem_mask = NARROW ? ((1u << l2.rem_bits) - 1u) : 0u;
    constexpr int PER = K2_CHUNK / K2_BLOCK;
    // chunk c -> (first key, #keys, level-1 bucket): thread 0, into slot `w`
    auto locate = [&](uint32_t c, uint32_t w) {
        uint32_t lo = 0, hi = B1;   // largest b with chunk_first[b] <= c  (empty buckets repeat a value)
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (chunk_first[mid] <= c) lo = mid; else hi = mid; }
        const ull st = b1_start[lo] + (ull)(c - chunk_first[lo]) * K2_CHUNK;
        const ull be = b1_end[lo];
        s_chunk[w * 3 + 0] = st; s_chunk[w * 3 + 1] = (be - st < (ull)K2_CHUNK) ? (be - st) : (ull)K2_CHUNK; s_chunk[w * 3 + 2] = lo;
    };
    // persistent block: chunks blockIdx.x, +gridDim.x, ...; the keys of the NEXT chunk are loaded into registers while the
    // current one is ranked, staged and written out
    uint32_t c = blockIdx.x;
    if (c >= nchunks) return;
    if (tid == 0) locate(c, 0);
    __syncthreads();
    uint64_t keys[PER], nkeys[PER];
    {
        const ull st = s_chunk[0]; const uint32_t n = (uint32_t)s_chunk[1];
#pragma unroll
        for (int q = 0; q < PER; q++) { const uint32_t idx = (uint32_t)q * K2_BLOCK + tid; keys[q] = (idx < n) ? l1_keys[st + idx] : SIMKA_EMPTY_KEY; }
    }
    for (uint32_t it = 0; c < nchunks; it++, c += gridDim.x) {
        const uint32_t w = it & 1u;
        const uint32_t n = (uint32_t)s_chunk[w * 3 + 1], b1 = (uint32_t)s_chunk[w * 3 + 2];
        const uint32_t cn = c + gridDim.x;
        if (tid == 0 && cn < nchunks) locate(cn, w ^ 1u);
        for (uint32_t i = tid; i < B2; i += K2_BLOCK) hist[i] = 0;
        __syncthreads();
        if (cn < nchunks) {      // prefetch
            const ull st = s_chunk[(w ^ 1u) * 3 + 0]; const uint32_t nn = (uint32_t)s_chunk[(w ^ 1u) * 3 + 1];
#pragma unroll
            for (int q = 0; q < PER; q++) { const uint32_t idx = (uint32_t)q * K2_BLOCK + tid; nkeys[q] = (idx < nn) ? l1_keys[st + idx] : SIMKA_EMPTY_KEY; }
        }
        uint32_t ranks[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const bool ok = keys[q] != SIMKA_EMPTY_KEY;
            const uint32_t rk = atomicAdd(&hist[ok ? simka_key_l2(keys[q], cfg) : 0u], ok ? 1u : 0u);
            ranks[q] = rk;
        }
        __syncthreads();
        // reserve the runs (B2 <= 2048 buckets, <= 2 per thread): returning global atomics whose results are only needed by
        // the copy-out -- they stay in registers while the block scans and stages
        uint32_t rh[2] = { 0, 0 }, rpos[2] = { 0, 0 };
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t b = tid + (uint32_t)u * K2_BLOCK;
            if (b < B2) {
                rh[u] = hist[b];
                if (rh[u]) rpos[u] = atomicAdd(&l2.p_count[(b1 << cfg.l2) | b], rh[u]);     // reserve the run in its partition
            }
        }
        __syncthreads();
        block_excl_scan<K2_BLOCK>(hist, B2, tmp);
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (keys[q] != SIMKA_EMPTY_KEY) stage[hist[simka_key_l2(keys[q], cfg)] + ranks[q]] = keys[q];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t b = tid + (uint32_t)u * K2_BLOCK;
            if (b < B2) {
                const uint32_t h = rh[u], pos = rpos[u];
                ull g = 0;
                if (h) {
                    const uint32_t part = (b1 << cfg.l2) | b;
                    if ((ull)pos + h <= l2.cap2) g = simka_region_index(part, cfg) * l2.cap2 + pos;
                    else {
                        atomicMin(&l2.p_valid[part], pos);                          // region holds [0,pos) only; the rest is spilled
                        const ull sp = atomicAdd(&l2.spill_cursor[0], (ull)h);
                        const ull sr = atomicAdd(&l2.spill_cursor[1], 1ull);
                        if (sp + h > l2.spill_cap || sr >= l2.spill_run_cap) { atomicOr(flag, 2u); g = ~0ull; }
                        else { SimkaSpillRun run; run.start = sp; run.part = part; run.len = h; l2.spill_runs[sr] = run; g = (1ull << 63) | sp; }
                    }
                }
                gpos[b] = g;
            }
        }
        __syncthreads();
        for (uint32_t idx = tid; idx < n; idx += K2_BLOCK) {
            const uint64_t key = stage[idx];
            const uint32_t b = simka_key_l2(key, cfg);
            const ull g = gpos[b];
            const uint32_t off = idx - hist[b];
            if (g == ~0ull) continue;
            if (g >> 63) l2.spill_keys[(g & ~(1ull << 63)) + off] = key;
            else if (NARROW) ((uint32_t *)l2.l2_keys)[g + off] = (uint32_t)key & rem_mask;    // the partition bits are implicit
            else l2.l2_keys[g + off] = key;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) keys[q] = nkeys[q];
    }
}

// --------------------------------------------------------------------------------------------
// K2b  k_count: one partition = one LDS hash table (64-bit CAS insert + counter), then
// SimkaCompressedProcessor::process (ref: src/minikc/MiniKC.hpp:54-79): abundance filter, emit
// (k-mer,count), nbDistinct++, nbKmers+=c, chord+=c^2.  Records go to the HBM arena where the reference
// gzips them to solid/part_<p>/__p__<i>.gz.
//
// k_count_fast: persistent blocks, partition p = contiguous keys l2_keys[p*cap2 .. +n): every thread issues
// K2F_UNROLL coalesced independent loads; the loads of partition p+1 are issued BEFORE the summary pass of
// partition p (HBM latency hides behind LDS work); summary = one pass in which each thread owns 4 table
// slots, a block scan places its solid records and the thread clears exactly those slots.
// Partitions that spilled, exceed the prefetch window or over-fill the table go to the redo list.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool table_insert(ull *tkeys, uint32_t *tcnt, uint32_t tmask, ull key) {
    uint32_t slot = simka_slot_hash(key) & tmask;
    for (uint32_t probe = 0; probe <= tmask; probe++) {
        const ull prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY, key);
        if (prev == SIMKA_EMPTY_KEY || prev == key) { atomicAdd(&tcnt[slot], 1u); return true; }
        slot = (slot + 1u) & tmask;
    }
    return false;
}
// narrow keys (W - pb <= 31 bits: the partition is implicit): 32-bit table, 32-bit multiplicative slot hash
#define SIMKA_EMPTY_KEY32 0xffffffffu
__device__ __forceinline__ bool table_insert(uint32_t *tkeys, uint32_t *tcnt, uint32_t tmask, uint32_t key) {
    uint32_t slot = ((key * 0x9E3779B1u) >> 16) & tmask;
    for (uint32_t probe = 0; probe <= tmask; probe++) {
        const uint32_t prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY32, key);
        if (prev == SIMKA_EMPTY_KEY32 || prev == key) { atomicAdd(&tcnt[slot], 1u); return true; }
        slot = (slot + 1u) & tmask;
    }
    return false;
}

// fast path: give up after K2F_PROBES slots -- the table is (nearly) full, the partition goes to the general kernel
template <typename KT>
__device__ __forceinline__ bool table_insert_capped(KT *tkeys, uint32_t *tcnt, uint32_t tmask, KT key) {
    constexpr KT EMPTY = (KT)~(KT)0;
    uint32_t slot;
    if (sizeof(KT) == 4) slot = (((uint32_t)key * 0x9E3779B1u) >> 16) & tmask; else slot = simka_slot_hash((ull)key) & tmask;
#pragma unroll 4
    for (uint32_t probe = 0; probe < K2F_PROBES; probe++) {
        const KT prev = atomicCAS(&tkeys[slot], EMPTY, key);
        if (prev == EMPTY || prev == key) { atomicAdd(&tcnt[slot], 1u); return true; }
        slot = (slot + 1u) & tmask;
    }
    return false;
}

// reserve `ns` arena records for one partition out of the block's private slab (thread 0 only)
__device__ __forceinline__ ull slab_take(ull &slab_pos, ull &slab_end, uint32_t ns, const SimkaCountOut &o, ull sample_base, uint32_t &ok) {
    if (slab_pos + ns > slab_end) {
        const ull want = ns > o.slab ? (ull)ns : (ull)o.slab;
        slab_pos = atomicAdd(o.arena_cursor, want);
        slab_end = slab_pos + want;
        if (slab_end > o.arena_cap) { atomicOr(o.err, SIMKA_DEVERR_ARENA_FULL); ok = 0; slab_end = slab_pos; return 0; }
    }
    const ull b = slab_pos;
    slab_pos += ns;
    if (b - sample_base + ns > 0xffffffffull) { atomicOr(o.err, SIMKA_DEVERR_SAMPLE_TOO_BIG); ok = 0; }
    return b;
}

__device__ __forceinline__ void count_hist(const SimkaCountOut &o, uint32_t *lhist, uint32_t c) {
    if (c < SIMKA_HIST_MAX) atomicAdd(&lhist[c], 1u);
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

template <uint32_t TS, bool NARROW>
__global__ void __launch_bounds__(K2F_BLOCK)
k_count_fast(SimkaKeyCfg cfg, SimkaL2 l2, uint32_t amin, uint32_t amax, SimkaCountOut o, const uint32_t *flag,
             uint32_t *redo_list, ull *redo_count) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                         // [4]
    ull &s_base = *(ull *)(smem + 32);
    uint32_t &s_ok = *(uint32_t *)(smem + 56);
    uint32_t &s_fail = *(uint32_t *)(smem + 60);
    ull *s_slab = (ull *)(smem + 64);                 // [2][2] (pos, end) of the block's arena slab, double-buffered by iteration parity
    uint32_t *tmp = (uint32_t *)(smem + 128);         // [K2F_BLOCK/64]
    constexpr uint32_t tmask = TS - 1u, SPT = TS / K2F_BLOCK;   // slots per thread
    using KT = typename std::conditional<NARROW, uint32_t, ull>::type;     // level-2 key as stored: remainder only, or whole key
    constexpr KT KEMPTY = (KT)~(KT)0;
    KT *tkeys = (KT *)(smem + SIMKA_LDS_HEAD);        // [TS]
    uint32_t *tcnt = (uint32_t *)(tkeys + TS);        // [TS]
    uint32_t *spos = tcnt + TS;                       // [K2F_BLOCK]
    uint32_t *lhist = spos + K2F_BLOCK;               // [SIMKA_HIST_MAX] (complex only)
    const KT *l2k = (const KT *)l2.l2_keys;

    const uint32_t tid = threadIdx.x;
    const uint32_t nparts = 1u << cfg.pb;
    const ull sample_base = *o.sample_base;
    {   // the table starts clean; afterwards every thread re-cleans the slots it read
        uint4 *k4 = (uint4 *)tkeys; uint4 *c4 = (uint4 *)tcnt;
        for (uint32_t i = tid; i < TS * sizeof(KT) / 16; i += K2F_BLOCK) k4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (uint32_t i = tid; i < TS / 4; i += K2F_BLOCK) c4[i] = make_uint4(0, 0, 0, 0);
    }
    if (o.hist) for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += K2F_BLOCK) lhist[i] = 0;
    if (tid < 4) { s_tot[tid] = 0; s_slab[tid] = 0; }
    if (tid == 0) s_fail = 0u;
    uint32_t iter = 0;
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0;

    KT kk[K2F_UNROLL];
    // issue the loads of partition p (n keys) -- only partitions the fast path can take in one batch
#define K2F_LOAD(p, n) {                                                                      \
    const ull rb_ = simka_region_index(p, cfg) * l2.cap2;                                     \
    _Pragma("unroll") for (int u = 0; u < K2F_UNROLL; u++) {                                  \
        const uint32_t i = tid + (uint32_t)u * K2F_BLOCK;                                     \
        kk[u] = (i < (n)) ? l2k[rb_ + i] : KEMPTY;                                            \
    } }
    uint32_t part = blockIdx.x;
    uint32_t n = 0, n_ahead = 0;            // key counts of this partition and of the next one: loaded one iteration early
    bool fastp = false;
    if (part < nparts) {
        n = l2.p_count[part];
        fastp = (ull)n <= l2.cap2;          // not spilled
        if (fastp) { K2F_LOAD(part, n) }
        if (part + gridDim.x < nparts) n_ahead = l2.p_count[part + gridDim.x];
    }
    __syncthreads();
    PH_DECL
    while (part < nparts) {
        const uint32_t next = part + gridDim.x;
        const uint32_t n_next = n_ahead;
        const bool fast_next = next < nparts && (ull)n_next <= l2.cap2;
        n_ahead = (next + gridDim.x < nparts) ? l2.p_count[next + gridDim.x] : 0u;      // consumed in the next iteration
        if (n == 0) { part = next; n = n_next; fastp = fast_next; if (fastp) { K2F_LOAD(part, n) } continue; }
        if (!fastp) {     // spilled: the general kernel finishes it
            if (tid == 0) { const ull w = atomicAdd(redo_count, 1ull); redo_list[w] = part; }
            part = next; n = n_next; fastp = fast_next; if (fastp) { K2F_LOAD(part, n) }
            continue;
        }
        // ---- insert the prefetched keys
        PH(0) PH_WAITVM PH(1)
        bool placed = true;
#pragma unroll
        for (int u = 0; u < K2F_UNROLL; u++) {
            const KT key = kk[u];
            if (key != KEMPTY) placed &= table_insert_capped<KT>(tkeys, tcnt, tmask, key);
        }
        for (uint32_t i = K2F_BLOCK * K2F_UNROLL + tid; i < n; i += K2F_BLOCK)      // beyond the prefetch window (rare)
            placed &= table_insert_capped<KT>(tkeys, tcnt, tmask, l2k[simka_region_index(part, cfg) * l2.cap2 + i]);
        if (!placed) s_fail = 1u;           // a key found no slot within K2F_PROBES: the table is too small for this partition
        PH(2)
        __syncthreads();
        PH(3)
        // ---- prefetch the next partition while this one is summarised
        if (fast_next) { K2F_LOAD(next, n_next) }
        // ---- one pass over this thread's slots: SimkaCompressedProcessor::process
        uint32_t cs[SPT]; KT ks[SPT];
        uint32_t nsol = 0, ndall = 0;
        ull D = 0, N = 0, Q = 0;
#pragma unroll
        for (uint32_t q = 0; q < SPT; q++) {
            const uint32_t sl = tid * SPT + q;
            cs[q] = tcnt[sl]; ks[q] = tkeys[sl];
            if (cs[q]) { tcnt[sl] = 0; tkeys[sl] = KEMPTY; }     // leave the table clean for the next partition
            const uint32_t c = cs[q];
            if (c) { ndall++; if (!(c < amin || c > amax)) { D++; N += c; Q += (ull)c * (ull)c; nsol++; } else cs[q] = 0; }
        }
        spos[tid] = nsol | (ndall << 16);
        PH(4)
        __syncthreads();
        const uint32_t tot = block_excl_scan<K2F_BLOCK>(spos, K2F_BLOCK, tmp);
        PH(5)
        const uint32_t total = tot & 0xffffu, dall_tot = tot >> 16;
        const bool ovf = dall_tot > (TS * 7u) / 8u || s_fail != 0u;     // (nearly) full or keys dropped -> the general kernel
        // arena space for the solid records: the block's slab state is double-buffered in LDS, so in the common case (the
        // run fits the current slab) every thread derives the base itself and no barrier is needed before the emit
        const uint32_t par = iter & 1u;
        iter++;
        const ull sp_ = s_slab[par * 2u], se_ = s_slab[par * 2u + 1u];
        const bool fits = !ovf && (total == 0 || (sp_ + total <= se_ && sp_ - sample_base + total <= 0xffffffffull));
        ull base_ = sample_base;
        bool ok_ = true;
        if (fits) {
            if (total) base_ = sp_;
            // the bookkeeping is spread over three waves so that no single wave becomes the straggler of the next barrier
            if (tid == 0) { s_slab[(par ^ 1u) * 2u] = sp_ + total; s_slab[(par ^ 1u) * 2u + 1u] = se_; }
            if (tid == 64) o.foff[part] = (uint32_t)(base_ - sample_base);
            if (tid == 128) o.fcnt[part] = total;
        } else {
            if (tid == 0) {
                uint32_t ok = 1;
                ull slab_pos = sp_, slab_end = se_;
                if (ovf) { const ull w = atomicAdd(redo_count, 1ull); redo_list[w] = part; ok = 0; }
                else {
                    const ull bb = slab_take(slab_pos, slab_end, total, o, sample_base, ok);
                    o.foff[part] = ok ? (uint32_t)(bb - sample_base) : 0u;
                    o.fcnt[part] = ok ? total : 0u;
                    s_base = bb;
                }
                s_slab[(par ^ 1u) * 2u] = slab_pos; s_slab[(par ^ 1u) * 2u + 1u] = slab_end;
                s_ok = ok;
            }
            __syncthreads();
            base_ = s_base; ok_ = s_ok != 0;
        }
        PH(6)
        if (ok_) {
            bt_dall += ndall; bt_D += D; bt_N += N; bt_Q += Q;
            ull pos = base_ + (spos[tid] & 0xffffu);
            const ull khigh = NARROW ? ((ull)part << l2.rem_bits) : 0ull;         // the arena holds whole keys
#pragma unroll
            for (uint32_t q = 0; q < SPT; q++) {
                if (cs[q]) {
                    o.solid_keys[pos] = khigh | (ull)ks[q]; o.solid_counts[pos] = cs[q]; pos++;
                    if (o.hist) count_hist(o, lhist, cs[q]);
                }
            }
        }
        PH(7)
        if (ovf) { __syncthreads(); if (tid == 0) s_fail = 0u; }      // (uniform) everyone has read the flag
        part = next; n = n_next; fastp = fast_next;
    }
    PH_FLUSH
#undef K2F_LOAD
    if (o.hist) {
        __syncthreads();
        for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += K2F_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    }
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    __syncthreads();
    if (tid == 0) {
        ull *t = o.totals + o.sample;
        const size_t ns_ = o.nb_samples;
        if (s_tot[0]) atomicAdd(&t[SIMKA_TOT_DALL * ns_], s_tot[0]);
        if (s_tot[1]) { atomicAdd(&t[SIMKA_TOT_D * ns_], s_tot[1]); atomicAdd(&t[SIMKA_TOT_N * ns_], s_tot[2]); atomicAdd(&t[SIMKA_TOT_Q * ns_], s_tot[3]); }
    }
}

// k_count: the general kernel for the partitions on the redo list (spilled, very large, or more distinct keys than
// table slots): streams the region + the partition's spill entries, and re-runs in 2,4,.. rounds on extra key bits
// until every round fits the table.  pass 0 counts (and emits when one round suffices), pass 1 emits.
__global__ void __launch_bounds__(K2C_BLOCK)
k_count(SimkaKeyCfg cfg, SimkaL2 l2, uint32_t table_log2, uint32_t amin, uint32_t amax, SimkaCountOut o, const uint32_t *flag,
        const uint32_t *part_list, const ull *part_count) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                         // [4] D_all, D, N, Q of the whole block
    ull &s_base = *(ull *)(smem + 32);
    ull &s_slab_pos = *(ull *)(smem + 40);
    ull &s_slab_end = *(ull *)(smem + 48);
    uint32_t &s_nsolid = *(uint32_t *)(smem + 56);
    uint32_t &s_cur = *(uint32_t *)(smem + 60);
    uint32_t &s_ovf = *(uint32_t *)(smem + 64);
    const uint32_t TS = 1u << table_log2, tmask = TS - 1u;
    ull *tkeys = (ull *)(smem + SIMKA_LDS_HEAD);      // [TS]
    uint32_t *tcnt = (uint32_t *)(tkeys + TS);        // [TS]
    uint32_t *mlist = tcnt + TS;                      // [K2C_MATCH] spill runs of the current partition
    uint32_t *lhist = mlist + K2C_MATCH;              // [SIMKA_HIST_MAX] solid-count histogram (complex only)
    uint32_t &s_nmatch = *(uint32_t *)(smem + 68);
    if (o.hist) for (uint32_t i = threadIdx.x; i < SIMKA_HIST_MAX; i += K2C_BLOCK) lhist[i] = 0;

    const uint32_t nparts = 1u << cfg.pb;
    const uint32_t tid = threadIdx.x;
    const uint32_t free_bits = cfg.W - cfg.pb;
    if (tid < 4) s_tot[tid] = 0;
    if (tid == 0) { s_slab_pos = 0; s_slab_end = 0; }
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0;
    const ull sample_base = *o.sample_base;
    const uint32_t nwork = part_list ? (uint32_t)(*part_count < (ull)nparts ? *part_count : (ull)nparts) : nparts;

    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t part = part_list ? part_list[wi] : wi;
        const uint32_t pc = l2.p_count[part];
        if (pc == 0) continue;
        const uint32_t pv = l2.p_valid[part];
        const uint32_t nreg = (uint32_t)((ull)(pv < pc ? pv : pc) < l2.cap2 ? (pv < pc ? pv : pc) : (uint32_t)l2.cap2);   // keys in the region
        const uint32_t nruns = (pc > nreg) ? (uint32_t)(l2.spill_cursor[1] < l2.spill_run_cap ? l2.spill_cursor[1] : l2.spill_run_cap) : 0u;   // spill runs to look through
        const ull *reg = l2.l2_keys + simka_region_index(part, cfg) * l2.cap2;
        const uint32_t *reg32 = (const uint32_t *)l2.l2_keys + simka_region_index(part, cfg) * l2.cap2;       // narrow level-2 keys: remainder only
        const ull khigh = (ull)part << l2.rem_bits;
        __syncthreads();
        if (tid == 0) { s_nsolid = 0; s_cur = 0; s_ovf = 0; s_nmatch = 0; }
        __syncthreads();
        // the spill runs of this partition, found once (the rounds below re-read only these)
        for (uint32_t i = tid; i < nruns; i += K2C_BLOCK)
            if (l2.spill_runs[i].part == part) { const uint32_t m = atomicAdd(&s_nmatch, 1u); if (m < K2C_MATCH) mlist[m] = i; }
        __syncthreads();
        const uint32_t nmatch_all = s_nmatch;
        const bool listed = nmatch_all <= K2C_MATCH;

        uint32_t nr_log2 = 0;
        ull emit_base = 0;
        bool part_done = false;
        for (int pass = 0; pass < 2 && !part_done; pass++) {
            bool restart = false;
            ull dall = 0, D = 0, N = 0, Q = 0;
            for (uint32_t r = 0; r < (1u << nr_log2); r++) {
                __syncthreads();
                {   // clear the table with 16-byte LDS stores
                    ulonglong2 *k2 = (ulonglong2 *)tkeys; uint4 *c4 = (uint4 *)tcnt;
                    const ulonglong2 ek = make_ulonglong2(SIMKA_EMPTY_KEY, SIMKA_EMPTY_KEY);
                    const uint4 z = make_uint4(0, 0, 0, 0);
                    for (uint32_t i = tid; i < TS / 2; i += K2C_BLOCK) k2[i] = ek;
                    for (uint32_t i = tid; i < TS / 4; i += K2C_BLOCK) c4[i] = z;
                }
                __syncthreads();
                const uint32_t rsh = free_bits - nr_log2;
                for (uint32_t i0 = tid; i0 < nreg; i0 += K2C_BLOCK * K2_UNROLL) {
                    ull keyv[K2_UNROLL];
#pragma unroll
                    for (int u = 0; u < K2_UNROLL; u++) {
                        const uint32_t i = i0 + (uint32_t)u * K2C_BLOCK;
                        keyv[u] = (i < nreg) ? (l2.narrow ? (khigh | (ull)reg32[i]) : reg[i]) : SIMKA_EMPTY_KEY;
                    }
#pragma unroll
                    for (int u = 0; u < K2_UNROLL; u++) {
                        const ull key = keyv[u];
                        if (key == SIMKA_EMPTY_KEY) continue;
                        if (nr_log2 && (uint32_t)((key >> rsh) & ((1ull << nr_log2) - 1ull)) != r) continue;
                        if (!table_insert(tkeys, tcnt, tmask, key)) s_ovf = 1;
                    }
                }
                for (uint32_t m = 0; m < (listed ? nmatch_all : nruns); m++) {
                    const SimkaSpillRun run = l2.spill_runs[listed ? mlist[m] : m];
                    if (run.part != part) continue;                       // (unlisted: every run is looked at)
                    for (uint32_t j = tid; j < run.len; j += K2C_BLOCK) {
                        const ull key = l2.spill_keys[run.start + j];
                        if (nr_log2 && (uint32_t)((key >> rsh) & ((1ull << nr_log2) - 1ull)) != r) continue;
                        if (!table_insert(tkeys, tcnt, tmask, key)) s_ovf = 1;
                    }
                }
                __syncthreads();
                if (s_ovf) { restart = true; break; }
                if (pass == 0) {
                    const uint4 *c4 = (const uint4 *)tcnt;
                    for (uint32_t i = tid; i < TS / 4; i += K2C_BLOCK) {
                        const uint4 q = c4[i];
                        const uint32_t cs[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t c = cs[j];
                            if (c) { dall++; if (!(c < amin || c > amax)) { D++; N += c; Q += (ull)c * (ull)c; } }
                        }
                    }
                    if (nr_log2 == 0) {   // single round: reserve now, emit from the live table
                        if (D) atomicAdd(&s_nsolid, (uint32_t)D);
                        __syncthreads();
                        if (tid == 0) {
                            const uint32_t ns = s_nsolid;
                            uint32_t ok = 1;
                            const ull bb = ns ? slab_take(s_slab_pos, s_slab_end, ns, o, sample_base, ok) : sample_base;
                            o.foff[part] = ok ? (uint32_t)(bb - sample_base) : 0u;
                            o.fcnt[part] = ok ? ns : 0u;
                            s_base = bb; s_ovf = ok ? 0u : 2u;
                        }
                        __syncthreads();
                        emit_base = s_base;
                    }
                }
                if ((pass == 1 || nr_log2 == 0) && s_ovf != 2u) {
                    const uint4 *c4 = (const uint4 *)tcnt;
                    for (uint32_t i = tid; i < TS / 4; i += K2C_BLOCK) {
                        const uint4 q = c4[i];
                        const uint32_t cs[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t c = cs[j];
                            if (c && !(c < amin || c > amax)) {
                                const uint32_t pos = atomicAdd(&s_cur, 1u);
                                o.solid_keys[emit_base + pos] = tkeys[i * 4 + j];
                                o.solid_counts[emit_base + pos] = c;
                                if (o.hist) count_hist(o, lhist, c);
                            }
                        }
                    }
                }
            }
            if (restart) {
                __syncthreads();
                if (nr_log2 >= free_bits || nr_log2 >= 16) { if (tid == 0) atomicOr(o.err, SIMKA_DEVERR_TABLE_OVERFLOW); part_done = true; continue; }
                if (tid == 0) s_ovf = 0;
                nr_log2++; pass = -1;       // start over with twice the rounds (nothing was emitted yet)
                continue;
            }
            if (pass == 0) {
                bt_dall += dall; bt_D += D; bt_N += N; bt_Q += Q;
                if (nr_log2 == 0) { part_done = true; continue; }     // single round: already emitted
                if (D) atomicAdd(&s_nsolid, (uint32_t)D);
                __syncthreads();
                if (tid == 0) {
                    const uint32_t ns = s_nsolid;
                    uint32_t ok = 1;
                    const ull bb = ns ? slab_take(s_slab_pos, s_slab_end, ns, o, sample_base, ok) : sample_base;
                    o.foff[part] = ok ? (uint32_t)(bb - sample_base) : 0u;
                    o.fcnt[part] = ok ? ns : 0u;
                    s_base = bb; s_ovf = ok ? 0u : 2u;
                }
                __syncthreads();
                if (s_ovf == 2u || s_nsolid == 0) { part_done = true; continue; }
                emit_base = s_base;
            }
        }
    }
    if (o.hist) {
        __syncthreads();
        for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += K2C_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    }
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    __syncthreads();
    if (tid == 0) {
        ull *t = o.totals + o.sample;
        const size_t ns_ = o.nb_samples;
        if (s_tot[0]) atomicAdd(&t[SIMKA_TOT_DALL * ns_], s_tot[0]);
        if (s_tot[1]) { atomicAdd(&t[SIMKA_TOT_D * ns_], s_tot[1]); atomicAdd(&t[SIMKA_TOT_N * ns_], s_tot[2]); atomicAdd(&t[SIMKA_TOT_Q * ns_], s_tot[3]); }
    }
}

// per-partition record totals over all samples (input of the host-side partition scan)
__global__ void __launch_bounds__(256)
k_part_totals(const uint32_t *fcnt, uint32_t nb_samples, uint64_t nparts, ull *part_total) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nparts) return;
    ull s = 0;
    for (uint32_t i = 0; i < nb_samples; i++) s += fcnt[(size_t)i * nparts + p];
    part_total[p] = s;
}

// --------------------------------------------------------------------------------------------
// K3  k_regroup: bring partition p's records of all N samples together, ordered by sub-range.
// (the reference opens the N files solid/part_p/__p__*.gz, ref: src/SimkaMerge.cpp:1082-1103)
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(K3_BLOCK)
k_regroup(SimkaMergeIn in, SimkaKeyCfg cfg, uint64_t part_begin, const ull *part_off, ull batch_base,
          uint32_t *fb_off, ull *mkeys, ull *mvals) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t tmp[K3_BLOCK];
    const uint32_t nsub = 1u << cfg.t;
    const uint64_t p = part_begin + blockIdx.x;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < nsub; i += K3_BLOCK) hist[i] = 0;
    __syncthreads();
    const uint32_t GS = 16, g = tid / GS, lane = tid % GS, ngroups = K3_BLOCK / GS;
    for (uint32_t s = g; s < in.nb_samples; s += ngroups) {
        const uint32_t n = in.fcnt[(size_t)s * in.nparts + p];
        const ull b = in.sample_base[s] + in.foff[(size_t)s * in.nparts + p];
        for (uint32_t i = lane; i < n; i += GS) atomicAdd(&hist[simka_key_sub(in.solid_keys[b + i], cfg)], 1u);
    }
    __syncthreads();
    block_excl_scan<K3_BLOCK>(hist, nsub, tmp);
    const uint32_t rel = (uint32_t)(part_off[p] - batch_base);
    for (uint32_t i = tid; i < nsub; i += K3_BLOCK) fb_off[(size_t)blockIdx.x * nsub + i] = rel + hist[i];
    __syncthreads();
    for (uint32_t s = g; s < in.nb_samples; s += ngroups) {
        const uint32_t n = in.fcnt[(size_t)s * in.nparts + p];
        const ull b = in.sample_base[s] + in.foff[(size_t)s * in.nparts + p];
        for (uint32_t i = lane; i < n; i += GS) {
            const ull key = in.solid_keys[b + i];
            const uint32_t pos = rel + atomicAdd(&hist[simka_key_sub(key, cfg)], 1u);
            mkeys[pos] = key;
            mvals[pos] = ((ull)s << 32) | (ull)in.solid_counts[b + i];
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
__global__ void __launch_bounds__(K3_BLOCK)
k_group(const ull *mkeys, const ull *mvals, const uint32_t *fb_off, uint32_t nfb, uint32_t batch_total,
        SimkaKeyCfg cfg, uint32_t min_share, SimkaCsrOut o) {
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
    uint32_t *tmp = (uint32_t *)(smem + 96);               // [K3_BLOCK/64]
    ull &s_open = *(ull *)(smem + 336);                    // slot of the block's open span (~0: none)
    uint32_t &s_open_nent = *(uint32_t *)(smem + 344);     // its entries / groups / largest count so far
    uint32_t &s_open_ngrp = *(uint32_t *)(smem + 348);
    uint32_t &s_open_maxc = *(uint32_t *)(smem + 352);
    uint32_t &s_soff = *(uint32_t *)(smem + 356);          // entry offset of this round inside its span
    ull *tkeys = (ull *)(smem + SIMKA_LDS_HEAD);           // [K3_TABLE]
    ull *rval = tkeys + K3_TABLE;                          // [K3_CAP]
    uint32_t *scnt = (uint32_t *)(rval + K3_CAP);          // [K3_TABLE] group size
    uint32_t *gpk = scnt + K3_TABLE;                       // [K3_TABLE] packed prefix: entries | groups << 20; later the fill cursor
    uint16_t *rslot = (uint16_t *)(gpk + K3_TABLE);        // [K3_CAP]
    ull *s_stack = (ull *)(rslot + K3_CAP);                // [2*K3_STACK] (selector bits, value) of the refinement DFS: one level per key bit

    const uint32_t tid = threadIdx.x;
    const uint32_t free_bits = cfg.W - cfg.pb - cfg.t;     // bits left to split an over-full sub-range
    if (tid < 6) s_slab[tid] = 0;
    if (tid == 0) { s_ndist = 0; s_nshared = 0; s_open = ~0ull; s_open_nent = 0; s_open_ngrp = 0; s_open_maxc = 0; }
    __syncthreads();

    for (uint32_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const uint32_t rb = fb_off[fb];
        const uint32_t re = (fb + 1 < nfb) ? fb_off[fb + 1] : batch_total;
        const uint32_t R = re - rb;
        if (R == 0) continue;
        uint32_t e0 = 0;
        while (((R >> e0) > K3_PRESPLIT) && e0 < free_bits) e0++;
        const uint32_t nvals0 = 1u << e0;
        for (uint32_t v0 = 0; v0 < nvals0; v0++) {
            __syncthreads();
            if (tid == 0) { s_sp = 1; s_stack[0] = e0; s_stack[1] = v0; }
            while (true) {
                __syncthreads();
                if (s_sp == 0) break;
                const uint32_t e = (uint32_t)s_stack[2 * (s_sp - 1)];
                const ull val = s_stack[2 * (s_sp - 1) + 1];          // up to free_bits (> 32) selector bits
                __syncthreads();
                if (tid == 0) { s_sp--; s_nrec = 0; s_ovf = 0; s_maxc = 0; }
                {
                    ulonglong2 *k2 = (ulonglong2 *)tkeys; uint4 *c4 = (uint4 *)scnt;
                    const ulonglong2 ek = make_ulonglong2(SIMKA_EMPTY_KEY, SIMKA_EMPTY_KEY);
                    const uint4 z = make_uint4(0, 0, 0, 0);
                    for (uint32_t i = tid; i < K3_TABLE / 2; i += K3_BLOCK) k2[i] = ek;
                    for (uint32_t i = tid; i < K3_TABLE / 4; i += K3_BLOCK) c4[i] = z;
                }
                __syncthreads();
                // ---- hash the records of this (sub-)range; K3_UNROLL independent loads per thread
                const uint32_t selshift = free_bits - e;
                uint32_t mymax = 0;
                for (uint32_t i0 = rb + tid; i0 < re; i0 += K3_BLOCK * K3_UNROLL) {
                    ull kk[K3_UNROLL], vv[K3_UNROLL];
#pragma unroll
                    for (int u = 0; u < K3_UNROLL; u++) {
                        const uint32_t i = i0 + (uint32_t)u * K3_BLOCK;
                        kk[u] = SIMKA_EMPTY_KEY; vv[u] = 0;
                        if (i < re) { kk[u] = mkeys[i]; vv[u] = mvals[i]; }
                    }
#pragma unroll
                    for (int u = 0; u < K3_UNROLL; u++) {
                        const ull key = kk[u];
                        if (key == SIMKA_EMPTY_KEY) continue;
                        if (e && ((key >> selshift) & ((1ull << e) - 1ull)) != val) continue;
                        const uint32_t idx = atomicAdd(&s_nrec, 1u);
                        if (idx >= K3_CAP) { s_ovf = 1; continue; }
                        uint32_t slot = simka_slot_hash(key) & (K3_TABLE - 1u);
                        for (;;) {   // 2*K3_CAP slots, at most K3_CAP records: always terminates
                            const ull prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY, key);
                            if (prev == SIMKA_EMPTY_KEY || prev == key) break;
                            slot = (slot + 1u) & (K3_TABLE - 1u);
                        }
                        atomicAdd(&scnt[slot], 1u);
                        rslot[idx] = (uint16_t)slot;
                        rval[idx] = vv[u];
                        if ((uint32_t)vv[u] > mymax) mymax = (uint32_t)vv[u];
                    }
                }
#pragma unroll
                for (int o_ = 32; o_ > 0; o_ >>= 1) { const uint32_t t_ = __shfl_xor(mymax, o_, 64); mymax = t_ > mymax ? t_ : mymax; }
                if ((tid & 63u) == 0 && mymax) atomicMax(&s_maxc, mymax);
                __syncthreads();
                if (s_ovf) {
                    if (e >= free_bits) {
                        // Every key bit is fixed: the selected records are ONE k-mer shared by more than K3_CAP samples.  Such a
                        // group cannot be staged in LDS; it becomes a span of its own on the huge list (k_pairs_global).
                        const uint32_t s_ = s_nrec;
                        __syncthreads();
                        if (tid == 0) {
                            const ull eb_ = atomicAdd(&o.cursors[0], (ull)s_), hs = atomicAdd(&o.cursors[3], 1ull);
                            if (eb_ + s_ > o.cap_entries || hs >= o.cap_huge) { atomicOr(o.err, SIMKA_DEVERR_CSR_FULL); s_ovf = 2; }
                            else {
                                SimkaSpan sp; sp.ebase = eb_; sp.gbase = 0; sp.nent = s_; sp.ngrp = 1; sp.maxc = 0; sp.pad = 0;
                                o.huge[hs] = sp;
                                s_ebase = eb_; s_ndist++; s_nshared++;
                            }
                            s_nrec = 0;      // now the fill cursor
                        }
                        __syncthreads();
                        if (s_ovf != 2) {
                            const ull eb_ = s_ebase;
                            for (uint32_t i = rb + tid; i < re; i += K3_BLOCK) {
                                const ull key = mkeys[i];
                                if (key == SIMKA_EMPTY_KEY) continue;
                                if (e && ((key >> selshift) & ((1ull << e) - 1ull)) != val) continue;
                                o.entries[eb_ + atomicAdd(&s_nrec, 1u)] = mvals[i];
                            }
                        }
                        continue;
                    }
                    if (tid == 0) {   // refine: two children with one more selector bit
                        if (s_sp + 2 > K3_STACK) { atomicOr(o.err, SIMKA_DEVERR_GROUP_OVERFLOW); s_sp = 0; }
                        else {
                            s_stack[2 * s_sp] = e + 1; s_stack[2 * s_sp + 1] = val * 2ull + 1ull; s_sp++;
                            s_stack[2 * s_sp] = e + 1; s_stack[2 * s_sp + 1] = val * 2ull; s_sp++;
                        }
                    }
                    continue;
                }
                const uint32_t nrec = s_nrec;
                if (nrec == 0) continue;
                // ---- group geometry: one packed prefix over the table slots (entries | groups << 20)
                uint32_t ndist = 0, nshared = 0;
                for (uint32_t i = tid; i < K3_TABLE; i += K3_BLOCK) {
                    const uint32_t c = scnt[i];
                    if (c) ndist++;
                    if (c > 1) nshared++;
                    gpk[i] = (c >= min_share) ? (c | (1u << 20)) : 0u;
                }
                if (ndist) atomicAdd(&s_ndist, ndist);
                if (nshared) atomicAdd(&s_nshared, nshared);
                __syncthreads();
                const uint32_t tot = block_excl_scan<K3_BLOCK>(gpk, K3_TABLE, tmp);
                const uint32_t nent = tot & 0xfffffu, ngrp = tot >> 20;
                if (ngrp == 0) continue;
                if (tid == 0) {
                    // slab reservations: one global atomic per K3_SLAB_* items.  A round whose output lands right behind the
                    // block's open span (same slabs) EXTENDS that span up to o.span_cap entries: k_pairs pays its per-span
                    // overhead (scans, barriers, pair-range search) once per span, so longer spans are cheaper.
                    uint32_t ok = 1, fresh = 0;
                    if (s_slab[0] + nent > s_slab[1]) { s_slab[0] = atomicAdd(&o.cursors[0], (ull)K3_SLAB_ENT); s_slab[1] = s_slab[0] + K3_SLAB_ENT; fresh = 1; if (s_slab[1] > o.cap_entries) ok = 0; }
                    if (s_slab[2] + ngrp > s_slab[3]) { s_slab[2] = atomicAdd(&o.cursors[1], (ull)K3_SLAB_GRP); s_slab[3] = s_slab[2] + K3_SLAB_GRP; fresh = 1; if (s_slab[3] > o.cap_groups) ok = 0; }
                    const bool extend = !fresh && s_open != ~0ull && s_open_nent + nent <= o.span_cap && s_open_ngrp + ngrp <= o.span_cap / 2u;
                    if (!extend && s_slab[4] + 1 > s_slab[5]) { s_slab[4] = atomicAdd(&o.cursors[2], (ull)K3_SLAB_SPAN); s_slab[5] = s_slab[4] + K3_SLAB_SPAN; if (s_slab[5] > o.cap_spans) ok = 0; }
                    if (!ok) { atomicOr(o.err, SIMKA_DEVERR_CSR_FULL); s_ovf = 2; }
                    else {
                        if (!extend) { s_open = s_slab[4]; s_slab[4] += 1; s_open_nent = 0; s_open_ngrp = 0; s_open_maxc = 0; }
                        s_soff = s_open_nent;
                        s_ebase = s_slab[0]; s_gbase = s_slab[2];
                        s_open_nent += nent; s_open_ngrp += ngrp; s_open_maxc = s_maxc > s_open_maxc ? s_maxc : s_open_maxc;
                        SimkaSpan sp; sp.ebase = s_slab[0] - s_soff; sp.gbase = s_slab[2] - (s_open_ngrp - ngrp); sp.nent = s_open_nent; sp.ngrp = s_open_ngrp; sp.maxc = s_open_maxc; sp.pad = 0;
                        o.spans[s_open] = sp;
                        s_slab[0] += nent; s_slab[2] += ngrp;
                    }
                }
                __syncthreads();
                if (s_ovf == 2) continue;
                const ull eb = s_ebase, gb = s_gbase;
                const uint32_t soff = s_soff;
                for (uint32_t i = tid; i < K3_TABLE; i += K3_BLOCK) {
                    const uint32_t c = scnt[i], g_ = gpk[i];
                    if (c >= min_share) o.groups[gb + (g_ >> 20)] = (((g_ & 0xfffffu) + soff) << 16) | c;    // start is relative to the span
                    gpk[i] = g_ & 0xfffffu;               // now: entry offset, advanced as fill cursor
                }
                __syncthreads();
                for (uint32_t i = tid; i < nrec; i += K3_BLOCK) {
                    const uint32_t slot = rslot[i];
                    if (scnt[slot] >= min_share) o.entries[eb + atomicAdd(&gpk[slot], 1u)] = rval[i];
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_ndist) atomicAdd(&o.glob[0], (ull)s_ndist);      // _nbDistinctKmers  (:1315)
        if (s_nshared) atomicAdd(&o.glob[1], (ull)s_nshared);  // _nbSharedKmers    (:1319-1321)
    }
    // unused span slots of this block's last slab: mark empty
    for (ull i = s_slab[4] + tid; i < s_slab[5]; i += K3_BLOCK) { SimkaSpan sp; sp.ebase = 0; sp.gbase = 0; sp.nent = 0; sp.ngrp = 0; sp.maxc = 0; sp.pad = 0; o.spans[i] = sp; }
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
__device__ __forceinline__ uint32_t pair_isqrt(ull x) {
    if (x >> 32) return (uint32_t)simka_isqrt(x);
    const uint32_t v = (uint32_t)x;
    uint32_t r = (uint32_t)__fsqrt_rn((float)v);
    if (r > 65535u) r = 65535u;
    if (r * r > v) r--;
    else if (r < 65535u && (r + 1u) * (r + 1u) <= v) r++;
    return r;
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
        ull *acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t CP = pc.ncell_pad, npk = pc.nacc32 >> 1;
    const bool cplx = pc.nacc64 != 0;
    ull *pk = (ull *)(smem + SIMKA_LDS_HEAD);                    // [npk][CP]     packed u32 pairs
    ull *c64 = pk + (size_t)npk * CP;                            // [nacc64][CP]  (whit, klfix)
    // a span holds <= EC = pc.span_cap entries in <= GC = EC/2 groups (every group has >= 2 entries)
    const uint32_t EC = pc.span_cap, GC = EC / 2u;
    ull *ent = c64 + (size_t)pc.nacc64 * CP;                     // [EC]          (sample<<32 | count)
    double *ep = (double *)(ent + EC);                           // [EC]          complex: p = c / N_sample
    double *eplp = ep + (cplx ? EC : 0);                         // [EC]          complex: p * ln p
    double *tn = eplp + (cplx ? EC : 0);                         // [SIMKA_PAIR_TN] complex: N of the samples of tile I, then tile J
    uint32_t *gdesc = (uint32_t *)(tn + (cplx ? SIMKA_PAIR_TN : 0));   // [GC]      list A of the group: (start<<16 | size)
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
        for (uint32_t i = tid; i < TD; i += K4_BLOCK) {
            tn[i] = (baseI + i < N) ? (double)pc.tot_n[baseI + i] : 1.0;
            if (rect) tn[T + i] = (baseJ + i < N) ? (double)pc.tot_n[baseJ + i] : 1.0;
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
    SimkaSpan span, nspan;
    span.ngrp = 0; span.nent = 0; nspan.ngrp = 0; nspan.nent = 0;
    ull sp = blockIdx.x;
    if (sp < nspans) span = spans[sp];
    if (sp + gridDim.x < nspans) nspan = spans[sp + gridDim.x];
    ull pre_e[EPT]; uint32_t pre_g[EPT];
#pragma unroll
    for (int q = 0; q < EPT; q++) {
        const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
        pre_e[q] = (i < span.nent) ? entries[span.ebase + i] : 0ull;
        pre_g[q] = (i < span.ngrp) ? groups[span.gbase + i] : 0u;
    }
    PP_DECL
    for (; sp < nspans; sp += gridDim.x) {
        // ---- current span: registers -> LDS
        PP(0)
        __syncthreads();
        const SimkaSpan cur = span;
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            if (i < cur.nent) {
                const ull e = pre_e[q];
                ent[i] = e;
                const uint32_t si = (uint32_t)(e >> 32);
                uint32_t fl = 0, loc = si;                       // membership flags (A | B<<16), index into tn
                if (TILED) {
                    const uint32_t li = si - baseI, lj = si - baseJ;
                    if (li < T) { fl = 1u; loc = li; }
                    else if (rect && lj < T) { fl = 1u << 16; loc = T + lj; }
                    epre[i] = fl;
                }
                if (cplx && (!TILED || fl)) {
                    const double p = (double)(uint32_t)e / tn[loc];
                    ep[i] = p; eplp[i] = p * log(p);
                }
            }
            if (i < cur.ngrp) { gdesc[i] = pre_g[q]; if (!TILED) { const uint32_t s_ = pre_g[q] & 0xffffu; gpref[i] = s_ * (s_ - 1u) / 2u; } }
        }
        // ---- issue the loads of the next span, fetch the descriptor after it
        span = nspan;
        nspan.ngrp = 0; nspan.nent = 0;
        if (sp + 2 * (ull)gridDim.x < nspans) nspan = spans[sp + 2 * (ull)gridDim.x];
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            pre_e[q] = (i < span.nent) ? entries[span.ebase + i] : 0ull;
            pre_g[q] = (i < span.ngrp) ? groups[span.gbase + i] : 0u;
        }
        PP(1)
        if (cur.ngrp == 0) continue;     // unused slot of a k_group span slab (uniform)
        const ull add = (ull)cur.ngrp * (ull)cur.maxc;
        const ull addq = (ull)cur.ngrp * (ull)cur.maxc * (ull)cur.maxc;
        // chord: packed non-returning adds as long as the span's products cannot wrap a half cell; else straight into the global u64 cell
        const bool chord_fast = cur.maxc < 46341u && addq < 0xffffffffull;
        if (bound + add >= 0xffffffffull || (chord_fast && bound_q + addq >= 0xffffffffull)) { pairs_flush<TILED, K4_BLOCK>(pk, c64, acc, pc, rect, baseI, baseJ); bound = 0; bound_q = 0; }
        bound += add;
        if (chord_fast) bound_q += addq;
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
            for (; p < pend; p++) {
                const uint32_t iy = rect ? (uint32_t)idxB[b0 + y] : (TILED ? (uint32_t)idxA[a0 + y] : a0 + y);
                const ull ey = ent[iy];
                uint32_t si = (uint32_t)(ex >> 32), sj = (uint32_t)(ey >> 32);
                uint32_t ci = (uint32_t)ex, cj = (uint32_t)ey;
                // off-diagonal tiles: every member of A precedes every member of B.  Elsewhere order the pair.
                if (!rect && si > sj) { uint32_t t_ = si; si = sj; sj = t_; t_ = ci; ci = cj; cj = t_; }
                const uint32_t li = si - baseI, lj = sj - baseJ;
                const uint32_t cell = rect ? li * T + lj : li * TD - ((li * (li + 1u)) >> 1) + (lj - li - 1u);   // TD <= 65535: fits 32 bits
                atomicAdd(&pk[0 * CP + cell], (ull)ci | ((ull)cj << 32));                       // S_ij | S_ji
                atomicAdd(&pk[1 * CP + cell], 1ull | ((ull)(ci < cj ? ci : cj) << 32));         // a | bc
                if (pc.simple) {
                    const ull prod = (ull)ci * (ull)cj;
                    const ull hell = (ull)pair_isqrt(prod) << 32;
                    if (chord_fast) atomicAdd(&pk[2 * CP + cell], (ull)(uint32_t)prod | hell);  // chord | hell
                    else {   // huge counts: the product goes straight to the global u64 cell
                        atomicAdd(&pk[2 * CP + cell], hell);
                        atomicAdd(&acc[SIMKA_ACC_CHORD * pc.nb_pairs + simka_pair_index(si, sj, N)], prod);
                    }
                }
                if (cplx) {
                    // updateDistanceComplex restricted to both-present pairs (ref: src/core/SimkaAlgorithm.hpp:437-446,477-481);
                    // the one-sided terms are closed forms of S / totals / count histograms, added on the host.
                    // KL: with p = ci/Ni, q = cj/Nj the reference's  p ln(2p/(p+q)) + q ln(2q/(p+q))  equals
                    // p ln p + q ln q - (p+q) ln((p+q)/2): one logarithm per pair, the p ln p terms are per entry.
                    const double h = ep[ix] + ep[iy];
                    double dd = eplp[ix] + eplp[iy] - h * log(h * 0.5);
                    dd = dd < 0.0 ? 0.0 : dd;            // >= 0 mathematically (Jensen); rounding noise must not drive a sum of near-identical samples negative
                    atomicAdd(&c64[1 * CP + cell], (ull)(long long)llrint(dd * SIMKA_KL_SCALE));
                    const ull uX = (ull)((double)ci * tn[(rect ? T : 0u) + lj]), uY = (ull)((double)cj * tn[li]);
                    atomicAdd(&c64[0 * CP + cell], simka_whit_abs(uX - uY) - simka_whit_abs(uX) - simka_whit_abs(uY));
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
//   computed once instead of once per tile pair.
//   k_pairs_tm: block row (I, J) stages only the two segments it owns; group runs are found from heads and tails (two 16-bit
//   LDS stores per run, no scan over the entries, no index lists) and pairs are enumerated exactly as in k_pairs.
// --------------------------------------------------------------------------------------------
#define KTM_WAVES 4                 // k_tile_major: waves (= spans in flight) per block
#define KTM_NT_MAX 256              // largest number of sample tiles the tile-major path handles

__global__ void __launch_bounds__(64 * KTM_WAVES)
k_tile_major(const SimkaSpan *spans, const ull *cursors, const ull *entries, const uint32_t *groups, SimkaPairCfg pc,
             ull *tm_ent, double2 *tm_p, uint32_t *tm_off) {
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
                    tm_p[span.ebase + pos] = make_double2(p, p * log(p));
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

__global__ void __launch_bounds__(K4_BLOCK_BIG)
k_pairs_tm(const SimkaSpan *spans, const ull *cursors, const ull *tm_ent, const double2 *tm_p, const uint32_t *tm_off, SimkaPairCfg pc,
           ull *acc) {
    constexpr int K4_BLOCK = K4_BLOCK_BIG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t CP = pc.ncell_pad, npk = pc.nacc32 >> 1;
    const bool cplx = pc.nacc64 != 0;
    ull *pk = (ull *)(smem + SIMKA_LDS_HEAD);                    // [npk][CP]     packed u32 pairs
    ull *c64 = pk + (size_t)npk * CP;                            // [nacc64][CP]  (whit, klfix)
    const uint32_t EC = pc.span_cap, GC = EC / 2u;
    ull *ent = c64 + (size_t)pc.nacc64 * CP;                     // [EC]          per staged span: segment I, then segment J: (g<<48 | sample<<32 | count)
    double *ep = (double *)(ent + EC);                           // [EC]          complex: p = c / N_sample
    double *eplp = ep + (cplx ? EC : 0);                         // [EC]          complex: p * ln p
    double *tn = eplp + (cplx ? EC : 0);                         // [SIMKA_PAIR_TN] complex: N of the samples of tile I, then tile J
    uint32_t *gdesc = (uint32_t *)(tn + (cplx ? SIMKA_PAIR_TN : 0));   // [GC]      run of the group in segment I: (start<<16 | size)
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
        for (uint32_t i = tid; i < T; i += K4_BLOCK) {
            tn[i] = (baseI + i < N) ? (double)pc.tot_n[baseI + i] : 1.0;
            if (rect) tn[T + i] = (baseJ + i < N) ? (double)pc.tot_n[baseJ + i] : 1.0;
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
    ull pre_e[EPT]; double2 pre_p[EPT]; uint32_t pre_j[EPT];
    // issue the loads of batch B; pre_j = the span (index in the range table) a staged entry belongs to
#define KTM_FETCH(B) {                                                                        \
    const KtmRange *R_ = s_rng + (B).rt;                                                      \
    _Pragma("unroll") for (int q = 0; q < EPT; q++) {                                         \
        const uint32_t i = tid + (uint32_t)q * K4_BLOCK;                                      \
        pre_e[q] = 0ull; pre_p[q] = make_double2(0.0, 0.0); pre_j[q] = 0u;                    \
        if (i < (B).nm) {                                                                     \
            uint32_t j = (B).bf;                                                              \
            while (j + 1u < (B).bl && (R_->d[j].ngrp == 0u || i >= R_->sl[j].st + R_->d[j].na + R_->d[j].nb)) j++;   \
            const KtmSlot sl = R_->sl[j];                                                     \
            const KtmSpan d = R_->d[j];                                                       \
            const ull src = d.ebase + (i < sl.mid ? d.a0 + (i - sl.st) : d.b0 + (i - sl.mid)); \
            pre_e[q] = tm_ent[src];                                                           \
            if (cplx) pre_p[q] = tm_p[src];                                                   \
            pre_j[q] = j;                                                                     \
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
        uint32_t cur_j[EPT];              // which span of the range my staged entries belong to
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            cur_j[q] = pre_j[q];
            if (i < cur.nm) { ent[i] = pre_e[q] + ((ull)CR->sl[pre_j[q]].gbase << 48); if (cplx) { ep[i] = pre_p[q].x; eplp[i] = pre_p[q].y; } }
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
        // group runs: the first entry of a run stores its index, the last one the index behind it.  Neighbouring segments of
        // different spans never share a group id; the I and J segments of one span may, hence the explicit boundary.
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const uint32_t i = tid + (uint32_t)q * K4_BLOCK;
            if (i < cur.nm) {
                const KtmSlot sl = CR->sl[cur_j[q]];
                const uint32_t g = (uint32_t)(ent[i] >> 48);
                const bool inA = i < sl.mid;
                uint16_t *run = (uint16_t *)(inA ? runA : runB) + 2u * g;
                const uint32_t lo = inA ? sl.st : sl.mid, hi = inA ? sl.mid : sl.mid + CR->d[cur_j[q]].nb;
                if (i == lo || (uint32_t)(ent[i - 1u] >> 48) != g) run[0] = (uint16_t)i;
                if (i + 1u == hi || (uint32_t)(ent[i + 1u] >> 48) != g) run[1] = (uint16_t)(i + 1u);
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
                if (pc.simple) {
                    const ull prod = (ull)ci * (ull)cj;
                    const ull hell = (ull)pair_isqrt(prod) << 32;
                    if (chord_fast) atomicAdd(&pk[2 * CP + cell], (ull)(uint32_t)prod | hell);  // chord | hell
                    else {
                        atomicAdd(&pk[2 * CP + cell], hell);
                        atomicAdd(&acc[SIMKA_ACC_CHORD * pc.nb_pairs + simka_pair_index(si, sj, N)], prod);
                    }
                }
                if (cplx) {
                    // same arithmetic as k_pairs (ref: src/core/SimkaAlgorithm.hpp:437-446,477-481)
                    const double h = ep[ix] + ep[iy];
                    double dd = eplp[ix] + eplp[iy] - h * log(h * 0.5);
                    dd = dd < 0.0 ? 0.0 : dd;
                    atomicAdd(&c64[1 * CP + cell], (ull)(long long)llrint(dd * SIMKA_KL_SCALE));
                    const ull uX = (ull)((double)ci * tn[(rect ? T : 0u) + lj]), uY = (ull)((double)cj * tn[li]);
                    atomicAdd(&c64[0 * CP + cell], simka_whit_abs(uX - uY) - simka_whit_abs(uX) - simka_whit_abs(uY));
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
                double dd = pi_ * log(pi_) + pj_ * log(pj_) - hh * log(hh * 0.5);       // same form as k_pairs
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
