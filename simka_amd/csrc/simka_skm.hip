// simka_skm.hip -- count side of the hot path as a SUPER-K-MER pipeline (gfx950, wave64).
//
// The reference's simkaCount runs gatb's SortingCountAlgorithm: reads are cut into super-k-mers (maximal runs of
// consecutive k-mers that share their minimizer), the super-k-mers are written to one file per minimizer partition, every
// partition is then counted on its own (ref: src/SimkaCount.cpp:291-297; minimizer size 7 forced at src/core/Simka.cpp:111;
// the partition of a k-mer is a function of its minimizer, ref: src/minikc/MiniKC.hpp:237-253).  This file is the same
// flow on the GPU, with HBM where the reference has its temp disk:
//
//   packed reads --k_skm_scan----> 16-byte super-k-mer records in 2^8 level-1 buckets (by the top bits of the partition id)
//                                  + a 4-byte array of their partition ids
//                --k_skm_split---> every partition contiguous, exact (start, count) table: one block per bucket, a histogram
//                                  pass over the ids, then 8192-record chunks ordered by partition in LDS
//                --k_skm_count_fast (k_skm_count for the partitions it gives up on)-->
//                                  per partition: records -> k-mers -> canonical -> LDS hash table -> abundance filter ->
//                                  solid (k-mer,count) records in the HBM arena + D/N/Q totals (MiniKC.hpp:54-79)
//
// A k-mer occurrence costs ~1.9 bytes per level instead of 8, and the per-k-mer work of the partitioning levels (rank atomics,
// staging) becomes per-record work (one record per ~10 k-mers).  Results do not depend on the minimizer scheme:
// the order on m-mers, m itself and the minimizer -> partition map are free choices; ours:
//   * m-mer order: a bijective multiply / xor-shift / multiply hash of the CANONICAL m-mer (min of the m-mer and its reverse
//     complement), so a k-mer and its reverse complement agree on the minimizer VALUE, hence on the partition;
//   * a record = maximal run of consecutive valid k-mers of a read with the same minimizer value, cut at SKM_NMAX k-mers;
//   * partition id = top bits of a multiplicative hash of the minimizer value.
//
// Record (uint4, 128 bits):  bits [0,102) the run's bases, 2 bits each (n + k - 1 <= 51 bases)
//                            bits [102,107) n - 1        bits [107,128) partition id (<= 21 bits)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "simka_device.h"
#include "simka_kernels.h"

#ifndef SIMKA_SKM_HIP
#define SIMKA_SKM_HIP

typedef unsigned long long ull;

#ifdef SKM_SCAN_STOP       // experiments: the scan returns after phase n (timing only, the results are garbage)
#define SKM_STOP_AT(n) if (SKM_SCAN_STOP == (n)) return;
#else
#define SKM_STOP_AT(n)
#endif

#define SKM_TILE 8192            // m-mer positions hashed per block (entry i <-> base position Q0 + i)
#define SKM_OWN_LO 32            // a tile emits the k-mers starting at entries [32, 8160): 254 words of 32 bases
#define SKM_OWN_HI (SKM_TILE - 32)
#define SKM_STRIDE (SKM_OWN_HI - SKM_OWN_LO)
#define SKM_BLOCK 512
#define SKM_MAXB1 256            // level-1 buckets at most (512: the scan's runs per bucket halve to ~30 bytes and the kernel doubles, the split gains 10 %)
#define SKM_CSTRIDE 16            // the global fill cursor of a bucket has a 128-byte line of its own (every tile of the scan bumps every cursor)
#ifndef SKM_SCAN_HEAD
#define SKM_SCAN_HEAD 64          // bytes of block scalars in front of k_skm_scan's tables (with gbase as 32-bit record indices: 40 304 bytes of LDS at
                                 // W = 16 = 32 granules of 1280 bytes -- FOUR blocks per CU; it was 41 776 = 33 granules = three)
#endif
#define SKM_NT (SKM_BLOCK + 4)    // thread columns of the chunk-major hash array (4 pad columns)
#define SKM_SEG 16               // entries per thread
#define SKM_MAXW 20
#define SKM_RTAB 2048            // variable-length reads: read starts of one tile staged in LDS
#define SKM_CHUNK 2048           // records per chunk of the level-2 kernels
#define SKM_CS_CHUNK 8192         // records per chunk of k_skm_chunksort (and the unit in which the cursors of a bucket share its region)
#define SKM_MAXSUB 3              // log2 of the fill cursors a level-1 bucket has at most
#define SKM_L2_BLOCK 512
#define SKM_CNT_BLOCK 512
#define SKM_CNT_TS 4096          // slots of the count kernel's LDS table
#define SKM_CNT_BATCH 256        // records expanded per batch
#define SKM_FAST_BLOCK 256       // k_skm_count_fast: 4 waves, 2048 slots, four blocks per CU
#define SKM_FAST_TS 2048
#define SKM_FAST_WCHUNK 8         // partitions a block takes per grab of the work counter
#ifndef SKM_FAST_TAIL
#define SKM_FAST_TAIL 12           // queue entries (at most) that are finished one CAS at a time instead of by another pass of the look-ahead drain; 0: off
#endif
#ifndef SKM_FAST_U
#define SKM_FAST_U 1             // k-mers per lane in flight in the insert loop (2: 1..5 % slower once the partitions are handed out dynamically; 4: three blocks per CU)
#endif
#define SKM_FAST_QCAP (64 + 64 * SKM_FAST_U)      // retry queue of a wave: what one iteration can add on top of an undrained rest
#define SKM_FAST_BMW 32          // u64 words of a wave's record-start bitmap (64 records x nmax <= 32 k-mers)
#define SKM_FAST_WREG ((SKM_FAST_BMW * 8 + SKM_FAST_QCAP * 10 + 15) / 16 * 16)     // bytes of a wave's private LDS region
#define SKM_FAST_HEAD 256        // bytes of block scalars in front of k_skm_count_fast's table
#define SKM_FAST_HBINS 512       // bins of the -complex-dist count histogram k_skm_count_fast keeps in LDS: with all SIMKA_HIST_MAX = 1024 the block took
                                 // 43 044 bytes = 34 LDS granules of 1280 bytes, i.e. THREE blocks per CU instead of four (40 740 bytes now: 32 granules)
#define SKM_SORT_BITS 4          // solid records leave the count kernels ordered by the top 4 bits of their key (SIMKA_SEG_BITS): the merge reads sub-ranges of a segment in place
#define SKM_NSORT (1 << SKM_SORT_BITS)

struct SimkaSkmCfg {
    uint32_t k, W, m, nmax;          // k-mer size, m-mers per k-mer (k - m + 1), minimizer size, k-mers per record at most
    uint32_t mmask;                  // 2^(2m) - 1
    uint32_t pb, l1, l2, d;          // log2 #partitions = l1 + l2;  d: the minimizer window of the k-mer that starts at base p covers the
                                     // m-mers at p + d .. p + d + W - 1 (k <= 31: d = 0; k >= 36: the window is CENTRED in the k-mer, so that a
                                     // k-mer and its reverse complement see the same m-mers)
    uint32_t shard_index, shard_count;   // this context keeps the partitions p with p % shard_count == shard_index
    uint32_t shard_magic, pad1_;         // floor(2^32 / shard_count) + 1 (0: plain modulo): pid / shard_count by one multiply, exact for pid < 2^21, count < 2^11
    uint64_t kmask;                  // 2^(2k) - 1
};

SIMKA_HD uint32_t skm_mm_hash(uint32_t c, uint32_t mmask, uint32_t m) {
    // (two multiplies: with one -- 6 % off the VALU-bound scan -- the partitions get uneven enough to cost the count kernel 4 %)
    uint32_t h = (c * 0x9E3779B1u) & mmask;
    h ^= h >> m;
    return (h * 0x85EBCA6Bu) & mmask;
}
SIMKA_HD uint32_t skm_pid(uint32_t minhash, uint32_t pb) { return pb ? (uint32_t)((minhash * 0xC2B2AE35u) >> (32u - pb)) : 0u; }
// (the run loops of the scan ask this four times per thread and tile: a modulo by a run-time divisor is ~30 instructions on the vector unit)
SIMKA_HD bool skm_owns(uint32_t pid, const SimkaSkmCfg &c) {
    if (c.shard_count == 1u) return true;
    if (!c.shard_magic) return (pid % c.shard_count) == c.shard_index;
    const uint32_t q = (uint32_t)(((uint64_t)pid * c.shard_magic) >> 32);
    return pid - q * c.shard_count == c.shard_index;
}
SIMKA_HD void skm_set_shard(SimkaSkmCfg &c, uint32_t index, uint32_t count) {
    c.shard_index = index; c.shard_count = count ? count : 1u;
    c.shard_magic = (c.shard_count > 1u && c.shard_count < 2048u) ? (uint32_t)(0x100000000ull / c.shard_count) + 1u : 0u;
}
SIMKA_HD uint32_t skm_rec_n(const uint4 &r) { return ((r.w >> 6) & 31u) + 1u; }
SIMKA_HD uint32_t skm_rec_pid(const uint4 &r) { return r.w >> 11; }
// (slot and sort order of the count kernels' tables: simka_key_hash32 of the canonical k-mer, simka_device.h -- what leaves a table in
// slot order is ordered by the top bits of that hash)

// reverse complement of the 32 bases of a word (code ^ 2 = complement)
// (32-bit halves: the bit pairs never cross the word boundary, so the pair swap needs no 64-bit shifts -- quarter-rate on the VALU;
//  swap + complement of one word = one shift each way and a 3-input bit operation: a ? b : ~c with a = 0x5555..)
__device__ __forceinline__ uint32_t skm_swapcomp32(uint32_t r) {
    return ((0x55555555u & (r >> 1)) | (0xAAAAAAAAu & (r << 1))) ^ 0xAAAAAAAAu;
}
__device__ __forceinline__ uint64_t skm_revcomp64(uint64_t x) {
    const uint32_t rl = skm_swapcomp32(__brev((uint32_t)(x >> 32))), rh = skm_swapcomp32(__brev((uint32_t)x));
    return ((uint64_t)rh << 32) | rl;
}
// reverse complement of the k-mer in the low 2k bits of x: skm_revcomp64(x) >> (64 - 2k), with 32-bit funnel shifts (sh = 64 - 2k, wave-uniform)
__device__ __forceinline__ uint64_t skm_revcomp_k(uint64_t x, uint32_t sh) {
    const uint32_t rl = skm_swapcomp32(__brev((uint32_t)(x >> 32))), rh = skm_swapcomp32(__brev((uint32_t)x));
    // (ONE 64-bit shift: it costs what one 32-bit funnel shift costs -- profiles/r05_valu_rate.txt -- where the two-case form with 32-bit
    //  shifts was compiled to two shifts, a funnel shift and two selects on the wave-uniform case: round 6)
    return (((uint64_t)rh << 32) | rl) >> sh;
}

// --------------------------------------------------------------------------------------------
// k_skm_scan: reads -> super-k-mer records.
//   phase 1: every thread hashes the canonical m-mers at its 16 entries -> LDS
//   phase 2: sliding-window minimum over W entries (van Herk / Gil-Werman on the thread's 40 loaded values) -> the minimizer
//            value of the k-mers at its 16 entries (+ the one before)
//   phase 3: run starts / breaks as bit masks -> LDS; the starts of the tile as one list; one lane per start: run length =
//            distance to the next break (own mask + three neighbours), records (one per <= nmax k-mers) per level-1 bucket
//   phase 4: exclusive scan of the bucket histogram = the tile's records in BUCKET order; one global atomic per bucket reserves
//            the tile's run; the records are cut out of the LDS-staged tile into that order (LDS cursor per bucket) and leave
//            as contiguous runs: consecutive lanes store consecutive records
// HIST: only count the records per level-1 bucket (the exact-sizing fallback when a capacity-sized bucket overflowed).
// --------------------------------------------------------------------------------------------
template <int W, bool FIXED, bool HIST>
__global__ void __launch_bounds__(SKM_BLOCK)
k_skm_scan(SimkaScanArgs a, SimkaSkmCfg cfg, ull *b1_count, ull *b1_cursor, uint4 *l1_recs, const ull *b1_limit, uint32_t *ovf_flag, uint32_t caprec,
           uint32_t rbytes, uint32_t lcap, uint32_t *l1_pid, uint32_t lsub, ull capb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t &s_nrec = *(uint32_t *)(smem + 0);        // extra records of long runs (beyond the first of a start)
    uint32_t &s_nstart = *(uint32_t *)(smem + 4);
    uint32_t *stmp = (uint32_t *)(smem + 16);          // [4] wave totals of the bucket scan
    // Region R (rbytes >= 16 * SKM_NT * 4) is used three times:
    //   phases 1-2: m-mer hashes, chunk-major: entry e = 16 t + 4 c + r lives at dword ((c * SKM_NT + t) * 4 + r), so the 16-byte
    //               accesses of consecutive lanes are consecutive in LDS (thread-major: every wide access a 4-way bank conflict);
    //   phases 3-4: [caprec] staged records | [lcap] run starts (entry index) | [lcap] their partition ids;
    //   a tile with more starts than the list takes (rare): partition ids of all entries, in the layout of the hashes.
    uint32_t *hm = (uint32_t *)(smem + SKM_SCAN_HEAD);              // [4][SKM_NT][4]
    uint4 *stage = (uint4 *)hm;                                     // [caprec] (!HIST)
    uint2 *slist = (uint2 *)(stage + (HIST ? 0u : caprec));         // [lcap] run starts: (entry index, minimizer value) -- one 8-byte LDS access per start (round 6; it
                                                                    // was a 2-byte and a 4-byte array: two stores with their addresses in each of the 16 unrolled tests)
    uint32_t *tb = (uint32_t *)(smem + SKM_SCAN_HEAD + rbytes);     // [TILE/16 + 8] the tile's bases, 16 per word
    uint32_t *smask = tb + SKM_TILE / 16 + 8;                       // [BLOCK] start | brk << 16
    uint32_t *hist = smask + SKM_BLOCK + 4;                         // [B1]   (smask has 4 pad words: all-break)
    uint32_t *lcur = hist + SKM_MAXB1;                              // [B1]
    uint32_t *gbase = lcur + SKM_MAXB1;                             // [B1] where the tile's run of the bucket starts (a lane's record buffer holds < 2^32 records); ~0u: overflow
    uint32_t *rtab = gbase + SKM_MAXB1;                             // [SKM_RTAB] (!FIXED)

    const uint32_t tid = threadIdx.x;
    const uint32_t B1 = 1u << cfg.l1;
    const long long Q0 = 32ll * ((long long)SKM_STRIDE / 32 * (long long)blockIdx.x - 1);     // base position of entry 0
    if (tid == 0) { s_nrec = 0; s_nstart = 0; }
    if (tid < SKM_MAXB1) { hist[tid] = 0; lcur[tid] = 0; }
    if (tid < 4) smask[SKM_BLOCK + tid] = 0xffff0000u;
    // ---- stage the tile's bases: 64-bit words Q0/32 .. (+ TILE/32 + 2)
    {
        const long long w0 = Q0 / 32;
        for (uint32_t i = tid; i < SKM_TILE / 32 + 4; i += SKM_BLOCK) {
            const long long wi = w0 + (long long)i;
            uint64_t v = 0;
            if (wi >= 0 && (uint64_t)wi < a.nb_words) v = a.packed[wi];
            tb[2 * i] = (uint32_t)v; tb[2 * i + 1] = (uint32_t)(v >> 32);
        }
    }
    // ---- variable-length reads: the read starts inside this tile (relative to max(Q0,0)), (one binary search per tile, k_tile_reads)
    const uint64_t T0 = Q0 < 0 ? 0ull : (uint64_t)Q0;
    uint32_t ntab = 0;
    if (!FIXED) {
        const uint64_t r0 = a.tile_r0[blockIdx.x], r1 = a.tile_r0[blockIdx.x + 1];
        uint64_t last = r1 + 64;
        if (last > a.nb_reads) last = a.nb_reads;
        const uint64_t cnt = last > r0 ? last - r0 : 0;
        const bool covers = cnt > 0 && (last == a.nb_reads || a.offsets[last] - T0 > (uint64_t)(SKM_TILE + 64));
        if (covers && cnt <= SKM_RTAB) {
            ntab = (uint32_t)cnt;
            for (uint32_t i = tid; i < ntab; i += SKM_BLOCK) {
                const uint64_t d = a.offsets[r0 + 1 + i] - T0;
                rtab[i] = d < 0xffffffffull ? (uint32_t)d : 0xffffffffu;
            }
        }
    }
    __syncthreads();

    // ---- phase 1: hashes of the canonical m-mers at entries 16t .. 16t+15
    {
        const uint32_t d0 = tb[tid], d1 = tb[tid + 1];
        const uint64_t F = ((uint64_t)d1 << 32) | d0;                       // bases of entries 16t .. 16t+31
        const uint64_t R = skm_revcomp64(F) >> (2u * (17u - cfg.m));        // m-mer at entry q of the window: (R >> 2(15-q)) & mmask
        const uint32_t f0 = (uint32_t)F, f1 = (uint32_t)(F >> 32), r0_ = (uint32_t)R, r1_ = (uint32_t)(R >> 32);
        uint32_t hv[SKM_SEG];
        if (cfg.mmask == 0xffffffffu) {       // (wave-uniform) m = 16, the headline geometry: a 32-bit m-mer needs none of the four masks
#pragma unroll
            for (int q = 0; q < SKM_SEG; q++) {
                const uint32_t fw = __builtin_amdgcn_alignbit(f1, f0, 2 * q), rv = __builtin_amdgcn_alignbit(r1_, r0_, 2 * (15 - q));
                uint32_t h = (fw < rv ? fw : rv) * 0x9E3779B1u;
                h ^= h >> 16;
                hv[q] = h * 0x85EBCA6Bu;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SKM_SEG; q++) {
                const uint32_t fw = __builtin_amdgcn_alignbit(f1, f0, 2 * q) & cfg.mmask;
                const uint32_t rv = __builtin_amdgcn_alignbit(r1_, r0_, 2 * (15 - q)) & cfg.mmask;
                hv[q] = skm_mm_hash(fw < rv ? fw : rv, cfg.mmask, cfg.m);
            }
        }
        uint4 *dst = (uint4 *)hm;
        dst[0 * SKM_NT + tid] = make_uint4(hv[0], hv[1], hv[2], hv[3]); dst[1 * SKM_NT + tid] = make_uint4(hv[4], hv[5], hv[6], hv[7]);
        dst[2 * SKM_NT + tid] = make_uint4(hv[8], hv[9], hv[10], hv[11]); dst[3 * SKM_NT + tid] = make_uint4(hv[12], hv[13], hv[14], hv[15]);
        if (tid < 16) dst[(tid >> 2) * SKM_NT + SKM_BLOCK + (tid & 3u)] = make_uint4(~0u, ~0u, ~0u, ~0u);       // pad columns read by the last threads
    }
    __syncthreads();
    SKM_STOP_AT(1)

    // ---- phase 2: minimizer value of the k-mers at entries e_j = 16t - 1 + j, j = 0..16 (window of W entries)
    const bool owner = tid >= SKM_OWN_LO / SKM_SEG && tid < SKM_OWN_HI / SKM_SEG;
    uint32_t mh[SKM_SEG + 1];
    {
        constexpr int NH = 40;                              // H[x] = hm[16t - 4 + x]; window j covers x in [3 + j, 3 + j + W)
        uint32_t H[NH];
        // chunks (t-1, 3), (t, 0..3), (t+1, 0..3), (t+2, 0)
        const uint4 *src = (const uint4 *)hm;
        const uint32_t tm = tid ? tid - 1u : 0u;             // (thread 0 owns nothing: any in-range column)
#pragma unroll
        for (int x = 0; x < NH / 4; x++) {
            const int cc = (x + 3) & 3, dt = (x + 3) / 4;    // x = 0 -> (t-1, 3); x = 1..4 -> (t, 0..3); ...
            const uint4 v = src[cc * SKM_NT + tm + dt];
            H[4 * x] = v.x; H[4 * x + 1] = v.y; H[4 * x + 2] = v.z; H[4 * x + 3] = v.w;
        }
        static_assert(3 + SKM_SEG + SKM_MAXW <= NH, "window fits the loaded values");
        // suffix minima up to the end of each W-aligned block, prefix minima from its start
        uint32_t suf[NH], pre[NH];
#pragma unroll
        for (int x = NH - 1; x >= 0; x--) suf[x] = (x % W == W - 1 || x == NH - 1) ? H[x] : (H[x] < suf[x + 1] ? H[x] : suf[x + 1]);
#pragma unroll
        for (int x = 0; x < NH; x++) pre[x] = (x % W == 0) ? H[x] : (H[x] < pre[x - 1] ? H[x] : pre[x - 1]);
#pragma unroll
        for (int j = 0; j <= SKM_SEG; j++) {
            const int lo = 3 + j, hi = 3 + j + W - 1;
            mh[j] = (lo % W == 0) ? pre[hi] : (suf[lo] < pre[hi] ? suf[lo] : pre[hi]);
        }
    }
    // ---- validity of the k-mers at e_0 .. e_16: start inside the data, end inside their read
    uint32_t valid = 0;
    {
        const long long Qk = Q0 - (long long)cfg.d;                          // where the k-mer of entry 0 starts (its window starts d bases in)
        const long long P0 = Qk + (long long)(SKM_SEG * tid) - 1;           // start of the k-mer of e_0
        const uint32_t k = cfg.k;
        if (FIXED) {
            const uint32_t L = a.fixed_len;
            // offset of e_0 inside its read: the tile's first position modulo L is wave-uniform (one 64-bit modulo on the scalar unit),
            // the thread's share 16 tid - 1 + L (< 2^14 + 2 L) is reduced with a float reciprocal and 24-bit multiplies -- a 64-bit modulo
            // per thread cost a dozen quarter-rate multiplies in a VALU-bound kernel
            const uint32_t relb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(Qk >= 0 ? (uint32_t)((uint64_t)Qk % L) : (L - (uint32_t)((uint64_t)(-Qk) % L)) % L));
            uint32_t rel;
            if (L < (1u << 23)) {
                const uint32_t x = relb + SKM_SEG * tid + (L - 1u);                 // Q0 % L + 16 tid - 1 + L  >= 0, < 2 L + 2^13
                uint32_t q_ = (uint32_t)((float)x * (1.0f / (float)L));
                uint32_t r_ = x - __umul24(q_, L);
                if ((int32_t)r_ < 0) r_ += L;                                       // the quotient estimate is off by at most one
                if (r_ >= L) r_ -= L;
                rel = r_;
            } else rel = (uint32_t)(((uint64_t)relb + SKM_SEG * tid + (L - 1u)) % L);
            if (L >= k + (uint32_t)SKM_SEG) {
                // A read holds at least 17 k-mers, so the 17 positions see at most ONE run of starts whose k-mer would cross the end
                // of its read: positions j with L - k < rel + j < L, i.e. j in [L - k - rel + 1, L - rel) -- a mask, not 17 compares.
                const int lo_ = (int)(L - k) - (int)rel + 1, hi_ = (int)(L - rel);
                const uint32_t lo = lo_ < 0 ? 0u : (uint32_t)lo_, hi = hi_ > SKM_SEG + 1 ? (uint32_t)(SKM_SEG + 1) : (uint32_t)hi_;
                const uint32_t bad = lo < hi ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
                valid = ((1u << (SKM_SEG + 1)) - 1u) & ~bad;
                // the ends of the data (first and last tile only: wave-uniform test)
                if (Qk < 0 || (uint64_t)(Qk + SKM_TILE + 64) > a.nb_bases) {
                    const long long l2 = P0 < 0 ? -P0 : 0ll, h2 = (long long)a.nb_bases - P0;
                    const uint32_t lo2 = l2 > SKM_SEG + 1 ? (uint32_t)(SKM_SEG + 1) : (uint32_t)l2, hi2 = h2 < 0 ? 0u : (h2 > SKM_SEG + 1 ? (uint32_t)(SKM_SEG + 1) : (uint32_t)h2);
                    valid &= lo2 < hi2 ? (((1u << hi2) - 1u) & ~((1u << lo2) - 1u)) : 0u;
                }
            } else {
#pragma unroll
                for (int j = 0; j <= SKM_SEG; j++) {
                    const long long P = P0 + j;
                    const bool ok = P >= 0 && (uint64_t)P < a.nb_bases && rel + k <= L;
                    valid |= (ok ? 1u : 0u) << j;
                    rel++; if (rel >= L) rel = 0;
                }
            }
        } else if (owner && (uint64_t)(P0 < 0 ? 0 : P0) < a.nb_bases) {
            // `next` = first read start beyond the current position: from the staged table (relative to T0) or the offsets array
            const uint64_t w0 = P0 < 0 ? 0ull : (uint64_t)P0;
            uint64_t rd = 0, next;
            uint32_t ti = 0;
            const bool tab = ntab && w0 >= T0;          // (d > 0: a k-mer may start before the tile's first base)
            if (tab) {
                const uint32_t w0rel = (uint32_t)(w0 - T0);
                uint32_t lo = 0, hi = ntab;          // smallest i with rtab[i] > w0rel
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rtab[mid] > w0rel) hi = mid; else lo = mid + 1; }
                ti = lo;
                next = ti < ntab ? T0 + rtab[ti] : a.nb_bases;
            } else {
                uint64_t lo = 0, hi = a.nb_reads;
                while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.offsets[mid] <= w0) lo = mid; else hi = mid; }
                rd = lo;
                next = a.offsets[rd + 1];
            }
#pragma unroll
            for (int j = 0; j <= SKM_SEG; j++) {
                const long long P = P0 + j;
                if (P >= 0 && (uint64_t)P < a.nb_bases) {
                    while (next <= (uint64_t)P) {            // e_j starts a later read (empty reads: several steps)
                        if (tab) { ti++; next = ti < ntab ? T0 + rtab[ti] : a.nb_bases; }
                        else { rd++; next = rd + 1 <= a.nb_reads ? a.offsets[rd + 1] : a.nb_bases; }
                        if (next >= a.nb_bases) break;
                    }
                    if ((uint64_t)P + k <= next && (uint64_t)P < next) valid |= 1u << j;
                }
            }
        }
    }
    // ---- phase 3a: run starts and breaks among e_1 .. e_16 (bit j-1)
    uint32_t start = 0, brk = 0xffffu;
    if (owner) {
        // bit j - 1: the minimizer changes between e_{j-1} and e_j; then start = valid & (previous invalid | change), as masks
        // (from the top down, neq = 2 neq + (mh[j] != mh[j - 1]): a compare and an add-with-carry per position -- written with shifts and ORs
        //  the compiler spends a select on every bit: round 6)
        uint32_t neq = 0;
#pragma unroll
        for (int j = SKM_SEG; j >= 1; j--)
            asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(neq) : "v"(mh[j]), "v"(mh[j - 1]) : "vcc");
        if (tid == SKM_OWN_LO / SKM_SEG) neq |= 1u;                                  // a tile never continues a run
        const uint32_t V = (valid >> 1) & 0xffffu, PV = valid & 0xffffu;
        start = V & (~PV | neq);
        brk = (start | ~V) & 0xffffu;
    }
    smask[tid] = start | (brk << 16);
    SKM_STOP_AT(2)
    __syncthreads();               // (A) hm is dead from here on
    {
        // a run without a break for 32 positions restarts at a thread's first entry, so the look-ahead below stays within 64 bits
        const uint32_t p1 = tid >= 1 ? smask[tid - 1] >> 16 : 0xffffu, p2 = tid >= 2 ? smask[tid - 2] >> 16 : 0xffffu;
        const bool forced = owner && ((valid >> 1) & 1u) && !(brk & 1u) && p1 == 0u && p2 == 0u;
        __syncthreads();           // every thread has read the masks of its predecessors
        if (forced) { start |= 1u; brk |= 1u; smask[tid] = start | (brk << 16); }
    }
    __syncthreads();               // (B)
    // ---- phase 3b: the run starts of the tile as one list (entry index, partition id), then ONE LANE PER START: a thread with
    // five starts no longer holds its wave back.  The list lives where the hashes were (dead since (A)).
    {
        const uint32_t ns = owner ? (uint32_t)__popc(start) : 0u;
        const uint32_t inc = wave_incl_scan(ns);
        uint32_t wb = 0;
        if ((tid & 63u) == 63u && inc) wb = atomicAdd(&s_nstart, inc);
        wb = __builtin_amdgcn_readlane(wb, 63);
        uint32_t p = wb + inc - ns;
        if (ns && p + ns <= lcap) {
#pragma unroll
            for (int j = 1; j <= SKM_SEG; j++)
                if ((start >> (j - 1)) & 1u) { slist[p] = make_uint2(SKM_SEG * tid + (uint32_t)(j - 1), mh[j]); p++; }      // (the minimizer value: its partition id is one multiply per START, not per entry)
        }
    }
    __syncthreads();
    SKM_STOP_AT(3)
    const uint32_t nstart = s_nstart;
    const bool listed = nstart <= lcap;
    if (!listed) {
        // more starts than the list takes: the partition ids of ALL entries go where the hashes were, every thread walks its own starts
        if (owner) {
            uint4 *dst = (uint4 *)hm;
            dst[0 * SKM_NT + tid] = make_uint4(skm_pid(mh[1], cfg.pb), skm_pid(mh[2], cfg.pb), skm_pid(mh[3], cfg.pb), skm_pid(mh[4], cfg.pb));
            dst[1 * SKM_NT + tid] = make_uint4(skm_pid(mh[5], cfg.pb), skm_pid(mh[6], cfg.pb), skm_pid(mh[7], cfg.pb), skm_pid(mh[8], cfg.pb));
            dst[2 * SKM_NT + tid] = make_uint4(skm_pid(mh[9], cfg.pb), skm_pid(mh[10], cfg.pb), skm_pid(mh[11], cfg.pb), skm_pid(mh[12], cfg.pb));
            dst[3 * SKM_NT + tid] = make_uint4(skm_pid(mh[13], cfg.pb), skm_pid(mh[14], cfg.pb), skm_pid(mh[15], cfg.pb), skm_pid(mh[16], cfg.pb));
        }
        __syncthreads();
    }
    // length of the run that starts at entry e: distance to the next break (own mask + three neighbours)
    auto len_of = [&](uint32_t e) {
        const uint32_t t_ = e >> 4, jb = e & 15u;
        const uint32_t own_ = smask[t_] >> 16;
        const ull n1 = smask[t_ + 1] >> 16, n2 = smask[t_ + 2] >> 16, n3 = (smask[t_ + 3] >> 16) & 1u;
        const ull look = ((ull)(own_ >> (jb + 1u))) | (n1 << (15u - jb)) | (n2 << (31u - jb)) | (n3 << (47u - jb));
        return (uint32_t)__ffsll((long long)look);       // look != 0: a break within 48 positions is guaranteed
    };
    // f(entry, run length, partition id) for every run of the tile that this shard owns.  The FIRST walk over the list (phase 3c) leaves
    // what it worked out -- run length from the break masks (five LDS reads), partition id, ownership -- in the list entry itself:
    // (entry | length << 16, partition id), length 0 = not this shard's; the later walks read it back with one 8-byte access.
    // (round 5 measured the same idea with a side array slower: the scan ran at the pace of its cursors then, see k_skm_scan's reservation)
    bool first_walk = true;
    auto for_runs = [&](auto &&f) {
        if (listed) {
            for (uint32_t si = tid; si < nstart; si += SKM_BLOCK) {
                const uint2 sl = slist[si];
                if (first_walk) {
                    const uint32_t e = sl.x, pid = skm_pid(sl.y, cfg.pb);
                    const uint32_t len = skm_owns(pid, cfg) ? len_of(e) : 0u;
                    slist[si] = make_uint2(e | (len << 16), pid);
                    if (len) f(e, len, pid);
                } else {
                    const uint32_t len = sl.x >> 16;
                    if (len) f(sl.x & 0xffffu, len, sl.y);
                }
            }
            first_walk = false;
        } else if (owner) {
            uint32_t todo = start;
            while (todo) {
                const uint32_t e = SKM_SEG * tid + (uint32_t)__ffs(todo) - 1u;
                todo &= todo - 1u;
                const uint32_t pid = hm[((((e & 15u) >> 2) * SKM_NT + (e >> 4)) << 2) | (e & 3u)];
                if (skm_owns(pid, cfg)) f(e, len_of(e), pid);
            }
        }
    };
    // ---- phase 3c: records per level-1 bucket
    for_runs([&](uint32_t, uint32_t len, uint32_t pid) { atomicAdd(&hist[cfg.pb ? pid >> (cfg.pb - cfg.l1) : 0u], 1u + (len > cfg.nmax ? 1u : 0u) + (len > 2u * cfg.nmax ? 1u : 0u) + (len > 3u * cfg.nmax ? (len - 1u) / cfg.nmax - 2u : 0u)); });       // = ceil(len / nmax), len <= 48: no division on the common path
    __syncthreads();               // (C)
    SKM_STOP_AT(4)
    if (HIST) {
        if (tid < B1 && hist[tid]) atomicAdd(&b1_count[tid], (ull)hist[tid]);
        return;
    }
    // ---- phase 4a: the tile's records are laid out BY BUCKET in the staging area (exclusive scan of the histogram), and every
    // bucket's run is reserved with one global atomic -- issued here, looked at after the records have been cut (phase 4b needs
    // only the LDS cursors), so its round trip is hidden.  hist[] becomes the start of the bucket inside the staging order, lcur[]
    // the bucket's fill cursor.
    ull g_res = 0; uint32_t h_res = 0;
    {
        const uint32_t h = tid < B1 ? hist[tid] : 0u;
        uint32_t excl = 0;
        if (tid < SKM_MAXB1) {
            const uint32_t inc = wave_incl_scan(h);
            if ((tid & 63u) == 63u) stmp[tid >> 6] = inc;
            excl = inc - h;
            h_res = h;
            // lsub > 0 (round 6): a bucket has 2^lsub fill cursors and the tile takes the one its number selects.  A returning atomic on ONE
            // word is served every ~11 ns, whoever asks (MI355X_MICROARCH.md: "one word saturates at ~88 dequeues/us"), and every tile of the
            // launch bumps every bucket's cursor: 184 000 tiles of a C3 sample kept each of the 256 words busy for 2.1 of the kernel's
            // 2.5 ms -- the scan ran at the pace of its cursors.  The cursors of a bucket fill disjoint sets of its chunks (sub_pos below).
            if (h) g_res = atomicAdd(&b1_cursor[(size_t)((tid << lsub) | (blockIdx.x & ((1u << lsub) - 1u))) * SKM_CSTRIDE], (ull)h);
        }
        __syncthreads();
        if (tid < SKM_MAXB1) {
            for (uint32_t w = 0; w < (tid >> 6); w++) excl += stmp[w];
            hist[tid] = excl; lcur[tid] = excl;
            if (tid == SKM_MAXB1 - 1u) { s_nrec = excl + h; s_nstart = 0; }          // records of the tile; s_nstart: now the "tile too large" flag
        }
    }
    __syncthreads();
    // cut one record out of the LDS-staged tile
    auto cut = [&](uint32_t e_, uint32_t n, uint32_t pid) {
        const uint32_t e = e_ - cfg.d;               // the run's first k-mer starts d bases before its window
        const uint32_t wi = e >> 4, sh = (e & 15u) * 2u;
        const uint32_t c0 = tb[wi], c1 = tb[wi + 1], c2 = tb[wi + 2], c3 = tb[wi + 3], c4 = tb[wi + 4];
        uint4 rec;
        rec.x = __builtin_amdgcn_alignbit(c1, c0, sh); rec.y = __builtin_amdgcn_alignbit(c2, c1, sh);
        rec.z = __builtin_amdgcn_alignbit(c3, c2, sh);
        rec.w = (__builtin_amdgcn_alignbit(c4, c3, sh) & 63u) | ((n - 1u) << 6) | (pid << 11);
        return rec;
    };
    // ---- phase 4b: the records, straight to their place in the bucket order.  direct: more records than the staging area takes,
    // or no list (the staging area holds the partition ids): every record goes to its reserved global slot instead (phase 4d)
    const bool direct = !listed || s_nrec > caprec;
    // where record u of the tile's reservation g in bucket b1 goes.  One cursor per bucket (lsub = 0): g is the absolute record index.
    // 2^lsub cursors: g counts the records of cursor `sub` of the bucket, whose chunk j is chunk (j << lsub) | sub of the bucket's region
    // [b1 * capb, (b1 + 1) * capb) -- the cursors fill the region's chunks side by side, so the chunks in use stay at its front and the
    // chunk sort / the gather of the count kernels see one bucket as before (k_skm_layout, k_skm_chunksort).
    const uint32_t sub = blockIdx.x & ((1u << lsub) - 1u);
    auto rec_pos = [&](uint32_t b1, uint32_t g, uint32_t u_) -> ull {
        if (lsub == 0u) return (ull)g + u_;
        const uint32_t u = g + u_;
        return (ull)b1 * capb + (((ull)(u / (uint32_t)SKM_CS_CHUNK) << lsub | sub) * (ull)SKM_CS_CHUNK) + (u % (uint32_t)SKM_CS_CHUNK);
    };
    if (!direct) {
        for_runs([&](uint32_t e, uint32_t len, uint32_t pid) {
            const uint32_t b1 = cfg.pb ? pid >> (cfg.pb - cfg.l1) : 0u;
            while (len) {
                const uint32_t n = len < cfg.nmax ? len : cfg.nmax;
                stage[atomicAdd(&lcur[b1], 1u)] = cut(e, n, pid);
                e += n; len -= n;
            }
        });
    }
    SKM_STOP_AT(5)
    if (tid < SKM_MAXB1) {
        if (h_res && b1_limit && g_res + h_res > b1_limit[(tid << lsub) | (blockIdx.x & ((1u << lsub) - 1u))]) { *ovf_flag = 1u; g_res = ~0ull; }
        gbase[tid] = g_res == ~0ull ? 0xffffffffu : (uint32_t)g_res;
    }
    __syncthreads();
    if (!direct) {
        // ---- phase 4c: the staged records leave in bucket order: consecutive lanes store consecutive 16-byte records of one
        // bucket's run
        const uint32_t ntot = s_nrec;
        for (uint32_t i = tid; i < ntot; i += SKM_BLOCK) {
            const uint4 rec = stage[i];
            const uint32_t b1 = cfg.pb ? skm_rec_pid(rec) >> (cfg.pb - cfg.l1) : 0u;
            const uint32_t g = gbase[b1];
            if (g != 0xffffffffu) { const ull at = rec_pos(b1, g, i - hist[b1]); l1_recs[at] = rec; if (l1_pid) l1_pid[at] = skm_rec_pid(rec); }       // (the exact split's first pass reads 4 bytes per record; the chunk sort needs no ids)
        }
    } else {
        // ---- phase 4d
        for_runs([&](uint32_t e, uint32_t len, uint32_t pid) {
            const uint32_t b1 = cfg.pb ? pid >> (cfg.pb - cfg.l1) : 0u;
            const uint32_t g32 = gbase[b1];
            while (len) {
                const uint32_t n = len < cfg.nmax ? len : cfg.nmax;
                const uint32_t pos = atomicAdd(&lcur[b1], 1u);
                if (g32 != 0xffffffffu) { const ull at = rec_pos(b1, g32, pos - hist[b1]); l1_recs[at] = cut(e, n, pid); if (l1_pid) l1_pid[at] = pid; }
                e += n; len -= n;
            }
        });
    }
}

// --------------------------------------------------------------------------------------------
// k_skm_layout: level-1 bucket geometry, one block.
//   mode 1 (before the capacity-sized scatter): bucket b owns [b*cap, (b+1)*cap)
//   mode 2 (after it): counts from the cursors
//   mode 0 (exact, after the histogram pass): starts from the counts
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SKM_MAXB1)
k_skm_layout(ull *b1_count, ull *b1_start, ull *b1_limit, ull *b1_cursor, uint32_t B1, uint32_t mode, ull cap,
             ull *arena_cursor, ull *sample_base, uint32_t first_pass, const uint32_t *skip_flag, ull *redo_count, uint32_t *cbase, uint32_t cs_chunk,
             uint32_t lsub, ull *sub_count) {
    __shared__ ull s_cnt[SKM_MAXB1];
    const uint32_t tid = threadIdx.x;
    if (mode != 2u && tid == 0) { redo_count[0] = 0ull; redo_count[1] = 0ull; }      // [1]: the work counter of k_skm_count_fast
    if (mode == 1u) {
        if (tid < B1) {
            b1_start[tid] = (ull)tid * cap;
            if (lsub == 0u) { b1_cursor[tid * SKM_CSTRIDE] = (ull)tid * cap; b1_limit[tid] = (ull)(tid + 1) * cap; }
            else      // 2^lsub cursors: each counts its own records from 0 and owns cap >> lsub of the bucket's slots (cap: a multiple of cs_chunk << lsub)
                for (uint32_t s_ = 0; s_ < (1u << lsub); s_++) { b1_cursor[(size_t)((tid << lsub) | s_) * SKM_CSTRIDE] = 0ull; b1_limit[(tid << lsub) | s_] = cap >> lsub; }
        }
        if (tid == 0 && first_pass) *sample_base = *arena_cursor;
        return;
    }
    if (mode == 2u) {
        if (skip_flag && *skip_flag) return;
        if (tid < B1) {
            if (lsub == 0u) { const ull c = b1_cursor[tid * SKM_CSTRIDE] - b1_start[tid]; b1_count[tid] = c; s_cnt[tid] = (c + cs_chunk - 1) / cs_chunk; }
            else {      // chunk j of cursor s is chunk (j << lsub) | s of the bucket: the bucket's chunks in use end behind the last one any cursor reached
                ull tot = 0, nch = 0;
                for (uint32_t s_ = 0; s_ < (1u << lsub); s_++) {
                    const ull c = b1_cursor[(size_t)((tid << lsub) | s_) * SKM_CSTRIDE];
                    sub_count[(tid << lsub) | s_] = c; tot += c;
                    if (c) { const ull last = ((((c - 1) / cs_chunk) << lsub) | s_) + 1; nch = last > nch ? last : nch; }
                }
                b1_count[tid] = tot; s_cnt[tid] = nch;
            }
        }
        if (cbase) {       // chunk numbering of k_skm_chunksort: bucket b owns the chunks [cbase[b], cbase[b + 1])
            __syncthreads();
            if (tid == 0) { uint32_t run = 0; for (uint32_t b = 0; b < B1; b++) { cbase[b] = run; run += (uint32_t)s_cnt[b]; } cbase[B1] = run; }
        }
        return;
    }
    if (tid < B1) s_cnt[tid] = b1_count[tid];
    __syncthreads();
    if (tid == 0) {
        ull run = 0;
        uint32_t crun = 0;
        for (uint32_t b = 0; b < B1; b++) {
            const ull c = s_cnt[b]; b1_start[b] = run; b1_cursor[b * SKM_CSTRIDE] = run; b1_limit[b] = run + c; run += c;
            if (cbase) { cbase[b] = crun; crun += (uint32_t)((c + cs_chunk - 1) / cs_chunk); }
        }
        if (cbase) cbase[B1] = crun;
        if (first_pass) *sample_base = *arena_cursor;
    }
}

// --------------------------------------------------------------------------------------------
// k_skm_split: level-1 bucket -> its 2^l2 partitions, exactly sized, ONE block per bucket.
//   pass 1: stream the partition ids of the bucket (a 4-byte side array the scan writes next to the 16-byte records), LDS
//           histogram of the partitions (non-returning atomics) -> exclusive scan -> partition table;
//   pass 2: stream it again, LDS cursor per partition (returning atomic = final position), 16-byte stores into the bucket's
//           range of the output buffer.
// 256 buckets x one 1024-thread block each: every CU streams its own bucket (8.8 MB on C3) twice and writes it once --
// 3 passes over the records where the three-level scheme (count, scatter, count + scatter) needed 6.
// --------------------------------------------------------------------------------------------
#define SKM_SPLIT_BLOCK 1024
#define SKM_SPLIT_UNROLL 8
__global__ void __launch_bounds__(SKM_SPLIT_BLOCK)
k_skm_split(const uint4 *l1_recs, const uint32_t *l1_pid, const ull *b1_start, const ull *b1_count, SimkaSkmCfg cfg, uint4 *out_recs, uint32_t *pstart, uint32_t *pcnt, const uint32_t *flag) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lh = (uint32_t *)smem;                   // [F2] counts, then cursors
    uint32_t *wsum = lh + (1u << (cfg.pb - cfg.l1));   // [16]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t l2 = cfg.pb - cfg.l1, F2 = 1u << l2, m2 = F2 - 1u;
    const uint32_t b1 = blockIdx.x;
    const ull st = b1_start[b1];
    const ull n = b1_count[b1];
    for (uint32_t i = tid; i < F2; i += SKM_SPLIT_BLOCK) lh[i] = 0;
    __syncthreads();
    constexpr uint32_t STEP = SKM_SPLIT_BLOCK * SKM_SPLIT_UNROLL;
    {
        uint32_t w[SKM_SPLIT_UNROLL], wn[SKM_SPLIT_UNROLL];
#pragma unroll
        for (int u = 0; u < SKM_SPLIT_UNROLL; u++) { const ull i = (ull)u * SKM_SPLIT_BLOCK + tid; w[u] = i < n ? l1_pid[st + i] : 0u; }
        for (ull i0 = 0; i0 < n; i0 += STEP) {
#pragma unroll
            for (int u = 0; u < SKM_SPLIT_UNROLL; u++) { const ull i = i0 + STEP + (ull)u * SKM_SPLIT_BLOCK + tid; wn[u] = i < n ? l1_pid[st + i] : 0u; }
#pragma unroll
            for (int u = 0; u < SKM_SPLIT_UNROLL; u++) { const ull i = i0 + (ull)u * SKM_SPLIT_BLOCK + tid; if (i < n) atomicAdd(&lh[w[u] & m2], 1u); }
#pragma unroll
            for (int u = 0; u < SKM_SPLIT_UNROLL; u++) w[u] = wn[u];
        }
    }
    __syncthreads();
    // exclusive scan of lh[F2] (F2 <= 4096: up to 4 per thread), partition table, cursors
    {
        const uint32_t per = (F2 + SKM_SPLIT_BLOCK - 1) / SKM_SPLIT_BLOCK;          // 1, 2 or 4
        uint32_t v[4] = { 0, 0, 0, 0 }, sum = 0;
        for (uint32_t q = 0; q < per; q++) { const uint32_t i = tid * per + q; v[q] = i < F2 ? lh[i] : 0u; sum += v[q]; }
        const uint32_t inc = wave_incl_scan(sum);
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t wpre = 0;
#pragma unroll
        for (uint32_t w_ = 0; w_ < SKM_SPLIT_BLOCK / 64; w_++) { const uint32_t t = wsum[w_]; if (w_ < wave) wpre += t; }
        uint32_t run = wpre + inc - sum;
        for (uint32_t q = 0; q < per; q++) {
            const uint32_t i = tid * per + q;
            if (i < F2) {
                const uint32_t p = (b1 << l2) | i;
                pstart[p] = (uint32_t)st + run; pcnt[p] = v[q];
                lh[i] = run;
                run += v[q];
            }
        }
    }
    __syncthreads();
    // pass 2, chunk by chunk (STEP records): the chunk is ordered by partition in LDS first, so that what goes to one partition
    // leaves as ONE run (direct 16-byte stores into 2048 open partitions were written back line by line: 3.7x the bytes)
    uint16_t *ch = (uint16_t *)(wsum + 16);            // [F2] records of the chunk per partition, then their start in the staging area
    uint4 *stage = (uint4 *)(smem + ((F2 * 4u + 64u + (F2 + 8u) * 2u + 15u) & ~15u));         // [STEP] (behind lh [F2], wsum [16], ch [F2 + 8]; 16-byte aligned)
    for (ull i0 = 0; i0 < n; i0 += STEP) {
        for (uint32_t i = tid; i < (F2 + 1) / 2; i += SKM_SPLIT_BLOCK) ((uint32_t *)ch)[i] = 0;
        uint4 rec[SKM_SPLIT_UNROLL]; uint32_t lr[SKM_SPLIT_UNROLL];
#pragma unroll
        for (int u = 0; u < SKM_SPLIT_UNROLL; u++) { const ull i = i0 + (ull)u * SKM_SPLIT_BLOCK + tid; rec[u] = make_uint4(0u, 0u, 0u, 0u); if (i < n) rec[u] = l1_recs[st + i]; }
        __syncthreads();
        // rank inside the chunk's run of the partition: 16-bit counters, two per word (the returning atomic works on the word)
#pragma unroll
        for (int u = 0; u < SKM_SPLIT_UNROLL; u++) {
            const ull i = i0 + (ull)u * SKM_SPLIT_BLOCK + tid;
            lr[u] = 0;
            if (i < n) {
                const uint32_t b2 = skm_rec_pid(rec[u]) & m2, sh = (b2 & 1u) * 16u;
                lr[u] = (atomicAdd((uint32_t *)ch + (b2 >> 1), 1u << sh) >> sh) & 0xffffu;
            }
        }
        __syncthreads();
        {   // exclusive scan of the chunk histogram (F2 <= 4096 counters, <= 4 per thread)
            const uint32_t per = (F2 + SKM_SPLIT_BLOCK - 1) / SKM_SPLIT_BLOCK;
            uint32_t v[4] = { 0, 0, 0, 0 }, sum = 0;
            for (uint32_t q = 0; q < per; q++) { const uint32_t i = tid * per + q; v[q] = i < F2 ? ch[i] : 0u; sum += v[q]; }
            const uint32_t inc = wave_incl_scan(sum);
            if (lane == 63u) wsum[wave] = inc;
            __syncthreads();
            uint32_t wpre = 0;
#pragma unroll
            for (uint32_t w_ = 0; w_ < SKM_SPLIT_BLOCK / 64; w_++) { const uint32_t t = wsum[w_]; if (w_ < wave) wpre += t; }
            uint32_t run = wpre + inc - sum;
            for (uint32_t q = 0; q < per; q++) { const uint32_t i = tid * per + q; if (i < F2) { ch[i] = (uint16_t)run; run += v[q]; } }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SKM_SPLIT_UNROLL; u++) {
            const ull i = i0 + (ull)u * SKM_SPLIT_BLOCK + tid;
            if (i < n) stage[ch[skm_rec_pid(rec[u]) & m2] + lr[u]] = rec[u];
        }
        __syncthreads();
        const uint32_t nc = (uint32_t)(n - i0 < (ull)STEP ? n - i0 : (ull)STEP);
        for (uint32_t sidx = tid; sidx < nc; sidx += SKM_SPLIT_BLOCK) {
            const uint4 r = stage[sidx];
            const uint32_t b2 = skm_rec_pid(r) & m2;
            out_recs[st + lh[b2] + (sidx - ch[b2])] = r;
        }
        __syncthreads();
        // the partitions' cursors move past this chunk: count of partition i = start of i + 1 (or the chunk's end) - start of i
        for (uint32_t i = tid; i < F2; i += SKM_SPLIT_BLOCK) lh[i] += (i + 1 < F2 ? (uint32_t)ch[i + 1] : nc) - (uint32_t)ch[i];
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// k_skm_chunksort: level-1 bucket -> its 2^l2 partitions WITHOUT moving the records to a second buffer: one block per chunk of
// SKM_CS_CHUNK records of a bucket orders the chunk by partition in LDS and writes it back IN PLACE, fully coalesced, together with
// one row of the chunk table: ctab[chunk][i] = first record of partition i inside the sorted chunk (i = 0 .. 2^l2, the last entry =
// records of the chunk).  The count kernels then GATHER a partition: piece c of partition i = records [ctab[c][i], ctab[c][i+1]) of
// chunk c of its bucket (C3: 67 pieces of ~4 records).  Against k_skm_split this reads the records once and writes them once in whole
// lines (the exact split read a 4-byte id array first and wrote 64-byte runs into 2048 open partitions: 1.9x its algorithmic bytes),
// needs no second record buffer and no id array -- the price is the gather in the count kernels (two 2-byte table entries per piece).
//   cbase[b] = first chunk of bucket b in the chunk numbering of the sample (k_skm_layout), cbase[B1] = chunks of the sample.
// --------------------------------------------------------------------------------------------
#define SKM_CS_BLOCK 1024
#define SKM_CS_UNROLL (SKM_CS_CHUNK / SKM_CS_BLOCK)
__global__ void __launch_bounds__(SKM_CS_BLOCK)
k_skm_chunksort(uint4 *recs, const ull *b1_start, const ull *b1_count, const uint32_t *cbase, SimkaSkmCfg cfg, uint16_t *ctab, uint32_t cstride, const uint32_t *flag,
                uint32_t lsub, const ull *sub_count) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t l2 = cfg.pb - cfg.l1, F2 = 1u << l2, m2 = F2 - 1u, B1 = 1u << cfg.l1;
    uint16_t *ch = (uint16_t *)smem;                     // [F2 + 2] records of the chunk per partition, then their start in the sorted chunk
    const uint32_t ch_bytes = (((F2 + 2u) * 2u) + 15u) & ~15u;
    uint32_t *wsum = (uint32_t *)(smem + ch_bytes);      // [16]
    uint4 *stage = (uint4 *)(smem + ch_bytes + 64);      // [CHUNK]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t gc = blockIdx.x;
    if (gc >= cbase[B1]) return;
    // bucket of this chunk: the last b with cbase[b] <= gc (wave-uniform search)
    uint32_t lo = 0, hi = B1;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (cbase[mid] <= gc) lo = mid; else hi = mid; }
    const uint32_t b1 = lo;
    const uint32_t cl = gc - cbase[b1];               // chunk of the bucket
    const ull st = b1_start[b1] + (ull)cl * SKM_CS_CHUNK;
    uint32_t nc;
    if (lsub == 0u) { const ull i0 = (ull)cl * SKM_CS_CHUNK, n = b1_count[b1]; nc = (uint32_t)(n - i0 < (ull)SKM_CS_CHUNK ? n - i0 : (ull)SKM_CS_CHUNK); }
    else {      // chunk (j << lsub) | s of the bucket is chunk j of its cursor s (k_skm_scan): full, the cursor's last, or beyond it (empty: an all-zero table row)
        const ull c = sub_count[(b1 << lsub) | (cl & ((1u << lsub) - 1u))], i0 = (ull)(cl >> lsub) * SKM_CS_CHUNK;
        nc = c > i0 ? (uint32_t)(c - i0 < (ull)SKM_CS_CHUNK ? c - i0 : (ull)SKM_CS_CHUNK) : 0u;
    }
    for (uint32_t i = tid; i < (F2 + 2u) / 2u; i += SKM_CS_BLOCK) ((uint32_t *)ch)[i] = 0;
    uint4 rec[SKM_CS_UNROLL]; uint32_t lr[SKM_CS_UNROLL];
#pragma unroll
    for (int u = 0; u < SKM_CS_UNROLL; u++) { const uint32_t i = (uint32_t)u * SKM_CS_BLOCK + tid; rec[u] = make_uint4(0u, 0u, 0u, 0u); if (i < nc) rec[u] = recs[st + i]; }
    __syncthreads();
    // rank inside the chunk's run of the partition: 16-bit counters, two per word (the returning atomic works on the word)
#pragma unroll
    for (int u = 0; u < SKM_CS_UNROLL; u++) {
        const uint32_t i = (uint32_t)u * SKM_CS_BLOCK + tid;
        lr[u] = 0;
        if (i < nc) {
            const uint32_t b2 = skm_rec_pid(rec[u]) & m2, sh = (b2 & 1u) * 16u;
            lr[u] = (atomicAdd((uint32_t *)ch + (b2 >> 1), 1u << sh) >> sh) & 0xffffu;
        }
    }
    __syncthreads();
    {   // exclusive scan of the chunk histogram (F2 <= 4096 counters, <= 4 per thread); the starts are the chunk's table row
        const uint32_t per = (F2 + SKM_CS_BLOCK - 1) / SKM_CS_BLOCK;
        uint32_t v[4] = { 0, 0, 0, 0 }, sum = 0;
        for (uint32_t q = 0; q < per; q++) { const uint32_t i = tid * per + q; v[q] = i < F2 ? ch[i] : 0u; sum += v[q]; }
        const uint32_t inc = wave_incl_scan(sum);
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t wpre = 0;
#pragma unroll
        for (uint32_t w_ = 0; w_ < SKM_CS_BLOCK / 64; w_++) { const uint32_t t = wsum[w_]; if (w_ < wave) wpre += t; }
        uint32_t run = wpre + inc - sum;
        for (uint32_t q = 0; q < per; q++) { const uint32_t i = tid * per + q; if (i < F2) { ch[i] = (uint16_t)run; run += v[q]; } }
        if (tid == 0) { ch[F2] = (uint16_t)nc; ch[F2 + 1u] = (uint16_t)nc; }
    }
    __syncthreads();
    {
        uint32_t *row = (uint32_t *)(ctab + (size_t)gc * cstride);       // (cstride is even: rows are 4-byte aligned)
        for (uint32_t i = tid; i < (F2 + 2u) / 2u; i += SKM_CS_BLOCK) row[i] = ((const uint32_t *)ch)[i];
    }
#pragma unroll
    for (int u = 0; u < SKM_CS_UNROLL; u++) {
        const uint32_t i = (uint32_t)u * SKM_CS_BLOCK + tid;
        if (i < nc) stage[ch[skm_rec_pid(rec[u]) & m2] + lr[u]] = rec[u];
    }
    __syncthreads();
    for (uint32_t sidx = tid; sidx < nc; sidx += SKM_CS_BLOCK) recs[st + sidx] = stage[sidx];
}

// Where the count kernels find the records of a partition.  Legacy (ctab == nullptr): ONE contiguous run of `recs` (k_skm_split:
// pstart / pcnt).  Gather: the pieces of the partition in the sorted chunks of its level-1 bucket (k_skm_chunksort): piece c =
// records [ctab[c][p2], ctab[c][p2 + 1]) of chunk c, c < cbase[b + 1] - cbase[b] <= SKM_G_MAXCH.
struct SimkaSkmSrc {
    const uint4 *recs;
    const uint32_t *pstart, *pcnt;
    const uint32_t *cbase; const ull *b1_start; const uint16_t *ctab; uint32_t cstride, pad_;
};
#define SKM_G_MAXCH 128          // chunks per level-1 bucket the gather takes (two per lane of a wave); beyond: the exact split
#define SKM_G_WBYTES (SKM_G_MAXCH / 4 * 6)      // per wave of k_skm_count_fast (it tabulates the chunks c = 4 l + wave): u32 prefix + u16 start of its pieces
#define SKM_G_BYTES(nw) ((SKM_MAXB1 + 1) * 4 + SKM_MAXB1 * 4 + 2 * (nw) * (SKM_G_WBYTES + 4))      // cbase + bucket starts + the waves' piece tables and totals, double-buffered

// --------------------------------------------------------------------------------------------
// k_skm_count: one partition at a time per (persistent) block.
//   records -> LDS; a wave scan of the record lengths + one LDS atomic per wave gives every record a range of k-mer slots,
//   map[slot] = (record, offset); then one LANE PER K-MER: cut the k-mer out of its record (funnel shift, no rolling state),
//   reverse complement (brev), canonical, 32-bit slot hash, insert (64-bit CAS on the canonical k-mer + counter).
//   Summary in slot order: slot = top bits of the hash, probing stays inside its 128-slot sort block, so the solid records
//   leave ordered by the top SKM_SORT_BITS bits of the hash.
//   A partition with more distinct k-mers than the table takes is redone in 2, 4, .. rounds on the top hash bits (first a
//   counting pass over all rounds, then the emitting pass), so the records of a (sample, partition) stay ONE contiguous,
//   ordered segment of the arena.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t skm_kmer_at(const uint4 &r, uint32_t j, const SimkaSkmCfg &cfg) {
    const uint32_t s = 2u * j, wd = s >> 5, sh = s & 31u;
    // (the record's last word goes in unmasked: what rides above its six base bits lands beyond bit 2 (j + k) of the window, and a k-mer
    //  of the record never reaches beyond its 102 bits -- the mask of the k-mer removes it)
    const uint32_t a0 = wd ? r.y : r.x, a1 = wd ? r.z : r.y, a2 = wd ? r.w : r.z;
    const uint32_t lo = __builtin_amdgcn_alignbit(a1, a0, sh), hi = __builtin_amdgcn_alignbit(a2, a1, sh);
    return (((uint64_t)hi << 32) | lo) & cfg.kmask;
}

template <bool GATHER>
__global__ void __launch_bounds__(SKM_CNT_BLOCK)
k_skm_count(SimkaSkmSrc src, SimkaSkmCfg cfg, SimkaKeyCfg kcfg, uint32_t amin, uint32_t amax, SimkaCountOut o,
            const uint32_t *flag, ull *kocc_owned, const uint32_t *part_list, const ull *part_count) {
    if (*flag) return;
    const uint4 *recs = src.recs; const uint32_t *pstart = src.pstart, *pcnt = src.pcnt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                          // [5] D_all, D, N, Q, K_occ of the whole block
    ull &s_base = *(ull *)(smem + 48);
    ull &s_slab_pos = *(ull *)(smem + 56);
    ull &s_slab_end = *(ull *)(smem + 64);
    uint32_t &s_kt = *(uint32_t *)(smem + 72);         // k-mer slots handed out in the current batch
    uint32_t &s_fail = *(uint32_t *)(smem + 76);
    uint32_t &s_ok = *(uint32_t *)(smem + 80);
    uint32_t *tmp = (uint32_t *)(smem + 128);          // [BLOCK/64]
    uint32_t *s_row = (uint32_t *)(smem + 192);        // [SIMKA_SEG_BLOCKS] solid records of the partition per key-hash block
    ull *tkeys = (ull *)(smem + SIMKA_LDS_HEAD);       // [TS]
    uint32_t *tcnt = (uint32_t *)(tkeys + SKM_CNT_TS); // [TS]
    uint4 *lrec = (uint4 *)(tcnt + SKM_CNT_TS);        // [BATCH]
    uint32_t *spos = (uint32_t *)(lrec + SKM_CNT_BATCH);     // [BLOCK]
    uint32_t *lhist = spos + SKM_CNT_BLOCK;            // [SIMKA_HIST_MAX] (complex only)
    uint16_t *map = (uint16_t *)(lhist + (o.hist ? SIMKA_HIST_MAX : 0));     // [BATCH * nmax]
    // GATHER: the piece table of the current partition (SimkaSkmSrc), one for the block
    uint32_t *gpre = (uint32_t *)(((uintptr_t)(map + SKM_CNT_BATCH * cfg.nmax) + 15u) & ~(uintptr_t)15u);      // [SKM_G_MAXCH + 1] records before piece c, [MAXCH]: all
    uint32_t *gcnt = gpre + SKM_G_MAXCH + 1;                                                                  // [SKM_G_MAXCH]
    uint16_t *gst = (uint16_t *)(gcnt + SKM_G_MAXCH);                                                         // [SKM_G_MAXCH]

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nparts = 1u << cfg.pb;
    const uint32_t gl2 = cfg.pb - cfg.l1, gm2 = (1u << gl2) - 1u;
    constexpr uint32_t TS = SKM_CNT_TS, SPT = TS / SKM_CNT_BLOCK, TSL = 12;          // log2 TS
    const ull sample_base = *o.sample_base;
    for (uint32_t i = tid; i < TS; i += SKM_CNT_BLOCK) { tkeys[i] = SIMKA_EMPTY_KEY; tcnt[i] = 0; }
    if (o.hist) for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_CNT_BLOCK) lhist[i] = 0;
    if (tid < 5) s_tot[tid] = 0;
    if (tid == 0) { s_slab_pos = 0; s_slab_end = 0; s_fail = 0; }
    if (tid < SIMKA_SEG_BLOCKS) s_row[tid] = 0;
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0, bt_kocc = 0;
    __syncthreads();

    const uint32_t nwork = part_list ? (uint32_t)(*part_count < (ull)nparts ? *part_count : (ull)nparts) : nparts;
    for (uint32_t wi_ = blockIdx.x; wi_ < nwork; wi_ += gridDim.x) {
        const uint32_t part = part_list ? part_list[wi_] : wi_;
        uint32_t nrec, rbase;
        if (GATHER) {
            const uint32_t b = part >> gl2, p2 = part & gm2, gc0 = src.cbase[b], nch = src.cbase[b + 1u] - gc0;
            __syncthreads();       // (the table of the partition before is done with)
            if (tid < SKM_G_MAXCH) {
                uint32_t t0 = 0, t1 = 0;
                if (tid < nch) { const uint16_t *row = src.ctab + (size_t)(gc0 + tid) * src.cstride + p2; t0 = row[0]; t1 = row[1]; }
                gcnt[tid] = t1 - t0; gst[tid] = (uint16_t)t0;
            }
            __syncthreads();
            if (tid == 0) { uint32_t run = 0; for (uint32_t c = 0; c < SKM_G_MAXCH; c++) { gpre[c] = run; run += gcnt[c]; } gpre[SKM_G_MAXCH] = run; }
            __syncthreads();
            nrec = gpre[SKM_G_MAXCH];
            rbase = (uint32_t)src.b1_start[b];
        } else { nrec = pcnt[part]; rbase = pstart[part]; }
        if (nrec == 0) continue;                       // (foff / fcnt of the sample were zeroed by the host)
        auto rec_at = [&](uint32_t i) -> uint4 {
            if (GATHER) {
                uint32_t c = 0;
#pragma unroll
                for (uint32_t stp = SKM_G_MAXCH / 2; stp; stp >>= 1) c += gpre[c + stp] <= i ? stp : 0u;
                return recs[(size_t)rbase + (size_t)c * SKM_CS_CHUNK + gst[c] + (i - gpre[c])];
            }
            return recs[rbase + i];
        };
        uint32_t rho = 0;                              // log2 #rounds
        bool done = false;
        while (!done) {
            const uint32_t R = 1u << rho;
            // pass 0: count (and, with a single round, emit); pass 1 (R > 1): emit
            uint32_t total_solid = 0;
            ull emit_pos = 0;
            bool overflow = false;
            ull pD_all = 0, pD = 0, pN = 0, pQ = 0, pK = 0;
            for (uint32_t pass = 0; pass < (R > 1 ? 2u : 1u) && !overflow; pass++) {
                const bool emit = (R == 1) || pass == 1;
                if (pass == 1) {
                    __syncthreads();
                    if (tid == 0) {
                        uint32_t ok = 1;
                        const ull bb = total_solid ? slab_take(s_slab_pos, s_slab_end, total_solid, o, sample_base, ok) : sample_base;
                        s_base = bb; s_ok = ok;
                    }
                    __syncthreads();
                    if (!s_ok) { overflow = false; break; }
                    emit_pos = s_base;
                }
                for (uint32_t r = 0; r < R && !overflow; r++) {
                    // ---- insert the k-mers of the partition whose hash belongs to round r
                    for (uint32_t b0 = 0; b0 < nrec; b0 += SKM_CNT_BATCH) {
                        __syncthreads();
                        if (tid == 0) s_kt = 0;
                        __syncthreads();
                        const uint32_t nb = nrec - b0 < (uint32_t)SKM_CNT_BATCH ? nrec - b0 : (uint32_t)SKM_CNT_BATCH;
                        uint32_t len = 0;
                        if (tid < nb) { const uint4 rc = rec_at(b0 + tid); lrec[tid] = rc; len = skm_rec_n(rc); }
                        uint32_t x = len;
#pragma unroll
                        for (int o_ = 1; o_ < 64; o_ <<= 1) { const uint32_t t = __shfl_up(x, o_, 64); if (lane >= (uint32_t)o_) x += t; }
                        uint32_t wbase = 0;
                        if (lane == 63u) wbase = atomicAdd(&s_kt, x);
                        wbase = __shfl(wbase, 63, 64);
                        const uint32_t off = wbase + x - len;
                        for (uint32_t j = 0; j < len; j++) map[off + j] = (uint16_t)((tid << 5) | j);
                        __syncthreads();
                        const uint32_t kt = s_kt;
                        if (r == 0 && pass == 0) pK += (tid == 0) ? kt : 0;
                        for (uint32_t f = tid; f < kt; f += SKM_CNT_BLOCK) {
                            const uint32_t e = map[f];
                            const uint4 rc = lrec[e >> 5];
                            const uint64_t fwd = skm_kmer_at(rc, e & 31u, cfg);
                            const uint64_t rev = skm_revcomp_k(fwd, 64u - 2u * cfg.k);
                            const uint64_t canon = fwd < rev ? fwd : rev;
                            const uint64_t mkey = canon;
                            const uint32_t h = simka_key_hash32(canon);
                            if (rho && (h >> (32u - rho)) != r) continue;
                            const uint32_t hs = rho ? (h << rho) : h;
                            uint32_t slot = hs >> (32u - TSL);
                            // probing stays inside the sort block (128 << rho slots, the whole table from rho = 5 on)
                            const uint32_t bmask = rho >= SKM_SORT_BITS ? (TS - 1u) : ((TS >> (SKM_SORT_BITS - rho)) - 1u);
                            const uint32_t bbase = slot & ~bmask;
                            bool placed = false;
                            for (uint32_t probe = 0; probe <= bmask; probe++) {
                                const ull prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY, (ull)mkey);
                                if (prev == SIMKA_EMPTY_KEY || prev == (ull)mkey) { atomicAdd(&tcnt[slot], 1u); placed = true; break; }
                                slot = bbase | ((slot + 1u) & bmask);
                            }
                            if (!placed) s_fail = 1u;
                        }
                    }
                    __syncthreads();
                    // ---- summary of the round in slot order (SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-79)
                    uint32_t cs[SPT]; ull ks[SPT];
                    uint32_t nsol = 0, ndall = 0;
                    ull D = 0, N = 0, Q = 0;
#pragma unroll
                    for (uint32_t q = 0; q < SPT; q++) {
                        const uint32_t sl = tid * SPT + q;
                        cs[q] = tcnt[sl]; ks[q] = tkeys[sl];
                        if (cs[q]) { tcnt[sl] = 0; tkeys[sl] = SIMKA_EMPTY_KEY; }
                        const uint32_t c = cs[q];
                        if (c) { ndall++; if (!(c < amin || c > amax)) { D++; N += c; Q += (ull)c * (ull)c; nsol++; } else cs[q] = 0; }
                    }
                    const bool failed = s_fail != 0u;
                    spos[tid] = nsol;
                    __syncthreads();
                    const uint32_t tot = block_excl_scan<SKM_CNT_BLOCK>(spos, SKM_CNT_BLOCK, tmp);
                    if (failed) { overflow = true; if (tid == 0) s_fail = 0u; continue; }     // (uniform) the table is clean again
                    if (pass == 0) { pD_all += ndall; pD += D; pN += N; pQ += Q; total_solid += tot; }
                    if (emit) {
                        if (R == 1) {      // single round: reserve now
                            if (tid == 0) {
                                uint32_t ok = 1;
                                const ull bb = tot ? slab_take(s_slab_pos, s_slab_end, tot, o, sample_base, ok) : sample_base;
                                s_base = bb; s_ok = ok;
                            }
                            __syncthreads();
                            emit_pos = s_base;
                            if (!s_ok) { overflow = false; pD_all = pD = pN = pQ = 0; total_solid = 0; continue; }
                        }
                        ull pos = emit_pos + spos[tid];
#pragma unroll
                        for (uint32_t q = 0; q < SPT; q++) {
                            if (cs[q]) {
                                o.solid_keys[pos] = ks[q]; o.solid_counts[pos] = cs[q]; pos++;
                                if (o.seg_rows) atomicAdd(&s_row[simka_key_hash32(ks[q]) >> (32u - SIMKA_SEG_BITS)], 1u);
                                if (o.hist) count_hist(o, lhist, cs[q]);
                            }
                        }
                        if (R > 1) emit_pos += tot;
                    }
                }
            }
            __syncthreads();
            if (overflow) {
                if (rho >= 16u) { if (tid == 0) atomicOr(o.err, SIMKA_DEVERR_TABLE_OVERFLOW); done = true; }
                else rho++;
                continue;
            }
            // the partition is done: its segment and totals
            if (tid == 0) {
                const bool ok = total_solid == 0 || s_ok;
                const ull seg = (R == 1) ? s_base : s_base;
                o.foff[part] = (ok && total_solid) ? (uint32_t)(seg - sample_base) : 0u;
                o.fcnt[part] = ok ? total_solid : 0u;
                if (o.seg_rows) {      // (the emit loops of this partition are behind the barrier above)
                    uint32_t run = 0;
                    for (uint32_t b_ = 0; b_ < SIMKA_SEG_BLOCKS; b_++) { run += s_row[b_]; s_row[b_] = 0; o.seg_rows[((size_t)part * o.nb_samples) * SIMKA_SEG_BLOCKS + b_] = (uint16_t)(ok ? (run > 0xffffu ? 0xffffu : run) : 0u); }
                    o.seg_abs[(size_t)part * o.nb_samples] = seg;
                    if (run > 0xffffu) atomicOr(o.err, SIMKA_DEVERR_SEGMENT_TOO_BIG);
                }
            }
            bt_dall += pD_all; bt_D += pD; bt_N += pN; bt_Q += pQ; bt_kocc += pK;
            done = true;
        }
        __syncthreads();
    }
    if (o.hist) {
        __syncthreads();
        for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_CNT_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    }
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    if (bt_kocc) atomicAdd(&s_tot[4], bt_kocc);
    __syncthreads();
    if (tid == 0) {
        ull *t = o.totals + o.sample;
        const size_t ns_ = o.nb_samples;
        if (s_tot[0]) atomicAdd(&t[SIMKA_TOT_DALL * ns_], s_tot[0]);
        if (s_tot[1]) { atomicAdd(&t[SIMKA_TOT_D * ns_], s_tot[1]); atomicAdd(&t[SIMKA_TOT_N * ns_], s_tot[2]); atomicAdd(&t[SIMKA_TOT_Q * ns_], s_tot[3]); }
        if (s_tot[4] && kocc_owned) atomicAdd(kocc_owned, s_tot[4]);
    }
}


// --------------------------------------------------------------------------------------------
// k_skm_count_wide: 32 <= k <= 51 -- the canonical k-mer is a pair of 64-bit words (hi, lo), counted per minimizer partition in an
// LDS hash table like k_skm_count, instead of sorting every k-mer occurrence of the sample (simka_wide.hip, which stays for k > 51
// and as the cross-check).  There is no 128-bit LDS atomic: a slot is CLAIMED with a 64-bit CAS on the high word (never all ones:
// it has at most 38 bits), the claimant then stores the low word and bumps the counter; a k-mer that finds its high word in a slot
// compares the low word once the counter says it is there (the LDS serves a wave's operations in order: counter > 0 implies the low
// word is written).  Within a wave the claimants' stores are issued before any lane looks, so no lane waits for a lane of its own wave.
// The solid records of the sample leave unordered, (hi, lo, count) into three arrays behind a global cursor; the host sorts them
// by k-mer into the sorted spectrum the wide merge works on.  A partition that fills the table flags the sample (the caller then
// counts it on the sort path).
// --------------------------------------------------------------------------------------------
struct SimkaWideOut {
    ull *hi, *lo; uint32_t *cnt;     // the sample's solid records
    ull *cursor;                     // [0] records written, [1] D_all, [2] D, [3] N, [4] Q, [5] K_occ, [6] flags: 1 = a table overflowed, 2 = the arrays are full
    ull cap;
    uint32_t shard_index, shard_count;   // the k-mers this context keeps (simka_wide_owns; the scan keeps every partition)
};
#define SKM_WIDE_TS 4096
#define SKM_WIDE_TSL 12

__device__ __forceinline__ void skm_wkmer_at(const uint4 &r, uint32_t j, uint32_t k, ull &hi, ull &lo) {
    const uint32_t s = 2u * j, wd = s >> 5, sh = s & 31u;         // j <= 19: wd is 0 or 1
    const uint32_t w3 = r.w & 63u;
    const uint32_t a0 = wd ? r.y : r.x, a1 = wd ? r.z : r.y, a2 = wd ? w3 : r.z, a3 = wd ? 0u : w3;
    const uint32_t o0 = __builtin_amdgcn_alignbit(a1, a0, sh), o1 = __builtin_amdgcn_alignbit(a2, a1, sh), o2 = __builtin_amdgcn_alignbit(a3, a2, sh), o3 = a3 >> sh;
    lo = ((ull)o1 << 32) | o0;
    hi = (((ull)o3 << 32) | o2) & ((2u * k > 64u) ? ((1ull << (2u * k - 64u)) - 1ull) : 0ull);
}

__global__ void __launch_bounds__(SKM_CNT_BLOCK)
k_skm_count_wide(const uint4 *recs, const uint32_t *pstart, const uint32_t *pcnt, SimkaSkmCfg cfg, uint32_t amin, uint32_t amax, SimkaWideOut wo, SimkaCountOut o,
                 const uint32_t *flag, const uint32_t *redo_list, const ull *redo_count) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                          // [5] D_all, D, N, Q, K_occ of the whole block
    ull &s_base = *(ull *)(smem + 48);
    uint32_t &s_kt = *(uint32_t *)(smem + 72);
    uint32_t &s_fail = *(uint32_t *)(smem + 76);
    uint32_t *tmp = (uint32_t *)(smem + 128);          // [BLOCK/64]
    constexpr uint32_t TS = SKM_WIDE_TS, SPT = TS / SKM_CNT_BLOCK, TSL = SKM_WIDE_TSL;
    ull *thi = (ull *)(smem + SIMKA_LDS_HEAD);         // [TS] high words: the claim (SIMKA_EMPTY_KEY: free)
    ull *tlo = thi + TS;                               // [TS] low words
    uint32_t *tcnt = (uint32_t *)(tlo + TS);           // [TS]
    uint4 *lrec = (uint4 *)(tcnt + TS);                // [BATCH]
    uint32_t *spos = (uint32_t *)(lrec + SKM_CNT_BATCH);     // [BLOCK]
    uint32_t *lhist = spos + SKM_CNT_BLOCK;            // [SIMKA_HIST_MAX] (complex only)
    uint16_t *map = (uint16_t *)(lhist + (o.hist ? SIMKA_HIST_MAX : 0));     // [BATCH * nmax]

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nparts = redo_list ? (uint32_t)*redo_count : 1u << cfg.pb;      // with a list: the partitions k_skm_count_wide_fast gave up on
    const uint32_t k = cfg.k, s2 = 128u - 2u * k;      // 26 .. 64
    for (uint32_t i = tid; i < TS; i += SKM_CNT_BLOCK) { thi[i] = SIMKA_EMPTY_KEY; tcnt[i] = 0; }
    if (o.hist) for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_CNT_BLOCK) lhist[i] = 0;
    if (tid < 5) s_tot[tid] = 0;
    if (tid == 0) s_fail = 0;
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0, bt_kocc = 0;
    __syncthreads();
    for (uint32_t pi = blockIdx.x; pi < nparts; pi += gridDim.x) {
        const uint32_t part = redo_list ? redo_list[pi] : pi;
        const uint32_t nrec = pcnt[part];
        if (nrec == 0) continue;
        const uint32_t rbase = pstart[part];
        for (uint32_t b0 = 0; b0 < nrec; b0 += SKM_CNT_BATCH) {
            __syncthreads();
            if (tid == 0) s_kt = 0;
            __syncthreads();
            const uint32_t nb = nrec - b0 < (uint32_t)SKM_CNT_BATCH ? nrec - b0 : (uint32_t)SKM_CNT_BATCH;
            uint32_t len = 0;
            if (tid < nb) { const uint4 rc = recs[rbase + b0 + tid]; lrec[tid] = rc; len = skm_rec_n(rc); }
            uint32_t x = len;
#pragma unroll
            for (int o_ = 1; o_ < 64; o_ <<= 1) { const uint32_t t = __shfl_up(x, o_, 64); if (lane >= (uint32_t)o_) x += t; }
            uint32_t wbase_ = 0;
            if (lane == 63u) wbase_ = atomicAdd(&s_kt, x);
            wbase_ = __shfl(wbase_, 63, 64);
            const uint32_t off = wbase_ + x - len;
            for (uint32_t j = 0; j < len; j++) map[off + j] = (uint16_t)((tid << 5) | j);
            __syncthreads();
            const uint32_t kt = s_kt;
            for (uint32_t f0 = 0; f0 < kt; f0 += SKM_CNT_BLOCK) {
                const uint32_t f = f0 + tid;
                bool pending = f < kt;
                ull hi = 0, lo = 0;
                uint32_t slot = 0;
                if (pending) {
                    const uint32_t e = map[f];
                    ull fh, fl;
                    skm_wkmer_at(lrec[e >> 5], e & 31u, k, fh, fl);
                    // reverse complement of the 2k-bit value: the 128-bit reversal, moved down by 128 - 2k bits
                    const ull rh_ = skm_revcomp64(fl), rl_ = skm_revcomp64(fh);
                    ull rl, rh;
                    if (s2 >= 64u) { rl = rh_; rh = 0; } else { rl = (rl_ >> s2) | (rh_ << (64u - s2)); rh = rh_ >> s2; }
                    const bool fsm = fh < rh || (fh == rh && fl < rl);
                    hi = fsm ? fh : rh; lo = fsm ? fl : rl;
                    if (wo.shard_count > 1u && !simka_wide_owns(hi, lo, wo.shard_index, wo.shard_count)) pending = false;
                    else bt_kocc++;
                    uint32_t h = (uint32_t)lo * 0x9E3779B1u + (uint32_t)(lo >> 32) * 0x85EBCA6Bu + (uint32_t)hi * 0xC2B2AE35u + (uint32_t)(hi >> 32) * 0x27D4EB2Fu;
                    h ^= h >> 15; h *= 0x2C1B3C6Du;
                    slot = h >> (32u - TSL);
                }
                // every lane of the wave runs the SAME straight-line probe step; a lane is done when its k-mer is counted
                for (uint32_t step = 0; step < 4u * TS; step++) {
                    if (!__any(pending)) break;
                    ull prev = 0;
                    if (pending) prev = atomicCAS(&thi[slot], SIMKA_EMPTY_KEY, hi);
                    const bool won = pending && prev == SIMKA_EMPTY_KEY;
                    if (won) tlo[slot] = lo;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the claimants of this wave have stored their low words
                    bool hit = won, wait = false;
                    if (pending && !won && prev == hi) {
                        const uint32_t c = ((volatile uint32_t *)tcnt)[slot];
                        if (c == 0u) wait = true;                              // claimed by another wave, its low word not counted in yet: look again
                        else hit = ((volatile ull *)tlo)[slot] == lo;
                    }
                    if (hit) { atomicAdd(&tcnt[slot], 1u); pending = false; }
                    else if (pending && !wait) slot = (slot + 1u) & (TS - 1u);
                }
                if (pending) s_fail = 1u;
            }
        }
        __syncthreads();
        // ---- summary (SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-79): the solid records leave unordered
        uint32_t cs[SPT]; ull kh[SPT], kl[SPT];
        uint32_t nsol = 0, ndall = 0;
        ull D = 0, N = 0, Q = 0;
#pragma unroll
        for (uint32_t q = 0; q < SPT; q++) {
            const uint32_t sl = tid * SPT + q;
            const uint32_t c = tcnt[sl];
            cs[q] = 0; kh[q] = 0; kl[q] = 0;
            if (c) {
                kh[q] = thi[sl]; kl[q] = tlo[sl];
                tcnt[sl] = 0; thi[sl] = SIMKA_EMPTY_KEY;
                ndall++;
                if (!(c < amin || c > amax)) { cs[q] = c; D++; N += c; Q += (ull)c * (ull)c; nsol++; }
            } else if (thi[sl] != SIMKA_EMPTY_KEY) thi[sl] = SIMKA_EMPTY_KEY;      // (claimed, never counted: only after a failure)
        }
        spos[tid] = nsol;
        __syncthreads();
        const uint32_t tot = block_excl_scan<SKM_CNT_BLOCK>(spos, SKM_CNT_BLOCK, tmp);
        if (tid == 0) { s_base = tot ? atomicAdd(wo.cursor, (ull)tot) : 0ull; if (s_fail) { atomicOr(&wo.cursor[6], 1ull); s_fail = 0; } }
        __syncthreads();
        const ull base = s_base;
        if (base + tot > wo.cap) { if (tid == 0) atomicOr(&wo.cursor[6], 2ull); }
        else {
            ull pos = base + spos[tid];
#pragma unroll
            for (uint32_t q = 0; q < SPT; q++) if (cs[q]) { wo.hi[pos] = kh[q]; wo.lo[pos] = kl[q]; wo.cnt[pos] = cs[q]; pos++; if (o.hist) count_hist(o, lhist, cs[q]); }
        }
        bt_dall += ndall; bt_D += D; bt_N += N; bt_Q += Q;
        __syncthreads();
    }
    if (o.hist) {
        __syncthreads();
        for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_CNT_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    }
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    if (bt_kocc) atomicAdd(&s_tot[4], bt_kocc);
    __syncthreads();
    if (tid < 5 && s_tot[tid]) atomicAdd(&wo.cursor[1 + tid], s_tot[tid]);
}

// --------------------------------------------------------------------------------------------
// k_skm_count_wide_fast: the common case of k_skm_count_wide -- EVERY WAVE counts partitions of its own in a private table of 512
// two-word slots, so nothing in the kernel waits for another wave (no barrier, no "claimed but not yet written" state: the LDS
// serves a wave's operations in order, so the low word a lane stores after winning a slot is there for every later read of the
// wave).  A slot is two 64-bit words: (high word << 26 | count) -- claimed, with count 1, by one compare-and-swap -- and the low word:
// 8 KB per table, sixteen waves per CU instead of four.  The partitions are sized for it (~130 k-mer occurrences); one that fills
// more than three quarters of the table (or could count beyond 2^26) is cleared and handed to k_skm_count_wide through the redo list.
// A wave is bound by latency, not by throughput, so the dependent chains are kept short: the records of the NEXT partition (and
// the table entry of the one after) are loaded before the current one is counted; two k-mers per lane are in flight in the probe
// loop; the solid records leave into a slab the wave reserves 512 at a time (one global atomic per ~12 partitions; what is left
// of a slab when a partition does not fit gets count 0 -- simka_wide_adopt drops those slots), one ballot per 64 slots gives
// every record its place, so consecutive lanes store consecutive records.
// --------------------------------------------------------------------------------------------
#define SKM_WF_BLOCK 256
#ifndef SKM_WF_TSL
#define SKM_WF_TSL 9
#endif
#define SKM_WF_TS (1 << SKM_WF_TSL)
#define SKM_WF_CHUNK 32
#define SKM_WF_SLAB 512
#define SKM_WF_CBITS 26
#ifndef SKM_WF_LOADX
#define SKM_WF_LOADX 4u            // half-slots per k-mer occurrence of a one-chunk partition
#endif
// (records + map: at least the 3 TS / 4 two-byte entries the summary's list of solid slots takes)
SIMKA_HD uint32_t skm_wf_wave_bytes(uint32_t nmax) { const uint32_t rm = SKM_WF_CHUNK * 16u + ((SKM_WF_CHUNK * nmax * 2u + 15u) & ~15u); const uint32_t lst = SKM_WF_TS * 3u / 2u; return SKM_WF_TS * 16u + (rm < lst ? lst : rm); }

__device__ __forceinline__ void skm_wcanon(const uint4 &r, uint32_t j, uint32_t k, uint32_t s2, ull &hi, ull &lo, uint32_t &slot) {
    ull fh, fl;
    skm_wkmer_at(r, j, k, fh, fl);
    // reverse complement of the 2k-bit value: the 128-bit reversal, moved down by 128 - 2k bits
    const ull rh_ = skm_revcomp64(fl), rl_ = skm_revcomp64(fh);
    ull rl, rh;
    if (s2 >= 64u) { rl = rh_; rh = 0; } else { rl = (rl_ >> s2) | (rh_ << (64u - s2)); rh = rh_ >> s2; }
    const bool fsm = fh < rh || (fh == rh && fl < rl);
    hi = fsm ? fh : rh; lo = fsm ? fl : rl;
    uint32_t h = (uint32_t)lo * 0x9E3779B1u + (uint32_t)(lo >> 32) * 0x85EBCA6Bu + (uint32_t)hi * 0xC2B2AE35u + (uint32_t)(hi >> 32) * 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    slot = h;
}

__global__ void __launch_bounds__(SKM_WF_BLOCK)
k_skm_count_wide_fast(const uint4 *recs, const uint32_t *pstart, const uint32_t *pcnt, SimkaSkmCfg cfg, uint32_t amin, uint32_t amax, SimkaWideOut wo, SimkaCountOut o,
                      const uint32_t *flag, uint32_t *redo_list, ull *redo_count) {
    if (*flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                          // [5] D_all, D, N, Q, K_occ of the whole block
    uint32_t *lhist = (uint32_t *)(smem + SIMKA_LDS_HEAD);      // [SIMKA_HIST_MAX] (complex only)
    constexpr uint32_t TS = SKM_WF_TS, SPT = TS / 64u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned char *wreg = smem + SIMKA_LDS_HEAD + (o.hist ? SIMKA_HIST_MAX * 4u : 0u) + wave * skm_wf_wave_bytes(cfg.nmax);
    ulonglong2 *tab = (ulonglong2 *)wreg;              // [TS] x: high word << 26 | count (SIMKA_EMPTY_KEY: free), y: low word
    uint4 *lrec = (uint4 *)(tab + TS);                 // [CHUNK]
    uint16_t *map = (uint16_t *)(lrec + SKM_WF_CHUNK); // [CHUNK * nmax] k-mer f of the chunk -> (record << 5) | index

    const uint32_t nparts = 1u << cfg.pb;
    const uint32_t k = cfg.k, s2 = 128u - 2u * k;      // 26 .. 64
    for (uint32_t i = lane; i < TS; i += 64u) tab[i].x = SIMKA_EMPTY_KEY;
    if (o.hist) for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_WF_BLOCK) lhist[i] = 0;
    if (tid < 5) s_tot[tid] = 0;
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0, bt_kocc = 0;
    ull slab_pos = 0, slab_end = 0;                    // the wave's slab of output slots (wave-uniform)
    __syncthreads();
    PH_DECL
    const uint32_t nwaves = gridDim.x * (SKM_WF_BLOCK / 64u);
    // (the table entries of the partition after the next stay in vector registers until they are needed: a readfirstlane here would
    // wait for the load it was just issued)
    auto meta = [&](uint32_t p, uint32_t &nrec_, uint32_t &rbase_) {
        nrec_ = 0; rbase_ = 0;
        if (p < nparts) { nrec_ = pcnt[p]; rbase_ = pstart[p]; }
    };
    // (a native vector, not HIP's uint4: the struct went through a private-memory temporary -- a dead 16-byte scratch store behind an
    // s_waitcnt vmcnt(0), i.e. the prefetch of the next partition's records was waited for where it was issued)
    typedef uint32_t skm_v4u __attribute__((ext_vector_type(4)));
    auto first_chunk = [&](uint32_t nrec_, uint32_t rbase_) {
        skm_v4u r = {0u, 0u, 0u, 0u};
        if (lane < nrec_ && lane < (uint32_t)SKM_WF_CHUNK) r = *(const skm_v4u *)(recs + rbase_ + lane);
        return r;
    };
    uint32_t part = blockIdx.x * (SKM_WF_BLOCK / 64u) + wave;
    uint32_t nrec_v, rbase_v, nrec_nv, rbase_nv;
    meta(part, nrec_v, rbase_v);
    uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)nrec_v), rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase_v);
    skm_v4u rc0 = first_chunk(nrec, rbase);
    meta(part + nwaves, nrec_nv, rbase_nv);
    constexpr int NF = 2;                              // k-mers a lane has in flight in the probe loop
    for (; part < nparts; part += nwaves) {
        // what the next two partitions need is under way while this one is counted
        const uint32_t nrec_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)nrec_nv), rbase_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase_nv);
        const skm_v4u rc0_n = first_chunk(nrec_n, rbase_n);
        meta(part + 2u * nwaves, nrec_nv, rbase_nv);
        PH(0)
        bool fail = nrec > 512u;          // a partition of many records (a few hot k-mers: one wave would walk them alone; and a count must stay below 2^26): the block kernel's
        uint32_t ndist = 0, my_k = 0;
        uint32_t tsl = SKM_WF_TSL;                                 // log2 of the slots this partition uses
        for (uint32_t b0 = 0; b0 < nrec && !fail; b0 += SKM_WF_CHUNK) {
            const uint32_t nb = nrec - b0 < (uint32_t)SKM_WF_CHUNK ? nrec - b0 : (uint32_t)SKM_WF_CHUNK;
            uint32_t len = 0;
            if (lane < nb) { const uint4 rc = b0 ? recs[rbase + b0 + lane] : make_uint4(rc0.x, rc0.y, rc0.z, rc0.w); lrec[lane] = rc; len = skm_rec_n(rc); }
            const uint32_t x = wave_incl_scan(len);
            const uint32_t kt = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
            const uint32_t off = x - len;
            // a partition of one chunk knows its k-mers now: a table of >= 2 slots per occurrence (>= 64) is enough, and the
            // summary reads only that many
            if (nrec <= (uint32_t)SKM_WF_CHUNK) { tsl = 6u; while (tsl < (uint32_t)SKM_WF_TSL && (2u << tsl) < SKM_WF_LOADX * kt) tsl++; }
            PH_WAITVM
            PH(1)
            for (uint32_t j = 0; j < len; j++) map[off + j] = (uint16_t)((lane << 5) | j);
            PH(2)
            const uint32_t tmask = (1u << tsl) - 1u, tfull = (3u << tsl) >> 2;
            for (uint32_t f0 = 0; f0 < kt && !fail; f0 += 64u * NF) {
                bool pd[NF]; ull hi[NF], lo[NF]; uint32_t sl[NF];
#pragma unroll
                for (int i = 0; i < NF; i++) {
                    const uint32_t f = f0 + 64u * i + lane;
                    pd[i] = f < kt; hi[i] = 0; lo[i] = 0; sl[i] = 0;
                    if (pd[i]) {
                        const uint32_t e = map[f];
                        skm_wcanon(lrec[e >> 5], e & 31u, k, s2, hi[i], lo[i], sl[i]);
                        sl[i] >>= 32u - tsl;
                        if (wo.shard_count > 1u && !simka_wide_owns(hi[i], lo[i], wo.shard_index, wo.shard_count)) pd[i] = false;
                    }
                    my_k += pd[i] ? 1u : 0u;
                }
                PH(3)
                // claim the slot (compare-and-swap of high word | count 1 into a free one: the winner stores the low word) or find it
                // taken: by the same high word -> compare the low word -> count; else the next slot
                while (__any(pd[0] || pd[1])) {
                    ull pv[NF]; bool wn[NF];
#pragma unroll
                    for (int i = 0; i < NF; i++) { pv[i] = 0; if (pd[i]) pv[i] = atomicCAS(&tab[sl[i]].x, SIMKA_EMPTY_KEY, (hi[i] << SKM_WF_CBITS) | 1ull); }
#pragma unroll
                    for (int i = 0; i < NF; i++) { wn[i] = pd[i] && pv[i] == SIMKA_EMPTY_KEY; if (wn[i]) tab[sl[i]].y = lo[i]; ndist += (uint32_t)__popcll(__ballot(wn[i])); }
                    asm volatile("" ::: "memory");          // the low words are stored (in program order: that is enough inside one wave) before any is compared
                    if (ndist > tfull) { fail = true; break; }
                    ull lw[NF]; bool cm[NF];
#pragma unroll
                    for (int i = 0; i < NF; i++) { cm[i] = pd[i] && !wn[i] && (pv[i] >> SKM_WF_CBITS) == hi[i]; lw[i] = 0; if (cm[i]) lw[i] = ((volatile ull *)&tab[sl[i]].y)[0]; }
#pragma unroll
                    for (int i = 0; i < NF; i++) {
                        const bool hit = cm[i] && lw[i] == lo[i];
                        if (hit) atomicAdd(&tab[sl[i]].x, 1ull);
                        if (wn[i] || hit) pd[i] = false; else if (pd[i]) sl[i] = (sl[i] + 1u) & tmask;
                    }
                }
                PH(4)
            }
            PH(5)
        }
        if (nrec) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (fail) {      // too many distinct k-mers for this table: the block kernel's (eight times the slots)
                for (uint32_t i = lane; i < TS; i += 64u) tab[i].x = SIMKA_EMPTY_KEY;
                if (lane == 0) redo_list[atomicAdd(redo_count, 1ull)] = part;
            } else {
                // ---- summary (SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-79): the solid records leave unordered,
                // one pass over the table (the slab has room for every distinct k-mer of it, solid or not)
                bool room = true;
                if (slab_pos + ndist > slab_end) {      // (wave-uniform) the rest of the slab stays empty: count 0
                    for (ull i = slab_pos + lane; i < slab_end; i += 64u) wo.cnt[i] = 0;
                    const ull want = ndist > (uint32_t)SKM_WF_SLAB ? (ull)ndist : (ull)SKM_WF_SLAB;
                    ull b_ = 0;
                    if (lane == 0) b_ = atomicAdd(wo.cursor, want);
                    slab_pos = ((ull)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b_ >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_);
                    slab_end = slab_pos + want;
                    if (slab_end > wo.cap) { room = false; slab_end = slab_pos; if (lane == 0) atomicOr(&wo.cursor[6], 2ull); }
                }
                // rows of 64 slots: the solid ones into a list (over the records and the map, which are done with), the others cleared;
                // then one lane per solid record: every lane of the (usually single) pass has work
                uint16_t *list = (uint16_t *)lrec;                 // [<= 3 TS / 4]
                uint32_t nl = 0;
                const uint32_t nrows = 1u << (tsl - 6u);
                for (uint32_t q = 0; q < nrows; q++) {
                    const uint32_t sl = q * 64u + lane;
                    const ull w = tab[sl].x;
                    const bool used = w != SIMKA_EMPTY_KEY;
                    const uint32_t c = (uint32_t)w & ((1u << SKM_WF_CBITS) - 1u);
                    const bool solid = used && !(c < amin || c > amax);
                    const ull m = __ballot(solid);
                    if (solid) list[nl + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)sl;
                    else if (used) tab[sl].x = SIMKA_EMPTY_KEY;
                    nl += (uint32_t)__popcll(m);
                }
                asm volatile("" ::: "memory");
                PH(6)
                ull D = 0, N = 0, Q = 0;
                for (uint32_t i = lane; i < nl; i += 64u) {
                    const uint32_t sl = ((volatile uint16_t *)list)[i];
                    const ulonglong2 e_ = tab[sl];
                    const ull w = e_.x, lo_ = e_.y;
                    tab[sl].x = SIMKA_EMPTY_KEY;
                    const uint32_t c = (uint32_t)w & ((1u << SKM_WF_CBITS) - 1u);
                    D++; N += c; Q += (ull)c * (ull)c;
                    if (room) {
                        const ull at = slab_pos + i;
                        wo.hi[at] = w >> SKM_WF_CBITS; wo.lo[at] = lo_; wo.cnt[at] = c;
                        if (o.hist) count_hist(o, lhist, c);
                    }
                }
                if (room) slab_pos += nl;
                asm volatile("" ::: "memory");          // (the list is read before the next partition's records overwrite it)
                PH(7)
                if (lane == 0) bt_dall += ndist;
                bt_D += D; bt_N += N; bt_Q += Q; bt_kocc += my_k;
            }
        }
        nrec = nrec_n; rbase = rbase_n; rc0 = rc0_n;
    }
    for (ull i = slab_pos + lane; i < slab_end; i += 64u) wo.cnt[i] = 0;
    PH_FLUSH
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    if (bt_kocc) atomicAdd(&s_tot[4], bt_kocc);
    __syncthreads();
    if (o.hist)
        for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += SKM_WF_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    if (tid < 5 && s_tot[tid]) atomicAdd(&wo.cursor[1 + tid], s_tot[tid]);
}

// --------------------------------------------------------------------------------------------
// k_skm_count_fast: the common case of k_skm_count -- the partition's distinct k-mers fit the table in ONE round.
//   * every wave expands its own 64 records: the records go to a wave-private LDS copy together with the index of their first
//     k-mer among the wave's k-mers, and ONE bit per record marks that index in a bitmap.  K-mer f of the wave then finds its
//     record with two mbcnt (records before f = set bits below f; the 64 bits of a chunk of k-mers are wave-uniform), cuts
//     itself out of the record (funnel shift), reverse complement, canonical, slot hash;
//   * SKM_FAST_U k-mers per lane are in flight through ONE straight-line insert (64-bit CAS + counter add).  A k-mer that finds
//     its slot taken by another k-mer does not loop: ballot + mbcnt compact the losers into a wave-private retry queue (key, next
//     slot), and the queue is drained 64 dense lanes at a time -- so a wave never idles 60 lanes while 4 keep probing;
//   * the records of the NEXT partition are loaded (registers) before the summary of this one;
//   * arena slab state double-buffered in LDS, so the emit needs no barrier of its own.
// A partition whose inserts overflow a sort block of the table goes to the redo list (k_skm_count takes it in rounds).
// --------------------------------------------------------------------------------------------
template <bool GATHER>
__global__ void __launch_bounds__(SKM_FAST_BLOCK, 4)       // (four blocks per CU = four waves per SIMD: at most 128 VGPRs)
k_skm_count_fast(SimkaSkmSrc src, SimkaSkmCfg cfg, SimkaKeyCfg kcfg, uint32_t amin, uint32_t amax, SimkaCountOut o,
                 const uint32_t *flag, ull *kocc_owned, uint32_t *redo_list, ull *redo_count) {
    if (*flag) return;
    const uint4 *recs = src.recs; const uint32_t *pstart = src.pstart, *pcnt = src.pcnt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ull *s_tot = (ull *)smem;                          // [5]
    ull &s_base = *(ull *)(smem + 48);
    uint32_t &s_fail = *(uint32_t *)(smem + 56);
    uint32_t &s_ok = *(uint32_t *)(smem + 60);
    ull *s_slab = (ull *)(smem + 64);                  // [2][2] (pos, end), double-buffered by iteration parity
    uint32_t *tmp = (uint32_t *)(smem + 128);          // [BLOCK/64]
    constexpr uint32_t TS = SKM_FAST_TS, SPT = TS / SKM_FAST_BLOCK, TSL = 11, NW = SKM_FAST_BLOCK / 64;
    ull *tkeys = (ull *)(smem + SKM_FAST_HEAD);        // [TS]
    uint32_t *tcnt = (uint32_t *)(tkeys + TS);         // [TS]
    uint4 *lrec = (uint4 *)(tcnt + TS);                // [BLOCK]: 64 per wave
    uint32_t *lhist = (uint32_t *)(lrec + SKM_FAST_BLOCK);     // [SKM_FAST_HBINS] (complex only)
    unsigned char *wreg0 = (unsigned char *)(lhist + (o.hist ? SKM_FAST_HBINS : 0));     // [NW][SKM_FAST_WREG] wave-private regions
    // GATHER: chunk numbering and start of every level-1 bucket, then per wave the piece table of the partition it loads next
    uint32_t *s_cb = (uint32_t *)(wreg0 + NW * SKM_FAST_WREG);     // [SKM_MAXB1 + 1]
    uint32_t *s_bs = s_cb + SKM_MAXB1 + 1;                         // [SKM_MAXB1] (a lane's record buffer holds < 2^32 records)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // (wave w TABULATES the pieces in the chunks c = NW l + w, l < GNL, of the next partition's bucket -- a quarter of the table each,
    //  no redundant loads; the records are then split evenly over the waves, whichever wave tabulated their piece: the tables are
    //  double-buffered by partition parity, written before the barrier that ends a partition's inserts and read by every wave after it)
    constexpr uint32_t GNL = SKM_G_MAXCH / NW;
    unsigned char *gtab0 = (unsigned char *)(s_bs + SKM_MAXB1);                                   // [2][NW] { u32 gpre[GNL]; u16 gst[GNL]; }
    uint32_t *s_gt = (uint32_t *)(gtab0 + 2 * NW * SKM_G_WBYTES);                                 // [2][NW] records each wave tabulated
    auto gpre_of = [&](uint32_t buf, uint32_t w) -> uint32_t * { return (uint32_t *)(gtab0 + (buf * NW + w) * SKM_G_WBYTES); };
    auto gst_of = [&](uint32_t buf, uint32_t w) -> uint16_t * { return (uint16_t *)(gtab0 + (buf * NW + w) * SKM_G_WBYTES + GNL * 4); };
    uint32_t gbuf = 0;                  // table buffer of the current partition
    uint32_t gT0 = 0, gT1 = 0, gT2 = 0; // records tabulated by the waves before wave 1 / 2 / 3 (current partition)
    const uint32_t nparts = 1u << cfg.pb;
    const uint32_t gl2 = cfg.pb - cfg.l1, gm2 = (1u << gl2) - 1u;
    if (GATHER) {
        const uint32_t B1 = 1u << cfg.l1;
        for (uint32_t i = tid; i <= B1; i += SKM_FAST_BLOCK) { s_cb[i] = src.cbase[i]; if (i < B1) s_bs[i] = (uint32_t)src.b1_start[i]; }
    }
    const ull sample_base = *o.sample_base;
    for (uint32_t i = tid; i < TS; i += SKM_FAST_BLOCK) { tkeys[i] = SIMKA_EMPTY_KEY; tcnt[i] = 0; }
    if (o.hist) for (uint32_t i = tid; i < SKM_FAST_HBINS; i += SKM_FAST_BLOCK) lhist[i] = 0;
    if (tid < 5) s_tot[tid] = 0;
    if (tid < 4) s_slab[tid] = 0;
    if (tid == 0) s_fail = 0;
    ull bt_dall = 0, bt_D = 0, bt_N = 0, bt_Q = 0, bt_kocc = 0;
    PH_DECL
#ifdef SIMKA_PHASE_PROF
    const ull blk_t0 = wall_clock64();
    ull dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DBG_ADD(i, v) { if (lane == 0) dbg[(i) - 8] += (ull)(v); }
#else
#define DBG_ADD(i, v)
#endif
    uint4 *wrec = lrec + wave * 64u;
    ull *bm64 = (ull *)(wreg0 + wave * SKM_FAST_WREG);              // [SKM_FAST_BMW] bit f set: a record starts at k-mer f
    ull *qk = bm64 + SKM_FAST_BMW;                                  // [QCAP] retry queue: canonical k-mers ...
    uint16_t *qm = (uint16_t *)(qk + SKM_FAST_QCAP);                // [QCAP] ... and the slot to try next
    constexpr uint32_t bmask = (TS >> SKM_SORT_BITS) - 1u;        // probing stays inside the sort block (TS / 8 slots)
    constexpr uint32_t U = SKM_FAST_U;
    uint32_t qn = 0;                                                // entries in the queue (wave-uniform)
    // one dense pass over the tail of the retry queue.  An entry = (k-mer, next slot to try | passes << 11).  The lane LOOKS at four
    // slots ahead (plain reads, one LDS round trip): a slot that holds another k-mer keeps it for the rest of the partition, so
    // it can be skipped without an atomic; the first slot that holds this k-mer takes a plain counter add, the first empty one a
    // CAS.  Only a CAS lost to another k-mer, or four occupied slots, send the entry back -- the tail of a wave's queue empties in
    // one or two passes instead of one pass per probe.  After 31 passes (124 slots) the k-mer gives up: redo list.
    auto drain = [&]() {
        const uint32_t n = qn < 64u ? qn : 64u;
        const uint32_t e = qn - n + lane;
        const bool a = lane < n;
        ull key = 0; uint32_t meta = 0;
        if (a) { key = qk[e]; meta = qm[e]; }
        qn -= n;
        bool again = false;
        uint32_t slot = meta & (TS - 1u);
        if (a) {
            const uint32_t bb = slot & ~bmask;
            const uint32_t s1 = bb | ((slot + 1u) & bmask), s2 = bb | ((slot + 2u) & bmask), s3 = bb | ((slot + 3u) & bmask);
            const ull w0 = tkeys[slot], w1 = tkeys[s1], w2 = tkeys[s2], w3 = tkeys[s3];
            const bool h0 = w0 == SIMKA_EMPTY_KEY || w0 == key, h1 = w1 == SIMKA_EMPTY_KEY || w1 == key, h2 = w2 == SIMKA_EMPTY_KEY || w2 == key, h3 = w3 == SIMKA_EMPTY_KEY || w3 == key;
            const uint32_t st = h0 ? slot : h1 ? s1 : h2 ? s2 : s3;
            const ull ws = h0 ? w0 : h1 ? w1 : h2 ? w2 : w3;
            if (!(h0 || h1 || h2 || h3)) { again = true; slot = bb | ((slot + 4u) & bmask); }
            else if (ws == key) atomicAdd(&tcnt[st], 1u);
            else {
                const ull prev = atomicCAS(&tkeys[st], SIMKA_EMPTY_KEY, key);
                if (prev == SIMKA_EMPTY_KEY || prev == key) atomicAdd(&tcnt[st], 1u);
                else { again = true; slot = bb | ((st + 1u) & bmask); }
            }
            if (again) {
                const uint32_t passes = (meta >> 11) + 1u;
                if (passes >= 32u) { s_fail = 1u; again = false; }
                meta = slot | (passes << 11);
            }
        }
        const ull am = __ballot(again);
        if (am) {
            if (again) { const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u)); qk[pos] = key; qm[pos] = (uint16_t)meta; }
            qn += (uint32_t)__popcll(am);
        }
    };

    // Work items = the partitions this shard (or pass) owns, handed out DYNAMICALLY in chunks of SKM_FAST_WCHUNK: chunk blockIdx.x
    // first, then whatever the global counter says (one word serves ~90 grabs per microsecond: one grab per partition would cap
    // the kernel at 6 ms per C3 sample, one per 8 partitions costs nothing).  Thread 0 grabs the chunk after next at the first item
    // of a chunk (a returning global atomic, long back when it is needed) and publishes it in LDS before the summary barrier.
    const uint32_t sh_c = cfg.shard_count, sh_i = cfg.shard_index;
    const uint32_t nwork = nparts > sh_i ? (nparts - sh_i + sh_c - 1u) / sh_c : 0u;
    auto part_of = [&](uint32_t j_) -> uint32_t { return j_ < nwork ? j_ * sh_c + sh_i : 0xffffffffu; };
    uint32_t *s_next = (uint32_t *)(smem + 96);       // [2] the chunk after next, double-buffered
    ull *work_counter = redo_count + 1;
    // (a handful of partitions per block -- small samples: nothing to balance, static stride without any atomic)
    const bool dyn = nwork >= gridDim.x * 16u;
    if (dyn && tid == 0) s_next[0] = gridDim.x + (uint32_t)atomicAdd(work_counter, 1ull);
    uint32_t tog = 0, wpos = 0;                        // wpos: position inside the chunk
    // (few partitions per block -- small samples: smaller chunks, down to one partition per grab)
    const uint32_t wchunk = dyn ? min((uint32_t)SKM_FAST_WCHUNK, max(1u, nwork / (gridDim.x * 4u))) : 1u;
    uint32_t item = blockIdx.x * wchunk, iter = 0;
    uint32_t part = part_of(item);
    uint32_t nrec = 0, rbase = 0;
    uint4 pre = make_uint4(0, 0, 0, 0);
    // record i of the partition the wave's piece table describes (GATHER) / of the run that starts at rb_
    auto rec_at = [&](uint32_t rb_, uint32_t i) -> uint4 {
        if (GATHER) {
            // the wave that tabulated record i (the waves' records follow each other), then the last piece of its table that starts at
            // or before the record (empty pieces share their successor's prefix)
            const uint32_t w = (i >= gT0 ? 1u : 0u) + (i >= gT1 ? 1u : 0u) + (i >= gT2 ? 1u : 0u);
            const uint32_t li = i - (w == 0u ? 0u : w == 1u ? gT0 : w == 2u ? gT1 : gT2);
            const uint32_t *gp = gpre_of(gbuf, w); const uint16_t *gs = gst_of(gbuf, w);
            uint32_t l = 0;
#pragma unroll
            for (uint32_t stp = GNL / 2; stp; stp >>= 1) l += gp[l + stp] <= li ? stp : 0u;
            return recs[(size_t)rb_ + (size_t)(l * NW + w) * SKM_CS_CHUNK + gs[l] + (li - gp[l])];
        }
        return recs[rb_ + i];
    };
    auto prefetch = [&](uint32_t n_, uint32_t rb_) {
        const uint32_t nb = n_ < (uint32_t)SKM_FAST_BLOCK ? n_ : (uint32_t)SKM_FAST_BLOCK;
        const uint32_t per = (nb + NW - 1u) / NW;
        if (lane < per && wave * per + lane < nb) pre = rec_at(rb_, wave * per + lane);
    };
    // (count, start) of the next partition travel as a VECTOR load of lanes 0 / 1: a scalar load of these wave-uniform words would
    // share the LDS counter (lgkmcnt), and the first LDS wait of the insert phase would sit out its HBM latency.
    // GATHER: the table entries (start | end << 16) of the wave's piece in chunk NW lane + wave of the partition's bucket instead;
    // take_desc() turns them into the wave's piece table.
    auto load_desc = [&](uint32_t p_) -> uint32_t {
        uint32_t v = 0;
        if (GATHER) {
            if (p_ < nparts) {
                const uint32_t b = p_ >> gl2, p2 = p_ & gm2, gc0 = s_cb[b], nch = s_cb[b + 1u] - gc0, c = lane * NW + wave;
                if (lane < GNL && c < nch) { const uint16_t *row = src.ctab + (size_t)(gc0 + c) * src.cstride + p2; v = (uint32_t)row[0] | ((uint32_t)row[1] << 16); }
            }
        } else if (p_ < nparts && lane < 2u) { const uint32_t *q = lane == 0u ? pcnt : pstart; v = q[p_]; }
        return v;
    };
    // descriptor -> (records, start) of partition p_; GATHER: also the wave's piece table (the table of the partition before is dead:
    // its last batch has been loaded)
    // GATHER, before the barrier: the wave's quarter of the piece table of the partition that comes next, into the other buffer
    auto tabulate = [&](uint32_t d) {
        const uint32_t cA = (d >> 16) - (d & 0xffffu);          // (0 beyond the wave's pieces)
        const uint32_t iA = wave_incl_scan(cA);
        if (lane < GNL) { gpre_of(gbuf ^ 1u, wave)[lane] = iA - cA; gst_of(gbuf ^ 1u, wave)[lane] = (uint16_t)(d & 0xffffu); }
        if (lane == 63u) s_gt[(gbuf ^ 1u) * NW + wave] = iA;
    };
    // descriptor -> (records, start) of partition p_; GATHER (after the barrier): the other buffer becomes the current one
    auto take_desc = [&](uint32_t p_, uint32_t d, uint32_t &n_, uint32_t &rb_) {
        if (GATHER) {
            gbuf ^= 1u;
            const uint32_t t0 = s_gt[gbuf * NW], t1 = s_gt[gbuf * NW + 1u], t2 = s_gt[gbuf * NW + 2u], t3 = s_gt[gbuf * NW + 3u];
            static_assert(NW == 4, "four waves tabulate");
            gT0 = t0; gT1 = t0 + t1; gT2 = gT1 + t2;
            n_ = gT2 + t3;
            rb_ = p_ < nparts ? s_bs[p_ >> gl2] : 0u;
        } else { n_ = (uint32_t)__builtin_amdgcn_readlane((int)d, 0); rb_ = (uint32_t)__builtin_amdgcn_readlane((int)d, 1); }
    };
    __syncthreads();           // (the tables and s_cb / s_bs are in place)
    if (GATHER) { tabulate(load_desc(part)); __syncthreads(); }
    if (part < nparts) { const uint32_t d0 = GATHER ? 0u : load_desc(part); take_desc(part, d0, nrec, rbase); prefetch(nrec, rbase); }
    __syncthreads();
    uint32_t chunk_n = dyn ? s_next[0] : blockIdx.x + gridDim.x, chunk_n2 = 0;    // (published before the barrier above)
    while (part < nparts) {
        const bool first = wpos == 0u, last = wpos + 1u == wchunk;
        const uint32_t item_n = last ? chunk_n * wchunk : item + 1u;
        const uint32_t next = part_of(item_n);
        const uint32_t desc_n = load_desc(next);
        ull grab = 0;
        if (dyn && first && tid == 0) grab = atomicAdd(work_counter, 1ull);
        if (!dyn) chunk_n2 = chunk_n + gridDim.x;
        if (!GATHER && nrec == 0) {       // an empty partition (rare among the owned ones; GATHER: it takes the common path -- the next table is built before a barrier)
            if (dyn && first) {       // agree on the chunk after next through LDS right away
                if (tid == 0) s_next[tog ^ 1u] = gridDim.x + (uint32_t)grab;
                __syncthreads();
                tog ^= 1u;
                chunk_n2 = s_next[tog];
            }
            if (last) { chunk_n = chunk_n2; wpos = 0; } else wpos++;
            item = item_n;
            part = next; take_desc(next, desc_n, nrec, rbase);
            prefetch(nrec, rbase);
            continue;
        }
        // ---- expand + insert: every wave takes an equal share of the records (at most 64 per batch)
        PH(0)
        ull my_k = 0;
        // (the records of the SECOND batch -- about every other C3 partition has a few beyond the first 256 -- are asked for now and
        //  arrive behind the inserts of the first: loaded where they are needed, every wave sat out a trip to memory)
        uint4 pre2 = make_uint4(0, 0, 0, 0);
        if (nrec > (uint32_t)SKM_FAST_BLOCK) {
            const uint32_t nb2 = nrec - SKM_FAST_BLOCK < (uint32_t)SKM_FAST_BLOCK ? nrec - SKM_FAST_BLOCK : (uint32_t)SKM_FAST_BLOCK;
            const uint32_t per2 = (nb2 + NW - 1u) / NW;
            if (lane < per2 && wave * per2 + lane < nb2) pre2 = rec_at(rbase, SKM_FAST_BLOCK + wave * per2 + lane);
        }
        for (uint32_t b0 = 0; b0 < nrec; b0 += SKM_FAST_BLOCK) {
            const uint32_t nb = nrec - b0 < (uint32_t)SKM_FAST_BLOCK ? nrec - b0 : (uint32_t)SKM_FAST_BLOCK;
            const uint32_t per = (nb + NW - 1u) / NW;                       // records of this wave: [b0 + wave*per, +per)
            const uint32_t i = b0 + wave * per + lane;
            const bool mine = lane < per && wave * per + lane < nb;
            uint4 rc = b0 == (uint32_t)SKM_FAST_BLOCK ? pre2 : pre;
            if (b0 > (uint32_t)SKM_FAST_BLOCK) { if (mine) rc = rec_at(rbase, i); }
            uint32_t len = mine ? skm_rec_n(rc) : 0u;
            const uint32_t x = wave_incl_scan(len);
            const uint32_t kt = __builtin_amdgcn_readlane(x, 63);
            const uint32_t off = x - len;
            // the wave's copy of the record carries the index of its first k-mer where the partition id was
            wrec[lane] = make_uint4(rc.x, rc.y, rc.z, (rc.w & 63u) | (off << 6));
            if (lane < SKM_FAST_BMW) bm64[lane] = 0ull;
            if (len) atomicOr((uint32_t *)bm64 + (off >> 5), 1u << (off & 31u));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) my_k += kt;
            PH(1)
            // the whole bitmap in registers (word c in lane c): a chunk's 64 bits are then two readlanes away, no LDS round trip
            ull bmreg = 0;
            if (lane < SKM_FAST_BMW) bmreg = bm64[lane];
            const uint32_t bmlo = (uint32_t)bmreg, bmhi = (uint32_t)(bmreg >> 32);
            uint32_t rbefore = 0;                                      // records that start before the current chunk of 64 k-mers
            // record of k-mer f0 + 64 u + lane, for the U chunks of one iteration; software pipeline: the reads of iteration i + 1
            // are issued behind the CASes of iteration i (LDS returns in order), so an iteration exposes ONE round trip
            uint4 rx[U]; bool act[U];
            auto fetch = [&](uint32_t f0_) {
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t c = (f0_ >> 6) + u;                  // (wave-uniform; c < SKM_FAST_BMW while f0_ < kt)
                    act[u] = f0_ + 64u * u + lane < kt;
                    const uint32_t mlo = __builtin_amdgcn_readlane(bmlo, c & (SKM_FAST_BMW - 1u)), mhi = __builtin_amdgcn_readlane(bmhi, c & (SKM_FAST_BMW - 1u));
                    // records that start at or below the lane's k-mer = the set bits of the chunk's mask up to and INCLUDING the lane: bit 0 plus the
                    // bits below the lane of the mask shifted down by one (scalar unit) -- two mbcnt, no per-lane extraction of the own bit (round 6)
                    const ull m1 = (((ull)mhi << 32) | mlo) >> 1;
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, rbefore + (mlo & 1u) - 1u)) & 63u;      // (a lane beyond the last k-mer: any record)
                    rbefore += (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
                    rx[u] = wrec[r];
                }
            };
            fetch(0);
            // one iteration: 64 U k-mers of the wave.  TAIL: the wave's last, partly filled iteration -- only there a lane can lie beyond the
            // last k-mer (it swaps EMPTY for EMPTY and adds 0: straight-line code); the full iterations before it carry no activity mask,
            // no select of the key and no masked counter value (round 6: four vector instructions less per 64 k-mers)
            auto step = [&](uint32_t f0, auto tail_c) {
                constexpr bool TAIL = decltype(tail_c)::value;
                ull cu[U]; uint32_t su[U]; bool actc[U];
                // cut, reverse complement, canonical, slot
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    actc[u] = TAIL ? act[u] : true;
                    const uint64_t fw = skm_kmer_at(rx[u], (f0 + 64u * u + lane - (rx[u].w >> 6)) & 31u, cfg);
                    const uint64_t rv = skm_revcomp_k(fw, 64u - 2u * cfg.k);
                    cu[u] = fw < rv ? fw : rv;
                    su[u] = simka_key_hash32(cu[u]) >> (32u - TSL);
                }
                // all U inserts in flight together, the next iteration's records behind them
                ull pu[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) { if (TAIL && !actc[u]) cu[u] = SIMKA_EMPTY_KEY; pu[u] = atomicCAS(&tkeys[su[u]], SIMKA_EMPTY_KEY, cu[u]); }
                fetch(f0 + 64u * U);                 // (beyond the last k-mer: no lane is active, every lane reads some record)
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const bool ok = pu[u] == SIMKA_EMPTY_KEY || pu[u] == cu[u];
                    atomicAdd(&tcnt[su[u]], (actc[u] && ok) ? 1u : 0u);
                    const bool lost = actc[u] && !ok;
                    const ull lm = __ballot(lost);
                    if (lm) {       // (bmask >= 1: the next slot is never the home slot)
                        if (lost) {
                            const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                            qk[pos] = cu[u]; qm[pos] = (uint16_t)((su[u] & ~bmask) | ((su[u] + 1u) & bmask));
                        }
                        qn += (uint32_t)__popcll(lm);
                    }
                }
                while (qn >= 64u) drain();
            };
            const uint32_t kfull = kt - kt % (64u * U);
            for (uint32_t f0 = 0; f0 < kfull; f0 += 64u * U) step(f0, std::false_type());
            if (kfull < kt) step(kfull, std::true_type());
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's bitmap / records are rewritten by the next batch
            PH(2)
        }
        DBG_ADD(8, my_k) DBG_ADD(12, 1) DBG_ADD(14, qn)
#if SKM_FAST_TAIL
        // the last few entries of the queue (what a pass re-queues: 1.2 passes of ~5 lanes per wave and partition on C3) are not worth a pass of
        // the look-ahead drain (~60 VALU for 64 lanes): each of them walks on with one CAS per slot until it is placed (round 6: -0.8 %)
        while (qn > SKM_FAST_TAIL) { drain(); DBG_ADD(13, 1) }
        if (qn) {
            const bool a = lane < qn;
            ull key = 0; uint32_t slot = 0, passes = 0;
            if (a) { key = qk[lane]; const uint32_t meta = qm[lane]; slot = meta & (TS - 1u); passes = meta >> 11; }
            bool pend_ = a;
            while (__ballot(pend_)) {
                if (pend_) {
                    const ull prev = atomicCAS(&tkeys[slot], SIMKA_EMPTY_KEY, key);
                    if (prev == SIMKA_EMPTY_KEY || prev == key) { atomicAdd(&tcnt[slot], 1u); pend_ = false; }
                    else {
                        slot = (slot & ~bmask) | ((slot + 1u) & bmask);
                        if (++passes >= 128u) { s_fail = 1u; pend_ = false; }      // (a sort block has 128 slots: full)
                    }
                }
            }
            qn = 0;
            DBG_ADD(13, 1)
        }
#else
        while (qn) { drain(); DBG_ADD(13, 1) }
#endif
        if (GATHER) tabulate(desc_n);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        PH(3)
        // ---- the next partition's records travel while this one is summarised
        uint32_t nrec_n, rbase_n;
        take_desc(next, desc_n, nrec_n, rbase_n);
        prefetch(nrec_n, rbase_n);
        // ---- summary in slot order (SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-79)
        uint32_t cs[SPT]; ull ks[SPT];
        uint32_t nsol = 0, ndall = 0;
        ull N = 0, Q = 0;
        {   // the thread's 8 slots: vector loads, then the slots are reset unconditionally (vector stores, no branches).
            // The 16 threads of a group own one sort block of 128 slots (only the order of the BLOCKS matters downstream): thread l of the
            // group takes the slot pairs 2 l + 32 m, m = 0..3 -- the 16-byte key vectors l + 16 m, consecutive over the lanes of every
            // access (eight consecutive slots per thread made every 16-byte access a 2- to 4-way bank conflict) -- and the halves of the
            // count vectors (l >> 1) + 8 m that hold them (neighbouring lanes read the same vector).
            static_assert(SPT == 8 && TS == 16u * 128u && SKM_FAST_BLOCK == 256, "16 groups of 16 threads, a sort block each");
            const uint32_t grp = tid >> 4, l = tid & 15u;
            // (round 6: a lane reads its own two counts per row as one 8-byte word -- consecutive over the lanes like the key vectors -- instead of
            //  the 16-byte vector it shared with its neighbour and a select per count)
            uint2 *c2 = (uint2 *)(tcnt + grp * 128u) + l; ulonglong2 *k2 = (ulonglong2 *)(tkeys + grp * 128u) + l;
            const uint2 ca = c2[0], cb = c2[16], cc = c2[32], cd = c2[48];
            const ulonglong2 ka = k2[0], kb = k2[16], kc = k2[32], kd = k2[48];
            const uint2 z = make_uint2(0, 0); const ulonglong2 ek = make_ulonglong2(SIMKA_EMPTY_KEY, SIMKA_EMPTY_KEY);
            c2[0] = z; c2[16] = z; c2[32] = z; c2[48] = z; k2[0] = ek; k2[16] = ek; k2[32] = ek; k2[48] = ek;
            cs[0] = ca.x; cs[1] = ca.y; cs[2] = cb.x; cs[3] = cb.y; cs[4] = cc.x; cs[5] = cc.y; cs[6] = cd.x; cs[7] = cd.y;
            ks[0] = ka.x; ks[1] = ka.y; ks[2] = kb.x; ks[3] = kb.y; ks[4] = kc.x; ks[5] = kc.y; ks[6] = kd.x; ks[7] = kd.y;
        }
        uint32_t w_ndall = 0;                   // distinct k-mers of the wave's slots: counted on the scalar unit from the compare masks
#pragma unroll
        for (uint32_t q = 0; q < SPT; q++) {
            const uint32_t c = cs[q];
            const bool any = c != 0u, sol = any && !(c < amin || c > amax);
            w_ndall += (uint32_t)__popcll(__ballot(any));
            nsol += sol ? 1u : 0u;
            cs[q] = sol ? c : 0u;
            // (the filtered count: no select per accumulator, D is nsol, one multiply-add per sum -- round 6: -3 % of the kernel)
            asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(N) : "v"(cs[q]) : "vcc");      // N += cs[q] (64-bit) in ONE vector instruction
            Q += (ull)cs[q] * (ull)cs[q];
        }
        ndall = lane == 0u ? w_ndall : 0u;
        const ull D = nsol;
        const bool failed = s_fail != 0u;
        // the wave's solid records, in slot order, go to its (now empty) retry queue: keys [0, cap), counts behind them
        constexpr uint32_t wcap = (SKM_FAST_QCAP * 10u) / 12u;                    // records the region takes
        ull *wk = qk; uint32_t *wc = (uint32_t *)(wk + wcap);
        const uint32_t winc = wave_incl_scan(nsol);
        const uint32_t wtot = __builtin_amdgcn_readlane(winc, 63);
        const bool staged = wtot <= wcap;
        if (staged) {
            uint32_t p = winc - nsol;
#pragma unroll
            for (uint32_t q = 0; q < SPT; q++) if (cs[q]) { wk[p] = ks[q]; wc[p] = cs[q]; p++; }
        }
        PH(4)
        const uint32_t par = iter & 1u;
        iter++;
        // block-level offsets of the waves (one barrier)
        if (lane == 63u) tmp[par * 16u + wave] = winc;
        if (dyn && first && tid == 0) s_next[tog ^ 1u] = gridDim.x + (uint32_t)grab;
        __syncthreads();
        if (dyn && first) { tog ^= 1u; chunk_n2 = s_next[tog]; }
        uint32_t wpre = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; w++) { const uint32_t t = tmp[par * 16u + w]; if (w < wave) wpre += t; total += t; }
        PH(5)
        const ull sp_ = s_slab[par * 2u], se_ = s_slab[par * 2u + 1u];
        const bool fits = !failed && (total == 0 || (sp_ + total <= se_ && sp_ - sample_base + total <= 0xffffffffull));
        ull base_ = sample_base;
        bool ok_ = !failed;
        if (fits) {
            if (total) base_ = sp_;
            if (tid == 0) { s_slab[(par ^ 1u) * 2u] = sp_ + total; s_slab[(par ^ 1u) * 2u + 1u] = se_; }
            if (tid == 64) o.foff[part] = (uint32_t)(base_ - sample_base);
            if (tid == 128) o.fcnt[part] = total;
        } else {
            if (tid == 0) {
                uint32_t ok = 1;
                ull slab_pos = sp_, slab_end = se_;
                if (failed) { const ull w = atomicAdd(redo_count, 1ull); redo_list[w] = part; ok = 0; s_fail = 0u; }
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
        if (ok_ && o.seg_rows) {      // the merge's index of this segment: thread 16 b + 15 owns the last slots of key-hash block b, its inclusive prefix is the block's end
            if ((tid & 15u) == 15u) o.seg_rows[((size_t)part * o.nb_samples) * SIMKA_SEG_BLOCKS + (tid >> 4)] = (uint16_t)(wpre + winc);
            if (tid == 192) o.seg_abs[(size_t)part * o.nb_samples] = base_;
        }
        PH(6)
        if (ok_) {
            bt_dall += ndall; bt_D += D; bt_N += N; bt_Q += Q; bt_kocc += my_k;
            if (staged) {      // one record per lane: coalesced stores, one key mix per record
                for (uint32_t r = lane; r < wtot; r += 64u) {
                    const ull key = wk[r]; const uint32_t c = wc[r];
                    const ull pos = base_ + wpre + r;
                    o.solid_keys[pos] = key; o.solid_counts[pos] = c;
                    if (o.hist) count_hist(o, lhist, c, SKM_FAST_HBINS);
                }
            } else {
                ull pos = base_ + wpre + winc - nsol;
#pragma unroll
                for (uint32_t q = 0; q < SPT; q++) {
                    if (cs[q]) {
                        o.solid_keys[pos] = ks[q]; o.solid_counts[pos] = cs[q]; pos++;
                        if (o.hist) count_hist(o, lhist, cs[q], SKM_FAST_HBINS);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staging region becomes the wave's retry queue again
        PH(7)
        if (last) { chunk_n = chunk_n2; wpos = 0; } else wpos++;
        item = item_n;
        part = next; nrec = nrec_n; rbase = rbase_n;
    }
    PH_FLUSH
#ifdef SIMKA_PHASE_PROF
    if (lane == 0 && o.phase) for (int i_ = 0; i_ < 8; i_++) if (i_ < 1 || i_ > 3) atomicAdd(&o.phase[8 + i_], dbg[i_]);
    if (tid == 0 && o.phase) { const ull t1_ = wall_clock64(), el = t1_ - blk_t0; atomicMax(&o.phase[9], el); atomicAdd(&o.phase[10], el); atomicAdd(&o.phase[11], 1ull);
                               if (blockIdx.x < 2048u) { o.phase[16 + 2 * blockIdx.x] = blk_t0; o.phase[17 + 2 * blockIdx.x] = t1_; } }
#endif
    if (o.hist) {
        __syncthreads();
        for (uint32_t i = tid; i < SKM_FAST_HBINS; i += SKM_FAST_BLOCK)
            if (lhist[i]) atomicAdd(&o.hist[(size_t)o.sample * SIMKA_HIST_MAX + i], (ull)lhist[i]);
    }
    if (bt_dall) atomicAdd(&s_tot[0], bt_dall);
    if (bt_D) { atomicAdd(&s_tot[1], bt_D); atomicAdd(&s_tot[2], bt_N); atomicAdd(&s_tot[3], bt_Q); }
    if (bt_kocc) atomicAdd(&s_tot[4], bt_kocc);
    __syncthreads();
    if (tid == 0) {
        ull *t = o.totals + o.sample;
        const size_t ns_ = o.nb_samples;
        if (s_tot[0]) atomicAdd(&t[SIMKA_TOT_DALL * ns_], s_tot[0]);
        if (s_tot[1]) { atomicAdd(&t[SIMKA_TOT_D * ns_], s_tot[1]); atomicAdd(&t[SIMKA_TOT_N * ns_], s_tot[2]); atomicAdd(&t[SIMKA_TOT_Q * ns_], s_tot[3]); }
        if (s_tot[4] && kocc_owned) atomicAdd(kocc_owned, s_tot[4]);
    }
}

#endif
