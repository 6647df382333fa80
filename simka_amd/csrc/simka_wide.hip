// simka_wide.hip -- 32 <= k <= 63: canonical k-mers are up to 126 bits (the reference's Kmer<span=64>
// instantiation, KSIZE_LIST of its CMakeLists.txt).  They do not fit the 64-bit keys of the hash pipeline
// (simka_kernels.hip), so this path keeps every k-mer as a (hi, lo) pair of 64-bit words and COUNTS BY SORTING:
//
//   count:  k_wscan (rolling forward / reverse-complement words per thread, ref: gatb ModelCanonical as used at
//           src/minikc/MiniKC.hpp:152-158)  ->  two stable LSD radix sorts (lo, then hi)  ->  run heads  ->
//           abundance filter + totals (SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-79)  ->
//           the sample's solid spectrum, sorted, appended to the wide arena
//   merge:  all samples' solid records sorted by k-mer (the N-way merge, ref: src/SimkaMerge.cpp:1164-1264)  ->
//           groups of >= 2 samples (the gate, ref: :1307-1326)  ->  the same CSR (entries / groups / spans) the hash
//           merge builds, consumed by the same k_pairs kernel.
//
// The radix sorts and prefix sums are hand-written too (simka_sort.hip: LSD radix sort, 8 bits per pass, wave-match ranking).
// Much slower than the hash pipeline (C2: ~100 ms against 12 ms per step; every base position goes through two 64-bit radix
// sorts), exact, and an independent cross-check of it: with SIMKA_SORT_PATH=1 the k <= 31 tests run through this path and
// must give bit-identical statistics (they do: goldens included).
#include <hip/hip_runtime.h>
#include "simka_efence.h"      // (test builds: -DSIMKA_EFENCE)
#include "simka_trace.h"       // SIMKA_FAULT_TRACE=1: registry of device ranges + ring of launches, dumped when the process dies
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>

#include "simka_device.h"
#include "simka_kernels.h"
#include "simka_wide.h"
#include "simka_sort.hip"

typedef unsigned long long ull;

#define WCHK(call)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) { w->err = std::string(#call) + " failed: " + hipGetErrorString(e_); return SIMKA_WIDE_ERR_HIP; } \
    } while (0)

struct SimkaWide {
    int device = 0;
    uint32_t nb_samples = 0, k = 0, W = 0;
    uint32_t shard_index = 0, shard_count = 1;   // partition shard: the k-mers this context keeps (wide_owns)
    hipStream_t stream = nullptr;
    std::string err;
    // wide arena: the solid spectra of all samples, each sorted by (hi, lo) -- or, for a sample adopted from the partitioned count
    // (simka_wide_adopt), in any order until someone asks for its order (export, partition runs): the merge sorts all records anyway
    ull *a_hi = nullptr, *a_lo = nullptr; uint32_t *a_cnt = nullptr;
    uint64_t a_cap = 0, a_used = 0;
    std::vector<uint64_t> s_off, s_n;            // per sample: offset and number of solid records
    std::vector<uint8_t> s_sorted;
    uint64_t nb_full_sorts = 0;                  // counts and merges that sorted by k-mer instead of bucketing (fallbacks, forced)
    // scratch (grown on demand)
    void *scratch[12] = { nullptr }; uint64_t scratch_bytes[12] = { 0 };
    // CSR handed to k_pairs (owned here, valid until the next reset / merge)
    ull *entries = nullptr; uint32_t *groups = nullptr; SimkaSpan *spans = nullptr, *huge = nullptr; ull *cursors = nullptr;
};

template <typename T>
static int wide_buf(SimkaWide *w, int slot, uint64_t n, T **out) {
    const uint64_t need = n * sizeof(T) + 256;
    if (w->scratch_bytes[slot] < need) {
        if (w->scratch[slot]) { WCHK(hipStreamSynchronize(w->stream)); WCHK(hipFree(w->scratch[slot])); w->scratch[slot] = nullptr; w->scratch_bytes[slot] = 0; }
        WCHK(hipMalloc(&w->scratch[slot], need + need / 8));
        w->scratch_bytes[slot] = need + need / 8;
    }
    *out = (T *)w->scratch[slot];
    return 0;
}

// --------------------------------------------------------------------------------------------
// k_wscan: thread = WSEG consecutive END positions of the concatenated base array.  It warms up on the k-1 bases before
// its first position (inside the same read), then rolls the forward and reverse-complement words one base at a time.
// Only whole k-mers are written, compacted (block scan of the per-thread counts + one global atomic per block).
// --------------------------------------------------------------------------------------------
#define WSEG 32

struct WideScanArgs { const uint64_t *packed; uint64_t nb_bases, nb_words; const uint64_t *offsets; uint64_t nb_reads; uint32_t fixed_len, k;
                      uint32_t shard_index, shard_count; };

#define wide_owns simka_wide_owns      // (simka_device.h: the hash path for k <= 51 applies the same rule)

__device__ __forceinline__ uint32_t wbase(const uint64_t *packed, uint64_t p) { return (uint32_t)(packed[p >> 5] >> ((p & 31u) * 2u)) & 3u; }

// roll the forward / reverse-complement words over the thread's positions [p0, pend); warm up on the k-1 bases before p0 (same
// read).  EMIT: write the (owned) canonical k-mers from slot wr on; returns how many there are.
template <bool EMIT, bool SHARDED>
__device__ __forceinline__ uint32_t wroll(const WideScanArgs &a, uint64_t p0, uint64_t pend, uint64_t rd, uint64_t rstart, uint64_t rend, ull *khi, ull *klo, ull wr) {
    const uint32_t k = a.k;
    const uint32_t W = 2u * k;
    const ull mlo = W >= 64u ? ~0ull : ((1ull << W) - 1ull);
    const ull mhi = W > 64u ? ((1ull << (W - 64u)) - 1ull) : 0ull;
    const uint32_t top = 2u * (k - 1u);                      // bit position of the first base in the reverse-complement word
    ull fh = 0, fl = 0, rh = 0, rl = 0;
    uint32_t have = 0;                                        // bases of the current read inside the window (capped at k)
    uint32_t n = 0;
    uint64_t p = p0 > rstart + (k - 1u) ? p0 - (k - 1u) : rstart;
    ull word = a.packed[p >> 5] >> ((p & 31u) * 2u);        // the bases from p on, refilled every 32 bases
    for (; p < pend; p++) {
        while (p >= rend) {        // next read (fixed length: arithmetic; else the offsets array)
            rd++;
            rstart = rend;
            rend = a.fixed_len ? rstart + a.fixed_len : a.offsets[rd + 1];
            have = 0; fh = fl = rh = rl = 0;
        }
        if ((p & 31u) == 0u) word = a.packed[p >> 5];
        const ull c = word & 3ull;
        word >>= 2;
        fh = ((fh << 2) | (fl >> 62)) & mhi;                 // forward: (f << 2 | c) & mask
        fl = ((fl << 2) | c) & mlo;
        rl = (rl >> 2) | (rh << 62);                         // reverse complement: (r >> 2) | (c ^ 2) << 2(k-1)
        rh >>= 2;
        const ull cc = c ^ 2ull;
        if (top >= 64u) rh |= cc << (top - 64u); else rl |= cc << top;
        if (have < k) have++;
        if (p >= p0 && have >= k) {
            const bool fsm = fh < rh || (fh == rh && fl < rl);
            const ull ch = fsm ? fh : rh, cl = fsm ? fl : rl;
            if (!SHARDED || wide_owns(ch, cl, a.shard_index, a.shard_count)) {
                if (EMIT) { khi[wr] = ch; klo[wr] = cl; wr++; }
                n++;
            }
        }
    }
    return n;
}

// ---- 64 <= k <= 127 (the reference's Kmer<span=96/128>, ref: CMakeLists.txt:66-71, src/SimkaPotara.cpp:132-141): the canonical k-mer is
// rolled in FOUR words and leaves the scan as a 126-bit FINGERPRINT (hi: 62 bits, lo: 64 bits -- the key geometry of k = 63), so that
// everything behind the scan (bucket count, sort fallback, bucket merge, export / import, shards) is the two-word machinery unchanged.
// Two independently keyed 64-bit mixes of the four words; two DISTINCT k-mers of a run share a fingerprint with probability
// < (number of distinct k-mers)^2 / 2^127 (1e-16 for 1e11 of them) -- not a proof of exactness like the word-exact paths of k <= 63,
// stated as such in DESIGN.md; the oracle keeps the whole k-mers and the tests compare bit for bit.
__device__ __forceinline__ ull wfmix(ull x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ void wfinger(const ull c[4], ull &hi, ull &lo) {
    const ull a = wfmix(c[0] ^ wfmix(c[1] ^ wfmix(c[2] ^ wfmix(c[3] ^ 0x9E3779B97F4A7C15ULL))));
    const ull b = wfmix(c[3] + 0xD6E8FEB86659FD93ULL * (wfmix(c[2] + 0xD6E8FEB86659FD93ULL * (wfmix(c[1] + 0xD6E8FEB86659FD93ULL * wfmix(c[0] ^ 0xBF58476D1CE4E5B9ULL))))));
    hi = a & 0x3fffffffffffffffULL; lo = b;
}
template <bool EMIT, bool SHARDED>
__device__ __forceinline__ uint32_t wroll4(const WideScanArgs &a, uint64_t p0, uint64_t pend, uint64_t rd, uint64_t rstart, uint64_t rend, ull *khi, ull *klo, ull wr) {
    const uint32_t k = a.k, W = 2u * k;                      // 128 <= W <= 254
    const uint32_t tw = (W - 1u) >> 6;                       // top word (1 for k = 64, else 2 or 3)
    const ull mtop = (W & 63u) ? ((1ull << (W & 63u)) - 1ull) : ~0ull;
    const uint32_t top = 2u * (k - 1u);                      // bit position of the first base in the reverse complement
    ull f[4] = { 0, 0, 0, 0 }, r[4] = { 0, 0, 0, 0 };
    uint32_t have = 0, n = 0;
    uint64_t p = p0 > rstart + (k - 1u) ? p0 - (k - 1u) : rstart;
    ull word = a.packed[p >> 5] >> ((p & 31u) * 2u);
    for (; p < pend; p++) {
        while (p >= rend) {
            rd++; rstart = rend;
            rend = a.fixed_len ? rstart + a.fixed_len : a.offsets[rd + 1];
            have = 0; f[0] = f[1] = f[2] = f[3] = 0; r[0] = r[1] = r[2] = r[3] = 0;
        }
        if ((p & 31u) == 0u) word = a.packed[p >> 5];
        const ull c = word & 3ull;
        word >>= 2;
        f[3] = (f[3] << 2) | (f[2] >> 62); f[2] = (f[2] << 2) | (f[1] >> 62); f[1] = (f[1] << 2) | (f[0] >> 62); f[0] = (f[0] << 2) | c;
        if (tw == 1u) { f[1] &= mtop; f[2] = 0; f[3] = 0; } else if (tw == 2u) { f[2] &= mtop; f[3] = 0; } else f[3] &= mtop;
        r[0] = (r[0] >> 2) | (r[1] << 62); r[1] = (r[1] >> 2) | (r[2] << 62); r[2] = (r[2] >> 2) | (r[3] << 62); r[3] >>= 2;
        r[top >> 6] |= (c ^ 2ull) << (top & 63u);
        if (have < k) have++;
        if (p >= p0 && have >= k) {
            bool fsm = false, decided = false;
#pragma unroll
            for (int q = 3; q >= 0; q--) if (!decided && f[q] != r[q]) { fsm = f[q] < r[q]; decided = true; }
            ull ch, cl;
            const ull cw[4] = { fsm ? f[0] : r[0], fsm ? f[1] : r[1], fsm ? f[2] : r[2], fsm ? f[3] : r[3] };
            wfinger(cw, ch, cl);
            if (!SHARDED || wide_owns(ch, cl, a.shard_index, a.shard_count)) {
                if (EMIT) { khi[wr] = ch; klo[wr] = cl; wr++; }
                n++;
            }
        }
    }
    return n;
}

template <bool SHARDED>
__global__ void __launch_bounds__(256)
k_wscan(WideScanArgs a, ull *khi, ull *klo, ull *nvalid) {
    __shared__ uint32_t s_wave[4];
    __shared__ ull s_base;
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * WSEG;
    const uint32_t k = a.k;
    const bool active = p0 < a.nb_bases;
    uint64_t rd = 0, rstart = 0, rend = 0;
    const uint64_t pend = active ? (p0 + WSEG < a.nb_bases ? p0 + WSEG : a.nb_bases) : 0;
    uint32_t cnt = 0;
    if (active) {
        // the read that holds base p0
        if (a.fixed_len) { rd = p0 / a.fixed_len; rstart = rd * a.fixed_len; rend = rstart + a.fixed_len; }
        else {
            uint64_t lo = 0, hi = a.nb_reads;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.offsets[mid] <= p0) lo = mid; else hi = mid; }
            rd = lo; rstart = a.offsets[rd]; rend = a.offsets[rd + 1];
        }
        if (SHARDED) cnt = k >= 64u ? wroll4<false, true>(a, p0, pend, rd, rstart, rend, khi, klo, 0ull) : wroll<false, true>(a, p0, pend, rd, rstart, rend, khi, klo, 0ull);      // ownership needs the k-mer itself
        else {
            // pass 1 (no bases touched): how many of my end positions close a whole k-mer inside one read
            uint64_t rr = rd, rs = rstart, re = rend;
            for (uint64_t p = p0; p < pend; p++) {
                while (p >= re) { rr++; rs = re; re = a.fixed_len ? rs + a.fixed_len : a.offsets[rr + 1]; }
                if (p - rs >= (uint64_t)(k - 1u)) cnt++;
            }
        }
    }
    // output slots: block scan of the counts + one global atomic per block (the order of the keys is irrelevant: they get sorted)
    uint32_t incl = cnt;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += t; }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wpre = 0, btot = 0;
    for (uint32_t i = 0; i < 4; i++) { if (i < wave) wpre += s_wave[i]; btot += s_wave[i]; }
    if (threadIdx.x == 0) s_base = btot ? atomicAdd(nvalid, (ull)btot) : 0ull;
    __syncthreads();
    if (!active || cnt == 0) return;
    // pass 2: write the k-mers
    if (k >= 64u) (void)wroll4<true, SHARDED>(a, p0, pend, rd, rstart, rend, khi, klo, s_base + wpre + incl - cnt);
    else (void)wroll<true, SHARDED>(a, p0, pend, rd, rstart, rend, khi, klo, s_base + wpre + incl - cnt);
}

__global__ void __launch_bounds__(256)
k_wgather(const ull *src, const uint32_t *idx, ull *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void __launch_bounds__(256)
k_wiota(uint32_t *idx, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}

// head of a run of equal keys
__global__ void __launch_bounds__(256)
k_wheads(const ull *khi, const ull *klo, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || khi[i] != khi[i - 1] || klo[i] != klo[i - 1]) ? 1u : 0u;
}

// start[rank of the run] = position of its head; start[nruns] = n
__global__ void __launch_bounds__(256)
k_wstarts(const uint32_t *flag, const uint32_t *rank, uint64_t n, uint32_t *start, const uint32_t *nruns_minus) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) start[rank[i]] = (uint32_t)i;
    if (i == n - 1) start[rank[i] + flag[i]] = (uint32_t)n;
    (void)nruns_minus;
}

// SimkaCompressedProcessor::process over the runs: abundance filter, totals, (complex) histogram of solid counts
__global__ void __launch_bounds__(256)
k_wfilter(const uint32_t *start, uint32_t nruns, uint32_t amin, uint32_t amax, uint32_t *sflag, ull *tot /* D N Q */,
          ull *hist, uint32_t *ovf_list, ull *ovf_cursor, ull ovf_cap, uint32_t sample) {
    __shared__ uint32_t lhist[SIMKA_HIST_MAX];      // -complex-dist: the block's histogram of solid counts, flushed once
    if (hist) { for (uint32_t i = threadIdx.x; i < SIMKA_HIST_MAX; i += blockDim.x) lhist[i] = 0; __syncthreads(); }
    ull D = 0, N = 0, Q = 0;
    // grid-stride: a few thousand waves add their partial totals, not one wave per 64 runs (same-address atomics serialise)
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nruns; j += gridDim.x * blockDim.x) {
        const uint32_t c = start[j + 1] - start[j];
        const bool solid = !(c < amin || c > amax);
        sflag[j] = solid ? 1u : 0u;
        if (solid) {
            D += 1; N += c; Q += (ull)c * (ull)c;
            if (hist) {
                if (c < SIMKA_HIST_MAX) atomicAdd(&lhist[c], 1u);
                else { const ull wq = atomicAdd(ovf_cursor, 1ull); if (wq < ovf_cap) { ovf_list[2 * wq] = sample; ovf_list[2 * wq + 1] = c; } }
            }
        }
    }
    if (hist) { __syncthreads(); for (uint32_t i = threadIdx.x; i < SIMKA_HIST_MAX; i += blockDim.x) if (lhist[i]) atomicAdd(&hist[i], (ull)lhist[i]); }
    for (int o = 32; o > 0; o >>= 1) { D += __shfl_down(D, o, 64); N += __shfl_down(N, o, 64); Q += __shfl_down(Q, o, 64); }
    if ((threadIdx.x & 63u) == 0 && D) { atomicAdd(&tot[0], D); atomicAdd(&tot[1], N); atomicAdd(&tot[2], Q); }
}

__global__ void __launch_bounds__(256)
k_wemit(const ull *khi, const ull *klo, const uint32_t *start, const uint32_t *sflag, const uint32_t *srank, uint32_t nruns,
        ull *ohi, ull *olo, uint32_t *ocnt) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nruns || !sflag[j]) return;
    const uint32_t s = start[j], o = srank[j];
    ohi[o] = khi[s]; olo[o] = klo[s]; ocnt[o] = start[j + 1] - s;
}

// ---- merge: grouping without a full sort ------------------------------------------------------------------------------------
// What the CSR needs is every k-mer's records side by side, not an order.  So the records are scattered into buckets of 450..900 by a
// hash of the k-mer (two or three 8-bit passes of the radix sort on the bucket number, not nine on the k-mer), and one block per bucket
// finishes in LDS: a table of the bucket's k-mers counts the records of each, a scan of the counts gives every group its place.
#define WL_BLOCK 256
#define WL_RPT 4              // records a thread keeps in registers between the two passes (buckets of up to 1024 records: all of them)
#define WL_TS 2048
#define WL_TSL 11
__device__ __forceinline__ uint32_t wide_hash32(ull hi, ull lo) {
    ull x = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
    return (uint32_t)(x >> 32);
}
// one sample's records: bucket number + index for the sort, and (hi, lo, sample << 32 | count) packed into 32 bytes -- k_wlocal_group
// fetches a record with one sector instead of three
__global__ void __launch_bounds__(256)
k_wbucket_key(const ull *hi, const ull *lo, const uint32_t *cnt, uint64_t off, uint64_t n, uint32_t sample, uint32_t bits, ull *key, uint32_t *idx, ulonglong4 *pack) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ull h_ = hi[off + i], l_ = lo[off + i];
    key[off + i] = (ull)(wide_hash32(h_, l_) >> (32u - bits)); idx[off + i] = (uint32_t)(off + i);
    pack[off + i] = make_ulonglong4(h_, l_, ((ull)sample << 32) | cnt[off + i], 0ull);
}
// bstart[b] = first position of the sorted bucket numbers that is >= b (b = 0 .. nb); the largest bucket
__global__ void __launch_bounds__(256)
k_wbucket_bounds(const ull *key, uint64_t n, uint32_t nb, uint32_t *bstart, uint32_t *maxsize) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    auto lower = [&](ull v) { uint64_t a = 0, e = n; while (a < e) { const uint64_t m = (a + e) >> 1; if (key[m] < v) a = m + 1; else e = m; } return (uint32_t)a; };
    const uint32_t lo_ = lower((ull)b);
    bstart[b] = lo_;
    if (b < nb) atomicMax(maxsize, lower((ull)b + 1ull) - lo_);
}
// the table of a bucket: claim a slot with a compare-and-swap on the high word (never all ones: <= 62 bits), store the low word, then
// count -- a lane that finds its high word in a slot compares the low word once the counter says it is there (the LDS serves a wave's
// operations in order: counter > 0 implies the low word is written).  Returns the slot (its counter one up), or ~0 when the table is
// (nearly) full.
template <bool LOOKUP>
__device__ __forceinline__ uint32_t wl_slot(ull *thi, ull *tlo, uint32_t *tcnt, ull h_, ull l_, uint32_t *ndist, bool active = true) {
    uint32_t slot = (wide_hash32(l_, h_) * 0x9E3779B1u) >> (32u - WL_TSL);          // (other bits than the bucket number's)
    if (LOOKUP) {      // every key is in the table and ready
        if (!active) return ~0u;
        for (uint32_t step = 0; step < WL_TS; step++) {
            if (thi[slot] == h_ && tlo[slot] == l_) return slot;
            slot = (slot + 1u) & (WL_TS - 1u);
        }
        return ~0u;
    }
    // Every lane of the wave runs the SAME straight-line step until none is pending: a lane that waits for a claimed slot to become
    // ready may be waiting for a lane of its own wave, whose store and count therefore must not sit behind the loop's exit.
    bool pending = active;
    uint32_t res = ~0u;
    for (uint32_t step = 0; step < 8u * WL_TS; step++) {
        if (!__any(pending)) break;
        ull prev = 0;
        if (pending) prev = atomicCAS(&thi[slot], ~0ull, h_);
        const bool won = pending && prev == ~0ull;
        if (won) { tlo[slot] = l_; atomicAdd(&tcnt[slot], 1u); }
        bool done = won, wait = false, full = false;
        if (won) full = atomicAdd(ndist, 1u) >= (uint32_t)(WL_TS * 3 / 4);
        if (pending && !won && prev == h_) {
            if (((volatile uint32_t *)tcnt)[slot] == 0u) wait = true;          // claimed, its low word not counted in yet: look again
            else if (((volatile ull *)tlo)[slot] == l_) { atomicAdd(&tcnt[slot], 1u); done = true; }
        }
        if (done) { pending = false; res = full ? ~0u : slot; }
        else if (pending && !wait) slot = (slot + 1u) & (WL_TS - 1u);
    }
    return res;
}

// one block per bucket, any number of records: pass A counts the records of every k-mer (the first WL_RPT rounds keep their record
// and slot in registers), a scan of the counts gives every group its place, pass B writes the records there (later rounds fetch
// their record again and look the slot up).  A bucket with more distinct k-mers than 3/4 of the table raises *fail.
__global__ void __launch_bounds__(WL_BLOCK)
k_wlocal_group(const uint32_t *bstart, const uint32_t *idx, const ulonglong4 *pack, ull *o_hi, ull *o_lo, ull *o_val, uint32_t *fail) {
    __shared__ ull thi[WL_TS], tlo[WL_TS];
    __shared__ uint32_t tcnt[WL_TS];
    __shared__ uint32_t wsum[WL_BLOCK / 64];
    __shared__ uint32_t s_ndist, s_fail;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = bstart[blockIdx.x], n = bstart[blockIdx.x + 1] - base;
    if (n == 0) return;
    constexpr uint32_t SPT = WL_TS / WL_BLOCK;
    for (uint32_t i = tid; i < WL_TS; i += WL_BLOCK) { thi[i] = ~0ull; tcnt[i] = 0; }
    if (tid == 0) { s_ndist = 0; s_fail = 0; }
    __syncthreads();
    ull my_hi[WL_RPT], my_lo[WL_RPT], my_val[WL_RPT]; uint32_t my_slot[WL_RPT];
#pragma unroll
    for (uint32_t q = 0; q < WL_RPT; q++) {
        const uint32_t r = q * WL_BLOCK + tid;
        my_slot[q] = ~0u; my_hi[q] = 0; my_lo[q] = 0; my_val[q] = 0;
        if (q * WL_BLOCK < n) {          // (block-uniform: whole waves enter wl_slot)
            if (r < n) { const ulonglong4 e = pack[idx[base + r]]; my_hi[q] = e.x; my_lo[q] = e.y; my_val[q] = e.z; }
            const uint32_t sl = wl_slot<false>(thi, tlo, tcnt, my_hi[q], my_lo[q], &s_ndist, r < n);
            my_slot[q] = sl;
            if (r < n && sl == ~0u) s_fail = 1u;
        }
    }
    for (uint32_t r0 = WL_RPT * WL_BLOCK; r0 < n; r0 += WL_BLOCK) {
        const uint32_t r = r0 + tid;
        ulonglong4 e = make_ulonglong4(0, 0, 0, 0);
        if (r < n) e = pack[idx[base + r]];
        if (wl_slot<false>(thi, tlo, tcnt, e.x, e.y, &s_ndist, r < n) == ~0u && r < n) s_fail = 1u;
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) *fail = 1u; return; }
    // exclusive scan of the group sizes over the slots: tcnt[] becomes every group's cursor
    {
        uint32_t c[SPT], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < SPT; j++) { c[j] = tcnt[tid * SPT + j]; sum += c[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((tid & 63u) >= (uint32_t)o) inc += t; }
        if ((tid & 63u) == 63u) wsum[tid >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (uint32_t w_ = 0; w_ < (tid >> 6); w_++) run += wsum[w_];
#pragma unroll
        for (uint32_t j = 0; j < SPT; j++) { tcnt[tid * SPT + j] = run; run += c[j]; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < WL_RPT; q++) {
        if (my_slot[q] != ~0u) {
            const uint32_t pos = base + atomicAdd(&tcnt[my_slot[q]], 1u);
            o_hi[pos] = my_hi[q]; o_lo[pos] = my_lo[q]; o_val[pos] = my_val[q];
        }
    }
    for (uint32_t r = WL_RPT * WL_BLOCK + tid; r < n; r += WL_BLOCK) {
        const ulonglong4 e = pack[idx[base + r]];
        const uint32_t sl = wl_slot<true>(thi, tlo, tcnt, e.x, e.y, nullptr);
        const uint32_t pos = base + atomicAdd(&tcnt[sl], 1u);
        o_hi[pos] = e.x; o_lo[pos] = e.y; o_val[pos] = e.z;
    }
}

// ---- merge ----
__global__ void __launch_bounds__(256)
k_wvals(const uint32_t *cnt, uint64_t off, uint64_t n, uint32_t sample, ull *val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) val[off + i] = ((ull)sample << 32) | cnt[off + i];
}

// groups: size, kept (>= 2 samples) flag, size of kept groups
// huge != 0: flag / size of the groups LARGER than maxg (they get a span of their own on the huge list); else of the groups of 2..maxg samples
__global__ void __launch_bounds__(256)
k_wgsizes(const uint32_t *gstart, uint32_t ngroups, uint32_t maxg, uint32_t huge, uint32_t *kflag, uint32_t *ksize) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint32_t s = gstart[g + 1] - gstart[g];
    const bool take = huge ? s > maxg : (s >= 2u && s <= maxg);
    kflag[g] = take ? 1u : 0u;
    ksize[g] = take ? s : 0u;
}

// a group shared by more samples than a span can hold: its entries behind the ordinary ones, one span on the huge list (k_pairs_global)
__global__ void __launch_bounds__(256)
k_whuge(const uint32_t *gstart, uint32_t ngroups, const uint32_t *hflag, const uint32_t *hrank, const uint32_t *hoff, const ull *val, ull ebase0,
        ull *entries, SimkaSpan *huge) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups || !hflag[g]) return;
    const uint32_t b = gstart[g], s = gstart[g + 1] - b;
    const ull e = ebase0 + hoff[g];
    for (uint32_t t = 0; t < s; t++) entries[e + t] = val[b + t];
    SimkaSpan sp; sp.ebase = e; sp.gbase = 0; sp.nent = s; sp.ngrp = 1; sp.maxc = 0; sp.pad = 0;
    huge[hrank[g]] = sp;
}

// kept group r: entries copied, span bookkeeping (span = the groups whose first entry falls into [sp*C, (sp+1)*C)).  The groups of
// a wave are neighbours, so nearly always they all belong to ONE span: its four counters then take one atomic each from the wave
// instead of one from every lane (thousands of groups per span: the per-lane atomics on four hot words were most of the merge).
__global__ void __launch_bounds__(256)
k_wgroups(const uint32_t *gstart, uint32_t ngroups, const uint32_t *kflag, const uint32_t *krank, const uint32_t *eoff, const ull *val,
          uint32_t C, ull *entries, uint32_t *ge, uint32_t *gs, uint32_t *sp_first, uint32_t *sp_ngrp, uint32_t *sp_nent, uint32_t *sp_maxc) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = g < ngroups && kflag[g];
    uint32_t s = 0, r = 0xffffffffu, sp = 0, maxc = 0;
    if (act) {
        const uint32_t b = gstart[g], e = eoff[g];
        s = gstart[g + 1] - b; r = krank[g]; sp = e / C;
        for (uint32_t t = 0; t < s; t++) { const ull v = val[b + t]; entries[e + t] = v; const uint32_t c = (uint32_t)v; maxc = c > maxc ? c : maxc; }
        ge[r] = e; gs[r] = s;
    }
    const ull am = __ballot(act);
    if (am == 0ull) return;
    const uint32_t sp0 = (uint32_t)__builtin_amdgcn_readlane((int)sp, __ffsll((long long)am) - 1);
    if (__all(!act || sp == sp0)) {
        uint32_t rmin = r, cnt = act ? 1u : 0u, ssum = s, cmax = maxc;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t a_ = __shfl_xor(rmin, o, 64), b_ = __shfl_xor(cnt, o, 64), c_ = __shfl_xor(ssum, o, 64), d_ = __shfl_xor(cmax, o, 64);
            rmin = a_ < rmin ? a_ : rmin; cnt += b_; ssum += c_; cmax = d_ > cmax ? d_ : cmax;
        }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&sp_first[sp0], rmin); atomicAdd(&sp_ngrp[sp0], cnt); atomicAdd(&sp_nent[sp0], ssum); atomicMax(&sp_maxc[sp0], cmax); }
    } else if (act) {
        atomicMin(&sp_first[sp], r);
        atomicAdd(&sp_ngrp[sp], 1u);
        atomicAdd(&sp_nent[sp], s);
        atomicMax(&sp_maxc[sp], maxc);
    }
}

__global__ void __launch_bounds__(256)
k_wspans(uint32_t nspans, const uint32_t *sp_first, const uint32_t *sp_ngrp, const uint32_t *sp_nent, const uint32_t *sp_maxc, const uint32_t *ge,
         SimkaSpan *spans, ull *cursors) {
    const uint32_t sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp == 0) { cursors[0] = 0; cursors[1] = 0; cursors[2] = nspans; }      // [3] (huge spans) is set by the host
    if (sp >= nspans) return;
    SimkaSpan s; s.ebase = 0; s.gbase = 0; s.nent = 0; s.ngrp = 0; s.maxc = 0; s.pad = 0;
    if (sp_ngrp[sp]) { const uint32_t f = sp_first[sp]; s.ebase = ge[f]; s.gbase = f; s.nent = sp_nent[sp]; s.ngrp = sp_ngrp[sp]; s.maxc = sp_maxc[sp]; }
    spans[sp] = s;
}

__global__ void __launch_bounds__(256)
k_wgdesc(uint32_t nkept, const uint32_t *ge, const uint32_t *gs, const uint32_t *sp_first, uint32_t C, uint32_t *groups) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nkept) return;
    const uint32_t sp = ge[r] / C;
    groups[r] = ((ge[r] - ge[sp_first[sp]]) << 16) | gs[r];       // start inside the span | size
}

// --------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255) / 256)); }


// exclusive prefix sum of n 32-bit values (in -> out, may alias) with scratch slot 11
static int wide_scan(SimkaWide *w, const uint32_t *in, uint32_t *out, uint64_t n) {
    if (n == 0) return 0;
    uint32_t *tmp; int rc = wide_buf(w, 11, wscan_tmp_u32(n), &tmp); if (rc) return rc;
    WCHK(wscan_u32(in, out, n, tmp, w->stream));
    return 0;
}

// stable sort of n (hi, lo) keys (+ optional 32-bit payload) by (hi, lo): LSD, lo first.  In: hi0/lo0(/p0); out: hi1/lo1(/p1).
static int wide_sort(SimkaWide *w, uint64_t n, uint32_t hi_bits, ull *hi0, ull *lo0, ull *hi1, ull *lo1, ull *tmp_key, uint32_t *idx0, uint32_t *idx1) {
    if (n == 0) return 0;
    if (n >= ((uint64_t)1 << 31)) { w->err = "wide-k path: more than 2^31 k-mers in one sort (sample too deep for k >= 32)"; return SIMKA_WIDE_ERR_LIMIT; }
    char *tmp; int rc = wide_buf(w, 11, wsort_tmp_bytes<uint32_t>(n), &tmp); if (rc) return rc;
    SIMKA_LAUNCH(k_wiota, grid_for(n), dim3(256), 0, w->stream, idx0, n);
    WCHK(wsort_pairs<uint32_t>(lo0, tmp_key, idx0, idx1, n, 64u, tmp, w->stream));                                   // by lo; idx1 = permutation
    SIMKA_LAUNCH(k_wgather, grid_for(n), dim3(256), 0, w->stream, hi0, idx1, lo1, n);                          // lo1 := hi in lo-order (scratch use)
    WCHK(wsort_pairs<uint32_t>(lo1, hi1, idx1, idx0, n, hi_bits, tmp, w->stream));                                   // by hi (stable); idx0 = final permutation
    SIMKA_LAUNCH(k_wgather, grid_for(n), dim3(256), 0, w->stream, lo0, idx0, lo1, n);
    WCHK(hipGetLastError());
    return 0;
}

// the same order without a permutation (count side): the other word rides along as the VALUE of each sort.
// In: hi0/lo0 (+ scratch hi1/lo1); out: sorted by (hi, lo) -- back in hi0/lo0, or in hi1/lo1 when there is no high word to sort by.
static int wide_sort_words(SimkaWide *w, uint64_t n, uint32_t hi_bits, ull *hi0, ull *lo0, ull *hi1, ull *lo1) {
    if (n == 0) return 0;
    if (n >= ((uint64_t)1 << 31)) { w->err = "wide-k path: more than 2^31 k-mers in one sort (sample too deep for k >= 32)"; return SIMKA_WIDE_ERR_LIMIT; }
    const int lo_bits = (int)std::min<uint32_t>(64u, w->W);
    char *tmp; int rc = wide_buf(w, 11, wsort_tmp_bytes<ull>(n), &tmp); if (rc) return rc;
    WCHK(wsort_pairs<ull>(lo0, lo1, hi0, hi1, n, (uint32_t)lo_bits, tmp, w->stream));         // by lo: (lo1, hi1)
    if (!hi_bits) return 0;                                                                       // k <= 32: done, result in hi1 / lo1
    WCHK(wsort_pairs<ull>(hi1, hi0, lo1, lo0, n, hi_bits, tmp, w->stream));                   // by hi, stable: (hi0, lo0)
    return 0;
}

int simka_wide_create(SimkaWide **out, int device, uint32_t nb_samples, uint32_t k, void *stream) {
    SimkaWide *w = new SimkaWide();
    // k >= 64: the keys are 126-bit fingerprints (wfinger)
    w->device = device; w->nb_samples = nb_samples; w->k = k; w->W = k >= 64 ? 126 : 2 * k;
    w->stream = (hipStream_t)stream;
    w->s_off.assign(nb_samples, 0); w->s_n.assign(nb_samples, 0); w->s_sorted.assign(nb_samples, 1);
    *out = w;
    return 0;
}

void simka_wide_set_shard(SimkaWide *w, uint32_t shard_index, uint32_t shard_count) { w->shard_index = shard_index; w->shard_count = shard_count ? shard_count : 1; }

const char *simka_wide_error(SimkaWide *w) { return w ? w->err.c_str() : ""; }

void simka_wide_destroy(SimkaWide *w) {
    if (!w) return;
    (void)hipStreamSynchronize(w->stream);
    for (void *p : w->scratch) if (p) (void)hipFree(p);
    void *own[] = { w->a_hi, w->a_lo, w->a_cnt, w->entries, w->groups, w->spans, w->huge, w->cursors };
    for (void *p : own) if (p) (void)hipFree(p);
    delete w;
}

int simka_wide_reset(SimkaWide *w) {
    w->a_used = 0;
    std::fill(w->s_off.begin(), w->s_off.end(), 0); std::fill(w->s_n.begin(), w->s_n.end(), 0); std::fill(w->s_sorted.begin(), w->s_sorted.end(), 1);
    return 0;
}

static int arena_reserve(SimkaWide *w, uint64_t extra) {
    if (w->a_used + extra <= w->a_cap) return 0;
    const uint64_t ncap = std::max<uint64_t>((w->a_used + extra) * 3 / 2, (uint64_t)1 << 20);
    ull *nh = nullptr, *nl = nullptr; uint32_t *nc = nullptr;
    if (hipMalloc(&nh, ncap * 8) != hipSuccess || hipMalloc(&nl, ncap * 8) != hipSuccess || hipMalloc(&nc, ncap * 4) != hipSuccess) {
        if (nh) (void)hipFree(nh); if (nl) (void)hipFree(nl); if (nc) (void)hipFree(nc);
        w->err = "wide-k path: cannot grow the solid-spectrum arena"; return SIMKA_WIDE_ERR_NOMEM;
    }
    if (w->a_used) {
        WCHK(hipMemcpyAsync(nh, w->a_hi, w->a_used * 8, hipMemcpyDeviceToDevice, w->stream));
        WCHK(hipMemcpyAsync(nl, w->a_lo, w->a_used * 8, hipMemcpyDeviceToDevice, w->stream));
        WCHK(hipMemcpyAsync(nc, w->a_cnt, w->a_used * 4, hipMemcpyDeviceToDevice, w->stream));
    }
    WCHK(hipStreamSynchronize(w->stream));
    if (w->a_hi) (void)hipFree(w->a_hi); if (w->a_lo) (void)hipFree(w->a_lo); if (w->a_cnt) (void)hipFree(w->a_cnt);
    w->a_hi = nh; w->a_lo = nl; w->a_cnt = nc; w->a_cap = ncap;
    return 0;
}

// ---- counting by buckets (k >= 52, and the last resort of the partitioned count): the occurrences of a sample are scattered into
// buckets by a hash of the k-mer like the records of the merge, and a block counts its bucket in the same LDS table: what leaves is
// the bucket's solid (hi, lo, count) records, unordered, behind a global cursor.
__global__ void __launch_bounds__(256)
k_wocc_key(const ull *hi, const ull *lo, uint64_t n, uint32_t bits, ull *key, uint32_t *idx, ulonglong2 *pack) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ull h_ = hi[i], l_ = lo[i];
    key[i] = (ull)(wide_hash32(h_, l_) >> (32u - bits)); idx[i] = (uint32_t)i;
    pack[i] = make_ulonglong2(h_, l_);
}
// part: [WL_NPART][8] partial totals -- [1] D [2] N [3] Q [4] D_all [6] flag "a table filled" -- spread over WL_NPART rows: tens of
// thousands of blocks adding to ONE set of words queue up behind each other in the L2 (13 ms of a 14-ms kernel).  The solid records
// of bucket b go to the slots [bstart[b], ...) of the output, the rest of the bucket's range gets count 0 (simka_wide_adopt drops
// those): no cursor to reserve from either.
#define WL_NPART 256
__global__ void __launch_bounds__(WL_BLOCK)
k_wlocal_count(const uint32_t *bstart, const uint32_t *idx, const ulonglong2 *pack, uint32_t amin, uint32_t amax, ull *o_hi, ull *o_lo, uint32_t *o_cnt, ull *part,
               ull *hist, uint32_t *ovf_list, ull *ovf_cursor, ull ovf_cap, uint32_t sample) {
    __shared__ ull thi[WL_TS], tlo[WL_TS];
    __shared__ uint32_t tcnt[WL_TS];
    __shared__ uint32_t wsum[WL_BLOCK / 64];
    __shared__ uint32_t s_ndist, s_fail;
    __shared__ uint32_t lhist[SIMKA_HIST_MAX];      // -complex-dist: the block's histogram of solid counts
    const uint32_t tid = threadIdx.x;
    const uint32_t base = bstart[blockIdx.x], n = bstart[blockIdx.x + 1] - base;
    if (n == 0) return;
    ull *mine = part + (size_t)(blockIdx.x % WL_NPART) * 8;
    constexpr uint32_t SPT = WL_TS / WL_BLOCK;
    for (uint32_t i = tid; i < WL_TS; i += WL_BLOCK) { thi[i] = ~0ull; tcnt[i] = 0; }
    if (hist) for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += WL_BLOCK) lhist[i] = 0;
    if (tid == 0) { s_ndist = 0; s_fail = 0; }
    __syncthreads();
    for (uint32_t r0 = 0; r0 < n; r0 += WL_BLOCK) {          // (block-uniform trip count: whole waves enter wl_slot)
        const uint32_t r = r0 + tid;
        ulonglong2 e = make_ulonglong2(0, 0);
        if (r < n) e = pack[idx[base + r]];
        if (wl_slot<false>(thi, tlo, tcnt, e.x, e.y, &s_ndist, r < n) == ~0u && r < n) s_fail = 1u;
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) atomicOr(&mine[6], 1ull); return; }
    uint32_t c[SPT], nsol = 0, ndall = 0;
    ull D = 0, N = 0, Q = 0;
#pragma unroll
    for (uint32_t j = 0; j < SPT; j++) {
        c[j] = tcnt[tid * SPT + j];
        if (c[j]) { ndall++; if (c[j] < amin || c[j] > amax) c[j] = 0; else { nsol++; D++; N += c[j]; Q += (ull)c[j] * (ull)c[j]; } }
    }
    uint32_t inc = nsol;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((tid & 63u) >= (uint32_t)o) inc += t; }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - nsol, tot = 0;
    for (uint32_t w_ = 0; w_ < WL_BLOCK / 64; w_++) { if (w_ < (tid >> 6)) run += wsum[w_]; tot += wsum[w_]; }
    ull pos = (ull)base + run;          // tot <= the bucket's distinct k-mers <= n
#pragma unroll
    for (uint32_t j = 0; j < SPT; j++)
        if (c[j]) {
            o_hi[pos] = thi[tid * SPT + j]; o_lo[pos] = tlo[tid * SPT + j]; o_cnt[pos] = c[j]; pos++;
            if (hist) {
                if (c[j] < SIMKA_HIST_MAX) atomicAdd(&lhist[c[j]], 1u);
                else { const ull wq = atomicAdd(ovf_cursor, 1ull); if (wq < ovf_cap) { ovf_list[2 * wq] = sample; ovf_list[2 * wq + 1] = c[j]; } }
            }
        }
    for (uint32_t i = tot + tid; i < n; i += WL_BLOCK) o_cnt[base + i] = 0;
    for (int o = 32; o > 0; o >>= 1) { D += __shfl_down(D, o, 64); N += __shfl_down(N, o, 64); Q += __shfl_down(Q, o, 64); ndall += __shfl_down(ndall, o, 64); }
    if ((tid & 63u) == 0) { if (D) { atomicAdd(&mine[1], D); atomicAdd(&mine[2], N); atomicAdd(&mine[3], Q); } if (ndall) atomicAdd(&mine[4], (ull)ndall); }
    if (hist) { __syncthreads(); for (uint32_t i = tid; i < SIMKA_HIST_MAX; i += WL_BLOCK) if (lhist[i]) atomicAdd(&hist[i], (ull)lhist[i]); }
}

int simka_wide_count_sample(SimkaWide *w, uint32_t sample, const void *packed, uint64_t nb_bases, uint64_t nb_words, const void *offsets, uint64_t nb_reads,
                            uint32_t fixed_len, uint32_t amin, uint32_t amax, unsigned long long totals5[5], void *d_hist_row, void *d_ovf_list,
                            void *d_ovf_cursor, uint64_t ovf_cap) {
    for (int i = 0; i < 5; i++) totals5[i] = 0;
    w->s_off[sample] = w->a_used; w->s_n[sample] = 0; w->s_sorted[sample] = 1;
    if (nb_bases == 0) return 0;
    const uint64_t n = nb_bases;
    ull *hi0, *lo0, *hi1, *lo1; uint32_t *idx0, *idx1; ull *d_small;
    int rc;
    if ((rc = wide_buf(w, 0, n, &hi0)) || (rc = wide_buf(w, 1, n, &lo0)) || (rc = wide_buf(w, 2, n, &hi1)) || (rc = wide_buf(w, 3, n, &lo1)) ||
        (rc = wide_buf(w, 5, n + 2, &idx0)) || (rc = wide_buf(w, 6, n + 2, &idx1))) return rc;
    {   // 16 small words + the partial totals + the bucket table of the bucket route (u32, at most 2^bits + 8 with n >> bits <= 1100)
        uint32_t bmax = 1;
        while ((n >> bmax) > 1100u && bmax < 24u) bmax++;
        if ((rc = wide_buf(w, 7, 16 + 2048 + (((uint64_t)1 << bmax) + 8) / 2 + 1, &d_small))) return rc;          // (2048 = WL_NPART * 8 partial totals)
    }
    WCHK(hipMemsetAsync(d_small, 0, 16 * 8, w->stream));
    WideScanArgs a; a.packed = (const uint64_t *)packed; a.nb_bases = nb_bases; a.nb_words = nb_words; a.offsets = (const uint64_t *)offsets;
    a.nb_reads = nb_reads; a.fixed_len = fixed_len; a.k = w->k; a.shard_index = w->shard_index; a.shard_count = w->shard_count;
    if (w->shard_count > 1) SIMKA_LAUNCH(k_wscan<true>, grid_for((n + WSEG - 1) / WSEG), dim3(256), 0, w->stream, a, hi0, lo0, d_small /* [0] = nvalid */);
    else SIMKA_LAUNCH(k_wscan<false>, grid_for((n + WSEG - 1) / WSEG), dim3(256), 0, w->stream, a, hi0, lo0, d_small /* [0] = nvalid */);
    ull nvalid = 0;
    WCHK(hipMemcpyAsync(&nvalid, d_small, 8, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    // ---- by buckets: hash -> bucket number, two or three radix passes on it, one LDS table per bucket (k_wlocal_count)
    if (nvalid && nvalid < ((uint64_t)1 << 31) && !simka_test_knob("SIMKA_WIDE_COUNT_SORT")) {
        uint32_t bits = 1;
        while ((nvalid >> bits) > 1100u && bits < 24u) bits++;
        const uint32_t nb = 1u << bits;
        const uint64_t ocap = nvalid + 16;          // (bucket b's solid records sit at the start of its range of the occurrences)
        ulonglong2 *pack; char *tmp; ull *ohi, *olo; uint32_t *ocnt;
        if ((rc = wide_buf(w, 4, nvalid, &pack)) || (rc = wide_buf(w, 11, wsort_tmp_bytes<uint32_t>(nvalid), &tmp)) || (rc = wide_buf(w, 8, ocap, &ohi)) ||
            (rc = wide_buf(w, 9, ocap, &olo)) || (rc = wide_buf(w, 10, ocap, &ocnt))) return rc;
        ull *part = d_small + 16;
        uint32_t *bstart = (uint32_t *)(part + WL_NPART * 8);
        WCHK(hipMemsetAsync(part, 0, (size_t)WL_NPART * 64, w->stream));
        SIMKA_LAUNCH(k_wocc_key, grid_for(nvalid), dim3(256), 0, w->stream, hi0, lo0, nvalid, bits, hi1, idx0, pack);
        WCHK(wsort_pairs<uint32_t>(hi1, lo1, idx0, idx1, nvalid, bits, tmp, w->stream));          // lo1: the sorted bucket numbers
        SIMKA_LAUNCH(k_wbucket_bounds, grid_for((uint64_t)nb + 1), dim3(256), 0, w->stream, lo1, nvalid, nb, bstart, (uint32_t *)(d_small + 15));
        ull before[2] = { 0, 0 };
        if (d_ovf_cursor) WCHK(hipMemcpyAsync(before, d_ovf_cursor, 16, hipMemcpyDeviceToHost, w->stream));
        SIMKA_LAUNCH(k_wlocal_count, dim3(nb), dim3(WL_BLOCK), 0, w->stream, bstart, idx1, pack, amin, amax, ohi, olo, ocnt, part,
                           (ull *)d_hist_row, (uint32_t *)d_ovf_list, (ull *)d_ovf_cursor, (ull)ovf_cap, sample);
        std::vector<ull> ph((size_t)WL_NPART * 8);
        WCHK(hipMemcpyAsync(ph.data(), part, ph.size() * 8, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
        ull sm[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (size_t i = 0; i < ph.size(); i++) sm[i & 7] += ph[i];
        if (sm[6] == 0 && !simka_test_knob("SIMKA_WIDE_COUNT_FAIL")) {
            totals5[SIMKA_TOT_KOCC] = nvalid; totals5[SIMKA_TOT_DALL] = sm[4];
            totals5[SIMKA_TOT_D] = sm[1]; totals5[SIMKA_TOT_N] = sm[2]; totals5[SIMKA_TOT_Q] = sm[3];
            return simka_wide_adopt(w, sample, ohi, olo, ocnt, nvalid, sm[1]);
        }
        // a bucket's table filled: by sorting, below -- with the totals and the abundance histogram as they were
        WCHK(hipMemsetAsync(d_small + 1, 0, 15 * 8, w->stream));
        if (d_hist_row) WCHK(hipMemsetAsync(d_hist_row, 0, (size_t)SIMKA_HIST_MAX * 8, w->stream));
        if (d_ovf_cursor) WCHK(hipMemcpyAsync(d_ovf_cursor, before, 16, hipMemcpyHostToDevice, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
    }
    const uint32_t hi_bits = w->W > 64 ? w->W - 64 : 0;
    if (nvalid) w->nb_full_sorts++;
    if ((rc = wide_sort_words(w, nvalid, hi_bits, hi0, lo0, hi1, lo1))) return rc;
    if (hi_bits) { hi1 = hi0; lo1 = lo0; }          // the sorted words (one sort only: they are in hi1 / lo1)
    totals5[SIMKA_TOT_KOCC] = nvalid;
    if (nvalid == 0) return 0;
    // runs of equal keys
    uint32_t *flag = idx0, *rank = idx1, *start, *sflag, *srank;
    if ((rc = wide_buf(w, 8, nvalid + 2, &start)) || (rc = wide_buf(w, 9, nvalid + 2, &sflag)) || (rc = wide_buf(w, 10, nvalid + 2, &srank))) return rc;
    SIMKA_LAUNCH(k_wheads, grid_for(nvalid), dim3(256), 0, w->stream, hi1, lo1, nvalid, flag);
    if ((rc = wide_scan(w, flag, rank, nvalid))) return rc;
    SIMKA_LAUNCH(k_wstarts, grid_for(nvalid), dim3(256), 0, w->stream, flag, rank, nvalid, start, (const uint32_t *)nullptr);
    uint32_t last_rank = 0, last_flag = 0;
    WCHK(hipMemcpyAsync(&last_rank, rank + (nvalid - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipMemcpyAsync(&last_flag, flag + (nvalid - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    const uint32_t nruns = last_rank + last_flag;
    totals5[SIMKA_TOT_DALL] = nruns;
    SIMKA_LAUNCH(k_wfilter, dim3((uint32_t)std::min<uint64_t>((nruns + 255) / 256, 2048)), dim3(256), 0, w->stream, start, nruns, amin, amax, sflag, d_small + 1, (ull *)d_hist_row, (uint32_t *)d_ovf_list,
                       (ull *)d_ovf_cursor, (ull)ovf_cap, sample);
    if ((rc = wide_scan(w, sflag, srank, nruns))) return rc;
    ull dnq[3]; uint32_t lr = 0, lf = 0;
    WCHK(hipMemcpyAsync(dnq, d_small + 1, 24, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipMemcpyAsync(&lr, srank + (nruns - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipMemcpyAsync(&lf, sflag + (nruns - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    const uint64_t nsolid = (uint64_t)lr + lf;
    totals5[SIMKA_TOT_D] = dnq[0]; totals5[SIMKA_TOT_N] = dnq[1]; totals5[SIMKA_TOT_Q] = dnq[2];
    if ((rc = arena_reserve(w, nsolid))) return rc;
    if (nsolid) SIMKA_LAUNCH(k_wemit, grid_for(nruns), dim3(256), 0, w->stream, hi1, lo1, start, sflag, srank, nruns, w->a_hi + w->a_used, w->a_lo + w->a_used,
                                   w->a_cnt + w->a_used);
    WCHK(hipGetLastError());
    w->s_n[sample] = nsolid;
    w->a_used += nsolid;
    return 0;
}

// a sample's solid records counted by k_skm_count_wide (simka_skm.hip): unordered (hi, lo, count) triples on the device.  They
// enter the arena as they are; wide_ensure_sorted() orders them when a caller needs the sample's own order.
__global__ void __launch_bounds__(256)
k_wgather32(const uint32_t *src, const uint32_t *idx, uint32_t *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void __launch_bounds__(256)
k_wused(const uint32_t *cnt, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cnt[i] ? 1u : 0u;
}
__global__ void __launch_bounds__(256)
k_wcompact(const ull *hi, const ull *lo, const uint32_t *cnt, const uint32_t *rank, uint64_t n, ull *o_hi, ull *o_lo, uint32_t *o_cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cnt[i];
    if (c) { const uint32_t r = rank[i]; o_hi[r] = hi[i]; o_lo[r] = lo[i]; o_cnt[r] = c; }
}

// nb_slots slots hold n records: the slots with count 0 are the unused ends of the count kernel's slabs
int simka_wide_adopt(SimkaWide *w, uint32_t sample, const void *d_hi, const void *d_lo, const void *d_cnt, uint64_t nb_slots, uint64_t n) {
    int rc = arena_reserve(w, n); if (rc) return rc;
    w->s_off[sample] = w->a_used; w->s_n[sample] = n; w->s_sorted[sample] = n <= 1;
    if (n == 0) return 0;
    if (nb_slots >= ((uint64_t)1 << 32)) { w->err = "wide-k path: more than 2^32 output slots of one sample"; return SIMKA_WIDE_ERR_LIMIT; }
    if (nb_slots == n) {
        WCHK(hipMemcpyAsync(w->a_hi + w->a_used, d_hi, n * 8, hipMemcpyDeviceToDevice, w->stream));
        WCHK(hipMemcpyAsync(w->a_lo + w->a_used, d_lo, n * 8, hipMemcpyDeviceToDevice, w->stream));
        WCHK(hipMemcpyAsync(w->a_cnt + w->a_used, d_cnt, n * 4, hipMemcpyDeviceToDevice, w->stream));
    } else {
        uint32_t *flag, *rank;
        if ((rc = wide_buf(w, 5, nb_slots + 2, &flag)) || (rc = wide_buf(w, 6, nb_slots + 2, &rank))) return rc;
        SIMKA_LAUNCH(k_wused, grid_for(nb_slots), dim3(256), 0, w->stream, (const uint32_t *)d_cnt, nb_slots, flag);
        if ((rc = wide_scan(w, flag, rank, nb_slots))) return rc;
        SIMKA_LAUNCH(k_wcompact, grid_for(nb_slots), dim3(256), 0, w->stream, (const ull *)d_hi, (const ull *)d_lo, (const uint32_t *)d_cnt, (const uint32_t *)rank, nb_slots,
                           w->a_hi + w->a_used, w->a_lo + w->a_used, w->a_cnt + w->a_used);
        WCHK(hipGetLastError());
    }
    w->a_used += n;
    return 0;
}

static int wide_ensure_sorted(SimkaWide *w, uint32_t sample) {
    if (w->s_sorted[sample]) return 0;
    const uint64_t n = w->s_n[sample], off = w->s_off[sample];
    int rc;
    ull *h0, *l0, *tkey; uint32_t *c0, *idx0, *idx1;
    if ((rc = wide_buf(w, 0, n, &h0)) || (rc = wide_buf(w, 1, n, &l0)) || (rc = wide_buf(w, 2, n, &c0)) || (rc = wide_buf(w, 3, n, &tkey)) ||
        (rc = wide_buf(w, 5, n + 2, &idx0)) || (rc = wide_buf(w, 6, n + 2, &idx1))) return rc;
    WCHK(hipMemcpyAsync(h0, w->a_hi + off, n * 8, hipMemcpyDeviceToDevice, w->stream));
    WCHK(hipMemcpyAsync(l0, w->a_lo + off, n * 8, hipMemcpyDeviceToDevice, w->stream));
    WCHK(hipMemcpyAsync(c0, w->a_cnt + off, n * 4, hipMemcpyDeviceToDevice, w->stream));
    const uint32_t hi_bits = std::max<uint32_t>(1u, w->W > 64 ? w->W - 64 : 0);
    if ((rc = wide_sort(w, n, hi_bits, h0, l0, w->a_hi + off, w->a_lo + off, tkey, idx0, idx1))) return rc;
    SIMKA_LAUNCH(k_wgather32, grid_for(n), dim3(256), 0, w->stream, (const uint32_t *)c0, (const uint32_t *)idx0, w->a_cnt + off, n);
    WCHK(hipGetLastError());
    w->s_sorted[sample] = 1;
    return 0;
}

int simka_wide_merge(SimkaWide *w, uint32_t span_cap, SimkaWideCsr *out) {
    const uint64_t M = w->a_used;
    const uint32_t N = w->nb_samples;
    out->nb_distinct = 0; out->nb_shared = 0; out->entries = nullptr; out->groups = nullptr; out->spans = nullptr; out->cursors = nullptr; out->nb_spans = 0; out->nb_entries = 0; out->huge = nullptr; out->nb_huge = 0;
    if (M == 0) return 0;
    const uint32_t maxg = std::min<uint32_t>(N, span_cap / 2 - 8);     // larger groups cannot share a span: huge list
    int rc;
    ull *val, *hi1, *lo1, *tkey, *val2; uint32_t *idx0, *idx1;
    if ((rc = wide_buf(w, 0, M, &val)) || (rc = wide_buf(w, 1, M, &hi1)) || (rc = wide_buf(w, 2, M, &lo1)) || (rc = wide_buf(w, 3, M, &tkey)) ||
        (rc = wide_buf(w, 4, M, &val2)) || (rc = wide_buf(w, 5, M + 2, &idx0)) || (rc = wide_buf(w, 6, M + 2, &idx1))) return rc;
    // the records of every k-mer side by side in hi1 / lo1 / val2: buckets by hash + grouping in LDS; a bucket beyond the LDS staging
    // (a k-mer in thousands of samples) or SIMKA_WIDE_MERGE_SORT: the full sort by k-mer
    bool grouped = false;
    if (!simka_test_knob("SIMKA_WIDE_MERGE_SORT") && M < ((uint64_t)1 << 31)) {
        uint32_t bits = 1;
        while ((M >> bits) > 900u && bits < 24u) bits++;          // (<= 900 records per bucket: at most 900 of the table's 1536 usable slots)
        const uint32_t nb = 1u << bits;
        uint32_t *bstart, *d_max;
        char *tmp;
        ulonglong4 *pack;
        if ((rc = wide_buf(w, 7, (uint64_t)nb + 8, &bstart)) || (rc = wide_buf(w, 11, wsort_tmp_bytes<uint32_t>(M), &tmp)) || (rc = wide_buf(w, 9, M, &pack))) return rc;
        d_max = bstart + nb + 2;
        WCHK(hipMemsetAsync(d_max, 0, 4, w->stream));
        for (uint32_t s = 0; s < N; s++)
            if (w->s_n[s]) SIMKA_LAUNCH(k_wbucket_key, grid_for(w->s_n[s]), dim3(256), 0, w->stream, w->a_hi, w->a_lo, w->a_cnt, w->s_off[s], w->s_n[s], s, bits, tkey, idx0, pack);
        WCHK(wsort_pairs<uint32_t>(tkey, hi1, idx0, idx1, M, bits, tmp, w->stream));          // (hi1: the sorted bucket numbers, until k_wlocal_group overwrites it)
        SIMKA_LAUNCH(k_wbucket_bounds, grid_for((uint64_t)nb + 1), dim3(256), 0, w->stream, hi1, M, nb, bstart, d_max);
        WCHK(hipMemsetAsync(d_max, 0, 4, w->stream));           // (from here on: the "a bucket's table filled" flag)
        SIMKA_LAUNCH(k_wlocal_group, dim3(nb), dim3(WL_BLOCK), 0, w->stream, bstart, idx1, pack, hi1, lo1, val2, d_max);
        uint32_t failed = 0;
        WCHK(hipMemcpyAsync(&failed, d_max, 4, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
        grouped = !failed && !simka_test_knob("SIMKA_WIDE_MERGE_FAIL");          // (tests: the fallback after a failed grouping)
    }
    if (!grouped) {
        w->nb_full_sorts++;
        for (uint32_t s = 0; s < N; s++)
            if (w->s_n[s]) SIMKA_LAUNCH(k_wvals, grid_for(w->s_n[s]), dim3(256), 0, w->stream, w->a_cnt, w->s_off[s], w->s_n[s], s, val);
        const uint32_t hi_bits = (w->W > 64 ? w->W - 64 : 0) + 1;
        if ((rc = wide_sort(w, M, hi_bits, w->a_hi, w->a_lo, hi1, lo1, tkey, idx0, idx1))) return rc;     // idx0 = final permutation
        SIMKA_LAUNCH(k_wgather, grid_for(M), dim3(256), 0, w->stream, val, idx0, val2, M);
    }
    // groups = runs of equal k-mers
    uint32_t *flag = idx0, *rank = idx1, *gstart, *kflag, *ksize, *krank, *eoff;
    if ((rc = wide_buf(w, 7, M + 2, &gstart)) || (rc = wide_buf(w, 8, M + 2, &kflag)) || (rc = wide_buf(w, 9, M + 2, &ksize))) return rc;
    SIMKA_LAUNCH(k_wheads, grid_for(M), dim3(256), 0, w->stream, hi1, lo1, M, flag);
    if ((rc = wide_scan(w, flag, rank, M))) return rc;
    SIMKA_LAUNCH(k_wstarts, grid_for(M), dim3(256), 0, w->stream, flag, rank, M, gstart, (const uint32_t *)nullptr);
    uint32_t lr = 0, lf = 0;
    WCHK(hipMemcpyAsync(&lr, rank + (M - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipMemcpyAsync(&lf, flag + (M - 1), 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    const uint32_t ngroups = lr + lf;
    out->nb_distinct = ngroups;
    // kept groups (2..maxg samples): rank and entry offset.  flag / rank (idx0 / idx1) are free again: reuse them.
    krank = idx0; eoff = idx1;
    auto scan_class = [&](uint32_t huge, uint32_t &count, uint64_t &nentries) -> int {
        SIMKA_LAUNCH(k_wgsizes, grid_for(ngroups), dim3(256), 0, w->stream, gstart, ngroups, maxg, huge, kflag, ksize);
        int r2;
        if ((r2 = wide_scan(w, kflag, krank, ngroups)) || (r2 = wide_scan(w, ksize, eoff, ngroups))) return r2;
        uint32_t a4[4] = { 0, 0, 0, 0 };
        WCHK(hipMemcpyAsync(&a4[0], krank + (ngroups - 1), 4, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipMemcpyAsync(&a4[1], kflag + (ngroups - 1), 4, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipMemcpyAsync(&a4[2], eoff + (ngroups - 1), 4, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipMemcpyAsync(&a4[3], ksize + (ngroups - 1), 4, hipMemcpyDeviceToHost, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
        count = a4[0] + a4[1]; nentries = (uint64_t)a4[2] + a4[3];
        return 0;
    };
    uint32_t nhuge = 0, nkept = 0; uint64_t hent = 0, nent = 0;
    if (maxg < N) { if ((rc = scan_class(1, nhuge, hent))) return rc; }
    // CSR (owned by the wide state); the huge groups' entries follow the ordinary ones -- sizes are needed first: ordinary class last
    void *old[] = { w->entries, w->groups, w->spans, w->huge, w->cursors };
    for (void *p : old) if (p) WCHK(hipFree(p));
    w->entries = nullptr; w->groups = nullptr; w->spans = nullptr; w->huge = nullptr; w->cursors = nullptr;
    ull *h_entries = nullptr;
    if (nhuge) {      // stage the huge groups now (the scan buffers are reused below)
        WCHK(hipMalloc(&h_entries, (hent + 16) * 8));
        WCHK(hipMalloc(&w->huge, ((uint64_t)nhuge + 4) * sizeof(SimkaSpan)));
        SIMKA_LAUNCH(k_whuge, grid_for(ngroups), dim3(256), 0, w->stream, gstart, ngroups, kflag, krank, eoff, val2, 0ull, h_entries, w->huge);
    }
    if ((rc = scan_class(0, nkept, nent))) { if (h_entries) (void)hipFree(h_entries); return rc; }
    out->nb_shared = (uint64_t)nkept + nhuge;
    if (nkept == 0 && nhuge == 0) return 0;
    const uint32_t C = span_cap - maxg - 8;                   // a span holds the groups that START inside a window of C entries: < span_cap entries
    const uint32_t nspans = (uint32_t)((nent + C - 1) / C);
    WCHK(hipMalloc(&w->entries, (nent + hent + 16) * 8)); WCHK(hipMalloc(&w->groups, ((uint64_t)nkept + 16) * 4));
    WCHK(hipMalloc(&w->spans, ((uint64_t)nspans + 4) * sizeof(SimkaSpan))); WCHK(hipMalloc(&w->cursors, 64));
    WCHK(hipMemsetAsync(w->cursors, 0, 64, w->stream));
    if (nhuge) {      // the huge entries move behind the ordinary ones; their spans' ebase shifts by nent (k_pairs_global reads entries + ebase)
        WCHK(hipMemcpyAsync(w->entries + nent, h_entries, hent * 8, hipMemcpyDeviceToDevice, w->stream));
        std::vector<SimkaSpan> hs(nhuge);
        WCHK(hipMemcpyAsync(hs.data(), w->huge, (size_t)nhuge * sizeof(SimkaSpan), hipMemcpyDeviceToHost, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
        for (auto &sp : hs) sp.ebase += nent;
        WCHK(hipMemcpyAsync(w->huge, hs.data(), (size_t)nhuge * sizeof(SimkaSpan), hipMemcpyHostToDevice, w->stream));
        const ull nh = nhuge;
        WCHK(hipMemcpyAsync(w->cursors + 3, &nh, 8, hipMemcpyHostToDevice, w->stream));
        WCHK(hipStreamSynchronize(w->stream));
        (void)hipFree(h_entries);
    }
    if (nkept) {
        uint32_t *ge, *gs, *spb;
        if ((rc = wide_buf(w, 0, (uint64_t)nkept + 2, &ge)) || (rc = wide_buf(w, 3, (uint64_t)nkept + 2, &gs)) || (rc = wide_buf(w, 10, (uint64_t)nspans * 4 + 8, &spb))) return rc;
        uint32_t *sp_first = spb, *sp_ngrp = spb + nspans, *sp_nent = spb + 2 * (uint64_t)nspans, *sp_maxc = spb + 3 * (uint64_t)nspans;
        WCHK(hipMemsetAsync(sp_first, 0xff, (size_t)nspans * 4, w->stream));
        WCHK(hipMemsetAsync(sp_ngrp, 0, (size_t)nspans * 12, w->stream));
        SIMKA_LAUNCH(k_wgroups, grid_for(ngroups), dim3(256), 0, w->stream, gstart, ngroups, kflag, krank, eoff, val2, C, w->entries, ge, gs, sp_first, sp_ngrp,
                           sp_nent, sp_maxc);
        SIMKA_LAUNCH(k_wspans, grid_for(nspans), dim3(256), 0, w->stream, nspans, sp_first, sp_ngrp, sp_nent, sp_maxc, ge, w->spans, w->cursors);
        SIMKA_LAUNCH(k_wgdesc, grid_for(nkept), dim3(256), 0, w->stream, nkept, ge, gs, sp_first, C, w->groups);
    }
    WCHK(hipGetLastError());
    WCHK(hipStreamSynchronize(w->stream));
    out->huge = w->huge; out->nb_huge = nhuge;
    out->entries = w->entries; out->groups = w->groups; out->spans = w->spans; out->cursors = w->cursors; out->nb_spans = nkept ? nspans : 0; out->nb_entries = nent;
    return 0;
}

// --------------------------------------------------------------------------------------------
// spectra out of / into the wide arena (-keep-tmp, -nb-gpus, -merge-ranges of the driver).  A sample's records are sorted by
// k-mer, so "partition p" = the records whose top log2_parts bits of the 2k-bit k-mer equal p: contiguous, in order.
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_wpartbounds(const ull *hi, const ull *lo, uint64_t n, uint32_t W, uint32_t log2_parts, uint32_t *bound /* [P+1] */) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t P = 1u << log2_parts;
    if (p > P) return;
    if (p == P) { bound[p] = (uint32_t)n; return; }
    // first record whose prefix is >= p
    uint64_t a = 0, b = n;
    while (a < b) {
        const uint64_t m = (a + b) >> 1;
        const uint32_t sh = W - log2_parts;                       // prefix = key >> sh over the (hi, lo) pair
        const ull pre = sh >= 64u ? (hi[m] >> (sh - 64u)) : ((hi[m] << (64u - sh)) | (lo[m] >> sh));
        if (pre < (ull)p) a = m + 1; else b = m;
    }
    bound[p] = (uint32_t)a;
}

int simka_wide_part_counts(SimkaWide *w, uint32_t sample, uint32_t log2_parts, uint32_t *host_counts) {
    const uint32_t P = 1u << log2_parts;
    const uint64_t n = w->s_n[sample], off = w->s_off[sample];
    if (n == 0) { std::fill(host_counts, host_counts + P, 0u); return 0; }
    int rc = wide_ensure_sorted(w, sample); if (rc) return rc;
    uint32_t *d_b; rc = wide_buf(w, 7, (uint64_t)P + 2, &d_b); if (rc) return rc;
    SIMKA_LAUNCH(k_wpartbounds, grid_for((uint64_t)P + 1), dim3(256), 0, w->stream, w->a_hi + off, w->a_lo + off, n, w->W, log2_parts, d_b);
    std::vector<uint32_t> b((size_t)P + 1);
    WCHK(hipMemcpyAsync(b.data(), d_b, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    for (uint32_t p = 0; p < P; p++) host_counts[p] = b[p + 1] - b[p];
    return 0;
}

uint64_t simka_wide_sample_records(SimkaWide *w, uint32_t sample) { return w->s_n[sample]; }
uint64_t simka_wide_full_sorts(SimkaWide *w) { return w->nb_full_sorts; }

// keys: [hi x n][lo x n]
int simka_wide_export(SimkaWide *w, uint32_t sample, void *keys, void *counts, int on_device) {
    const uint64_t n = w->s_n[sample], off = w->s_off[sample];
    if (n == 0) return 0;
    { const int rc = wide_ensure_sorted(w, sample); if (rc) return rc; }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    WCHK(hipMemcpyAsync(keys, w->a_hi + off, n * 8, kind, w->stream));
    WCHK(hipMemcpyAsync((ull *)keys + n, w->a_lo + off, n * 8, kind, w->stream));
    WCHK(hipMemcpyAsync(counts, w->a_cnt + off, n * 4, kind, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    return 0;
}

int simka_wide_import(SimkaWide *w, uint32_t sample, const void *keys, const void *counts, uint64_t n, int on_device) {
    int rc = arena_reserve(w, n); if (rc) return rc;
    w->s_off[sample] = w->a_used; w->s_n[sample] = n; w->s_sorted[sample] = 1;
    if (n == 0) return 0;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    WCHK(hipMemcpyAsync(w->a_hi + w->a_used, keys, n * 8, kind, w->stream));
    WCHK(hipMemcpyAsync(w->a_lo + w->a_used, (const ull *)keys + n, n * 8, kind, w->stream));
    WCHK(hipMemcpyAsync(w->a_cnt + w->a_used, counts, n * 4, kind, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    w->a_used += n;
    return 0;
}

// ---- batch forms (multi-GPU exchange): many samples, caller-chosen destination of every (sample, partition) run -------------------
__global__ void __launch_bounds__(256)
k_wgather_runs(const ull *a_hi, const ull *a_lo, const uint32_t *a_cnt, ull s_off, const uint32_t *bound, const ull *out_off, uint32_t P,
               ull *o_hi, ull *o_lo, uint32_t *o_cnt) {
    for (uint32_t p = blockIdx.x; p < P; p += gridDim.x) {
        const uint32_t b = bound[p], n = bound[p + 1] - b;
        const ull src = s_off + b, dst = out_off[p];
        for (uint32_t i = threadIdx.x; i < n; i += 256) { o_hi[dst + i] = a_hi[src + i]; o_lo[dst + i] = a_lo[src + i]; o_cnt[dst + i] = a_cnt[src + i]; }
    }
}

// out_offsets: host [nb][P]; d_hi / d_lo / d_counts: device buffers addressed by those offsets
int simka_wide_gather(SimkaWide *w, const uint32_t *samples, uint32_t nb, uint32_t log2_parts, const uint64_t *out_offsets, void *d_hi, void *d_lo, void *d_counts) {
    const uint32_t P = 1u << log2_parts;
    uint32_t *d_b; ull *d_off;
    int rc;
    if ((rc = wide_buf(w, 7, (uint64_t)P + 2, &d_b)) || (rc = wide_buf(w, 8, (uint64_t)P + 2, &d_off))) return rc;
    for (uint32_t j = 0; j < nb; j++) {
        const uint32_t s = samples[j];
        const uint64_t n = w->s_n[s], off = w->s_off[s];
        if (n == 0) continue;
        if ((rc = wide_ensure_sorted(w, s))) return rc;
        SIMKA_LAUNCH(k_wpartbounds, grid_for((uint64_t)P + 1), dim3(256), 0, w->stream, w->a_hi + off, w->a_lo + off, n, w->W, log2_parts, d_b);
        WCHK(hipMemcpyAsync(d_off, out_offsets + (size_t)j * P, (size_t)P * 8, hipMemcpyHostToDevice, w->stream));
        SIMKA_LAUNCH(k_wgather_runs, dim3(std::min<uint32_t>(P, 1024u)), dim3(256), 0, w->stream, w->a_hi, w->a_lo, w->a_cnt, (ull)off, d_b, d_off, P,
                           (ull *)d_hi, (ull *)d_lo, (uint32_t *)d_counts);
        WCHK(hipStreamSynchronize(w->stream));          // d_off is reused by the next sample
    }
    WCHK(hipGetLastError());
    return 0;
}

// one sample's (sorted) slice out of a received block: separate word arrays
int simka_wide_import_words(SimkaWide *w, uint32_t sample, const void *d_hi, const void *d_lo, const void *d_counts, uint64_t n) {
    int rc = arena_reserve(w, n); if (rc) return rc;
    w->s_off[sample] = w->a_used; w->s_n[sample] = n; w->s_sorted[sample] = 1;
    if (n == 0) return 0;
    WCHK(hipMemcpyAsync(w->a_hi + w->a_used, d_hi, n * 8, hipMemcpyDeviceToDevice, w->stream));
    WCHK(hipMemcpyAsync(w->a_lo + w->a_used, d_lo, n * 8, hipMemcpyDeviceToDevice, w->stream));
    WCHK(hipMemcpyAsync(w->a_cnt + w->a_used, d_counts, n * 4, hipMemcpyDeviceToDevice, w->stream));
    w->a_used += n;
    return 0;
}
