// simka_sort.hip -- the two primitives of the sort-based path (simka_wide.hip), hand-written for gfx950 (wave64):
//
//   wscan_u32        exclusive prefix sum of n 32-bit values (reduce per 2048-item tile -> recursive scan of the tile sums ->
//                    scan inside the tiles), in place or out of place;
//   wsort_pairs<V>   stable LSD radix sort of 64-bit keys with a 32- or 64-bit payload on the key bits [0, nbits): 8 bits per pass,
//                    per pass  k_rs_hist (LDS histogram of every 4096-item tile, written digit-major)  ->  wscan_u32 over the
//                    256 x tiles table (= the global base of every (digit, tile))  ->  k_rs_scatter (the tile's items ranked in
//                    ORDER: every wave owns a contiguous quarter of the tile, walks it 64 items at a time, and ranks the items of a
//                    step with the wave-match of their digit -- eight ballots -- on top of the wave's running LDS counters).
//
// The input arrays are left untouched (the callers gather through them afterwards); passes ping-pong between the output arrays
// and a scratch pair of the same size.
#ifndef SIMKA_SORT_HIP
#define SIMKA_SORT_HIP
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long ull;

#define WS_BLOCK 256
#define WS_ITEMS 8                       // scan: items per thread
#define WS_TILE (WS_BLOCK * WS_ITEMS)
#define RS_BLOCK 256
#define RS_ITEMS 16                      // sort: items per thread (16 steps of 64 per wave)
#define RS_TILE (RS_BLOCK * RS_ITEMS)
#define RS_DIGITS 256

__device__ __forceinline__ uint32_t ws_wave_incl(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if ((threadIdx.x & 63u) >= (uint32_t)o) v += t; }
    return v;
}

// sums[b] = sum of tile b
static __global__ void __launch_bounds__(WS_BLOCK)
k_ws_reduce(const uint32_t *in, uint64_t n, uint32_t *sums) {
    __shared__ uint32_t s_w[WS_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * WS_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < WS_ITEMS; q++) { const uint64_t i = base + (uint64_t)q * WS_BLOCK + threadIdx.x; if (i < n) s += in[i]; }
    const uint32_t inc = ws_wave_incl(s);
    if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < WS_BLOCK / 64; w++) t += s_w[w]; sums[blockIdx.x] = t; }
}

// out[i] = offs[tile] + exclusive prefix inside the tile (offs == NULL: 0); thread t owns WS_ITEMS CONSECUTIVE items
static __global__ void __launch_bounds__(WS_BLOCK)
k_ws_scan(const uint32_t *in, uint32_t *out, uint64_t n, const uint32_t *offs) {
    __shared__ uint32_t s_w[WS_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * WS_TILE + (uint64_t)threadIdx.x * WS_ITEMS;
    uint32_t v[WS_ITEMS], s = 0;
#pragma unroll
    for (int q = 0; q < WS_ITEMS; q++) { v[q] = (base + q < n) ? in[base + q] : 0u; s += v[q]; }
    const uint32_t inc = ws_wave_incl(s);
    if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = (offs ? offs[blockIdx.x] : 0u) + inc - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += s_w[w];
#pragma unroll
    for (int q = 0; q < WS_ITEMS; q++) { if (base + q < n) out[base + q] = run; run += v[q]; }
}

// exclusive prefix sum of in[0, n) -> out (may alias in).  tmp: scratch of wscan_tmp_u32(n) 32-bit words.
static inline uint64_t wscan_tmp_u32(uint64_t n) {
    uint64_t t = 0;
    while (n > WS_TILE) { n = (n + WS_TILE - 1) / WS_TILE; t += n + 16; }
    return t + 16;
}
static hipError_t wscan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *tmp, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint64_t nt = (n + WS_TILE - 1) / WS_TILE;
    if (nt == 1) {
        SIMKA_LAUNCH(k_ws_scan, dim3(1), dim3(WS_BLOCK), 0, st, in, out, n, (const uint32_t *)nullptr);
        return hipGetLastError();
    }
    SIMKA_LAUNCH(k_ws_reduce, dim3((uint32_t)nt), dim3(WS_BLOCK), 0, st, in, n, tmp);
    hipError_t e = wscan_u32(tmp, tmp, nt, tmp + nt + 16, st);          // tile sums -> tile offsets, in place
    if (e != hipSuccess) return e;
    SIMKA_LAUNCH(k_ws_scan, dim3((uint32_t)nt), dim3(WS_BLOCK), 0, st, in, out, n, (const uint32_t *)tmp);
    return hipGetLastError();
}

// ---- radix sort ------------------------------------------------------------------------------------------------------------
// hist[d * ntiles + tile] = items of the tile whose digit is d
static __global__ void __launch_bounds__(RS_BLOCK)
k_rs_hist(const ull *keys, uint64_t n, uint32_t shift, uint32_t ntiles, uint32_t *hist) {
    __shared__ uint32_t s_h[RS_DIGITS];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
        const uint64_t i = base + (uint64_t)q * RS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & (RS_DIGITS - 1u)], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}

// lanes of the wave whose digit equals mine (digit < 256; inactive lanes pass valid = false)
__device__ __forceinline__ ull rs_match(uint32_t d, bool valid) {
    ull m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) { const ull bal = __ballot((d >> b) & 1u); m &= ((d >> b) & 1u) ? bal : ~bal; }
    return m;
}

// bases[d * ntiles + tile] = global position of the first item of (digit d, tile).  Stable: wave w of the block owns the items
// [w * 1024, (w + 1) * 1024) of the tile and walks them in order.
template <typename V>
static __global__ void __launch_bounds__(RS_BLOCK)
k_rs_scatter(const ull *kin, const V *vin, ull *kout, V *vout, uint64_t n, uint32_t shift, uint32_t ntiles, const uint32_t *bases) {
    __shared__ uint32_t s_cnt[RS_BLOCK / 64][RS_DIGITS];       // per wave: items of each digit, then the wave's running output cursor
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (RS_BLOCK / 64) * RS_DIGITS; i += RS_BLOCK) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint64_t wbase = (uint64_t)blockIdx.x * RS_TILE + (uint64_t)wave * (RS_TILE / (RS_BLOCK / 64));
    ull k[RS_ITEMS]; V v[RS_ITEMS];
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
        const uint64_t i = wbase + (uint64_t)q * 64u + lane;
        k[q] = 0; v[q] = V();
        if (i < n) { k[q] = kin[i]; v[q] = vin[i]; atomicAdd(&s_cnt[wave][(uint32_t)(k[q] >> shift) & (RS_DIGITS - 1u)], 1u); }
    }
    __syncthreads();
    {   // every digit: global base of the tile + the waves before (thread d handles digit d)
        const uint32_t d = threadIdx.x;
        uint32_t run = bases[(uint64_t)d * ntiles + blockIdx.x];
#pragma unroll
        for (uint32_t w = 0; w < RS_BLOCK / 64; w++) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
        const uint64_t i = wbase + (uint64_t)q * 64u + lane;
        const bool valid = i < n;
        const uint32_t d = (uint32_t)(k[q] >> shift) & (RS_DIGITS - 1u);
        const ull m = rs_match(d, valid);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t pos = 0;
        if (valid) pos = s_cnt[wave][d] + below;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (valid && below == 0u) s_cnt[wave][d] += (uint32_t)__popcll(m);       // the first lane of each digit moves the wave's cursor on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (valid) { kout[pos] = k[q]; vout[pos] = v[q]; }
    }
}

// scratch bytes of wsort_pairs: a key + payload ping-pong pair, the digit x tile table and its scan scratch
template <typename V>
static inline uint64_t wsort_tmp_bytes(uint64_t n) {
    const uint64_t nt = (n + RS_TILE - 1) / RS_TILE;
    return (n + 32) * (8 + sizeof(V)) + (nt * RS_DIGITS + 32 + wscan_tmp_u32(nt * RS_DIGITS)) * 4 + 256;
}

// stable sort of (kin, vin) by the key bits [0, nbits) into (kout, vout); kin / vin stay untouched.  n < 2^32.
template <typename V>
static hipError_t wsort_pairs(const ull *kin, ull *kout, const V *vin, V *vout, uint64_t n, uint32_t nbits, void *tmp, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (nbits == 0) {      // nothing to sort by: the stable order is the input order
        hipError_t e = hipMemcpyAsync(kout, kin, n * 8, hipMemcpyDeviceToDevice, st);
        return e != hipSuccess ? e : hipMemcpyAsync(vout, vin, n * sizeof(V), hipMemcpyDeviceToDevice, st);
    }
    const uint32_t passes = (nbits + 7u) / 8u;
    const uint32_t nt = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    ull *kt = (ull *)tmp;
    V *vt = (V *)(kt + n + 32);
    uint32_t *hist = (uint32_t *)(((uintptr_t)(vt + n + 32) + 15u) & ~(uintptr_t)15u);
    uint32_t *stmp = hist + (uint64_t)nt * RS_DIGITS + 32;
    const ull *ks = kin; const V *vs = vin;
    for (uint32_t p = 0; p < passes; p++) {
        // the last pass writes the output arrays; the ones before alternate so that it does
        const bool to_out = ((passes - 1u - p) & 1u) == 0u;
        ull *kd = to_out ? kout : kt; V *vd = to_out ? vout : vt;
        SIMKA_LAUNCH(k_rs_hist, dim3(nt), dim3(RS_BLOCK), 0, st, ks, n, 8u * p, nt, hist);
        hipError_t e = wscan_u32(hist, hist, (uint64_t)nt * RS_DIGITS, stmp, st);
        if (e != hipSuccess) return e;
        SIMKA_LAUNCH((k_rs_scatter<V>), dim3(nt), dim3(RS_BLOCK), 0, st, ks, vs, kd, vd, n, 8u * p, nt, (const uint32_t *)hist);
        ks = kd; vs = vd;
    }
    return hipGetLastError();
}

#endif
