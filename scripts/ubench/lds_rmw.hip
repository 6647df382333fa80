// Micro-benchmark: cost of one LDS wave instruction on RANDOM slots (gfx950), for the instruction kinds a hash-count step can be
// built from.  Block = 256 threads, BPC blocks per CU (grid = 256 * BPC), table 2048 64-bit slots per block (as k_skm_count_fast).
// Index generation is a murmur-style finaliser per access; the loop is unrolled 4x so that 4 independent accesses are in flight per lane.
// Prints cycles per wave instruction per CU (2.4 GHz nominal) = CU cycles / (wave instructions issued on that CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long ull;
#define ITER 4096
#define TS 2048
enum { M_NONE = 0, M_READ32, M_READ64, M_READ128, M_WRITE64, M_ADD32, M_ADD32_RTN, M_ADD64, M_MIN64, M_CAS32, M_CAS64, M_CAS64_ADD32, M_READ64_ADD32, M_MIN64_READ64_ADD32, M_XCHG64, M_RESET_ONLY, M_CAS64_RESET, M_CAS64_ADD32_RESET };
template <int MODE>
__global__ void __launch_bounds__(256) k(ull *out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) ull t64[TS];
    __shared__ uint32_t t32[TS];
    for (int i = threadIdx.x; i < TS; i += 256) { t64[i] = ~0ull; t32[i] = 0; }
    __syncthreads();
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 97u;
    ull acc = 0;
    for (int i = 0; i < ITER; i += 4) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {      // a real hash per lane: an affine generator is an arithmetic progression ACROSS lanes (few bank conflicts)
            x += 0x9E3779B9u; uint32_t h = x ^ (x >> 16); h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; c[u] = h & (TS - 1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const ull key = ((ull)c[u] << 32) | (x & 0xffu);
            if (MODE == M_NONE) acc += c[u];
            if (MODE == M_READ32) acc += t32[c[u]];
            if (MODE == M_READ64) acc += t64[c[u]];
            if (MODE == M_READ128) { const uint4 v = ((const uint4 *)t64)[c[u] >> 1]; acc += v.x + v.w; }
            if (MODE == M_WRITE64) t64[c[u]] = key;
            if (MODE == M_ADD32) atomicAdd(&t32[c[u]], 1u);
            if (MODE == M_ADD32_RTN) acc += atomicAdd(&t32[c[u]], 1u);
            if (MODE == M_ADD64) atomicAdd(&t64[c[u]], 1ull);
            if (MODE == M_MIN64) (void)__hip_atomic_fetch_min(&t64[c[u]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == M_CAS32) acc += atomicCAS(&t32[c[u]], 0u, (uint32_t)key | 1u);
            if (MODE == M_CAS64) acc += atomicCAS(&t64[c[u]], ~0ull, key);
            if (MODE == M_CAS64_ADD32) { const ull p = atomicCAS(&t64[c[u]], ~0ull, key); if (p == ~0ull || p == key) atomicAdd(&t32[c[u]], 1u); }
            if (MODE == M_READ64_ADD32) { const ull p = t64[c[u]]; if (p == ~0ull || p == key) atomicAdd(&t32[c[u]], 1u); }
        }
        if (MODE == M_XCHG64) {
#pragma unroll
            for (int u = 0; u < 4; u++) acc += atomicExch(&t64[c[u]], ((ull)c[u] << 32) | (x & 0xffu));
        }
        if (MODE == M_RESET_ONLY || MODE == M_CAS64_RESET || MODE == M_CAS64_ADD32_RESET) {
            // the table is emptied every 2048 inserts of the block (as a partition of the count kernel): 1/3 .. 1/2 of the swaps succeed
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const ull key = ((ull)(c[u] * 7u + (x & 3u)) << 20) | (x & 0xfffu);       // several keys per slot
                if (MODE == M_RESET_ONLY) acc += c[u];
                if (MODE == M_CAS64_RESET) acc += atomicCAS(&t64[c[u]], ~0ull, key);
                if (MODE == M_CAS64_ADD32_RESET) { const ull p = atomicCAS(&t64[c[u]], ~0ull, key); if (p == ~0ull || p == key) atomicAdd(&t32[c[u]], 1u); }
            }
            if ((i & 4) != 0) {
                __syncthreads();
                for (int q = threadIdx.x; q < TS; q += 256) { acc += t32[q]; t64[q] = ~0ull; t32[q] = 0; }
                __syncthreads();
            }
        }
        if (MODE == M_MIN64_READ64_ADD32) {
#pragma unroll
            for (int u = 0; u < 4; u++) (void)__hip_atomic_fetch_min(&t64[c[u]], ((ull)c[u] << 32) | (x & 0xffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int u = 0; u < 4; u++) { const ull p = t64[c[u]]; if (p == (((ull)c[u] << 32) | (x & 0xffu))) atomicAdd(&t32[c[u]], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TS; i += 256) acc += t64[i] + t32[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> float run1(ull *d, int bpc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * bpc, 256>>>(d, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<256 * bpc, 256>>>(d, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int MODE> void run(const char *name, ull *d, int nops) {
    static float base[9] = {0};
    printf("%-40s", name);
    for (int bpc : { 1, 2, 4, 8 }) {
        const float ms = run1<MODE>(d, bpc);
        if (MODE == M_NONE) base[bpc] = ms;
        const double waveinstr = 4.0 * bpc * ITER * nops;       // LDS wave instructions per CU
        printf("  bpc %d: %7.3f ms %6.1f cyc/instr (net %6.1f)", bpc, ms, ms * 1e-3 * 2.4e9 / waveinstr, (ms - base[bpc]) * 1e-3 * 2.4e9 / waveinstr);
    }
    printf("\n");
}
int main() {
    ull *d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    run<M_NONE>("index generation only", d, 1);
    run<M_READ32>("ds_read_b32", d, 1); run<M_READ64>("ds_read_b64", d, 1); run<M_READ128>("ds_read_b128", d, 1); run<M_WRITE64>("ds_write_b64", d, 1);
    run<M_ADD32>("ds_add_u32", d, 1); run<M_ADD32_RTN>("ds_add_rtn_u32", d, 1); run<M_ADD64>("ds_add_u64", d, 1); run<M_MIN64>("ds_min_u64", d, 1);
    run<M_CAS32>("ds_cmpst_rtn_b32", d, 1); run<M_CAS64>("ds_cmpst_rtn_b64", d, 1);
    run<M_CAS64_ADD32>("cmpst_rtn_b64 + add_u32 (count step)", d, 2); run<M_READ64_ADD32>("read_b64 + add_u32", d, 2);
    run<M_MIN64_READ64_ADD32>("min_u64, read_b64, add_u32", d, 3);
    run<M_XCHG64>("ds_wrxchg_rtn_b64", d, 1);
    run<M_RESET_ONLY>("table reset every 2048 inserts, no insert", d, 1);
    run<M_CAS64_RESET>("cmpst_rtn_b64, table reset every 2048", d, 1);
    run<M_CAS64_ADD32_RESET>("cmpst_rtn_b64 + add_u32, table reset", d, 2);
    return 0;
}
