// Micro-benchmark for the count step: ways to insert a realistic key stream into an LDS hash table on gfx950.
// Stream per "partition": NKEYS keys, a fraction SINGLE of them unique (sequencing errors), the rest drawn from HOT hot keys
// (solid k-mers, ~20x coverage).  Block = 512 threads, table TS slots, persistent over `rounds` partitions; each round:
// insert, barrier, per-thread clear of its slots (+ count of distinct, as a checksum), barrier.
//   S0  CAS64 (returning) + add32            (round-1 k_count_fast)
//   S1  read64 first: match -> add32 ; empty -> CAS64 (+add32) ; else next slot
//   S2  CAS32 on the low word (returning) + add32, high word stored by the claimer, verified after the barrier
//   S3  read32 first on the low word: match -> add32 ; empty -> CAS32 ; high word as S2
//   S4  as S1 with the count packed into the key cell: cell = key<<20 | count: claim = CAS64(empty -> key<<20|1), repeat = ds_add_u64
//   S5  two phases, no returning atomic on the common path: NON-returning ds_min_u64 of the key into its home slot, barrier,
//       read64: the home slot holds my key -> add32; else (my key lost its home slot) CAS64 probing from the next slot on
//   S6  as S5, but the losers of the first table go to a second, smaller table with another hash (min / barrier / read), CAS only after that
// Prints ns per partition and inserts per cycle per CU (2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned long long ull;
#define BLOCK 512
#define EMPTY64 0xffffffffffffffffULL
#define EMPTY32 0xffffffffu

__device__ __forceinline__ ull mix64(ull z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

template <int S, int TS, int KPT>
__global__ void __launch_bounds__(BLOCK) k_ins(ull *out, uint32_t rounds, uint32_t hot, uint32_t single_thr16, uint32_t seed) {
    __shared__ ull tk[TS];            // S0/S1/S4: keys (S4: key<<20|count)   S2/S3: low words in the first TS u32, high words after
    __shared__ uint32_t tc[TS];
    __shared__ uint32_t s_bad;
    uint32_t *lo = (uint32_t *)tk, *hi = lo + TS;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t tmask = TS - 1, SPT = TS / BLOCK;
    for (uint32_t i = tid; i < TS; i += BLOCK) { if (S == 2 || S == 3) { lo[i] = EMPTY32; hi[i] = 0; } else tk[i] = EMPTY64; tc[i] = 0; }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    ull checksum = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        const ull base = ((ull)blockIdx.x * rounds + r) * 0x9E3779B97F4A7C15ULL + seed;
        ull keys[KPT];
        // cheap generator (the real kernel gets its keys from loads / a rolling update): 32-bit LCG, hot ids by mask
        uint32_t x = (uint32_t)base ^ (tid * 2654435761u);
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t hsel = x >> 16;
            const bool single = hsel < single_thr16;
            x = x * 1664525u + 1013904223u;
            const uint32_t id = single ? (x | 0x80000000u) : ((x >> 8) & (hot - 1u));
            keys[u] = ((ull)(id * 0x9E3779B1u) << 30) ^ (ull)(id * 0x85EBCA6Bu) ^ ((base & 0xffffffu) << 38);     // < 2^62, distinct ids -> distinct keys (w.h.p.)
        }
        uint32_t vslot[KPT];
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            const ull key = keys[u];
            uint32_t slot = (uint32_t)((key * 0xd6e8feb86659fd93ULL) >> 40) & tmask;
            if (S == 6) slot = (uint32_t)(((ull)(uint32_t)((key * 0xd6e8feb86659fd93ULL) >> 32) * (ull)(TS - TS / 4)) >> 32);
            if (S < 0) { tc[slot] = 1; tk[slot] = key; }
            else if (S == 0) {
                for (int pr = 0; pr < 128; pr++) {
                    const ull prev = atomicCAS(&tk[slot], EMPTY64, key);
                    if (prev == EMPTY64 || prev == key) { atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & tmask;
                }
            } else if (S == 1) {
                for (int pr = 0; pr < 128; pr++) {
                    ull cur = tk[slot];
                    if (cur == EMPTY64) cur = atomicCAS(&tk[slot], EMPTY64, key);
                    if (cur == EMPTY64 || cur == key) { atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & tmask;
                }
            } else if (S == 2) {
                const uint32_t kl = (uint32_t)key, kh = (uint32_t)(key >> 32);
                for (int pr = 0; pr < 128; pr++) {
                    const uint32_t prev = atomicCAS(&lo[slot], EMPTY32, kl);
                    if (prev == EMPTY32) { hi[slot] = kh; atomicAdd(&tc[slot], 1u); break; }
                    if (prev == kl) { atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & tmask;
                }
                vslot[u] = slot;
            } else if (S == 3) {
                const uint32_t kl = (uint32_t)key, kh = (uint32_t)(key >> 32);
                for (int pr = 0; pr < 128; pr++) {
                    uint32_t cur = lo[slot];
                    if (cur == EMPTY32) { cur = atomicCAS(&lo[slot], EMPTY32, kl); if (cur == EMPTY32) { hi[slot] = kh; atomicAdd(&tc[slot], 1u); break; } }
                    if (cur == kl) { atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & tmask;
                }
                vslot[u] = slot;
            } else if (S == 5 || S == 6) {
                (void)__hip_atomic_fetch_min(&tk[slot], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                vslot[u] = slot;
            } else if (S == 4) {
                const ull cell1 = (key << 20) | 1ull;       // 44 key bits in this toy; the real kernel would size it
                for (int pr = 0; pr < 128; pr++) {
                    ull cur = tk[slot];
                    if (cur == EMPTY64) { cur = atomicCAS(&tk[slot], EMPTY64, cell1); if (cur == EMPTY64) break; }
                    if ((cur >> 20) == (key & 0xfffffffffffULL)) { atomicAdd(&tk[slot], 1ull); break; }
                    slot = (slot + 1) & tmask;
                }
            }
        }
        if (S == 5) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < KPT; u++) {
                const ull key = keys[u];
                uint32_t slot = vslot[u];
                if (tk[slot] == key) { atomicAdd(&tc[slot], 1u); continue; }
                for (int pr = 0; pr < 128; pr++) {
                    slot = (slot + 1) & tmask;
                    const ull prev = atomicCAS(&tk[slot], EMPTY64, key);
                    if (prev == EMPTY64 || prev == key) { atomicAdd(&tc[slot], 1u); break; }
                }
            }
        }
        if (S == 6) {
            // second table: the top quarter of the slots is reserved for the losers (first-level hash uses 3/4 of the table)
            __syncthreads();
            bool lost[KPT];
#pragma unroll
            for (int u = 0; u < KPT; u++) {
                const ull key = keys[u];
                lost[u] = tk[vslot[u]] != key;
                if (!lost[u]) atomicAdd(&tc[vslot[u]], 1u);
                else {
                    const uint32_t s2 = (TS - TS / 4) + ((uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 44) & (TS / 4 - 1));
                    (void)__hip_atomic_fetch_min(&tk[s2], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    vslot[u] = s2;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < KPT; u++) {
                if (!lost[u]) continue;
                const ull key = keys[u];
                uint32_t slot = vslot[u];
                if (tk[slot] == key) { atomicAdd(&tc[slot], 1u); continue; }
                for (int pr = 0; pr < 128; pr++) {       // probe inside the second table
                    slot = (TS - TS / 4) + ((slot + 1) & (TS / 4 - 1));
                    const ull prev = atomicCAS(&tk[slot], EMPTY64, key);
                    if (prev == EMPTY64 || prev == key) { atomicAdd(&tc[slot], 1u); break; }
                }
            }
        }
        __syncthreads();
        if (S == 2 || S == 3) {
            bool bad = false;
#pragma unroll
            for (int u = 0; u < KPT; u++) bad |= hi[vslot[u]] != (uint32_t)(keys[u] >> 32);
            if (bad) s_bad = 1;
        }
        // summary + clear: every thread owns SPT consecutive slots
        uint32_t nd = 0, nocc = 0;
#pragma unroll
        for (uint32_t q = 0; q < SPT; q++) {
            const uint32_t sl = tid * SPT + q;
            if (S == 4) { const ull c = tk[sl]; if (c != EMPTY64) { nd++; nocc += (uint32_t)(c & 0xfffffu); tk[sl] = EMPTY64; } }
            else {
                const uint32_t c = tc[sl];
                if (c) { nd++; nocc += c; tc[sl] = 0; if (S == 2 || S == 3) lo[sl] = EMPTY32; else tk[sl] = EMPTY64; }
            }
        }
        checksum += ((ull)nd << 32) | nocc;
        __syncthreads();
    }
    atomicAdd(&out[0], checksum);
    if (tid == 0 && s_bad) atomicAdd(&out[1], 1ull);
}

template <int S, int TS, int KPT>
static void run(const char *name, ull *d, int bpc, uint32_t hot, double single) {
    const uint32_t rounds = 200;
    const int grid = 256 * bpc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(d, 0, 16);
    k_ins<S, TS, KPT><<<grid, BLOCK>>>(d, rounds, hot, (uint32_t)(single * 65536), 1); (void)hipDeviceSynchronize();
    (void)hipMemset(d, 0, 16);
    (void)hipEventRecord(e0); k_ins<S, TS, KPT><<<grid, BLOCK>>>(d, rounds, hot, (uint32_t)(single * 65536), 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ull h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double parts = (double)grid * rounds, keys = parts * BLOCK * KPT;
    fflush(stdout); printf("%-44s TS %5d keys/part %5d blocks/CU %d hot %4u single %.2f: %7.3f ms  %6.0f ns/partition/block  %5.2f inserts/cycle/CU  distinct/part %.0f occ/part %.0f bad %llu\n",
           name, TS, BLOCK * KPT, bpc, hot, single, ms, ms * 1e6 / rounds, keys / 256.0 / (ms * 1e-3 * 2.4e9), (double)(h[0] >> 32) / parts, (double)(h[0] & 0xffffffffu) / parts, h[1]);
}

int main(int argc, char **argv) {
    ull *d; (void)hipMalloc(&d, 64);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t hot = pass == 0 ? 128 : 1024; const double single = pass == 0 ? 0.27 : 0.9;      // C3-like coverage / nearly all distinct
        printf("---- stream: %u hot keys, %.0f %% unique\n", hot, single * 100);
        run<-1, 4096, 6>("SN plain stores (no atomics): floor", d, 3, hot, single);
        run<0, 4096, 6>("S0 CAS64+add", d, 3, hot, single);
        run<1, 4096, 6>("S1 read64, CAS64 if empty, add", d, 3, hot, single);
        run<2, 4096, 6>("S2 CAS32 lo + add, hi verified later", d, 3, hot, single);
        run<3, 4096, 6>("S3 read32, CAS32 if empty, add", d, 3, hot, single);
        run<4, 4096, 6>("S4 read64, packed key|count cell", d, 3, hot, single);
        run<5, 4096, 6>("S5 min64 / barrier / read64, CAS for losers", d, 3, hot, single);
        run<6, 4096, 6>("S6 min64 x2 tables, CAS after", d, 3, hot, single);
        if (pass == 0) {
        run<0, 2048, 6>("S0 small table", d, 4, hot, single);
        run<1, 2048, 6>("S1 small table", d, 4, hot, single);
        run<3, 2048, 6>("S3 small table", d, 4, hot, single);
        run<4, 2048, 6>("S4 small table", d, 4, hot, single);
        run<5, 2048, 6>("S5 small table", d, 4, hot, single);
        run<6, 2048, 6>("S6 small table", d, 4, hot, single);
        }
        run<-1, 8192, 12>("SN 8192 slots floor", d, 1, hot * 2, single);
        run<0, 8192, 12>("S0 8192 slots, 6144 keys", d, 1, hot * 2, single);
        run<1, 8192, 12>("S1 8192 slots, 6144 keys", d, 1, hot * 2, single);
        run<3, 8192, 12>("S3 8192 slots, 6144 keys", d, 1, hot * 2, single);
        run<4, 8192, 12>("S4 8192 slots, 6144 keys", d, 1, hot * 2, single);
        run<4, 8192, 12>("S4 8192 slots, 6144 keys, 2 blocks/CU", d, 2, hot * 2, single);
    }
    return 0;
}
