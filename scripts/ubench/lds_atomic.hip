// Micro-benchmark: LDS atomic-add throughput on gfx950 (non-returning ds_add_u32 / ds_add_u64), 1024-thread blocks,
// one block per CU, random vs conflict-free cell indices.  Prints lane-atomics per cycle per CU (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define CELLS 8192
template <typename T, int MODE>
__global__ void __launch_bounds__(1024) k(uint64_t *out, uint32_t seed) {
    __shared__ T cells[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += 1024) cells[i] = 0;
    __syncthreads();
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 97u;
    for (int i = 0; i < ITER; i++) {
        uint32_t c;
        if (MODE == 0) c = (threadIdx.x + i * 64) & (CELLS - 1);            // consecutive lanes -> consecutive cells
        else if (MODE == 1) { x = x * 1664525u + 1013904223u; c = (x >> 10) & (CELLS - 1); }   // random cell
        else { x = x * 1664525u + 1013904223u; c = ((x >> 10) & (CELLS - 1) & ~63u) | (threadIdx.x & 63u); }   // random row, lane-distinct bank
        atomicAdd(&cells[c], (T)1);
    }
    __syncthreads();
    uint64_t s = 0;
    for (int i = threadIdx.x; i < CELLS; i += 1024) s += cells[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <typename T, int MODE> void run(const char *name, uint64_t *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<T, MODE><<<256, 1024>>>(d, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<T, MODE><<<256, 1024>>>(d, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = 1024.0 * ITER;        // lane-atomics per CU
    printf("%-34s %8.3f ms  %6.2f lane-atomics/cycle/CU   %5.1f cycles per wave instruction\n", name, ms, per_cu / (ms * 1e-3 * 2.4e9),
           ms * 1e-3 * 2.4e9 / (16.0 * ITER));
}
int main() {
    uint64_t *d; (void)hipMalloc(&d, 256 * 1024 * 8);
    run<uint32_t, 0>("u32 consecutive", d); run<uint32_t, 1>("u32 random", d); run<uint32_t, 2>("u32 random row, distinct banks", d);
    run<unsigned long long, 0>("u64 consecutive", d); run<unsigned long long, 1>("u64 random", d); run<unsigned long long, 2>("u64 random row, distinct banks", d);
    return 0;
}
