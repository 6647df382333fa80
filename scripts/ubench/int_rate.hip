// Micro-benchmark: issue rate of the integer ops the key mix is built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o int_rate int_rate.hip ; prints ns per wave-instruction-equivalent.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint64_t seed) {
    uint64_t x = seed + threadIdx.x + blockIdx.x * 256, y = x * 3 + 1;
    uint32_t a = (uint32_t)x, b = (uint32_t)y | 1u;
#pragma unroll 16
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { a = a * b + 1u; }                                  // v_mul_lo_u32 (+add / mad)
        if (OP == 1) { a = __umulhi(a, b) + a; }                          // v_mul_hi_u32
        if (OP == 2) { a = __umul24(a, b) + 1u; }                         // v_mul_u32_u24 / mad_u32_u24
        if (OP == 3) { x = x * 0xff51afd7ed558ccdULL; x ^= x >> 29; }     // 64-bit multiply by constant + xorshift
        if (OP == 4) { a ^= a >> 7; a += b; }                             // xor-shift + add (2 simple ops)
        if (OP == 5) { x ^= x >> 29; x += y; }                            // 64-bit xorshift + add
        if (OP == 6) { x = (uint64_t)(uint32_t)x * 0x9E3779B1u + (x >> 32); }  // v_mad_u64_u32
        if (OP == 7) { a = __builtin_amdgcn_alignbit(a, a, 13) ^ b; b += a; }   // rotate + xor + add
    }
    out[threadIdx.x + blockIdx.x * 256] = x + a;
}
template <int OP> void run(const char *name, uint64_t *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-iterations per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) * ITER
    const double wi = (double)blocks * 4 / (256.0 * 4) * ITER;
    printf("%-28s %8.3f ms   %6.2f ns per wave-iteration per SIMD (~%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
}
int main() {
    uint64_t *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("mul_lo_u32+add", d); run<1>("mul_hi_u32+add", d); run<2>("mul_u24+add", d);
    run<3>("mul64 const + xorshift64", d); run<4>("xorshift32+add", d); run<5>("xorshift64+add64", d);
    run<6>("mad_u64_u32 + shift", d); run<7>("rot+xor+add", d);
    return 0;
}
