// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU / SALU instruction classes the count-side
// kernels are built from, gfx950.  Eight INDEPENDENT instruction streams per wave (no dependent-chain latency), 4 waves per SIMD
// (the occupancy of k_skm_count_fast), every CU busy.  Cycles = s_memtime ticks of a wave / instructions of that wave x the waves
// that share its SIMD (the SIMD issues one wave's instruction at a time), cross-checked with wall time.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#define ITER 2048
#define REP8(x) x x x x x x x x

// one "round" = 8 independent instructions; BODY(d, a, b) expands to the asm text for destination/accumulator d
#define KERNEL(NAME, ASM8)                                                                                                   \
__global__ void __launch_bounds__(256, 4) NAME(uint64_t *out, uint32_t seed) {                                               \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
    uint32_t b0 = a0 ^ 0x9e3779b1u, b1 = (a1 | 1u) & 31u;                                                                    \
    uint64_t q0 = ((uint64_t)a0 << 32) | a1, q1 = ((uint64_t)a2 << 32) | a3, q2 = ((uint64_t)a4 << 32) | a5, q3 = ((uint64_t)a6 << 32) | a7; \
    const uint64_t t0 = __builtin_readcyclecounter();                                                                        \
    for (int i = 0; i < ITER; i++) {                                                                                         \
        REP8(asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b0), "v"(b1) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) \
    }                                                                                                                        \
    const uint64_t t1 = __builtin_readcyclecounter();                                                                        \
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q0 + q1 + q2 + q3;                  \
    if (threadIdx.x == 0) out[(size_t)gridDim.x * 256 + blockIdx.x] = t1 - t0;                                               \
}

// operands: %0..%7 = a0..a7 (32-bit), %8..%11 = q0..q3 (64-bit pairs), %12 = b0, %13 = b1
#define I32_8(op) op " %0, %0, %12\n" op " %1, %1, %12\n" op " %2, %2, %12\n" op " %3, %3, %12\n" op " %4, %4, %12\n" op " %5, %5, %12\n" op " %6, %6, %12\n" op " %7, %7, %12\n"
#define I32S_8(op) op " %0, %13, %0\n" op " %1, %13, %1\n" op " %2, %13, %2\n" op " %3, %13, %3\n" op " %4, %13, %4\n" op " %5, %13, %5\n" op " %6, %13, %6\n" op " %7, %13, %7\n"
#define I32_3_8(op) op " %0, %0, %12, %13\n" op " %1, %1, %12, %13\n" op " %2, %2, %12, %13\n" op " %3, %3, %12, %13\n" op " %4, %4, %12, %13\n" op " %5, %5, %12, %13\n" op " %6, %6, %12, %13\n" op " %7, %7, %12, %13\n"
#define I32_1_8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define I64S_8(op) op " %8, %13, %8\n" op " %9, %13, %9\n" op " %10, %13, %10\n" op " %11, %13, %11\n" op " %8, %13, %8\n" op " %9, %13, %9\n" op " %10, %13, %10\n" op " %11, %13, %11\n"

KERNEL(k_add, I32_8("v_add_u32"))
KERNEL(k_and, I32_8("v_and_b32"))
KERNEL(k_xor, I32_8("v_xor_b32"))
KERNEL(k_lshl, I32S_8("v_lshlrev_b32"))
KERNEL(k_lshr, I32S_8("v_lshrrev_b32"))
KERNEL(k_alignbit, I32_3_8("v_alignbit_b32"))
KERNEL(k_bfrev, I32_1_8("v_bfrev_b32"))
KERNEL(k_mov, I32_1_8("v_mov_b32"))
KERNEL(k_bfe, I32_3_8("v_bfe_u32"))
KERNEL(k_bfi, I32_3_8("v_bfi_b32"))
KERNEL(k_perm, I32_3_8("v_perm_b32"))
KERNEL(k_add3, I32_3_8("v_add3_u32"))
KERNEL(k_lshl_add, I32_3_8("v_lshl_add_u32"))
KERNEL(k_lshl_or, I32_3_8("v_lshl_or_b32"))
KERNEL(k_and_or, I32_3_8("v_and_or_b32"))
KERNEL(k_or3, I32_3_8("v_or3_b32"))
KERNEL(k_xad, I32_3_8("v_xad_u32"))
KERNEL(k_min, I32_8("v_min_u32"))
KERNEL(k_mul_lo, I32_8("v_mul_lo_u32"))
KERNEL(k_mul_hi, I32_8("v_mul_hi_u32"))
KERNEL(k_mul_u24, I32_8("v_mul_u32_u24"))
KERNEL(k_mad_u24, I32_3_8("v_mad_u32_u24"))
KERNEL(k_mul_f32, I32_8("v_mul_f32"))
KERNEL(k_fma_f32, I32_3_8("v_fma_f32"))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %12, vcc\nv_cndmask_b32 %1, %1, %12, vcc\nv_cndmask_b32 %2, %2, %12, vcc\nv_cndmask_b32 %3, %3, %12, vcc\nv_cndmask_b32 %4, %4, %12, vcc\nv_cndmask_b32 %5, %5, %12, vcc\nv_cndmask_b32 %6, %6, %12, vcc\nv_cndmask_b32 %7, %7, %12, vcc\n")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %12\nv_cmp_lt_u32 vcc, %1, %12\nv_cmp_lt_u32 vcc, %2, %12\nv_cmp_lt_u32 vcc, %3, %12\nv_cmp_lt_u32 vcc, %4, %12\nv_cmp_lt_u32 vcc, %5, %12\nv_cmp_lt_u32 vcc, %6, %12\nv_cmp_lt_u32 vcc, %7, %12\n")
KERNEL(k_cmp_u32_sgpr, "v_cmp_lt_u32 s[20:21], %0, %12\nv_cmp_lt_u32 s[22:23], %1, %12\nv_cmp_lt_u32 s[24:25], %2, %12\nv_cmp_lt_u32 s[26:27], %3, %12\nv_cmp_lt_u32 s[20:21], %4, %12\nv_cmp_lt_u32 s[22:23], %5, %12\nv_cmp_lt_u32 s[24:25], %6, %12\nv_cmp_lt_u32 s[26:27], %7, %12\n")
KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %8, %9\nv_cmp_lt_u64 vcc, %9, %10\nv_cmp_lt_u64 vcc, %10, %11\nv_cmp_lt_u64 vcc, %11, %8\nv_cmp_lt_u64 vcc, %8, %10\nv_cmp_lt_u64 vcc, %9, %11\nv_cmp_lt_u64 vcc, %10, %8\nv_cmp_lt_u64 vcc, %11, %9\n")
KERNEL(k_cmp_eq_u64, "v_cmp_eq_u64 vcc, %8, %9\nv_cmp_eq_u64 vcc, %9, %10\nv_cmp_eq_u64 vcc, %10, %11\nv_cmp_eq_u64 vcc, %11, %8\nv_cmp_eq_u64 vcc, %8, %10\nv_cmp_eq_u64 vcc, %9, %11\nv_cmp_eq_u64 vcc, %10, %8\nv_cmp_eq_u64 vcc, %11, %9\n")
KERNEL(k_lshl64, I64S_8("v_lshlrev_b64"))
KERNEL(k_lshr64, I64S_8("v_lshrrev_b64"))
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 2, %9\nv_lshl_add_u64 %9, %9, 2, %10\nv_lshl_add_u64 %10, %10, 2, %11\nv_lshl_add_u64 %11, %11, 2, %8\nv_lshl_add_u64 %8, %8, 2, %9\nv_lshl_add_u64 %9, %9, 2, %10\nv_lshl_add_u64 %10, %10, 2, %11\nv_lshl_add_u64 %11, %11, 2, %8\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %0, %12, %8\nv_mad_u64_u32 %9, vcc, %1, %12, %9\nv_mad_u64_u32 %10, vcc, %2, %12, %10\nv_mad_u64_u32 %11, vcc, %3, %12, %11\nv_mad_u64_u32 %8, vcc, %4, %12, %8\nv_mad_u64_u32 %9, vcc, %5, %12, %9\nv_mad_u64_u32 %10, vcc, %6, %12, %10\nv_mad_u64_u32 %11, vcc, %7, %12, %11\n")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %12\nv_addc_co_u32 %1, vcc, %1, %12, vcc\nv_add_co_u32 %2, vcc, %2, %12\nv_addc_co_u32 %3, vcc, %3, %12, vcc\nv_add_co_u32 %4, vcc, %4, %12\nv_addc_co_u32 %5, vcc, %5, %12, vcc\nv_add_co_u32 %6, vcc, %6, %12\nv_addc_co_u32 %7, vcc, %7, %12, vcc\n")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0\nv_mbcnt_hi_u32_b32 %1, -1, %1\nv_mbcnt_lo_u32_b32 %2, -1, %2\nv_mbcnt_hi_u32_b32 %3, -1, %3\nv_mbcnt_lo_u32_b32 %4, -1, %4\nv_mbcnt_hi_u32_b32 %5, -1, %5\nv_mbcnt_lo_u32_b32 %6, -1, %6\nv_mbcnt_hi_u32_b32 %7, -1, %7\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\nv_readlane_b32 s24, %4, 11\nv_readlane_b32 s25, %5, 13\nv_readlane_b32 s26, %6, 15\nv_readlane_b32 s27, %7, 17\n")
KERNEL(k_readfirstlane, "v_readfirstlane_b32 s20, %0\nv_readfirstlane_b32 s21, %1\nv_readfirstlane_b32 s22, %2\nv_readfirstlane_b32 s23, %3\nv_readfirstlane_b32 s24, %4\nv_readfirstlane_b32 s25, %5\nv_readfirstlane_b32 s26, %6\nv_readfirstlane_b32 s27, %7\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %12, %13 bitop3:0x34\nv_bitop3_b32 %1, %1, %12, %13 bitop3:0x34\nv_bitop3_b32 %2, %2, %12, %13 bitop3:0x34\nv_bitop3_b32 %3, %3, %12, %13 bitop3:0x34\nv_bitop3_b32 %4, %4, %12, %13 bitop3:0x34\nv_bitop3_b32 %5, %5, %12, %13 bitop3:0x34\nv_bitop3_b32 %6, %6, %12, %13 bitop3:0x34\nv_bitop3_b32 %7, %7, %12, %13 bitop3:0x34\n")
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %4, %3 row_shr:8 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %4, %5, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\nv_add_u32_dpp %5, %6, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\nv_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %7, %0, %7 row_shr:2 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sdwa, "v_and_b32_sdwa %0, %0, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %1, %1, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %2, %2, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %3, %3, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %4, %4, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %5, %5, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %6, %6, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\nv_and_b32_sdwa %7, %7, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n")


KERNEL(k_or, I32_8("v_or_b32"))
KERNEL(k_sub, I32_8("v_sub_u32"))
KERNEL(k_max, I32_8("v_max_u32"))
KERNEL(k_ashr, I32S_8("v_ashrrev_i32"))
KERNEL(k_not, I32_1_8("v_not_b32"))
KERNEL(k_lshl_c, "v_lshlrev_b32 %0, 3, %0\nv_lshlrev_b32 %1, 3, %1\nv_lshlrev_b32 %2, 3, %2\nv_lshlrev_b32 %3, 3, %3\nv_lshlrev_b32 %4, 3, %4\nv_lshlrev_b32 %5, 3, %5\nv_lshlrev_b32 %6, 3, %6\nv_lshlrev_b32 %7, 3, %7\n")
KERNEL(k_lshr_c, "v_lshrrev_b32 %0, 3, %0\nv_lshrrev_b32 %1, 3, %1\nv_lshrrev_b32 %2, 3, %2\nv_lshrrev_b32 %3, 3, %3\nv_lshrrev_b32 %4, 3, %4\nv_lshrrev_b32 %5, 3, %5\nv_lshrrev_b32 %6, 3, %6\nv_lshrrev_b32 %7, 3, %7\n")
KERNEL(k_alignbit_c, "v_alignbit_b32 %0, %0, %12, 6\nv_alignbit_b32 %1, %1, %12, 6\nv_alignbit_b32 %2, %2, %12, 6\nv_alignbit_b32 %3, %3, %12, 6\nv_alignbit_b32 %4, %4, %12, 6\nv_alignbit_b32 %5, %5, %12, 6\nv_alignbit_b32 %6, %6, %12, 6\nv_alignbit_b32 %7, %7, %12, 6\n")
KERNEL(k_mov_b64, "v_mov_b64 %8, %9\nv_mov_b64 %9, %10\nv_mov_b64 %10, %11\nv_mov_b64 %11, %8\nv_mov_b64 %8, %9\nv_mov_b64 %9, %10\nv_mov_b64 %10, %11\nv_mov_b64 %11, %8\n")
KERNEL(k_cnd_after_cmp, "v_cmp_lt_u32 vcc, %0, %12\nv_cndmask_b32 %1, %1, %12, vcc\nv_cndmask_b32 %2, %2, %12, vcc\nv_cndmask_b32 %3, %3, %12, vcc\nv_cmp_lt_u32 vcc, %4, %12\nv_cndmask_b32 %5, %5, %12, vcc\nv_cndmask_b32 %6, %6, %12, vcc\nv_cndmask_b32 %7, %7, %12, vcc\n")
KERNEL(k_cnd_cmp_pair, "v_cmp_lt_u32 vcc, %0, %12\nv_cndmask_b32 %1, %1, %12, vcc\nv_cmp_lt_u32 vcc, %2, %12\nv_cndmask_b32 %3, %3, %12, vcc\nv_cmp_lt_u32 vcc, %4, %12\nv_cndmask_b32 %5, %5, %12, vcc\nv_cmp_lt_u32 vcc, %6, %12\nv_cndmask_b32 %7, %7, %12, vcc\n")
KERNEL(k_cnd_sgpr, "v_cmp_lt_u32 s[20:21], %0, %12\nv_cndmask_b32_e64 %1, %1, %12, s[20:21]\nv_cndmask_b32_e64 %2, %2, %12, s[20:21]\nv_cndmask_b32_e64 %3, %3, %12, s[20:21]\nv_cmp_lt_u32 s[22:23], %4, %12\nv_cndmask_b32_e64 %5, %5, %12, s[22:23]\nv_cndmask_b32_e64 %6, %6, %12, s[22:23]\nv_cndmask_b32_e64 %7, %7, %12, s[22:23]\n")
KERNEL(k_cnd_const, "v_cndmask_b32 %0, 0, %0, vcc\nv_cndmask_b32 %1, 0, %1, vcc\nv_cndmask_b32 %2, 0, %2, vcc\nv_cndmask_b32 %3, 0, %3, vcc\nv_cndmask_b32 %4, 0, %4, vcc\nv_cndmask_b32 %5, 0, %5, vcc\nv_cndmask_b32 %6, 0, %6, vcc\nv_cndmask_b32 %7, 0, %7, vcc\n")
KERNEL(k_cmp_class, "v_cmp_eq_u32 vcc, 0, %0\nv_cmp_eq_u32 vcc, 0, %1\nv_cmp_eq_u32 vcc, 0, %2\nv_cmp_eq_u32 vcc, 0, %3\nv_cmp_eq_u32 vcc, 0, %4\nv_cmp_eq_u32 vcc, 0, %5\nv_cmp_eq_u32 vcc, 0, %6\nv_cmp_eq_u32 vcc, 0, %7\n")
// 4 fast + 4 slow interleaved: do the classes overlap?
KERNEL(k_mix_fs, "v_add_u32 %0, %0, %12\nv_alignbit_b32 %1, %1, %12, %13\nv_xor_b32 %2, %2, %12\nv_bfe_u32 %3, %3, %12, %13\nv_and_b32 %4, %4, %12\nv_add3_u32 %5, %5, %12, %13\nv_lshrrev_b32 %6, %13, %6\nv_mul_lo_u32 %7, %7, %12\n")

// SALU: eight independent scalar instructions (operands chosen by the compiler)
__global__ void __launch_bounds__(256, 4) k_salu(uint64_t *out, uint32_t seed) {
    uint32_t s0 = seed, s1 = seed * 3u, s2 = seed * 5u, s3 = seed * 7u, s4 = seed * 11u, s5 = seed * 13u, s6 = seed * 17u, s7 = seed * 19u;
    for (int i = 0; i < ITER; i++) {
        REP8(asm volatile("s_add_u32 %0, %0, %1\ns_and_b32 %1, %1, %2\ns_lshl_b32 %2, %2, 1\ns_bcnt1_i32_b32 %3, %4\ns_add_u32 %4, %4, %5\ns_or_b32 %5, %5, %6\ns_xor_b32 %6, %6, %7\ns_add_u32 %7, %7, %0\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");)
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
__global__ void __launch_bounds__(256, 4) k_valu_salu(uint64_t *out, uint32_t seed) {
    uint32_t s0 = seed, s1 = seed * 3u, s2 = seed * 5u, s3 = seed * 7u;
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
    for (int i = 0; i < ITER; i++) {
        REP8(asm volatile("v_add_u32 %4, %4, %5\ns_add_u32 %0, %0, %1\nv_add_u32 %5, %5, %6\ns_and_b32 %1, %1, %2\nv_add_u32 %6, %6, %7\ns_lshl_b32 %2, %2, 1\nv_add_u32 %7, %7, %4\ns_add_u32 %3, %3, %0\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "scc");)
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + a0 + a1 + a2 + a3;
}
typedef void (*kfn)(uint64_t *, uint32_t);
struct Row { const char *name; kfn f; int n_inst; };

int main(int argc, char **argv) {
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    uint64_t *d; hipMalloc(&d, (size_t)(cus * 8 * 256 + cus * 8) * 8 + 4096);
    uint64_t *h = (uint64_t *)malloc((size_t)cus * 8 * 8);
    Row rows[] = {
#define R(k) { #k, k, 8 }
        R(k_add), R(k_and), R(k_xor), R(k_lshl), R(k_lshr), R(k_alignbit), R(k_bfrev), R(k_mov), R(k_bfe), R(k_bfi), R(k_perm), R(k_add3), R(k_lshl_add), R(k_lshl_or),
        R(k_and_or), R(k_or3), R(k_xad), R(k_min), R(k_mul_lo), R(k_mul_hi), R(k_mul_u24), R(k_mad_u24), R(k_mul_f32), R(k_fma_f32), R(k_cndmask), R(k_cmp_u32), R(k_cmp_u32_sgpr),
        R(k_cmp_u64), R(k_cmp_eq_u64), R(k_lshl64), R(k_lshr64), R(k_lshl_add_u64), R(k_mad_u64_u32), R(k_add_co), R(k_mbcnt), R(k_readlane), R(k_readfirstlane), R(k_bitop3),
        R(k_dpp_mov), R(k_add_dpp), R(k_sdwa), R(k_or), R(k_sub), R(k_max), R(k_ashr), R(k_not), R(k_lshl_c), R(k_lshr_c), R(k_alignbit_c), R(k_mov_b64), R(k_cnd_after_cmp), R(k_cnd_cmp_pair), R(k_cnd_sgpr), R(k_cnd_const), R(k_cmp_class), R(k_mix_fs), R(k_salu), R(k_valu_salu),
    };
    printf("# %s, %d CUs; blocks of 256 threads, W blocks per CU = W waves per SIMD; %d x 64 instructions per wave\n", p.name, cus, ITER);
    printf("%-18s %10s %10s %10s %10s\n", "kernel", "cyc/inst@1", "cyc/inst@2", "cyc/inst@4", "ns/inst@4");
    for (const Row &r : rows) {
        double res[3]; double nsi = 0;
        int wi = 0;
        for (int w : { 1, 2, 4 }) {
            const int blocks = cus * w;
            r.f<<<blocks, 256>>>(d, 1u); hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); r.f<<<blocks, 256>>>(d, 2u); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d + (size_t)blocks * 256, (size_t)blocks * 8, hipMemcpyDeviceToHost);
            double ticks = 0; for (int i = 0; i < blocks; i++) ticks += (double)h[i];
            ticks /= blocks;
            const double inst = (double)ITER * 8 * r.n_inst;       // per wave
            // s_memtime counts at a fixed 100 MHz reference on gfx9 (REFCLK) -- use wall time for cycles instead, ticks as a sanity column
            (void)ticks;
            const double ns_per_inst_simd = (double)ms * 1e6 / (inst * w);      // the SIMD's time per instruction (w waves share it)
            res[wi++] = ns_per_inst_simd;
            nsi = ns_per_inst_simd;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
        printf("%-18s %10.2f %10.2f %10.2f %10.3f\n", r.name, res[0] * ghz, res[1] * ghz, res[2] * ghz, nsi); fflush(stdout);
    }
    return 0;
}
