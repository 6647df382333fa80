// test harness of simka_amd/csrc/simka_trace.h (no GPU needed): registers ranges and launches by hand, then dies the way the ROCm
// runtime does on a memory access fault (a message on stderr + abort()).  tests/test_fault_trace.py checks the dump and
// scripts/fault_resolve.py's reading of it.
#include "../../simka_amd/csrc/simka_trace.h"
int main(int argc, char **argv) {
    const unsigned long long addr = argc > 1 ? strtoull(argv[1], nullptr, 0) : 0x7f0040000000ull;
    if (!simka_trace::enabled()) { fprintf(stderr, "trace disabled\n"); return 3; }
    const uint32_t c = simka_trace::new_ctx();
    simka_trace::add_range(0, "&ctx->d_foff", "simka_ctx.hip", 436, (void *)0x7f0000100000ull, 4096);
    simka_trace::add_range(0, "&L.d_skm_a", "simka_ctx.hip", 823, (void *)0x7f0000200000ull, 1 << 20);
    simka_trace::del_range((void *)0x7f0000200000ull);
    simka_trace::add_range(1, "arena keys: virtual range", "simka_ctx.hip", 470, (void *)0x7f0040000000ull, 2ull << 30);
    simka_trace::add_range(2, "arena keys chunk", "simka_ctx.hip", 0, (void *)0x7f0040000000ull, 1ull << 30);
    simka_trace::set_arena(c, 3, 1ull << 27, 5000000, 1ull << 28);
    for (int i = 0; i < 300; i++) simka_trace::note_launch(i % 2 ? "k_skm_count_fast<true>" : "k_skm_scan<16, true, false>", dim3(1024), dim3(256), 38000, (hipStream_t)(uintptr_t)(0x1000 + (i & 1)));
    fprintf(stderr, "Memory access fault by GPU node-2 (Agent handle: 0x5f769b657750) on address 0x%llx. Reason: Unknown.\n", addr);
    abort();
}
