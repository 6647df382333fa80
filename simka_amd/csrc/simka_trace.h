// simka_trace.h -- SIMKA_FAULT_TRACE=1: what the library owns on the device and what it launched last, dumped when the process dies.
//
// A GPU memory access fault ends the process from inside the ROCm runtime ("Memory access fault by GPU node-N ... on address X", then
// abort()): no HIP error code, no kernel name, no buffer.  With SIMKA_FAULT_TRACE set (any value) the library keeps
//   * a registry of every device range it owns: each hipMalloc / hipFree of the library (name = the expression that was allocated, file:line,
//     freed ranges are kept and marked), the reserved virtual ranges of the solid arena and every chunk mapped into them with its map time;
//   * a ring of the last SIMKA_TRACE_RING kernel launches (kernel expression, stream, grid, block, dynamic LDS, time, and the arena state
//     the launch saw: records mapped / upper bound of the cursor / capacity);
// and dumps both from a SIGABRT / SIGSEGV / SIGBUS handler (and on request: simka_trace::dump) to stderr and to
// $SIMKA_FAULT_TRACE_DIR/simka_fault_trace.<pid>.txt (default: the working directory).  scripts/fault_resolve.py maps the runtime's fault
// address to a range of the dump and lists the launches that were in flight.  Without the variable every hook is one predictable branch.
// The handler formats with snprintf and write(2) only and takes no lock (the process is dying; a torn entry is better than a deadlock).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>

#ifndef SIMKA_TRACE_RING
#define SIMKA_TRACE_RING 256
#endif
#define SIMKA_TRACE_RANGES 8192

namespace simka_trace {

struct Range {
    const char *name, *file; int line;
    uint64_t base, bytes;
    double t_add, t_del;            // seconds since the first hook; t_del < 0: live
    uint32_t kind;                  // 0 hipMalloc, 1 reserved virtual range, 2 chunk mapped into a reserved range
};
struct Launch {
    const char *kernel; uint64_t stream; uint32_t gx, gy, bx, lds; double t;
    uint64_t seq, arena_mapped, arena_hi, arena_cap; uint32_t ctx_id, sample;
};
struct State {
    Range ranges[SIMKA_TRACE_RANGES];
    std::atomic<uint32_t> nranges{0};
    Launch ring[SIMKA_TRACE_RING];
    std::atomic<uint64_t> seq{0};
    // what the next launches see (set by the count / import paths before they launch)
    std::atomic<uint64_t> arena_mapped{0}, arena_hi{0}, arena_cap{0};
    std::atomic<uint32_t> ctx_id{0}, sample{0};
    std::atomic<uint32_t> next_ctx{0};
    struct sigaction old_abrt, old_segv, old_bus;
    std::atomic<int> dumped{0};
};

inline State &state() { static State s; return s; }

inline double now() {
    static const double t0 = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }();
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec - t0;
}

inline void dump(int sig);
inline void on_signal(int sig) {
    State &s = state();
    dump(sig);
    const struct sigaction *old = sig == SIGABRT ? &s.old_abrt : sig == SIGSEGV ? &s.old_segv : &s.old_bus;
    sigaction(sig, old, nullptr);          // whoever was there before (Python's faulthandler, the default action) takes it from here
    raise(sig);
}

inline bool enabled() {
    static const bool on = [] {
        const char *e = getenv("SIMKA_FAULT_TRACE");
        if (!e || !*e || (e[0] == '0' && !e[1])) return false;
        State &s = state();
        (void)now();
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_signal;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, &s.old_abrt);
        sigaction(SIGSEGV, &sa, &s.old_segv);
        sigaction(SIGBUS, &sa, &s.old_bus);
        return true;
    }();
    return on;
}

inline void add_range(uint32_t kind, const char *name, const char *file, int line, const void *base, uint64_t bytes) {
    if (!enabled() || !base) return;
    State &s = state();
    const uint32_t i = s.nranges.fetch_add(1);
    if (i >= SIMKA_TRACE_RANGES) return;           // (the dump says so)
    Range &r = s.ranges[i];
    r.name = name; r.file = file; r.line = line; r.base = (uint64_t)(uintptr_t)base; r.bytes = bytes; r.t_add = now(); r.t_del = -1.0; r.kind = kind;
}
inline void del_range(const void *base) {
    if (!enabled() || !base) return;
    State &s = state();
    const uint32_t n = s.nranges.load() < SIMKA_TRACE_RANGES ? s.nranges.load() : SIMKA_TRACE_RANGES;
    for (uint32_t i = n; i-- > 0;) {               // the youngest live range with this base
        Range &r = s.ranges[i];
        if (r.base == (uint64_t)(uintptr_t)base && r.t_del < 0 && r.kind != 1u) { r.t_del = now(); return; }
    }
}
// every chunk mapped inside [base, base + bytes) is gone (the arena of a context that is destroyed)
inline void del_chunks(const void *base, uint64_t bytes) {
    if (!enabled() || !base) return;
    State &s = state();
    const uint32_t n = s.nranges.load() < SIMKA_TRACE_RANGES ? s.nranges.load() : SIMKA_TRACE_RANGES;
    const uint64_t b = (uint64_t)(uintptr_t)base;
    for (uint32_t i = 0; i < n; i++) { Range &r = s.ranges[i]; if (r.t_del < 0 && r.base >= b && r.base < b + bytes) r.t_del = now(); }
}
inline uint32_t new_ctx() { return enabled() ? state().next_ctx.fetch_add(1) + 1u : 0u; }
inline void set_arena(uint32_t ctx_id, uint32_t sample, uint64_t mapped, uint64_t hi, uint64_t cap) {
    if (!enabled()) return;
    State &s = state();
    s.ctx_id = ctx_id; s.sample = sample; s.arena_mapped = mapped; s.arena_hi = hi; s.arena_cap = cap;
}
inline void note_launch(const char *kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (!enabled()) return;
    State &s = state();
    const uint64_t q = s.seq.fetch_add(1);
    Launch &l = s.ring[q % SIMKA_TRACE_RING];
    l.kernel = kernel; l.stream = (uint64_t)(uintptr_t)st; l.gx = grid.x; l.gy = grid.y; l.bx = block.x; l.lds = (uint32_t)lds; l.t = now(); l.seq = q;
    l.arena_mapped = s.arena_mapped; l.arena_hi = s.arena_hi; l.arena_cap = s.arena_cap; l.ctx_id = s.ctx_id; l.sample = s.sample;
}

inline void dump(int sig) {
    State &s = state();
    if (s.dumped.fetch_add(1) != 0) return;
    char path[512];
    const char *dir = getenv("SIMKA_FAULT_TRACE_DIR");
    snprintf(path, sizeof path, "%s/simka_fault_trace.%d.txt", dir && *dir ? dir : ".", (int)getpid());
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    char line[768];
    auto out = [&](int n) { if (n <= 0) return; if (n > (int)sizeof line) n = (int)sizeof line; (void)!write(2, line, (size_t)n); if (fd >= 0) (void)!write(fd, line, (size_t)n); };
    const double t = now();
    out(snprintf(line, sizeof line, "[simka-trace] ==== signal %d at t = %.6f s, pid %d (dump: %s) ====\n", sig, t, (int)getpid(), path));
    const uint32_t nr = s.nranges.load(), n = nr < SIMKA_TRACE_RANGES ? nr : SIMKA_TRACE_RANGES;
    out(snprintf(line, sizeof line, "[simka-trace] %u device ranges registered%s (kind: malloc / reserved = virtual range without memory / chunk = memory mapped into a reserved range)\n",
                 nr, nr > SIMKA_TRACE_RANGES ? " -- TABLE FULL, the youngest are missing" : ""));
    for (uint32_t i = 0; i < n; i++) {
        const Range &r = s.ranges[i];
        char life[48];
        if (r.t_del < 0) snprintf(life, sizeof life, "live"); else snprintf(life, sizeof life, "freed@%.6f", r.t_del);
        out(snprintf(line, sizeof line, "[simka-trace] range 0x%012llx - 0x%012llx %12llu B %-8s added@%.6f %-18s %s (%s:%d)\n", (unsigned long long)r.base,
                     (unsigned long long)(r.base + r.bytes), (unsigned long long)r.bytes, r.kind == 0 ? "malloc" : r.kind == 1 ? "reserved" : "chunk", r.t_add, life,
                     r.name ? r.name : "?", r.file ? r.file : "?", r.line));
    }
    const uint64_t q = s.seq.load(), first = q > SIMKA_TRACE_RING ? q - SIMKA_TRACE_RING : 0;
    out(snprintf(line, sizeof line, "[simka-trace] %llu kernel launches so far; the last %llu, oldest first (arena: records mapped / bound of the cursor / capacity as the launch saw them)\n",
                 (unsigned long long)q, (unsigned long long)(q - first)));
    for (uint64_t i = first; i < q; i++) {
        const Launch &l = s.ring[i % SIMKA_TRACE_RING];
        out(snprintf(line, sizeof line, "[simka-trace] launch #%llu t=%.6f (%.3f ms ago) stream 0x%llx grid %u x %u block %u lds %u ctx %u sample %u arena %llu / %llu / %llu  %s\n",
                     (unsigned long long)l.seq, l.t, (t - l.t) * 1e3, (unsigned long long)l.stream, l.gx, l.gy, l.bx, l.lds, l.ctx_id, l.sample,
                     (unsigned long long)l.arena_mapped, (unsigned long long)l.arena_hi, (unsigned long long)l.arena_cap, l.kernel ? l.kernel : "?"));
    }
    out(snprintf(line, sizeof line, "[simka-trace] ==== end of dump ====\n"));
    if (fd >= 0) close(fd);
}

inline hipError_t traced_malloc(const char *name, const char *file, int line, void **p, size_t bytes) {
    const hipError_t e = hipMalloc(p, bytes);          // (SIMKA_EFENCE builds: the fenced allocator -- its macro is in force here)
    if (e == hipSuccess) add_range(0, name, file, line, *p, bytes);
    return e;
}
inline hipError_t traced_free(void *p) {
    del_range(p);
    return hipFree(p);
}

}  // namespace simka_trace

// every allocation, release and launch of the library goes through the hooks
#undef hipMalloc
#undef hipFree
#define hipMalloc(p, n) simka_trace::traced_malloc(#p, __FILE__, __LINE__, (void **)(p), (n))
#define hipFree(p) simka_trace::traced_free((void *)(p))
#define SIMKA_LAUNCH(kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                      \
        simka_trace::note_launch(#kernel, dim3(grid), dim3(block), (size_t)(lds), (stream)); \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                    \
    } while (0)
