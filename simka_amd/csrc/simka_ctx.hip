// simka_ctx.hip -- the C ABI of include/simka_hip.h: context, device memory, kernel sequencing.
// This is the only translation unit that launches kernels; simka_host.cpp holds the pure-host
// pieces (finalisation, CSV, packing).
#include <hip/hip_runtime.h>
#include "simka_efence.h"      // (test builds: -DSIMKA_EFENCE)
#include "simka_trace.h"       // SIMKA_FAULT_TRACE=1: registry of device ranges + ring of launches, dumped when the process dies
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>

#include "../../include/simka_hip.h"
#include "simka_kernels.hip"
#include "simka_skm.hip"
#include "simka_sort.hip"
#include "simka_ingest.hip"
#include "simka_wide.h"

#define SIMKA_EXPORT extern "C" __attribute__((visibility("default")))

// kernel ids for the profiler
enum { KID_SCAN_HIST = 0, KID_LAYOUT, KID_SKM_SCAN, KID_SKM_SPLIT, KID_SKM_COUNT, KID_COUNT, KID_PART_TOTALS, KID_SEG_ROWS, KID_GROUP,
       KID_PAIRS, KID_PAIRS_GLOBAL, KID_NB };
static const char *const KID_NAMES[KID_NB] = { "k_skm_scan<hist>", "k_skm_layout", "k_skm_scan", "k_skm_split", "k_skm_count_fast", "k_skm_count",
                                               "k_part_totals", "k_segment_rows", "k_group", "k_pairs", "k_pairs_global" };

static thread_local std::string g_create_error;

struct simka_ctx {
    simka_config cfg;
    SimkaKeyCfg key;
    SimkaSkmCfg skm;                 // super-k-mer pipeline (the count side for k <= 31)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    bool geometry_ready = false;
    uint32_t B1 = 1;
    uint64_t nparts = 1;
    std::string err;

    // Two lanes (stream + private per-sample scratch): consecutive samples alternate between them, so the scan of sample
    // i+1 (VALU/LDS + writes) overlaps the split/count of sample i (HBM / LDS bound) on the GPU.
    struct Lane {
        hipStream_t stream = nullptr;
        ull *d_b1_count = nullptr, *d_b1_start = nullptr, *d_b1_end = nullptr, *d_b1_cursor = nullptr;   // level-1 buckets
        ull *d_b1_sub = nullptr;                                   // records per fill cursor of a bucket ([B1 << SKM_MAXSUB], k_skm_layout)
        uint32_t *d_tile_r0 = nullptr; uint64_t tile_r0_cap = 0;  // variable-length reads: first read of every scan tile
        uint32_t *d_redo_list = nullptr; ull *d_redo_count = nullptr;   // partitions k_skm_count_fast hands to k_skm_count
        // super-k-mer pipeline: two record buffers (level 1 / level 3 share one), level-2 counters, partition table
        uint4 *d_skm_a = nullptr, *d_skm_b = nullptr; uint64_t skm_a_cap = 0, skm_b_cap = 0;
        uint32_t *d_skm_p = nullptr; uint64_t skm_p_cap = 0;      // partition id of every level-1 record (4-byte side array)
        uint32_t *d_pstart = nullptr, *d_pcnt = nullptr;
        // gather mode (k_skm_chunksort): chunk numbering of the level-1 buckets, chunk table
        uint32_t *d_cbase = nullptr; uint16_t *d_ctab = nullptr; uint64_t ctab_cap = 0;
    };
    static constexpr uint32_t MAX_LANES = 4;
    Lane lanes[MAX_LANES];
    uint32_t nlanes = 2;
    // staging for host-provided reads: one buffer per lane (double buffering), filled through a copy stream of its own, so the
    // host -> device copy of sample i + 1 runs while the kernels of sample i (the other lane) are still busy
    uint64_t *d_reads[MAX_LANES] = {}; uint64_t reads_cap[MAX_LANES] = {};      // (words)
    uint64_t *d_offsets[MAX_LANES] = {}; uint64_t offsets_cap[MAX_LANES] = {};
    hipStream_t copy_stream = nullptr;
    // device-side ingest (simka_ingest_*): per lane the file's text, its line table and the per-line counts; what has been appended so far
    struct Ingest {
        unsigned char *d_text = nullptr; uint64_t text_cap = 0;
        uint32_t *d_lines = nullptr, *d_lb = nullptr, *d_lf = nullptr, *d_tmp = nullptr; uint64_t lines_cap = 0, tmp_cap = 0;      // (d_lb, d_lf: inside d_lines' block)
        ull *d_tot = nullptr;
        uint32_t sample = ~0u; uint64_t nb_bases = 0, nb_frags = 0, nb_reads = 0; bool open = false;
    } ing[MAX_LANES];
    uint32_t *d_l1_ovf = nullptr;                             // [N] capacity-mode scatter overflow flag per sample
    uint64_t nb_exact_fallbacks = 0;
    uint32_t nb_counted_this_run = 0;
    struct Pending { uint32_t sample; SimkaScanArgs a; uint32_t pass = 0, npass = 1; };     // device-resident samples whose flag has not been read yet
    std::vector<Pending> pending;
    // solid spectra of all samples
    ull *d_solid_keys = nullptr; uint32_t *d_solid_counts = nullptr; uint64_t arena_cap = 0;
    // The arena is a RESERVED virtual range of arena_cap records whose physical memory is mapped chunk by chunk as the samples arrive
    // (hipMemAddressReserve / hipMemCreate / hipMemMap): a 138-GB hipMalloc costs 4..6 s, the 10 GB a run at a tenth of C3's depth
    // really needs a fraction of a second.  arena_mapped: records backed by memory (what the kernels may use); arena_hi: upper bound
    // of the arena cursor once everything enqueued has finished; lane_bound: what the sample in flight on a lane may still add.
    bool arena_vmm = false; uint64_t arena_mapped = 0, arena_hi = 0, arena_reserved = 0;      // (arena_reserved: arena_cap rounded up to whole chunks)
    uint64_t lane_bound[MAX_LANES] = {};
    bool used_gather = false;                   // the partitioning kernel of this context is k_skm_chunksort (profile name)
    bool arena_plain = false;                   // plain hipMalloc arena instead of a lazily mapped virtual range (decided in simka_create)
    bool live_counted = false;                  // this context is part of g_live_ctx[device]
    std::vector<char> arena_accounted;          // per sample: its bound is part of arena_hi (a redo or a further pass adds nothing)
    std::vector<hipMemGenericAllocationHandle_t> arena_hk, arena_hc;
    ull *d_arena_cursor = nullptr, *d_sample_base = nullptr;
    uint32_t *d_foff = nullptr, *d_fcnt = nullptr;            // [N][nparts]
    // statistics
    uint64_t *d_stats = nullptr; uint64_t stats_n = 0;
    uint32_t *d_err = nullptr;
    // merge buffers
    ull *d_part_total = nullptr, *d_part_off = nullptr;
    struct PinSlot { void *p = nullptr; size_t cap = 0; } pin[4];       // pinned staging for the large host <-> device copies of the spectrum exchange (see stage_h2d)
    ull *h_part = nullptr; uint64_t h_part_n = 0;             // pinned: the merge's per-partition totals and offsets (a pageable copy of megabytes is pinned and unpinned by the runtime at every merge)
    ull *d_work = nullptr;                                    // [2] work counters of the persistent merge-side kernels (zeroed before a launch)
    ull *d_seg_abs = nullptr; uint4 *d_seg_rows = nullptr;    // [partitions][N] first record of a segment, ends of its 16 key-hash blocks: all partitions, written by the count kernels
                                                               // (seg_all), or -- too many segments, or imported spectra (seg_dirty) -- one merge batch at a time by k_segment_rows
    bool seg_all = false, seg_dirty = false;
    ull *d_entries = nullptr; uint32_t *d_groups = nullptr;
    SimkaSpan *d_spans = nullptr; ull *d_cursors = nullptr; SimkaSpan *d_huge = nullptr;
    uint64_t merge_cap = 0, seg_cap = 0, span_cap = 0, huge_cap = 0;
    // tile-major copy of the CSR for the tiled pair kernel (N too large for one LDS tile): entries, (p, p ln p), segment offsets
    ull *d_tm_ent = nullptr; ktm_p_t *d_tm_p = nullptr; uint32_t *d_tm_off = nullptr;
    uint64_t tm_ent_cap = 0, tm_p_cap = 0, tm_off_cap = 0;
    // -complex-dist: per-sample histogram of solid counts + list of the counts above the histogram
    // 32 <= k <= 51: the samples are counted on the super-k-mer pipeline (k_skm_count_wide), their solid records sorted into the wide arena
    bool wide_hash = false;
    ull *d_wh_hi = nullptr, *d_wh_lo = nullptr, *d_wh_cursor = nullptr; uint32_t *d_wh_cnt = nullptr;
    uint64_t wh_hi_cap = 0, wh_lo_cap = 0, wh_cnt_cap = 0;
    uint64_t nb_wide_hash = 0, nb_wide_sort = 0;                // samples counted on either path (tests / -verbose)
    SimkaWide *wide = nullptr;                                  // 32 <= k <= 63 (or SIMKA_SORT_PATH): the sort-based path of simka_wide.hip
    ull *d_xoff = nullptr; uint64_t xoff_cap = 0;               // simka_gather_samples_device: destination offsets
    // simka_pack_plan / _run, simka_import_block_device: the exchange tables on the device
    ull *d_xrows = nullptr; uint64_t xrows_cap = 0;             // [nb][nb_ranges] records of (sample slot, destination); then their starts
    uint32_t *d_xsamples = nullptr; uint64_t xsamples_cap = 0;  // sample of every slot
    std::vector<uint32_t> plan_samples; uint32_t plan_ranges = 0; uint64_t plan_total = 0;
    uint32_t trace_id = 0;          // SIMKA_FAULT_TRACE: the context's number in the dump
    bool plan_valid = false;        // simka_pack_plan succeeded and nothing has reused d_xrows / d_xsamples or reset the context since
    void drop_plan() { plan_valid = false; plan_ranges = 0; plan_total = 0; plan_samples.clear(); }
    ull *d_hist = nullptr; uint32_t *d_ovf_list = nullptr; ull *d_ovf_cursor = nullptr; uint64_t ovf_cap = 0;

    std::vector<uint8_t> counted;
    std::vector<uint64_t> nb_reads;
    bool merged = false;

    // profiling
    bool profiling = false;
    uint32_t prof_mask = ~0u;                                 // kernel ids (KID_*) whose launches are timed
    struct Ev { int kid; hipEvent_t a, b; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> event_pool;                       // recycled profiling events
    size_t mem_total = 0;                                     // hipMemGetInfo's total, read once
    double prof_ms[KID_NB] = {0};
    uint64_t prof_n[KID_NB] = {0};

    int fail(int code, const char *fmt, ...) {
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};

static int resolve_pending(simka_ctx *ctx, int lane = -1);
#define ARENA_CHUNK ((uint64_t)1 << 27)        // records per physical chunk of the arena: 1 GiB of keys + 512 MiB of counts
static double wall_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

#define HIPCHK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return ctx->fail(SIMKA_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// tests: see launch_timed (one 160 KB block per CU)
__global__ void __launch_bounds__(1024) k_poison_lds(uint32_t mode) {
    extern __shared__ uint32_t poison_smem[];
    for (uint32_t i = threadIdx.x; i < 160u * 1024u / 4u; i += 1024u) poison_smem[i] = mode == 0u ? 0u : (mode == 1u ? 0xffffffffu : (i * 2654435761u) ^ (blockIdx.x * 40503u));
    __syncthreads();
    if (poison_smem[(threadIdx.x * 37u) % (160u * 1024u / 4u)] == 0x12345678u && mode == 77u) poison_smem[0] = 1u;      // (keeps the stores alive)
}
template <typename F>
static inline void launch_timed(simka_ctx *ctx, int kid, F &&f, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    // tests (SIMKA_POISON_LDS=0|1|2): every CU's LDS is overwritten before each kernel -- zeros, ones or an address hash -- so that a kernel
    // which reads LDS it has not written (what an earlier kernel left there) fails every time instead of once in a while
    static const char *poison = simka_test_knob("SIMKA_POISON_LDS");
    if (poison) SIMKA_LAUNCH(k_poison_lds, dim3((uint32_t)ctx->num_cus), dim3(1024), 160 * 1024, st, (uint32_t)atoi(poison));
    static const bool dbg = simka_test_knob("SIMKA_DEBUG_SYNC") != nullptr;     // synchronise after every launch, name the kernel
    if (dbg) {
        fprintf(stderr, "[simka] launch %s\n", KID_NAMES[kid]); fflush(stderr);
        f();
        hipError_t e = hipStreamSynchronize(st);
        fprintf(stderr, "[simka]   -> %s\n", hipGetErrorString(e)); fflush(stderr);
        return;
    }
    if (ctx->profiling && ((ctx->prof_mask >> kid) & 1u)) {
        simka_ctx::Ev ev; ev.kid = kid;
        auto take = [&](hipEvent_t *e) {      // events are recycled: creating two per launch shows up with many small samples
            if (!ctx->event_pool.empty()) { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else (void)hipEventCreate(e);
        };
        take(&ev.a); take(&ev.b);
        (void)hipEventRecord(ev.a, st);
        f();
        (void)hipEventRecord(ev.b, st);
        ctx->events.push_back(ev);
    } else f();
}

static void profile_collect(simka_ctx *ctx) {
    for (auto &ev : ctx->events) {
        float ms = 0;
        (void)hipEventSynchronize(ev.b);
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        ctx->prof_ms[ev.kid] += ms; ctx->prof_n[ev.kid]++;
        ctx->event_pool.push_back(ev.a); ctx->event_pool.push_back(ev.b);
    }
    ctx->events.clear();
}

template <typename T>
static hipError_t dev_alloc_(const char *name, const char *file, int line, T **p, uint64_t n) { return simka_trace::traced_malloc(name, file, line, (void **)p, std::max<uint64_t>(n, 1) * sizeof(T)); }
#define dev_alloc(p, n) dev_alloc_(#p, __FILE__, __LINE__, (p), (n))      // (the fault trace names a range after the expression that holds it)

// ---- large copies between PAGEABLE host memory and the device.  The runtime pins such a buffer for the copy and unpins it afterwards
// (a few GB/s, and the deferred unpin holds the process' address-space lock: the next malloc or page fault of any thread waits --
// measured as 10-25 ms of host stall after every merge in a process that runs after another GPU process).  Copies of 256 KB and more
// therefore go through pinned staging slots of the context; callers that pass pinned memory (simka_host_alloc) are copied from directly.
#define SIMKA_STAGE_MIN ((size_t)256 << 10)
static size_t stage_min() { const char *e = simka_test_knob("SIMKA_STAGE_MIN"); return e ? (size_t)atoll(e) : SIMKA_STAGE_MIN; }      // (tests: stage small copies too)
static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
static void *pin_slot(simka_ctx *ctx, int slot, size_t bytes) {
    auto &ps = ctx->pin[slot];
    if (ps.cap >= bytes && ps.p) return ps.p;
    if (ps.p) { (void)hipHostFree(ps.p); ps.p = nullptr; ps.cap = 0; }
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&ps.p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ps.p = nullptr; return nullptr; }
    ps.cap = want;
    return ps.p;
}
// host source of an upload: src itself (small, or already pinned, or no staging memory), else its copy in the slot.  The slot is
// free again after the next synchronisation of the stream the copy is queued on.
static const void *stage_h2d(simka_ctx *ctx, int slot, const void *src, size_t bytes) {
    if (bytes < stage_min() || host_is_pinned(src)) return src;
    void *st = pin_slot(ctx, slot, bytes);
    if (!st) return src;
    memcpy(st, src, bytes);
    return st;
}
// host destination of a download: dst itself or the slot (then the caller copies slot -> dst after synchronising)
static void *stage_d2h(simka_ctx *ctx, int slot, void *dst, size_t bytes) {
    if (bytes < stage_min() || host_is_pinned(dst)) return dst;
    void *st = pin_slot(ctx, slot, bytes);
    return st ? st : dst;
}

static uint32_t ceil_log2_u64(uint64_t x) { uint32_t l = 0; while (((uint64_t)1 << l) < x && l < 63) l++; return l; }

// ---- flat statistics layout ---------------------------------------------------------------
//   [0,8) header | pair arrays nacc x P | per-sample totals 5 x N | host-derived (complex): canberra P, kl(f64) P
// Totals sit AFTER the pair arrays so that a multi-GPU caller can all-reduce them on their own before the
// merge (needed by -complex-dist) and the head [0, 8 + nacc*P) after it.
static inline uint32_t stats_nacc32(uint32_t flags) { return (flags & SIMKA_DIST_SIMPLE) ? 6u : 4u; }
static inline uint32_t stats_nacc64(uint32_t flags) { return (flags & SIMKA_DIST_COMPLEX) ? 2u : 0u; }
static inline uint32_t stats_nacc(uint32_t flags) { return stats_nacc32(flags) + stats_nacc64(flags); }
static inline uint64_t stats_pairs(uint32_t N) { return (uint64_t)N * (N - 1) / 2; }
static inline uint64_t stats_off_acc(uint32_t N, uint32_t a) { return 8 + (uint64_t)a * stats_pairs(N); }
static inline uint64_t stats_off_tot(uint32_t N, uint32_t flags, uint32_t t) { return stats_off_acc(N, stats_nacc(flags)) + (uint64_t)t * N; }
static inline uint64_t stats_off_derived(uint32_t N, uint32_t flags) { return stats_off_tot(N, flags, SIMKA_NB_TOTALS); }

SIMKA_EXPORT uint64_t simka_stats_nb_u64(uint32_t N, uint32_t flags) {
    return stats_off_derived(N, flags) + ((flags & SIMKA_DIST_COMPLEX) ? 2 * stats_pairs(N) : 0);
}

// out[0..7] = { nacc, offset of pair array 0, offset of totals row 0, offset of derived, nb_pairs, head words, total words, 0 }
SIMKA_EXPORT int simka_stats_layout(uint32_t N, uint32_t flags, uint64_t *out) {
    if (!out || N == 0) return SIMKA_ERR_INVALID;
    out[0] = stats_nacc(flags); out[1] = stats_off_acc(N, 0); out[2] = stats_off_tot(N, flags, 0); out[3] = stats_off_derived(N, flags);
    out[4] = stats_pairs(N); out[5] = stats_off_tot(N, flags, 0); out[6] = simka_stats_nb_u64(N, flags); out[7] = 0;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_stats_describe(uint32_t N, uint32_t flags, uint64_t *h, uint64_t n, simka_stats_view *v) {
    if (!h || !v || N == 0 || n < simka_stats_nb_u64(N, flags)) return SIMKA_ERR_INVALID;
    memset(v, 0, sizeof *v);
    const uint64_t P = stats_pairs(N);
    v->nb_samples = N; v->dist_flags = flags; v->nb_pairs = P;
    v->nb_distinct_kmers = h[0]; v->nb_shared_kmers = h[1];
    v->nb_distinct = h + stats_off_tot(N, flags, SIMKA_TOT_D);
    v->nb_kmers = h + stats_off_tot(N, flags, SIMKA_TOT_N);
    v->sum_sq = h + stats_off_tot(N, flags, SIMKA_TOT_Q);
    v->shared_ij = h + stats_off_acc(N, SIMKA_ACC_SIJ);
    v->shared_ji = h + stats_off_acc(N, SIMKA_ACC_SJI);
    v->distinct_shared = h + stats_off_acc(N, SIMKA_ACC_A);
    v->bray_curtis = h + stats_off_acc(N, SIMKA_ACC_BC);
    if (flags & SIMKA_DIST_SIMPLE) {
        v->chord = h + stats_off_acc(N, SIMKA_ACC_CHORD);
        v->hellinger = h + stats_off_acc(N, SIMKA_ACC_HELL);
    }
    if (flags & SIMKA_DIST_COMPLEX) {
        // derive what updateDistanceComplex accumulates k-mer by k-mer (ref: src/core/SimkaAlgorithm.hpp:404-516) from the
        // both-present sums of the device plus closed forms of the one-sided terms (identities of SURVEY.md App. A.6):
        //   canberra[i][j]  = #k-mers present in exactly one of i,j = D_i + D_j - 2a            (each adds floor(0+1) = 1)
        //   KL[i][j]        = both-present sum + ln2 * ((N_i - S_ij)/N_i + (N_j - S_ji)/N_j)     (one-sided log term is ln 2)
        const uint32_t a32 = stats_nacc32(flags);
        v->whittaker = h + stats_off_acc(N, a32 + 0);
        const int64_t *klfix = (const int64_t *)(h + stats_off_acc(N, a32 + 1));
        uint64_t *canb = h + stats_off_derived(N, flags);
        double *kl = (double *)(canb + P);
        const long double ln2 = logl(2.0L);
        uint64_t cell = 0;
        for (uint64_t i = 0; i < N; i++)
            for (uint64_t j = i + 1; j < N; j++, cell++) {
                canb[cell] = v->nb_distinct[i] + v->nb_distinct[j] - 2 * v->distinct_shared[cell];
                const long double Ni = (long double)v->nb_kmers[i], Nj = (long double)v->nb_kmers[j];
                // an empty sample next to a non-empty one: the reference divides 0 by N = 0 for every k-mer of the other sample
                // (ref: src/core/SimkaAlgorithm.hpp:437-446,500-506), the sum is NaN -- the same 0/0 is left to happen here;
                // two empty samples never reach updateDistanceComplex: 0
                long double one = 0;
                if (v->nb_kmers[i] || v->nb_kmers[j])
                    one = (long double)(v->nb_kmers[i] - v->shared_ij[cell]) / Ni + (long double)(v->nb_kmers[j] - v->shared_ji[cell]) / Nj;
                kl[cell] = (double)((long double)klfix[cell] / (long double)SIMKA_KL_SCALE + ln2 * one);
            }
        v->canberra = canb; v->kl = kl;
    }
    return SIMKA_OK;
}

// ---- lifecycle ----------------------------------------------------------------------------
SIMKA_EXPORT int simka_abi_version(void) { return SIMKA_ABI_VERSION; }

// Reserving, mapping and unmapping virtual ranges from several threads at once (contexts of one process on one device: `simka -nb-gpus
// -gpu-shared`, one worker thread per context) ended, once in ten runs, in a memory access fault at the base of a freshly mapped arena:
// the virtual-memory calls of a process go one at a time.
// LDS budgets that decide how many blocks a CU holds (granules of 1280 bytes, 128 per CU: scripts/ubench/lds_occupancy.hip) -- a few bytes
// more and a persistent grid silently runs in two waves
static_assert(K3_LDS_BYTES(K3_BLOCK) <= 25 * 1280, "k_group<256>: five blocks per CU");
static_assert(K3_LDS_BYTES(2 * K3_BLOCK) <= 64 * 1280, "k_group<512>: two blocks per CU");
static_assert((size_t)SKM_FAST_HEAD + (size_t)SKM_FAST_TS * 12 + (size_t)SKM_FAST_BLOCK * 16 + (size_t)SKM_FAST_HBINS * 4 + (size_t)(SKM_FAST_BLOCK / 64) * SKM_FAST_WREG +
              (size_t)SKM_G_BYTES(SKM_FAST_BLOCK / 64) <= 32 * 1280, "k_skm_count_fast with the complex histogram: four blocks per CU");
static std::mutex g_vmm_lock;
static int arena_ensure(simka_ctx *ctx, uint64_t need);
static int g_live_ctx[64] = {0};        // contexts alive per device (under g_vmm_lock): a second one on a device gets a plain arena
// Virtual ranges of destroyed contexts are RETIRED, never given back: on ROCm 7.0 / gfx950 a range that was unmapped and is mapped
// again (hipMemAddressFree -> a later hipMemAddressReserve returns the same addresses -> hipMemMap) is read and written through stale
// translations now and then -- whole 256-MiB chunks read back wrong, or a GPU memory access fault (scripts/ubench/vmm_two_contexts.hip:
// the same range remapped every round fails within 10 rounds with one thread or two, with or without draining the device around the
// mapping calls; a fresh range every round passes).  A retired range costs address space only; past 2^45 bytes of it the arenas
// of new contexts are plain allocations.
static uint64_t g_vmm_retired_bytes = 0;

SIMKA_EXPORT const char *simka_last_error(const simka_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// k_skm_scan<W, FIXED, HIST>
using SkmScanFn = void (*)(SimkaScanArgs, SimkaSkmCfg, ull *, ull *, uint4 *, const ull *, uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t *, uint32_t, ull);
static int skm_w_index(uint32_t W) { switch (W) { case 1: return 0; case 4: return 1; case 8: return 2; case 12: return 3; case 16: return 4; default: return 5; } }
static SkmScanFn skm_scan_kernel(int wi, bool fixed, bool hist) {
#define SKM_ROW(W) { k_skm_scan<W, false, false>, k_skm_scan<W, false, true>, k_skm_scan<W, true, false>, k_skm_scan<W, true, true> }
    static const SkmScanFn t[6][4] = { SKM_ROW(1), SKM_ROW(4), SKM_ROW(8), SKM_ROW(12), SKM_ROW(16), SKM_ROW(20) };
#undef SKM_ROW
    return t[wi][(fixed ? 2 : 0) | (hist ? 1 : 0)];
}

static int set_lds_attr(simka_ctx *ctx) {
    const int big = 160 * 1024;
    HIPCHK(hipFuncSetAttribute((const void *)k_group<K3_BLOCK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_group<2 * K3_BLOCK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_group<K3_BLOCK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_group<2 * K3_BLOCK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs<false, K4_BLOCK_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs<false, K4_BLOCK_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
#ifdef SIMKA_DEBUG_KNOBS
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
#endif
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs<true, K4_BLOCK_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs_tm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_pairs_tm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    for (int wi = 0; wi < 6; wi++) for (int v = 0; v < 4; v++) HIPCHK(hipFuncSetAttribute((const void *)skm_scan_kernel(wi, v & 2, v & 1), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_split, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_chunksort, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count_fast<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count_fast<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count_wide, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute((const void *)k_skm_count_wide_fast, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    return SIMKA_OK;
}

// ~SIMKA_TARGET_PER_PART k-mer occurrences of the largest sample per partition (one LDS table of k_skm_count_fast)
static uint32_t default_log2_partitions(uint64_t max_kmers, uint32_t shard_count) {
    const uint64_t per_shard = std::max<uint64_t>(1, max_kmers / std::max(1u, shard_count));
    const uint64_t target = SIMKA_TARGET_PER_PART;
    uint32_t pb = ceil_log2_u64((per_shard + target - 1) / target) + ceil_log2_u64(std::max(1u, shard_count));
    return std::min(pb, 20u);
}

SIMKA_EXPORT uint32_t simka_default_log2_partitions(uint64_t max_kmers_per_sample, uint32_t kmer_size) {
    if (kmer_size > 31) return std::min(std::max(8u, default_log2_partitions(max_kmers_per_sample, 1)), 16u);      // sort path: key-prefix ranges
    return std::max(1u, std::min(default_log2_partitions(max_kmers_per_sample, 1), 2u * kmer_size));
}

// partition geometry from the largest sample's k-mer count (all samples must share it: the merge
// joins partition p of every sample, as simkaMerge joins solid/part_p/ of every sample)
static int setup_geometry(simka_ctx *ctx, uint64_t max_kmers) {
    const double tgeo = wall_now();
    const simka_config &c = ctx->cfg;
    SimkaKeyCfg &k = ctx->key;
    uint32_t pb = c.log2_partitions;
    if (pb == 0) pb = default_log2_partitions(max_kmers, c.shard_count);
    if (pb > 20) pb = 20;
    if (pb > k.W) pb = k.W;
    k.l1 = 0; k.l2 = 0; k.pb = pb; k.t = 0;
    ctx->nparts = (uint64_t)1 << k.pb;
    {   // super-k-mer pipeline: the same partition count over two levels (<= 256 level-1 buckets x <= 4096 partitions each)
        SimkaSkmCfg &sk = ctx->skm;
        sk.pb = k.pb;
        static const uint32_t l1_env = simka_exp_knob("SIMKA_SKM_L1") ? (uint32_t)atoi(simka_exp_knob("SIMKA_SKM_L1")) : 8u;      // experiments
        sk.l1 = std::min<uint32_t>(sk.pb, std::min<uint32_t>(l1_env, 8u));
        sk.l2 = sk.pb - sk.l1;
        if (sk.l2 > 12) return ctx->fail(SIMKA_ERR_INVALID, "log2_partitions %u is beyond the two partitioning levels", sk.pb);
        ctx->B1 = 1u << sk.l1;
    }

    const uint32_t N = c.nb_samples;
    // SIMKA_LANES=2 alternates samples between two streams with private scratch (+4 % end to end on C2/C3: the kernels
    // of neighbouring samples overlap); the default single lane keeps per-kernel timings free of overlap.
    // samples alternate between two streams with private scratch: the scan of one sample overlaps the count of another (the
    // kernels are bound by different things: c3_10 212.7 -> 181.9 ms/step); SIMKA_LANES=1 keeps per-kernel timings free of overlap
    const uint32_t want_lanes = simka_test_knob("SIMKA_LANES") ? (uint32_t)std::max(1, atoi(simka_test_knob("SIMKA_LANES"))) : 2u;      // (read per context: bench.py profiles with one lane)
    ctx->nlanes = std::min<uint32_t>(std::min<uint32_t>(want_lanes, simka_ctx::MAX_LANES), std::max<uint32_t>(1u, c.nb_samples));
    for (uint32_t li = 0; li < ctx->nlanes; li++) {
        simka_ctx::Lane &L = ctx->lanes[li];
        if (!L.stream) HIPCHK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIPCHK(dev_alloc(&L.d_b1_count, ctx->B1 + 1));
        HIPCHK(dev_alloc(&L.d_b1_start, ctx->B1 + 1));
        HIPCHK(dev_alloc(&L.d_b1_end, ((uint64_t)ctx->B1 << SKM_MAXSUB) + 1));
        HIPCHK(dev_alloc(&L.d_b1_sub, ((uint64_t)ctx->B1 << SKM_MAXSUB) + 1));
        HIPCHK(dev_alloc(&L.d_b1_cursor, (((uint64_t)ctx->B1 << SKM_MAXSUB) + 1) * SKM_CSTRIDE));
        HIPCHK(dev_alloc(&L.d_redo_list, ctx->nparts + 1));
        HIPCHK(dev_alloc(&L.d_redo_count, 2));
        HIPCHK(dev_alloc(&L.d_pstart, ctx->nparts + 1)); HIPCHK(dev_alloc(&L.d_pcnt, ctx->nparts + 1));
        HIPCHK(dev_alloc(&L.d_cbase, SKM_MAXB1 + 2));
    }
    HIPCHK(dev_alloc(&ctx->d_l1_ovf, c.nb_samples + 1));
    HIPCHK(hipMemsetAsync(ctx->d_l1_ovf, 0, (size_t)(c.nb_samples + 1) * 4, ctx->stream));
    HIPCHK(dev_alloc(&ctx->d_foff, (uint64_t)N * ctx->nparts));
    HIPCHK(dev_alloc(&ctx->d_fcnt, (uint64_t)N * ctx->nparts));
    HIPCHK(hipMemsetAsync(ctx->d_foff, 0, (uint64_t)N * ctx->nparts * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_fcnt, 0, (uint64_t)N * ctx->nparts * 4, ctx->stream));
    // the merge's index of the arena for ALL partitions (40 bytes per segment) if that stays below 1/32 of the device memory
    {
        size_t fr = 0, tot_ = 0;
        const uint64_t nseg = (uint64_t)N * ctx->nparts;
        if (N >= 2 && hipMemGetInfo(&fr, &tot_) == hipSuccess && nseg * 40 <= tot_ / 32 && dev_alloc(&ctx->d_seg_abs, nseg) == hipSuccess) {
            if (dev_alloc(&ctx->d_seg_rows, nseg * 2) == hipSuccess) {
                ctx->seg_all = true; ctx->seg_cap = nseg;
                HIPCHK(hipMemsetAsync(ctx->d_seg_rows, 0, nseg * 32, ctx->stream));
            } else { (void)hipFree(ctx->d_seg_abs); ctx->d_seg_abs = nullptr; (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    HIPCHK(dev_alloc(&ctx->d_part_total, ctx->nparts + 1));
    HIPCHK(dev_alloc(&ctx->d_part_off, ctx->nparts + 1));

    // solid arena: explicit, or a share of what is free now
    uint64_t cap = c.solid_capacity;
    if (cap == 0) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        // worst case: every occurrence distinct & solid, plus the unused tails of the per-block slab reservations
        const uint64_t want = (uint64_t)N * (std::max<uint64_t>(max_kmers, 1) + (uint64_t)ctx->num_cus * 6 * K2_SLAB);
        const uint64_t budget = (uint64_t)(fr * 0.45) / 12;
        cap = std::min(want, budget);
    }
    ctx->arena_cap = cap;
    const double tdbg0 = simka_test_knob("SIMKA_DEBUG_SYNC") ? wall_now() : 0;
    {   // reserve the range; memory comes with arena_ensure().  Without the virtual-memory API: one allocation, as before.
        std::lock_guard<std::mutex> vmm_guard(g_vmm_lock);
        void *vk = nullptr, *vc = nullptr;
        const uint64_t capr = (cap + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
        // plain allocation on request, and whenever another context is alive on this device: chunks being mapped while another
        // context's kernels run is the pattern that faulted (root cause unknown; scripts/ubench/vmm_two_contexts.hip)
        const int dv = c.device >= 0 && c.device < 64 ? c.device : 0;
        (void)dv;
        const bool plain = ctx->arena_plain || g_vmm_retired_bytes > ((uint64_t)1 << 45);      // (decided in simka_create, not at the lazy geometry setup)
        if (!plain && hipMemAddressReserve(&vk, capr * 8, 0, nullptr, 0) == hipSuccess) {
            if (hipMemAddressReserve(&vc, capr * 4, 0, nullptr, 0) == hipSuccess) {
                ctx->arena_vmm = true; ctx->d_solid_keys = (ull *)vk; ctx->d_solid_counts = (uint32_t *)vc; ctx->arena_reserved = capr; ctx->arena_mapped = 0;
                simka_trace::add_range(1, "arena keys: virtual range", __FILE__, __LINE__, vk, capr * 8);
                simka_trace::add_range(1, "arena counts: virtual range", __FILE__, __LINE__, vc, capr * 4);
            } else { (void)hipMemAddressFree(vk, capr * 8); (void)hipGetLastError(); }
        } else (void)hipGetLastError();
        if (!ctx->arena_vmm) {
            HIPCHK(dev_alloc(&ctx->d_solid_keys, cap));
            HIPCHK(dev_alloc(&ctx->d_solid_counts, cap));
            ctx->arena_mapped = cap;
        }
        ctx->arena_hi = 0;
    }
    // A small arena (<= 4 chunks: 6 GB) is backed right here, before any kernel of this context runs.  Mapping chunks WHILE kernels run
    // (the other lane's, with the default two lanes) is the pattern behind a rare GPU memory access fault at a chunk boundary -- twice in
    // ~25 runs of the GPU suite in round 5, never reproduced on demand (docs/rounds/r05.md); larger arenas still grow on demand, behind a
    // device synchronisation (arena_ensure).
    // (SIMKA_ARENA_LAZY: the policy of rounds 1-4 -- every chunk mapped on demand, no synchronisation -- for scripts/stress_fault_trace.sh)
    if (ctx->arena_vmm && !simka_test_knob("SIMKA_ARENA_LAZY") && ctx->arena_reserved <= 4 * ARENA_CHUNK) { const int rca = arena_ensure(ctx, cap); if (rca) return rca; }
    if (simka_test_knob("SIMKA_DEBUG_SYNC")) fprintf(stderr, "[simka] arena of %llu records allocated (%.3f s)\n", (unsigned long long)cap, wall_now() - tdbg0);
    HIPCHK(hipStreamSynchronize(ctx->stream));       // the lanes' streams do not order against the main stream
    if (simka_test_knob("SIMKA_DEBUG_SYNC")) fprintf(stderr, "[simka] geometry ready (%.3f s since the arena, %.3f s in all)\n", wall_now() - tdbg0, wall_now() - tgeo);
    ctx->geometry_ready = true;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_create(const simka_config *cfg, simka_ctx **out) {
    if (!cfg || !out) { g_create_error = "simka_create: null argument"; return SIMKA_ERR_INVALID; }
    if (cfg->struct_size != sizeof(simka_config)) { g_create_error = "simka_create: struct_size mismatch (ABI)"; return SIMKA_ERR_INVALID; }
    if (cfg->nb_samples == 0 || cfg->nb_samples > 65535) { g_create_error = "simka_create: nb_samples must be in [1,65535]"; return SIMKA_ERR_INVALID; }
    if (cfg->kmer_size < 1 || cfg->kmer_size > 127) { g_create_error = "simka_create: kmer_size must be in [1,127]"; return SIMKA_ERR_INVALID; }
    const bool want_wide = cfg->kmer_size > 31 || simka_test_knob("SIMKA_SORT_PATH") != nullptr;
    if (cfg->shard_count == 0 || cfg->shard_index >= cfg->shard_count) { g_create_error = "simka_create: bad shard_index/shard_count"; return SIMKA_ERR_INVALID; }
    if (cfg->flags & ~(uint32_t)SIMKA_CFG_ARENA_PLAIN) { g_create_error = "simka_create: unknown bits in simka_config.flags (an ABI <= 5 caller must zero reserved0)"; return SIMKA_ERR_INVALID; }
    if (cfg->log2_subranges > 8) { g_create_error = "simka_create: log2_subranges must be <= 8"; return SIMKA_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "simka_create: no HIP device available (the HIP path has no CPU fallback)";
        return SIMKA_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "simka_create: bad device ordinal"; return SIMKA_ERR_INVALID; }
    simka_ctx *ctx = new simka_ctx();
    ctx->cfg = *cfg;
    ctx->trace_id = simka_trace::new_ctx();
    {   // the arena mode is fixed HERE, from what is alive at creation: a context that finds another one on its device takes a plain arena
        // (the known fault needs a range that is unmapped and mapped again, which the library no longer does -- retired ranges --, so
        // this is a second line of defence; callers that create several contexts on one device up front set SIMKA_CFG_ARENA_PLAIN
        // on all of them, see include/simka_hip.h)
        std::lock_guard<std::mutex> g_(g_vmm_lock);
        if (cfg->device < 64) g_live_ctx[cfg->device]++;
        ctx->live_counted = true;
        ctx->arena_plain = (cfg->flags & SIMKA_CFG_ARENA_PLAIN) || simka_test_knob("SIMKA_ARENA_MALLOC") || (cfg->device < 64 && g_live_ctx[cfg->device] > 1);
    }
    if (ctx->cfg.abundance_max > 999999999u) ctx->cfg.abundance_max = 999999999u;   // ref: src/core/SimkaAlgorithm.cpp:188
    auto bail = [&](int rc) { g_create_error = ctx->err; simka_destroy(ctx); return rc; };
    if (hipSetDevice(cfg->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(SIMKA_ERR_HIP); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (cfg->stream) ctx->stream = (hipStream_t)cfg->stream;
    else { if (hipStreamCreate(&ctx->stream) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(SIMKA_ERR_HIP); } ctx->own_stream = true; }

    SimkaKeyCfg &k = ctx->key;
    memset(&k, 0, sizeof k);
    k.k = cfg->kmer_size; k.W = 2 * cfg->kmer_size; k.mask = k.W >= 64 ? ~0ull : (1ull << k.W) - 1ull; k.xs = (k.W + 1) / 2;      // (hash path: W <= 62)
    k.shard_index = cfg->shard_index; k.shard_count = cfg->shard_count;
    if (!want_wide) {
        // minimizer geometry: W m-mers per k-mer from {20,16,12,8,4}, the largest that leaves m = k - W + 1 >= 14 (k < 17: every
        // k-mer is its own minimizer).  m must be large: the partition is a function of the minimizer, so 4^m / 2 minimizer
        // classes -- of which only the low-hash tenth ever wins a window -- have to spread over up to 2^20 partitions (m = 12
        // on C3's 2^19 partitions: one or two heavy classes per partition, sizes all over the place, 10 % of the partitions redone).
        // A record holds n + k - 1 <= 51 bases.
        SimkaSkmCfg &sk = ctx->skm;
        memset(&sk, 0, sizeof sk);
        sk.k = cfg->kmer_size;
        sk.W = 1;
        for (uint32_t w : { 20u, 16u, 12u, 8u, 4u }) if (sk.k >= w + 13u) { sk.W = w; break; }
        sk.m = sk.k - sk.W + 1;
        sk.nmax = std::min<uint32_t>(32u, 52u - sk.k);
        sk.mmask = (uint32_t)((1ull << (2 * sk.m)) - 1ull);
        sk.kmask = k.mask;
        skm_set_shard(sk, cfg->shard_index, cfg->shard_count);
    } else if (cfg->kmer_size >= 32 && cfg->kmer_size <= 51 && !simka_test_knob("SIMKA_SORT_PATH") && !simka_test_knob("SIMKA_WIDE_SORT")) {
        // 32 <= k <= 51: a record (<= 51 bases) still holds a k-mer.  The minimizer is the smallest of W = 20 m-mers in the MIDDLE of
        // the k-mer: m-mers d .. d + W - 1 with 2 d = k - (W + m - 1), so that a k-mer and its reverse complement look at the same
        // m-mers (m is lowered by one where the parity demands it).
        SimkaSkmCfg &sk = ctx->skm;
        memset(&sk, 0, sizeof sk);
        sk.k = cfg->kmer_size;
        sk.W = 20;
        sk.m = std::min<uint32_t>(16u, sk.k - 19u);
        if ((sk.k - (sk.W + sk.m - 1u)) & 1u) sk.m--;
        sk.d = (sk.k - (sk.W + sk.m - 1u)) / 2u;
        sk.nmax = 52u - sk.k;
        sk.mmask = (uint32_t)((1ull << (2 * sk.m)) - 1ull);
        sk.kmask = ~0ull;
        skm_set_shard(sk, 0, 1);          // (shards keep k-mers, not partitions: k_skm_count_wide)
        ctx->wide_hash = true;
    }

    const double tlds = wall_now();
    int rc = set_lds_attr(ctx);
    if (rc) return bail(rc);
    if (simka_test_knob("SIMKA_DEBUG_SYNC")) fprintf(stderr, "[simka] kernel attributes set (%.3f s)\n", wall_now() - tlds);
    const uint32_t N = cfg->nb_samples;
    ctx->stats_n = simka_stats_nb_u64(N, cfg->dist_flags);
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; };
    if (!chk(dev_alloc(&ctx->d_stats, ctx->stats_n), "hipMalloc(stats)")) return bail(SIMKA_ERR_NOMEM);
    if (!chk(hipMemsetAsync(ctx->d_stats, 0, ctx->stats_n * 8, ctx->stream), "memset(stats)")) return bail(SIMKA_ERR_HIP);
    if (!chk(dev_alloc(&ctx->d_err, 4), "hipMalloc(err)")) return bail(SIMKA_ERR_NOMEM);
    if (!chk(hipMemsetAsync(ctx->d_err, 0, 16, ctx->stream), "memset(err)")) return bail(SIMKA_ERR_HIP);
    if (!chk(dev_alloc(&ctx->d_arena_cursor, 2), "hipMalloc(cursor)")) return bail(SIMKA_ERR_NOMEM);
    if (!chk(hipMemsetAsync(ctx->d_arena_cursor, 0, 16, ctx->stream), "memset(cursor)")) return bail(SIMKA_ERR_HIP);
    if (!chk(dev_alloc(&ctx->d_sample_base, N + 1), "hipMalloc(sample_base)")) return bail(SIMKA_ERR_NOMEM);
    if (!chk(dev_alloc(&ctx->d_cursors, 4), "hipMalloc(cursors)")) return bail(SIMKA_ERR_NOMEM);
    if (!chk(dev_alloc(&ctx->d_work, 2), "hipMalloc(work)")) return bail(SIMKA_ERR_NOMEM);
    if (cfg->dist_flags & SIMKA_DIST_COMPLEX) {
        ctx->ovf_cap = simka_test_knob("SIMKA_OVF_CAP") ? (uint64_t)atoll(simka_test_knob("SIMKA_OVF_CAP")) : (uint64_t)1 << 22;      // (tests shrink it)
        if (!chk(dev_alloc(&ctx->d_hist, (uint64_t)N * SIMKA_HIST_MAX), "hipMalloc(hist)")) return bail(SIMKA_ERR_NOMEM);
        if (!chk(dev_alloc(&ctx->d_ovf_list, 2 * ctx->ovf_cap), "hipMalloc(ovf)")) return bail(SIMKA_ERR_NOMEM);
        if (!chk(dev_alloc(&ctx->d_ovf_cursor, 2), "hipMalloc(ovf cursor)")) return bail(SIMKA_ERR_NOMEM);
        if (!chk(hipMemsetAsync(ctx->d_hist, 0, (uint64_t)N * SIMKA_HIST_MAX * 8, ctx->stream), "memset(hist)")) return bail(SIMKA_ERR_HIP);
        if (!chk(hipMemsetAsync(ctx->d_ovf_cursor, 0, 16, ctx->stream), "memset(ovf)")) return bail(SIMKA_ERR_HIP);
    }
    ctx->counted.assign(N, 0);
    ctx->nb_reads.assign(N, 0);
    if (want_wide) {
        if (simka_wide_create(&ctx->wide, cfg->device, N, cfg->kmer_size, ctx->stream) != SIMKA_WIDE_OK) { ctx->err = "cannot create the wide-k state"; return bail(SIMKA_ERR_NOMEM); }
        simka_wide_set_shard(ctx->wide, cfg->shard_index, cfg->shard_count);
    } else if (cfg->max_kmers_per_sample) { rc = setup_geometry(ctx, cfg->max_kmers_per_sample); if (rc) return bail(rc); }
    *out = ctx;
    return SIMKA_OK;
}

SIMKA_EXPORT void simka_destroy(simka_ctx *ctx) {
    if (!ctx) return;
    if (ctx->live_counted) { std::lock_guard<std::mutex> g_(g_vmm_lock); if (ctx->cfg.device >= 0 && ctx->cfg.device < 64) g_live_ctx[ctx->cfg.device]--; ctx->live_counted = false; }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->wide) { simka_wide_destroy(ctx->wide); ctx->wide = nullptr; }
    for (auto &L : ctx->lanes) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        void *lp[] = { L.d_b1_count, L.d_b1_start, L.d_b1_end, L.d_b1_cursor, L.d_b1_sub, L.d_tile_r0, L.d_redo_list, L.d_redo_count,
                       L.d_skm_a, L.d_skm_b, L.d_skm_p, L.d_pstart, L.d_pcnt, L.d_cbase, L.d_ctab };
        for (void *q : lp) if (q) (void)hipFree(q);
        if (L.stream && L.stream != ctx->stream) (void)hipStreamDestroy(L.stream);
    }
    for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    for (uint32_t li = 0; li < simka_ctx::MAX_LANES; li++) { if (ctx->d_reads[li]) (void)hipFree(ctx->d_reads[li]); if (ctx->d_offsets[li]) (void)hipFree(ctx->d_offsets[li]); }
    for (auto &g : ctx->ing) { void *q[] = { g.d_text, g.d_lines, g.d_tmp, g.d_tot }; for (void *p_ : q) if (p_) (void)hipFree(p_); }
    if (ctx->arena_vmm) {      // unmap and release the chunks, give the ranges back
        std::lock_guard<std::mutex> vmm_guard(g_vmm_lock);
        (void)hipDeviceSynchronize();
        // chunk by chunk, mirroring the hipMemMap calls (one unmap over several mappings is not guaranteed to release them all)
        for (uint64_t at = 0; at < ctx->arena_mapped; at += ARENA_CHUNK) {
            const hipError_t e1 = hipMemUnmap((char *)ctx->d_solid_keys + at * 8, ARENA_CHUNK * 8), e2 = hipMemUnmap((char *)ctx->d_solid_counts + at * 4, ARENA_CHUNK * 4);
            if ((e1 != hipSuccess || e2 != hipSuccess) && simka_test_knob("SIMKA_DEBUG_SYNC")) fprintf(stderr, "[simka] hipMemUnmap of arena chunk %llu failed: %s\n", (unsigned long long)(at / ARENA_CHUNK), hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            (void)hipGetLastError();
        }
        simka_trace::del_chunks(ctx->d_solid_keys, ctx->arena_reserved * 8); simka_trace::del_chunks(ctx->d_solid_counts, ctx->arena_reserved * 4);
        for (auto h : ctx->arena_hk) (void)hipMemRelease(h);
        for (auto h : ctx->arena_hc) (void)hipMemRelease(h);
        g_vmm_retired_bytes += ctx->arena_reserved * 12;      // (no hipMemAddressFree: see g_vmm_retired_bytes)
        ctx->d_solid_keys = nullptr; ctx->d_solid_counts = nullptr;
    }
    void *ptrs[] = { ctx->d_wh_hi, ctx->d_wh_lo, ctx->d_wh_cnt, ctx->d_wh_cursor, ctx->d_l1_ovf, ctx->d_solid_keys, ctx->d_solid_counts, ctx->d_arena_cursor,
                     ctx->d_sample_base, ctx->d_foff, ctx->d_fcnt, ctx->d_stats, ctx->d_err, ctx->d_part_total,
                     ctx->d_part_off, ctx->d_work, ctx->d_seg_abs, ctx->d_seg_rows, ctx->d_entries, ctx->d_groups,
                     ctx->d_spans, ctx->d_cursors, ctx->d_huge, ctx->d_xoff, ctx->d_hist, ctx->d_ovf_list, ctx->d_ovf_cursor,
                     ctx->d_tm_ent, ctx->d_tm_p, ctx->d_tm_off, ctx->d_xrows, ctx->d_xsamples };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (ctx->h_part) (void)hipHostFree(ctx->h_part);
    for (auto &ps : ctx->pin) if (ps.p) (void)hipHostFree(ps.p);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

SIMKA_EXPORT int simka_sync(simka_ctx *ctx) {
    if (!ctx) return SIMKA_ERR_INVALID;
    { int rcp = resolve_pending(ctx); if (rcp) return rcp; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_reset(simka_ctx *ctx) {
    if (!ctx) return SIMKA_ERR_INVALID;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    if (ctx->wide) simka_wide_reset(ctx->wide);
    for (uint32_t li = 0; li < ctx->nlanes; li++) if (ctx->lanes[li].stream) HIPCHK(hipStreamSynchronize(ctx->lanes[li].stream));
    const uint32_t N = ctx->cfg.nb_samples;
    HIPCHK(hipMemsetAsync(ctx->d_stats, 0, ctx->stats_n * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, 16, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_arena_cursor, 0, 16, ctx->stream));
    if (ctx->geometry_ready) {
        HIPCHK(hipMemsetAsync(ctx->d_foff, 0, (uint64_t)N * ctx->nparts * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_fcnt, 0, (uint64_t)N * ctx->nparts * 4, ctx->stream));
        if (ctx->seg_all) HIPCHK(hipMemsetAsync(ctx->d_seg_rows, 0, (uint64_t)N * ctx->nparts * 32, ctx->stream));
    }
    ctx->seg_dirty = false;
    ctx->arena_hi = 0;
    for (auto &b_ : ctx->lane_bound) b_ = 0;
    std::fill(ctx->arena_accounted.begin(), ctx->arena_accounted.end(), 0);
    if (ctx->d_hist) {
        HIPCHK(hipMemsetAsync(ctx->d_hist, 0, (uint64_t)N * SIMKA_HIST_MAX * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_ovf_cursor, 0, 16, ctx->stream));
    }
    ctx->pending.clear();
    ctx->drop_plan();
    ctx->nb_counted_this_run = 0;
    if (ctx->d_l1_ovf) HIPCHK(hipMemsetAsync(ctx->d_l1_ovf, 0, (size_t)(N + 1) * 4, ctx->stream));
    std::fill(ctx->counted.begin(), ctx->counted.end(), 0);
    std::fill(ctx->nb_reads.begin(), ctx->nb_reads.end(), 0);
    ctx->merged = false;
    HIPCHK(hipStreamSynchronize(ctx->stream));       // the lanes' streams do not order against the main stream
    return SIMKA_OK;
}

static int check_device_error(simka_ctx *ctx) {
    uint32_t e = 0;
    HIPCHK(hipMemcpyAsync(&e, ctx->d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (e & SIMKA_DEVERR_TABLE_OVERFLOW) return ctx->fail(SIMKA_ERR_OVERFLOW, "a partition could not be counted in 2^16 rounds of its LDS table (%d slots)", SKM_CNT_TS);
    if (e & SIMKA_DEVERR_ARENA_FULL) return ctx->fail(SIMKA_ERR_NOMEM, "solid-spectrum arena exhausted (%llu records): raise solid_capacity", (unsigned long long)ctx->arena_cap);
    if (e & SIMKA_DEVERR_SAMPLE_TOO_BIG) return ctx->fail(SIMKA_ERR_OVERFLOW, "a sample holds more than 2^32 solid k-mers");
    if (e & SIMKA_DEVERR_GROUP_OVERFLOW) return ctx->fail(SIMKA_ERR_OVERFLOW, "merge: a sub-range could not be split below the LDS capacity");
    if (e & SIMKA_DEVERR_CSR_FULL) return ctx->fail(SIMKA_ERR_NOMEM, "merge: group buffer exhausted");
    // (SIMKA_DEVERR_SEGMENT_TOO_BIG is not an error: simka_merge finds the largest segment itself and takes 32-bit rows then)
    if (e & SIMKA_DEVERR_UNORDERED) return ctx->fail(SIMKA_ERR_INVALID, "merge: a spectrum is not ordered by key prefix inside its partitions (imported from another version?)");
    return SIMKA_OK;
}

template <typename T>
static int ensure_cap_(const char *name, int line, simka_ctx *ctx, T **p, uint64_t *cap, uint64_t need) {
    if (*cap >= need && *p) return SIMKA_OK;
    if (*p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(*p)); *p = nullptr; *cap = 0; }
    const uint64_t n = need + need / 8 + 16;
    hipError_t e = dev_alloc_(name, __FILE__, line, p, n);
    if (e != hipSuccess) return ctx->fail(SIMKA_ERR_NOMEM, "hipMalloc of %llu bytes failed: %s", (unsigned long long)(n * sizeof(T)), hipGetErrorString(e));
    *cap = n;
    return SIMKA_OK;
}
#define ensure_cap(ctx, p, cap, need) ensure_cap_(#p, __LINE__, (ctx), (p), (cap), (need))

// back the first `need` records of the arena with memory (no-op when they already are)
static int arena_ensure(simka_ctx *ctx, uint64_t need) {
    need = std::min(need, ctx->arena_cap);                                 // (beyond the capacity: the kernels flag SIMKA_DEVERR_ARENA_FULL)
    if (!ctx->arena_vmm || need <= ctx->arena_mapped) return SIMKA_OK;
    // no kernel of THIS device in flight while its page tables change (see the geometry setup; a chunk is 1.3e8 records: rare).  The wait
    // happens before the process-wide lock is taken -- other contexts and devices are not held up behind it -- and a sticky
    // asynchronous error surfaces here instead of being swallowed.
    if (!simka_test_knob("SIMKA_ARENA_LAZY")) HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> vmm_guard(g_vmm_lock);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->cfg.device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    while (ctx->arena_mapped < need) {
        hipMemGenericAllocationHandle_t hk, hc;
        const uint64_t at = ctx->arena_mapped;
        if (hipMemCreate(&hk, ARENA_CHUNK * 8, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return ctx->fail(SIMKA_ERR_NOMEM, "the solid-spectrum arena cannot grow beyond %llu records (device memory exhausted)", (unsigned long long)at); }
        if (hipMemCreate(&hc, ARENA_CHUNK * 4, &prop, 0) != hipSuccess) { (void)hipMemRelease(hk); (void)hipGetLastError(); return ctx->fail(SIMKA_ERR_NOMEM, "the solid-spectrum arena cannot grow beyond %llu records (device memory exhausted)", (unsigned long long)at); }
        // (a failure after hipMemCreate must not leak the handles: undo this chunk, keep the chunks before it)
        hipError_t me = hipMemMap((char *)ctx->d_solid_keys + at * 8, ARENA_CHUNK * 8, 0, hk, 0);
        bool mk = me == hipSuccess, mc = false;
        if (mk) { me = hipMemMap((char *)ctx->d_solid_counts + at * 4, ARENA_CHUNK * 4, 0, hc, 0); mc = me == hipSuccess; }
        if (mc) me = hipMemSetAccess((char *)ctx->d_solid_keys + at * 8, ARENA_CHUNK * 8, &acc, 1);
        if (mc && me == hipSuccess) me = hipMemSetAccess((char *)ctx->d_solid_counts + at * 4, ARENA_CHUNK * 4, &acc, 1);
        if (me != hipSuccess) {
            if (mk) (void)hipMemUnmap((char *)ctx->d_solid_keys + at * 8, ARENA_CHUNK * 8);
            if (mc) (void)hipMemUnmap((char *)ctx->d_solid_counts + at * 4, ARENA_CHUNK * 4);
            (void)hipMemRelease(hk); (void)hipMemRelease(hc); (void)hipGetLastError();
            return ctx->fail(SIMKA_ERR_HIP, "mapping chunk %llu of the solid-spectrum arena failed: %s", (unsigned long long)(at / ARENA_CHUNK), hipGetErrorString(me));
        }
        ctx->arena_hk.push_back(hk); ctx->arena_hc.push_back(hc);
        ctx->arena_mapped = at + ARENA_CHUNK;
        simka_trace::add_range(2, "arena keys chunk", __FILE__, (int)(at / ARENA_CHUNK), (char *)ctx->d_solid_keys + at * 8, ARENA_CHUNK * 8);       // (line = chunk number)
        simka_trace::add_range(2, "arena counts chunk", __FILE__, (int)(at / ARENA_CHUNK), (char *)ctx->d_solid_counts + at * 4, ARENA_CHUNK * 4);
    }
    return SIMKA_OK;
}
// the arena cursor as of now (every stream that was synchronised by the caller is accounted for)
static int arena_cursor_now(simka_ctx *ctx, ull *cur, hipStream_t st) {
    HIPCHK(hipMemcpyAsync(cur, ctx->d_arena_cursor, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SIMKA_OK;
}

// enqueue the count-side kernels of one sample.  exact=false: capacity-sized level-1 buckets, no histogram pass; the
// kernels after the scatter skip themselves if it flags an overflow, and resolve_pending() redoes the sample exactly.
// ---- super-k-mer pipeline: enqueue the count-side kernels of one sample (see simka_skm.hip) ----------------------------------
// exact=false: capacity-sized level-1 buckets; if one overflows the later kernels skip themselves and resolve_pending() redoes
// the sample with exact=true (a histogram-only scan first).
// scan + split of one sample (or one pass over it) on lane L: the partitioned records end up in L.d_skm_b, described by L.d_pstart /
// L.d_pcnt.  `sk` carries the partition geometry (and the pass's shard).  Returns the sample's k-mer bound in *kocc_up_out.
// *gather (in: allowed, out: used): the records stay in L.d_skm_a, every chunk of a level-1 bucket ordered by partition in place
// (k_skm_chunksort, table L.d_ctab / L.d_cbase) and the count kernels gather a partition's pieces -- unless a bucket could hold more
// chunks than their piece tables take, then the exact split as before.
static int skm_scan_split(simka_ctx *ctx, simka_ctx::Lane &L, uint32_t sample, const SimkaScanArgs &a_in, const SimkaSkmCfg &sk, bool exact, uint32_t pass,
                          uint64_t *kocc_up_out, bool *gather) {
    const hipStream_t st = L.stream;
    int rc;
    SimkaScanArgs a = a_in;
    a.tile_r0 = nullptr;
    const uint32_t B1 = 1u << sk.l1;
    const uint32_t ntiles = (uint32_t)((a.nb_bases + SKM_STRIDE - 1) / SKM_STRIDE);
    if (!a.fixed_len && a.nb_reads && ntiles) {
        rc = ensure_cap(ctx, &L.d_tile_r0, &L.tile_r0_cap, (uint64_t)ntiles + 2); if (rc) return rc;
        SIMKA_LAUNCH(k_tile_reads, dim3((ntiles + 1 + 255) / 256), dim3(256), 0, st, a.offsets, a.nb_reads, a.nb_bases, ntiles, (uint64_t)SKM_STRIDE, (uint64_t)(sk.d ? sk.d + 1u : 0u), L.d_tile_r0);
        a.tile_r0 = L.d_tile_r0;
    }
    uint32_t *flag = ctx->d_l1_ovf + sample;
    const uint64_t kocc_up = a.fixed_len ? (a.fixed_len >= sk.k ? a.nb_reads * (uint64_t)(a.fixed_len - sk.k + 1) : 0) : a.nb_bases;
    *kocc_up_out = kocc_up;
    // records staged per tile: what a tile yields at the expected run length (W + 1) / 2, + 35 %
    const uint32_t caprec = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(1280, (uint64_t)(SKM_STRIDE / (std::min<double>((sk.W + 1) / 2.0, sk.nmax)) * 1.35) / 128 * 128 + 128));
    const int wi = skm_w_index(sk.W);
    const bool fixed = a.fixed_len != 0;
    // LDS region R of the scan kernel: the m-mer hashes first, then staged records + the list of run starts (6 bytes each, as many
    // as records fit)
    const uint32_t rbytes = (uint32_t)std::max<size_t>((size_t)16 * SKM_NT * 4, (size_t)caprec * 22 + 16);
    auto scan_lcap = [&](bool hist) { return (uint32_t)((rbytes - (hist ? 0 : (size_t)caprec * 16) - 16) / 8); };      // (8 bytes per run start)
    auto scan_lds = [&](bool) {
        // 40 304 bytes at W = 16, fixed-length reads: 32 LDS granules of 1280 bytes, FOUR blocks per CU -- which pays for a launch of a
        // few waves of tiles (c5_50, 1845 tiles: 21.1 -> 18.8 ms per 500 samples) and costs a long one 5 % (C3, 184 000 tiles: 28.5 ->
        // 30.0 ms per 12 samples): a long launch asks for a 33rd granule and runs three blocks per CU
        size_t b = (size_t)SKM_SCAN_HEAD + rbytes + (SKM_TILE / 16 + 8) * 4 + (SKM_BLOCK + 4) * 4 + SKM_MAXB1 * 4 * 2 + SKM_MAXB1 * 4 + (fixed ? 0 : SKM_RTAB * 4);
        if (ntiles > (uint32_t)ctx->num_cus * 32u && b <= 32 * 1280) b = 32 * 1280 + 16;
        if (simka_exp_knob("SIMKA_SCAN_BPC")) { const size_t want = (size_t)(128 / std::max(1, atoi(simka_exp_knob("SIMKA_SCAN_BPC")))) * 1280; if (b < want - 1279) b = want - 1279; }      // experiments: blocks per CU by LDS padding
        return b;
    };
    static const bool no_gather = simka_test_knob("SIMKA_SKM_SPLIT") != nullptr;      // tests: the exact split instead of chunk sort + gather
    bool use_gather = *gather && !no_gather && L.d_cbase;
    // Fill cursors per level-1 bucket (k_skm_scan): every tile of the launch bumps every bucket's cursor with a returning atomic, and one
    // word serves one such atomic per ~11 ns -- a launch of T tiles cannot end before T x 11 ns, which is what a C3 sample's 184 000
    // tiles ran at.  A launch that long gives every bucket 2^lsub cursors (the tile's number picks one); they share the bucket's region
    // chunk by chunk.  Only where the chunk sort + gather take the records (the exact split wants a bucket contiguous) and the buckets
    // are capacity-sized (the exact redo of an overflowed sample is the rare path); SIMKA_SCAN_SUB=n forces 2^n (tests: small inputs).
    static const char *sub_env = simka_test_knob("SIMKA_SCAN_SUB");
    uint32_t lsub = 0;
    auto layout = [&](uint32_t mode, ull capb) {
        launch_timed(ctx, KID_LAYOUT, [&] {
            SIMKA_LAUNCH(k_skm_layout, dim3(1), dim3(SKM_MAXB1), 0, st, L.d_b1_count, L.d_b1_start, L.d_b1_end, L.d_b1_cursor, B1, mode, capb,
                               ctx->d_arena_cursor, ctx->d_sample_base + sample, pass == 0 ? 1u : 0u, (const uint32_t *)flag, L.d_redo_count,
                               use_gather ? L.d_cbase : (uint32_t *)nullptr, (uint32_t)SKM_CS_CHUNK, lsub, L.d_b1_sub);
        }, st);
    };
    ull scan_capb = 0;
    auto scan = [&](bool hist, const ull *limit) {
        launch_timed(ctx, hist ? KID_SCAN_HIST : KID_SKM_SCAN, [&] {
            SIMKA_LAUNCH(skm_scan_kernel(wi, fixed, hist), dim3(ntiles), dim3(SKM_BLOCK), scan_lds(hist), st, a, sk, L.d_b1_count, L.d_b1_cursor, L.d_skm_a, limit,
                               hist ? (uint32_t *)nullptr : flag, caprec, rbytes, scan_lcap(hist), use_gather ? (uint32_t *)nullptr : L.d_skm_p, hist ? 0u : lsub, scan_capb);
        }, st);
    };
    const uint32_t cstride = ((1u << sk.l2) + 2u + 1u) & ~1u;      // u16 entries per row of the chunk table (even: rows are 4-byte aligned)
    uint64_t rec_cap;
    if (!exact) {
        // expected records: one per (W + 1) / 2 k-mers (or per nmax, if a record takes fewer); a bucket gets its share + 12 % + slack.
        // An overflow flags the sample.
        const uint64_t est = (uint64_t)((double)kocc_up / std::max(1.0, std::min<double>((sk.W + 1) / 2.0, sk.nmax)) * 1.30 / sk.shard_count);
        uint64_t capb = est / B1 + est / B1 / 8 + 4096;
        if ((capb + SKM_CS_CHUNK - 1) / SKM_CS_CHUNK > SKM_G_MAXCH) use_gather = false;
        if (use_gather && L.d_b1_sub) {
            // Measured (12 C3 samples, ms): 1 / 2 / 4 / 8 cursors: 30.7 / 25.9 / 25.7 / 26.8 -- the cursors ran just past their limit, two
            // lift it, and every further one costs a partial chunk per bucket (C2, 0.8 chunks per bucket: 8 cursors = the chunk sort +45 %).
            // c3_10 (100 x 1M x 150 bp, 7 chunks per bucket): two cursors 27.7 -> 22.7 ms per 100 samples, the chunk sort + 0.2; C2 (k = 21): neutral.
            // Two cursors where a bucket is sized for >= 4 chunks, one otherwise (small inputs: a second partial chunk per bucket for nothing).
            lsub = sub_env ? (uint32_t)std::min(std::max(atoi(sub_env), 0), SKM_MAXSUB) : (est / B1 >= 4ull * SKM_CS_CHUNK ? 1u : 0u);
            while (lsub && ((capb + (((uint64_t)SKM_CS_CHUNK << lsub) - 1)) / ((uint64_t)SKM_CS_CHUNK << lsub) << lsub) > SKM_G_MAXCH) lsub--;      // (the rounded region must stay within the gather's chunk tables)
            if (lsub) capb = (capb + (((uint64_t)SKM_CS_CHUNK << lsub) - 1)) / ((uint64_t)SKM_CS_CHUNK << lsub) * ((uint64_t)SKM_CS_CHUNK << lsub);
        }
        scan_capb = capb;
        rec_cap = capb * B1;
        rc = ensure_cap(ctx, &L.d_skm_a, &L.skm_a_cap, rec_cap); if (rc) return rc;
        if (!use_gather) {
            rc = ensure_cap(ctx, &L.d_skm_b, &L.skm_b_cap, rec_cap); if (rc) return rc;
            rc = ensure_cap(ctx, &L.d_skm_p, &L.skm_p_cap, rec_cap); if (rc) return rc;
        }
        layout(1, capb);
        scan(false, (const ull *)L.d_b1_end);
        layout(2, capb);
    } else {
        HIPCHK(hipMemsetAsync(L.d_b1_count, 0, (size_t)(B1 + 1) * 8, st));
        scan(true, nullptr);
        std::vector<ull> cnt(B1);
        HIPCHK(hipMemcpyAsync(cnt.data(), L.d_b1_count, (size_t)B1 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        rec_cap = 16;
        for (ull c : cnt) { rec_cap += c; if ((c + SKM_CS_CHUNK - 1) / SKM_CS_CHUNK > SKM_G_MAXCH) use_gather = false; }
        if (rec_cap >= 0xffffffffull) return ctx->fail(SIMKA_ERR_OVERFLOW, "a sample yields more than 2^32 super-k-mer records in one pass");
        rc = ensure_cap(ctx, &L.d_skm_a, &L.skm_a_cap, rec_cap); if (rc) return rc;
        if (!use_gather) {
            rc = ensure_cap(ctx, &L.d_skm_b, &L.skm_b_cap, rec_cap); if (rc) return rc;
            rc = ensure_cap(ctx, &L.d_skm_p, &L.skm_p_cap, rec_cap); if (rc) return rc;
        }
        layout(0, 0);
        scan(false, (const ull *)L.d_b1_end);
    }
    if (rec_cap >= 0xffffffffull) return ctx->fail(SIMKA_ERR_OVERFLOW, "a sample needs more than 2^32 super-k-mer record slots in one pass");
    if (simka_exp_knob("SIMKA_DEBUG_MERGE")) {
        std::vector<ull> cnt(B1);
        HIPCHK(hipMemcpyAsync(cnt.data(), L.d_b1_count, (size_t)B1 * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
        ull mx = 0, sum = 0; for (ull c : cnt) { mx = std::max(mx, c); sum += c; }
        fprintf(stderr, "level-1 buckets: %u, records %llu, largest bucket %.1f %% above the mean\n", B1, sum, 100.0 * ((double)mx * B1 / std::max<ull>(1, sum) - 1.0));
    }
    *gather = use_gather;
    if (use_gather) ctx->used_gather = true;
    if (use_gather) {
        // every bucket's records in chunks of SKM_CS_CHUNK: at most rec_cap / chunk + one partial chunk per bucket
        const uint64_t nch_max = rec_cap / SKM_CS_CHUNK + B1 + 1;
        rc = ensure_cap(ctx, &L.d_ctab, &L.ctab_cap, nch_max * cstride); if (rc) return rc;
        launch_timed(ctx, KID_SKM_SPLIT, [&] {
            const size_t lds_cs = (((size_t)cstride * 2 + 15) & ~(size_t)15) + 64 + (size_t)SKM_CS_CHUNK * 16;
            SIMKA_LAUNCH(k_skm_chunksort, dim3((uint32_t)nch_max), dim3(SKM_CS_BLOCK), lds_cs, st, L.d_skm_a, (const ull *)L.d_b1_start, (const ull *)L.d_b1_count,
                               (const uint32_t *)L.d_cbase, sk, L.d_ctab, cstride, (const uint32_t *)flag, lsub, (const ull *)L.d_b1_sub);
        }, st);
        HIPCHK(hipGetLastError());
        return SIMKA_OK;
    }
    launch_timed(ctx, KID_SKM_SPLIT, [&] {
        const size_t lds_split = ((size_t)1 << sk.l2) * 4 + 64 + ((size_t)1 << sk.l2) * 2 + 48 + (size_t)SKM_SPLIT_BLOCK * SKM_SPLIT_UNROLL * 16;      // (the staging area starts 16-byte aligned behind F2 + 8 shorts)
        SIMKA_LAUNCH(k_skm_split, dim3(B1), dim3(SKM_SPLIT_BLOCK), lds_split, st, (const uint4 *)L.d_skm_a, (const uint32_t *)L.d_skm_p, (const ull *)L.d_b1_start, (const ull *)L.d_b1_count, sk,
                           L.d_skm_b, L.d_pstart, L.d_pcnt, (const uint32_t *)flag);
    }, st);
    HIPCHK(hipGetLastError());
    return SIMKA_OK;
}

static int run_count_kernels(simka_ctx *ctx, uint32_t sample, const SimkaScanArgs &a_in, bool exact, uint32_t pass = 0, uint32_t npass = 1) {
    const uint32_t N = ctx->cfg.nb_samples;
    simka_ctx::Lane &L = ctx->lanes[sample % ctx->nlanes];
    const hipStream_t st = L.stream;
    int rc;
    const SimkaScanArgs &a = a_in;
    SimkaSkmCfg sk = ctx->skm;
    if (npass > 1) skm_set_shard(sk, ctx->skm.shard_index + ctx->skm.shard_count * pass, ctx->skm.shard_count * npass);
    uint32_t *flag = ctx->d_l1_ovf + sample;
    static const bool force_exact = simka_test_knob("SIMKA_EXACT_SIZING") != nullptr;
    if (force_exact) exact = true;
    uint64_t kocc_up = 0;
    bool gather = true;
    rc = skm_scan_split(ctx, L, sample, a_in, sk, exact, pass, &kocc_up, &gather); if (rc) return rc;
    SimkaSkmSrc src;
    memset(&src, 0, sizeof src);
    if (gather) { src.recs = L.d_skm_a; src.cbase = L.d_cbase; src.b1_start = L.d_b1_start; src.ctab = L.d_ctab; src.cstride = ((1u << sk.l2) + 2u + 1u) & ~1u; }
    else { src.recs = L.d_skm_b; src.pstart = L.d_pstart; src.pcnt = L.d_pcnt; }
    if (!exact) {
        simka_ctx::Pending p; p.sample = sample; p.a = a_in; p.pass = pass; p.npass = npass;
        ctx->pending.push_back(p);
    }
    if (ctx->arena_accounted.size() != N) ctx->arena_accounted.assign(N, 0);
    if (ctx->arena_vmm && !ctx->arena_accounted[sample]) {
        ctx->arena_accounted[sample] = 1;
        // what this sample can add to the arena at most: every solid k-mer has >= abundance_min occurrences, plus the tails of the
        // blocks' slab reservations
        const uint64_t kocc_b = a.fixed_len ? (a.fixed_len >= sk.k ? a.nb_reads * (uint64_t)(a.fixed_len - sk.k + 1) : 0) : a.nb_bases;
        // (slab_take loses at most a third on top of the records it places; every block of the two count kernels leaves one open slab)
        const uint64_t solid_up = kocc_b / std::max<uint32_t>(1u, ctx->cfg.abundance_min);
        const uint64_t bound = solid_up + solid_up / 3 + (uint64_t)ctx->num_cus * 6 * K2_SLAB;
        ctx->arena_hi = std::min<uint64_t>(ctx->arena_cap, ctx->arena_hi + bound);
        ctx->lane_bound[sample % ctx->nlanes] = bound;
        rc = arena_ensure(ctx, ctx->arena_hi); if (rc) return rc;
    }
    simka_trace::set_arena(ctx->trace_id, sample, ctx->arena_mapped, ctx->arena_hi, ctx->arena_cap);
    SimkaCountOut o;
    o.arena_cursor = ctx->d_arena_cursor; o.sample_base = ctx->d_sample_base + sample; o.arena_cap = ctx->arena_vmm ? std::min(ctx->arena_mapped, ctx->arena_cap) : ctx->arena_cap;
    o.solid_keys = ctx->d_solid_keys; o.solid_counts = ctx->d_solid_counts;
    o.foff = ctx->d_foff + (uint64_t)sample * ctx->nparts; o.fcnt = ctx->d_fcnt + (uint64_t)sample * ctx->nparts;
    o.totals = (ull *)ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, 0); o.sample = sample; o.nb_samples = N; o.err = ctx->d_err;
    o.phase = nullptr; o.pad_ = 0;
    o.seg_rows = ctx->seg_all ? (uint16_t *)ctx->d_seg_rows + (size_t)sample * SIMKA_SEG_BLOCKS : nullptr; o.seg_abs = ctx->seg_all ? ctx->d_seg_abs + sample : nullptr;
    // slabs several partitions long (what is left of a slab when the next partition does not fit is lost), but never more than a
    // small share of the arena per block
    o.slab = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(K2_SLAB, std::max<uint64_t>(256, kocc_up / 10 / ((uint64_t)ctx->num_cus * 4 * 8))),
                                          std::max<uint64_t>(64, ctx->arena_cap / ((uint64_t)ctx->num_cus * 4 * 8)));
    o.hist = ctx->d_hist; o.ovf_list = ctx->d_ovf_list; o.ovf_cursor = ctx->d_ovf_cursor; o.ovf_cap = ctx->ovf_cap;
    ull *kocc = (ull *)ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, SIMKA_TOT_KOCC) + sample;
#ifdef SIMKA_PHASE_PROF
    {   // debug build: per-phase wall_clock64 ticks of thread 0 of every k_skm_count_fast block, printed per sample
        static ull *d_phase = nullptr;
        const size_t phase_bytes = (16 + 2 * 2048) * 8;
        if (!d_phase) { HIPCHK(hipMalloc(&d_phase, phase_bytes)); HIPCHK(hipMemset(d_phase, 0, phase_bytes)); }
        else {
            HIPCHK(hipDeviceSynchronize());
            ull h[16]; HIPCHK(hipMemcpy(h, d_phase, 128, hipMemcpyDeviceToHost));
            {   // when the blocks started and ended, relative to the first start, as deciles
                std::vector<ull> se(2 * 2048);
                HIPCHK(hipMemcpy(se.data(), d_phase + 16, se.size() * 8, hipMemcpyDeviceToHost));
                std::vector<ull> st_, en_;
                for (size_t b_ = 0; b_ < 2048; b_++) if (se[2 * b_ + 1]) { st_.push_back(se[2 * b_]); en_.push_back(se[2 * b_ + 1]); }
                if (!st_.empty()) {
                    const ull t0_ = *std::min_element(st_.begin(), st_.end());
                    std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end());
                    fprintf(stderr, "k_skm_count_fast blocks: %zu; starts (ticks after the first) p50 %llu p90 %llu max %llu; ends p10 %llu p25 %llu p50 %llu p75 %llu p90 %llu max %llu\n", st_.size(),
                            st_[st_.size() / 2] - t0_, st_[st_.size() * 9 / 10] - t0_, st_.back() - t0_, en_[en_.size() / 10] - t0_, en_[en_.size() / 4] - t0_, en_[en_.size() / 2] - t0_,
                            en_[en_.size() * 3 / 4] - t0_, en_[en_.size() * 9 / 10] - t0_, en_.back() - t0_);
                }
            }
            HIPCHK(hipMemset(d_phase, 0, phase_bytes));
            if (h[12]) fprintf(stderr, "k_skm_count_fast counters: per wave and partition %.1f k-mers, %.1f queue entries left after the last batch, %.2f final drain passes\n",
                               (double)h[8] / h[12], (double)h[14] / h[12], (double)h[13] / h[12]);
            if (h[11]) fprintf(stderr, "k_skm_count_fast blocks: busy time of the slowest block %.0f ticks, mean %.0f (%.1f %% above the mean)\n", (double)h[9], (double)h[10] / h[11], 100.0 * ((double)h[9] * h[11] / h[10] - 1.0));
            ull t_ = 0; for (int i_ = 0; i_ < 8; i_++) t_ += h[i_];
            if (t_) fprintf(stderr, "k_skm_count_fast phases %%: top %.1f map %.1f insert %.1f sync %.1f summary %.1f scan %.1f slab %.1f stores %.1f  (ticks/block %.0f)\n",
                    100.0 * h[0] / t_, 100.0 * h[1] / t_, 100.0 * h[2] / t_, 100.0 * h[3] / t_, 100.0 * h[4] / t_, 100.0 * h[5] / t_, 100.0 * h[6] / t_, 100.0 * h[7] / t_, (double)t_ / (ctx->num_cus * 2));
        }
        o.phase = d_phase;
    }
#endif
    const size_t hist_lds = ctx->d_hist ? (size_t)SIMKA_HIST_MAX * 4 : 0;
    const size_t lds_fast = (size_t)SKM_FAST_HEAD + (size_t)SKM_FAST_TS * 12 + (size_t)SKM_FAST_BLOCK * 16 + (ctx->d_hist ? (size_t)SKM_FAST_HBINS * 4 : 0) + (size_t)(SKM_FAST_BLOCK / 64) * SKM_FAST_WREG +
                            (gather ? (size_t)SKM_G_BYTES(SKM_FAST_BLOCK / 64) : 64);
    const size_t lds_count = (size_t)SIMKA_LDS_HEAD + (size_t)SKM_CNT_TS * 12 + (size_t)SKM_CNT_BATCH * 16 + (size_t)SKM_CNT_BLOCK * 4 + hist_lds + (size_t)SKM_CNT_BATCH * sk.nmax * 2 + 64 +
                             (gather ? (size_t)(SKM_G_MAXCH * 10 + 32) : 0);
    static const bool general_only = simka_test_knob("SIMKA_SKM_GENERAL") != nullptr;      // tests: every partition through the general kernel
    if (!general_only)
        launch_timed(ctx, KID_SKM_COUNT, [&] {
            static const uint32_t bpc_env = simka_exp_knob("SIMKA_SKM_BPC") ? (uint32_t)atoi(simka_exp_knob("SIMKA_SKM_BPC")) : 0u;     // experiments
            // (the dispatcher hands out LDS in granules of 1280 bytes, 128 per CU: scripts/ubench/lds_occupancy.hip)
            const uint32_t bpc = bpc_env ? bpc_env : (uint32_t)std::max<size_t>(1, std::min<size_t>(4, 128 / ((lds_fast + 1279) / 1280)));
            SIMKA_LAUNCH(gather ? k_skm_count_fast<true> : k_skm_count_fast<false>, dim3((uint32_t)std::min<uint64_t>(ctx->nparts, (uint64_t)ctx->num_cus * bpc)), dim3(SKM_FAST_BLOCK), lds_fast, st,
                               src, sk, ctx->key, ctx->cfg.abundance_min, ctx->cfg.abundance_max, o, (const uint32_t *)flag, kocc,
                               L.d_redo_list, L.d_redo_count);
        }, st);
    launch_timed(ctx, KID_COUNT, [&] {
        const uint32_t grid = general_only ? (uint32_t)std::min<uint64_t>(ctx->nparts, (uint64_t)ctx->num_cus * 2) : (uint32_t)ctx->num_cus;
        SIMKA_LAUNCH(gather ? k_skm_count<true> : k_skm_count<false>, dim3(grid), dim3(SKM_CNT_BLOCK), lds_count, st,
                           src, sk, ctx->key, ctx->cfg.abundance_min, ctx->cfg.abundance_max, o, (const uint32_t *)flag, kocc,
                           general_only ? (const uint32_t *)nullptr : (const uint32_t *)L.d_redo_list, general_only ? (const ull *)nullptr : (const ull *)L.d_redo_count);
    }, st);
    HIPCHK(hipGetLastError());
    return SIMKA_OK;
}

// read the overflow flags of the samples enqueued in capacity mode; redo the flagged ones exactly (their later kernels
// skipped themselves, so no state was touched).  Synchronises every lane -- or, with lane >= 0, that lane only: its pending
// samples are settled (their reads sit in the lane's staging buffer, which the caller is about to overwrite), the other lane
// keeps running.
static int resolve_pending(simka_ctx *ctx, int lane) {
    for (uint32_t li = 0; li < ctx->nlanes; li++)
        if (ctx->lanes[li].stream && (lane < 0 || (uint32_t)lane == li)) HIPCHK(hipStreamSynchronize(ctx->lanes[li].stream));
    if (ctx->arena_vmm && ctx->geometry_ready) {      // what the finished samples really took replaces their worst-case bounds
        ull cur = 0;
        HIPCHK(hipMemcpy(&cur, ctx->d_arena_cursor, 8, hipMemcpyDeviceToHost));
        uint64_t hi = cur;
        for (uint32_t li = 0; li < ctx->nlanes; li++) { if (lane < 0 || (uint32_t)lane == li) ctx->lane_bound[li] = 0; hi += ctx->lane_bound[li]; }
        ctx->arena_hi = std::min(ctx->arena_hi, hi);
    }
    if (ctx->pending.empty()) return SIMKA_OK;
    const uint32_t N = ctx->cfg.nb_samples;
    std::vector<simka_ctx::Pending> todo, keep;
    for (auto &p : ctx->pending) ((lane < 0 || p.sample % ctx->nlanes == (uint32_t)lane) ? todo : keep).push_back(p);
    if (todo.empty()) return SIMKA_OK;
    std::vector<uint32_t> flags(N);
    hipStream_t cs = lane < 0 ? ctx->stream : ctx->lanes[lane].stream;
    HIPCHK(hipMemcpyAsync(flags.data(), ctx->d_l1_ovf, (size_t)N * 4, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipStreamSynchronize(cs));
    ctx->pending.swap(keep);
    for (auto &p : todo) {
        if (!flags[p.sample]) continue;
        ctx->nb_exact_fallbacks++;
        if (ctx->arena_accounted.size() == N) ctx->arena_accounted[p.sample] = 0;      // its bound left arena_hi above: the redo adds it again
        HIPCHK(hipMemsetAsync(ctx->d_l1_ovf + p.sample, 0, 4, cs));
        HIPCHK(hipStreamSynchronize(cs));
        int rc = run_count_kernels(ctx, p.sample, p.a, true, p.pass, p.npass);
        if (rc) return rc;
    }
    for (uint32_t li = 0; li < ctx->nlanes; li++)
        if (lane < 0 || (uint32_t)lane == li) HIPCHK(hipStreamSynchronize(ctx->lanes[li].stream));
    return SIMKA_OK;
}

// ---- 32 <= k <= 63: sort-based path (simka_wide.hip) --------------------------------------------
static int wide_fail(simka_ctx *ctx, int wrc) {
    const int rc = wrc == SIMKA_WIDE_ERR_NOMEM ? SIMKA_ERR_NOMEM : wrc == SIMKA_WIDE_ERR_LIMIT ? SIMKA_ERR_UNSUPPORTED : SIMKA_ERR_HIP;
    return ctx->fail(rc, "%s", simka_wide_error(ctx->wide));
}

// 32 <= k <= 51: scan + split + k_skm_count_wide on lane 0, the solid records sorted into the wide arena.  *done = false: the sample
// does not fit this path (a partition beyond the LDS table, output arrays too small) -- nothing was kept, the caller sorts instead.
static int wide_hash_count(simka_ctx *ctx, uint32_t sample, const void *d_packed, const void *d_offsets, const simka_reads *r, unsigned long long tot[SIMKA_NB_TOTALS], bool *done) {
    *done = false;
    const uint32_t N = ctx->cfg.nb_samples;
    simka_ctx::Lane &L = ctx->lanes[0];
    if (!L.stream) L.stream = ctx->stream;            // (the wide path is synchronous: everything on the context's stream)
    const hipStream_t st = L.stream;
    int rc;
    SimkaScanArgs a;
    a.nb_bases = r->nb_bases; a.nb_words = (r->nb_bases + 31) / 32; a.nb_reads = r->nb_reads; a.fixed_len = r->fixed_len;
    a.packed = (const uint64_t *)d_packed; a.offsets = (const uint64_t *)d_offsets; a.tile_r0 = nullptr;
    SimkaSkmCfg sk = ctx->skm;
    const uint64_t kocc_b = a.fixed_len ? (a.fixed_len >= sk.k ? a.nb_reads * (uint64_t)(a.fixed_len - sk.k + 1) : 0) : a.nb_bases;
    if (kocc_b == 0) { *done = true; for (int i = 0; i < SIMKA_NB_TOTALS; i++) tot[i] = 0; return simka_wide_adopt(ctx->wide, sample, nullptr, nullptr, nullptr, 0, 0) ? wide_fail(ctx, 1) : SIMKA_OK; }
    // the partition count is the sample's own (the arena holds spectra, not partitions): ~100-190 k-mer occurrences per
    // partition, three eighths of a wave's table (k_skm_count_wide_fast) even if all of them are distinct
    const uint32_t per_part = simka_test_knob("SIMKA_WIDE_PER_PART") ? (uint32_t)std::max(1, atoi(simka_test_knob("SIMKA_WIDE_PER_PART"))) : 192u;      // (tests: partitions beyond the tables)
    sk.pb = std::min<uint32_t>(20u, ceil_log2_u64((kocc_b + per_part - 1) / per_part));
    sk.l1 = std::min<uint32_t>(sk.pb, 8u); sk.l2 = sk.pb - sk.l1;
    const uint32_t B1 = 1u << sk.l1;
    const uint64_t nparts = (uint64_t)1 << sk.pb;
    if (!L.d_b1_count) {
        HIPCHK(dev_alloc(&L.d_b1_count, SKM_MAXB1 + 1)); HIPCHK(dev_alloc(&L.d_b1_start, SKM_MAXB1 + 1)); HIPCHK(dev_alloc(&L.d_b1_end, SKM_MAXB1 + 1));
        HIPCHK(dev_alloc(&L.d_b1_cursor, (uint64_t)(SKM_MAXB1 + 1) * SKM_CSTRIDE));
        HIPCHK(dev_alloc(&L.d_redo_count, 2)); HIPCHK(dev_alloc(&L.d_redo_list, ((uint64_t)1 << 20) + 1));
        HIPCHK(dev_alloc(&L.d_pstart, ((uint64_t)1 << 20) + 1)); HIPCHK(dev_alloc(&L.d_pcnt, ((uint64_t)1 << 20) + 1));
        HIPCHK(dev_alloc(&ctx->d_l1_ovf, N + 1));
        HIPCHK(hipMemsetAsync(ctx->d_l1_ovf, 0, (size_t)(N + 1) * 4, st));
        HIPCHK(dev_alloc(&ctx->d_wh_cursor, 8));
    }
    (void)B1;
    // capacity-sized buckets first; an overflow (the later kernels skipped themselves) is redone with exact sizes
    uint64_t kocc_up = 0;
    // (every solid k-mer has >= abundance_min occurrences; + the unused ends of the waves' slabs)
    const uint64_t solid_up = kocc_b / std::max<uint32_t>(1u, ctx->cfg.abundance_min);
    const uint64_t out_cap = solid_up + solid_up / 8 + (uint64_t)ctx->num_cus * 16 * SKM_WF_SLAB + 16;
    rc = ensure_cap(ctx, &ctx->d_wh_hi, &ctx->wh_hi_cap, out_cap); if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_wh_lo, &ctx->wh_lo_cap, out_cap); if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_wh_cnt, &ctx->wh_cnt_cap, out_cap); if (rc) return rc;
    static const bool force_exact = simka_test_knob("SIMKA_EXACT_SIZING") != nullptr;
    ull ovf_before[2] = { 0, 0 };
    if (ctx->d_ovf_cursor) { HIPCHK(hipMemcpyAsync(ovf_before, ctx->d_ovf_cursor, 16, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    for (int attempt = force_exact ? 1 : 0; attempt < 2; attempt++) {
        HIPCHK(hipMemsetAsync(ctx->d_wh_cursor, 0, 64, st));
        bool gather_ = false;          // (the two-word count kernels read contiguous partitions)
        rc = skm_scan_split(ctx, L, sample, a, sk, attempt == 1, 0, &kocc_up, &gather_); if (rc) return rc;
        SimkaWideOut wo;
        wo.hi = ctx->d_wh_hi; wo.lo = ctx->d_wh_lo; wo.cnt = ctx->d_wh_cnt; wo.cursor = ctx->d_wh_cursor; wo.cap = out_cap;
        wo.shard_index = ctx->cfg.shard_index; wo.shard_count = ctx->cfg.shard_count;
        SimkaCountOut o;
        memset(&o, 0, sizeof o);
        o.sample = sample; o.nb_samples = N; o.err = ctx->d_err;
        o.hist = ctx->d_hist; o.ovf_list = ctx->d_ovf_list; o.ovf_cursor = ctx->d_ovf_cursor; o.ovf_cap = ctx->ovf_cap;
        const size_t hist_lds = ctx->d_hist ? (size_t)SIMKA_HIST_MAX * 4 : 0;
        const size_t lds_wide = (size_t)SIMKA_LDS_HEAD + (size_t)SKM_WIDE_TS * 20 + (size_t)SKM_CNT_BATCH * 16 + (size_t)SKM_CNT_BLOCK * 4 + hist_lds + (size_t)SKM_CNT_BATCH * sk.nmax * 2 + 64;
#ifdef SIMKA_PHASE_PROF
        {   // debug build: per-phase wall_clock64 ticks of thread 0 of every k_skm_count_wide_fast block, printed per sample
            static ull *d_phase = nullptr;
            if (!d_phase) { HIPCHK(hipMalloc(&d_phase, 128)); HIPCHK(hipMemset(d_phase, 0, 128)); }
            else {
                HIPCHK(hipDeviceSynchronize());
                ull h[16]; HIPCHK(hipMemcpy(h, d_phase, 128, hipMemcpyDeviceToHost)); HIPCHK(hipMemset(d_phase, 0, 128));
                ull t_ = 0; for (int i_ = 0; i_ < 8; i_++) t_ += h[i_];
                if (t_) fprintf(stderr, "k_skm_count_wide_fast phases %%: prefetch+summary-tail %.1f load+scan %.1f map %.1f canonical %.1f probe %.1f - %.1f rows %.1f records %.1f\n",
                        100.0 * h[0] / t_, 100.0 * h[1] / t_, 100.0 * h[2] / t_, 100.0 * h[3] / t_, 100.0 * h[4] / t_, 100.0 * h[5] / t_, 100.0 * h[6] / t_, 100.0 * h[7] / t_);
            }
            o.phase = d_phase;
        }
#endif
        const bool general_only = simka_test_knob("SIMKA_SKM_GENERAL") != nullptr;      // tests: every partition through the block kernel
        if (!general_only)
            launch_timed(ctx, KID_SKM_COUNT, [&] {
                const size_t lds_wf = (size_t)SIMKA_LDS_HEAD + hist_lds + (size_t)(SKM_WF_BLOCK / 64) * skm_wf_wave_bytes(sk.nmax);
                const uint32_t bpc = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, 128 / ((lds_wf + 1279) / 1280)));      // (LDS granules of 1280 bytes, 128 per CU)
                SIMKA_LAUNCH(k_skm_count_wide_fast, dim3((uint32_t)std::min<uint64_t>((nparts + 3) / 4, (uint64_t)ctx->num_cus * bpc)), dim3(SKM_WF_BLOCK), lds_wf, st, (const uint4 *)L.d_skm_b,
                                   (const uint32_t *)L.d_pstart, (const uint32_t *)L.d_pcnt, sk, ctx->cfg.abundance_min, ctx->cfg.abundance_max, wo, o, (const uint32_t *)(ctx->d_l1_ovf + sample),
                                   L.d_redo_list, L.d_redo_count);
            }, st);
        launch_timed(ctx, KID_COUNT, [&] {
            SIMKA_LAUNCH(k_skm_count_wide, dim3((uint32_t)std::min<uint64_t>(nparts, (uint64_t)ctx->num_cus)), dim3(SKM_CNT_BLOCK), lds_wide, st, (const uint4 *)L.d_skm_b,
                               (const uint32_t *)L.d_pstart, (const uint32_t *)L.d_pcnt, sk, ctx->cfg.abundance_min, ctx->cfg.abundance_max, wo, o, (const uint32_t *)(ctx->d_l1_ovf + sample),
                               general_only ? (const uint32_t *)nullptr : (const uint32_t *)L.d_redo_list, general_only ? (const ull *)nullptr : (const ull *)L.d_redo_count);
        }, st);
        uint32_t flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, ctx->d_l1_ovf + sample, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!flag) break;
        if (attempt == 1) return ctx->fail(SIMKA_ERR_OVERFLOW, "the exactly sized level-1 buckets of sample %u overflowed", sample);
        ctx->nb_exact_fallbacks++;
        HIPCHK(hipMemsetAsync(ctx->d_l1_ovf + sample, 0, 4, st));
    }
    ull cur[8];
    HIPCHK(hipMemcpyAsync(cur, ctx->d_wh_cursor, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (cur[6]) {      // nothing was kept; the abundance histogram of -complex-dist was touched: as before, for the sort path
        if (ctx->d_hist) HIPCHK(hipMemsetAsync(ctx->d_hist + (uint64_t)sample * SIMKA_HIST_MAX, 0, (size_t)SIMKA_HIST_MAX * 8, st));
        if (ctx->d_ovf_cursor) HIPCHK(hipMemcpyAsync(ctx->d_ovf_cursor, ovf_before, 16, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return SIMKA_OK;
    }
    tot[SIMKA_TOT_DALL] = cur[1]; tot[SIMKA_TOT_D] = cur[2]; tot[SIMKA_TOT_N] = cur[3]; tot[SIMKA_TOT_Q] = cur[4]; tot[SIMKA_TOT_KOCC] = cur[5];
    const int wrc = simka_wide_adopt(ctx->wide, sample, ctx->d_wh_hi, ctx->d_wh_lo, ctx->d_wh_cnt, cur[0], cur[2]);
    if (wrc) return wide_fail(ctx, wrc);
    *done = true;
    return SIMKA_OK;
}

static int wide_count_sample(simka_ctx *ctx, uint32_t sample, const simka_reads *r) {
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    ctx->nb_reads[sample] = r->nb_input_reads ? r->nb_input_reads : r->nb_reads;
    ctx->counted[sample] = 1;
    if (r->nb_bases == 0) return SIMKA_OK;
    const uint64_t nb_words = (r->nb_bases + 31) / 32;
    const void *d_packed = r->packed, *d_offsets = r->offsets;
    int rc;
    if (!r->on_device) {
        rc = ensure_cap(ctx, &ctx->d_reads[0], &ctx->reads_cap[0], nb_words + 2); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ctx->d_reads[0], r->packed, nb_words * 8, hipMemcpyHostToDevice, ctx->stream));
        d_packed = ctx->d_reads[0]; d_offsets = nullptr;
        if (!r->fixed_len) {
            rc = ensure_cap(ctx, &ctx->d_offsets[0], &ctx->offsets_cap[0], r->nb_reads + 1); if (rc) return rc;
            HIPCHK(hipMemcpyAsync(ctx->d_offsets[0], r->offsets, (r->nb_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            d_offsets = ctx->d_offsets[0];
        }
    }
    unsigned long long tot[SIMKA_NB_TOTALS];
    bool hashed = false;
    if (ctx->wide_hash) { rc = wide_hash_count(ctx, sample, d_packed, d_offsets, r, tot, &hashed); if (rc) return rc; }
    (hashed ? ctx->nb_wide_hash : ctx->nb_wide_sort)++;
    const int wrc = hashed ? 0 : simka_wide_count_sample(ctx->wide, sample, d_packed, r->nb_bases, nb_words, d_offsets, r->nb_reads, r->fixed_len, ctx->cfg.abundance_min,
                                            ctx->cfg.abundance_max, tot, ctx->d_hist ? (void *)(ctx->d_hist + (uint64_t)sample * SIMKA_HIST_MAX) : nullptr,
                                            ctx->d_ovf_list, ctx->d_ovf_cursor, ctx->ovf_cap);
    if (wrc) return wide_fail(ctx, wrc);
    for (int i = 0; i < SIMKA_NB_TOTALS; i++)
        HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, i) + sample, &tot[i], 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}

// ---- count side ---------------------------------------------------------------------------
SIMKA_EXPORT int simka_count_sample(simka_ctx *ctx, uint32_t sample, const simka_reads *r) {
    if (!ctx || !r) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples;
    if (sample >= N) return ctx->fail(SIMKA_ERR_INVALID, "simka_count_sample: sample index %u out of range", sample);
    if (ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "simka_count_sample: sample %u was already counted", sample);
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_count_sample: merge already ran");
    if (r->nb_bases && !r->packed) return ctx->fail(SIMKA_ERR_INVALID, "simka_count_sample: packed is NULL");
    if (!r->fixed_len && r->nb_bases && !r->offsets) return ctx->fail(SIMKA_ERR_INVALID, "simka_count_sample: offsets required when fixed_len==0");
    if (r->fixed_len && r->nb_bases != r->nb_reads * (uint64_t)r->fixed_len) return ctx->fail(SIMKA_ERR_INVALID, "simka_count_sample: nb_bases != nb_reads*fixed_len");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc;
    if (ctx->wide) return wide_count_sample(ctx, sample, r);
    if (!ctx->geometry_ready) { rc = setup_geometry(ctx, std::max<uint64_t>(r->nb_bases, 1)); if (rc) return rc; }
    ctx->nb_reads[sample] = r->nb_input_reads ? r->nb_input_reads : r->nb_reads;
    ctx->counted[sample] = 1;
    if (r->nb_bases == 0) {   // empty sample: tables stay zero, sample_base = cursor
        HIPCHK(hipMemcpyAsync(ctx->d_sample_base + sample, ctx->d_arena_cursor, 8, hipMemcpyDeviceToDevice, ctx->stream));
        return SIMKA_OK;
    }
    const uint64_t nb_words = (r->nb_bases + 31) / 32;
    SimkaScanArgs a;
    a.nb_bases = r->nb_bases; a.nb_words = nb_words; a.nb_reads = r->nb_reads; a.fixed_len = r->fixed_len;
    if (r->on_device) { a.packed = r->packed; a.offsets = r->offsets; }
    else {
        // The lane's staging buffer still holds the host-provided sample counted two calls ago: settle THAT sample (overflow flag,
        // exact redo) before it is overwritten.  The other lane -- the previous sample -- keeps running: its kernels overlap this
        // copy (pinned host memory, simka_host_alloc(): a DMA at PCIe speed; pageable memory works, through the runtime's bounce
        // buffer) and the caller's parsing of the sample after this one.
        const uint32_t li = sample % ctx->nlanes;
        rc = resolve_pending(ctx, (int)li); if (rc) return rc;
        if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        rc = ensure_cap(ctx, &ctx->d_reads[li], &ctx->reads_cap[li], nb_words + 2); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ctx->d_reads[li], r->packed, nb_words * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        a.packed = ctx->d_reads[li]; a.offsets = nullptr;
        if (!r->fixed_len) {
            rc = ensure_cap(ctx, &ctx->d_offsets[li], &ctx->offsets_cap[li], r->nb_reads + 1); if (rc) return rc;
            HIPCHK(hipMemcpyAsync(ctx->d_offsets[li], r->offsets, (r->nb_reads + 1) * 8, hipMemcpyHostToDevice, ctx->copy_stream));
            a.offsets = ctx->d_offsets[li];
        }
        HIPCHK(hipStreamSynchronize(ctx->copy_stream));   // the host buffers may be reused by the caller right away; the kernels below are ordered behind the copy
    }
    // scratch per k-mer occurrence: two buffers of 16-byte super-k-mer records, ~2 B per occurrence each at the expected run
    // length (+ 30 % head room): say 6 B.  A sample whose share does not fit
    // a third of the device is counted in several passes over its reads, each keeping a subset of the level-1 buckets.
    uint32_t npass = 1;
    {
        static const char *force = simka_test_knob("SIMKA_FORCE_PASSES");          // tests
        const uint32_t max_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, ctx->nparts / std::max<uint32_t>(1, ctx->cfg.shard_count)));
        if (force) npass = (uint32_t)atoi(force);
        else {
            if (!ctx->mem_total) { size_t fr = 0, tot_ = 0; HIPCHK(hipMemGetInfo(&fr, &tot_)); ctx->mem_total = tot_; }
            const size_t tot = ctx->mem_total;
            const uint64_t kocc_up = r->fixed_len ? r->nb_reads * (uint64_t)r->fixed_len : r->nb_bases;
            const double need = (double)kocc_up * 6.0 / std::max<uint32_t>(1, ctx->cfg.shard_count);
            while (need / npass > 0.30 * (double)tot && npass < max_pass) npass *= 2;
        }
        npass = std::max<uint32_t>(1, std::min(npass, max_pass));
    }
    for (uint32_t j = 0; j < npass; j++) {
        if (j && ctx->arena_accounted.size() == N) ctx->arena_accounted[sample] = 0;      // resolve_pending() dropped the bound of the pass before
        rc = run_count_kernels(ctx, sample, a, false, j, npass);
        if (rc) return rc;
        if (npass > 1) { rc = resolve_pending(ctx); if (rc) return rc; }       // the passes share the scratch buffers
    }
    ctx->nb_counted_this_run++;
    return SIMKA_OK;
}

// ---- device-side ingest (kernels: simka_ingest.hip) -------------------------------------------------------
// grow a device buffer and KEEP its first `keep` elements
template <typename T>
static int grow_keep(simka_ctx *ctx, T **p, uint64_t *cap, uint64_t need, uint64_t keep, hipStream_t st) {
    if (*cap >= need && *p) return SIMKA_OK;
    T *q = nullptr;
    const uint64_t n = need + need / 8 + 16;
    hipError_t e = dev_alloc(&q, n);
    if (e != hipSuccess) return ctx->fail(SIMKA_ERR_NOMEM, "hipMalloc of %llu bytes failed: %s", (unsigned long long)(n * sizeof(T)), hipGetErrorString(e));
    if (*p && keep) HIPCHK(hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    if (*p) HIPCHK(hipFree(*p));
    *p = q; *cap = n;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_ingest_begin(simka_ctx *ctx, uint32_t sample) {
    if (!ctx) return SIMKA_ERR_INVALID;
    if (sample >= ctx->cfg.nb_samples) return ctx->fail(SIMKA_ERR_INVALID, "simka_ingest_begin: sample index %u out of range", sample);
    if (ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "simka_ingest_begin: sample %u was already counted", sample);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    // (the lane count is fixed with the geometry, at the first count: until then, what setup_geometry will decide)
    const uint32_t want_lanes = simka_test_knob("SIMKA_LANES") ? (uint32_t)std::max(1, atoi(simka_test_knob("SIMKA_LANES"))) : 2u;
    const uint32_t li = sample % (ctx->nlanes ? ctx->nlanes : std::min<uint32_t>(std::min<uint32_t>(want_lanes, simka_ctx::MAX_LANES), std::max<uint32_t>(1u, ctx->cfg.nb_samples)));
    if (ctx->nlanes) { int rc = resolve_pending(ctx, (int)li); if (rc) return rc; }      // the lane's staging buffers still feed the sample counted two calls ago
    if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    simka_ctx::Ingest &g = ctx->ing[li];
    g.sample = sample; g.nb_bases = 0; g.nb_frags = 0; g.nb_reads = 0; g.open = true;
    if (!g.d_tot) HIPCHK(dev_alloc(&g.d_tot, 4));
    return SIMKA_OK;
}

// one word written by a kernel: a 4-byte hipMemcpy would queue on the host-to-device DMA engine behind the 64-MiB text blocks the
// loader threads are uploading (the main thread then waits milliseconds for a few bytes: simka_ingest_text_device)
__global__ void k_store_u32(uint32_t *p, uint32_t v) { *p = v; }
__global__ void k_store_u64(ull *p, ull v) { *p = v; }

static int ingest_text_impl(simka_ctx *ctx, uint32_t sample, const char *text, uint64_t nb_bytes, int format, uint64_t *nb_reads, int *irregular, bool on_device) {
    if (!ctx || (!text && nb_bytes) || !irregular) return SIMKA_ERR_INVALID;
    simka_ctx::Ingest *gp = nullptr; uint32_t li = 0;
    for (uint32_t l = 0; l < simka_ctx::MAX_LANES; l++) if (ctx->ing[l].open && ctx->ing[l].sample == sample) { gp = &ctx->ing[l]; li = l; }
    if (!gp) return ctx->fail(SIMKA_ERR_STATE, "simka_ingest_text: simka_ingest_begin was not called for sample %u", sample);
    simka_ctx::Ingest &g = *gp;
    *irregular = 0;
    if (nb_reads) *nb_reads = 0;
    if (nb_bytes == 0) return SIMKA_OK;
    if (format != ING_FASTA && format != ING_FASTQ) return ctx->fail(SIMKA_ERR_INVALID, "simka_ingest_text: format must be 0 (FASTA) or 1 (FASTQ)");
    if (nb_bytes >= 0xfffffff0ull) { *irregular = 1; return SIMKA_OK; }        // 32-bit line offsets
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const hipStream_t st = ctx->copy_stream;
    int rc;
    // the text: copied into the lane's buffer -- or, already on the device (simka_ingest_text_device), parsed where it is
    unsigned char *const d_text = on_device ? (unsigned char *)const_cast<char *>(text) : nullptr;
    if (!on_device) {
        rc = ensure_cap(ctx, &g.d_text, &g.text_cap, nb_bytes + 64); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(g.d_text, text, nb_bytes, hipMemcpyHostToDevice, st));
    }
    const unsigned char *const txt = on_device ? d_text : g.d_text;
    // ---- the line table
    const uint64_t ntiles = (nb_bytes + ING_TILE - 1) / ING_TILE;
    rc = ensure_cap(ctx, &g.d_tmp, &g.tmp_cap, ntiles + 16 + wscan_tmp_u32(ntiles) + 16); if (rc) return rc;
    uint32_t *d_cnt = g.d_tmp;
    HIPCHK(hipMemsetAsync(g.d_tot, 0, 32, st));
    SIMKA_LAUNCH(k_ing_nl_count, dim3((uint32_t)ntiles), dim3(ING_BLOCK), 0, st, txt, nb_bytes, d_cnt);
    uint32_t last_cnt = 0, last_off = 0;
    HIPCHK(hipMemcpyAsync(&last_cnt, d_cnt + ntiles - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(wscan_u32(d_cnt, d_cnt, ntiles, g.d_tmp + ntiles + 16, st));
    HIPCHK(hipMemcpyAsync(&last_off, d_cnt + ntiles - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t nlines = (uint64_t)last_off + last_cnt + 1;
    // one block for the per-line arrays: line starts [nlines + 1], bases, fragments, prefix of the bases, scan scratch
    const uint64_t la = (nlines + 2 + 15) & ~(uint64_t)15, block_need = 4 * la + wscan_tmp_u32(nlines) + 32;
    rc = ensure_cap(ctx, &g.d_lines, &g.lines_cap, block_need); if (rc) return rc;
    g.d_lb = g.d_lines + la; g.d_lf = g.d_lb + la;
    uint32_t *d_lbo = g.d_lf + la, *d_scan = d_lbo + la;
    SIMKA_LAUNCH(k_ing_nl_fill, dim3((uint32_t)ntiles), dim3(ING_BLOCK), 0, st, txt, nb_bytes, (const uint32_t *)d_cnt, g.d_lines);
    const uint32_t sentinel = (uint32_t)nb_bytes + 1u;
    SIMKA_LAUNCH(k_store_u32, dim3(1), dim3(1), 0, st, g.d_lines + nlines, sentinel);
    // ---- per line: bases, fragments that start in it, reads; prefix sums (the counts of the last line are read before the in-place scan)
    const uint32_t lgrid = (uint32_t)((nlines + ING_BLOCK - 1) / ING_BLOCK);
    SIMKA_LAUNCH(k_ing_lines, dim3(lgrid), dim3(ING_BLOCK), 0, st, txt, (const uint32_t *)g.d_lines, (uint32_t)nlines, format, g.d_lb, g.d_lf, g.d_tot);
    uint32_t lastb = 0, lastf = 0, sumb = 0, sumf = 0;
    HIPCHK(hipMemcpyAsync(&lastf, g.d_lf + nlines - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(wscan_u32(g.d_lf, g.d_lf, nlines, d_scan, st));
    HIPCHK(hipMemcpyAsync(&sumf, g.d_lf + nlines - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&lastb, g.d_lb + nlines - 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(wscan_u32(g.d_lb, d_lbo, nlines, d_scan, st));            // (the bases per line stay: k_ing_pack skips the lines without any)
    HIPCHK(hipMemcpyAsync(&sumb, d_lbo + nlines - 1, 4, hipMemcpyDeviceToHost, st));
    ull tot[2] = { 0, 0 };
    HIPCHK(hipMemcpyAsync(tot, g.d_tot, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (tot[1]) { *irregular = 1; return SIMKA_OK; }
    const uint64_t fbases = (uint64_t)sumb + lastb, ffrags = (uint64_t)sumf + lastf;
    if (nb_reads) *nb_reads = tot[0];
    // ---- append to the lane's staging buffers
    const uint64_t words_now = (g.nb_bases + 31) / 32, words_need = (g.nb_bases + fbases + 31) / 32 + 2;
    rc = grow_keep(ctx, &ctx->d_reads[li], &ctx->reads_cap[li], words_need, words_now, st); if (rc) return rc;
    rc = grow_keep(ctx, &ctx->d_offsets[li], &ctx->offsets_cap[li], g.nb_frags + ffrags + 2, g.nb_frags, st); if (rc) return rc;
    if (g.nb_bases == 0) HIPCHK(hipMemsetAsync(ctx->d_reads[li], 0, words_need * 8, st));
    else HIPCHK(hipMemsetAsync(ctx->d_reads[li] + words_now, 0, (words_need - words_now) * 8, st));      // (the last word so far keeps its bases: its upper bits are zero)
    if (fbases)
        SIMKA_LAUNCH(k_ing_pack, dim3(lgrid), dim3(ING_BLOCK), 0, st, txt, (const uint32_t *)g.d_lines, (uint32_t)nlines, format, (const uint32_t *)g.d_lb,
                           (const uint32_t *)d_lbo, (const uint32_t *)g.d_lf, (ull)g.nb_bases, (ull)g.nb_frags, (ull *)ctx->d_reads[li], (ull *)ctx->d_offsets[li]);
    HIPCHK(hipGetLastError());
    g.nb_bases += fbases; g.nb_frags += ffrags; g.nb_reads += tot[0];
    HIPCHK(hipStreamSynchronize(st));          // the caller's text buffer is free again
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_ingest_text(simka_ctx *ctx, uint32_t sample, const char *text, uint64_t nb_bytes, int format, uint64_t *nb_reads, int *irregular) {
    return ingest_text_impl(ctx, sample, text, nb_bytes, format, nb_reads, irregular, false);
}
SIMKA_EXPORT int simka_ingest_text_device(simka_ctx *ctx, uint32_t sample, const void *d_text, uint64_t nb_bytes, int format, uint64_t *nb_reads, int *irregular) {
    if (d_text && ((uintptr_t)d_text & 15u)) return ctx ? ctx->fail(SIMKA_ERR_INVALID, "simka_ingest_text_device: the text must be 16-byte aligned") : SIMKA_ERR_INVALID;
    return ingest_text_impl(ctx, sample, (const char *)d_text, nb_bytes, format, nb_reads, irregular, true);
}

SIMKA_EXPORT int simka_ingest_count(simka_ctx *ctx, uint32_t sample, uint64_t *nb_bases, uint64_t *nb_reads) {
    if (!ctx) return SIMKA_ERR_INVALID;
    simka_ctx::Ingest *gp = nullptr; uint32_t li = 0;
    for (uint32_t l = 0; l < simka_ctx::MAX_LANES; l++) if (ctx->ing[l].open && ctx->ing[l].sample == sample) { gp = &ctx->ing[l]; li = l; }
    if (!gp) return ctx->fail(SIMKA_ERR_STATE, "simka_ingest_count: simka_ingest_begin was not called for sample %u", sample);
    simka_ctx::Ingest &g = *gp;
    g.open = false;
    if (nb_bases) *nb_bases = g.nb_bases;
    if (nb_reads) *nb_reads = g.nb_reads;
    simka_reads r;
    memset(&r, 0, sizeof r);
    r.nb_bases = g.nb_bases; r.nb_reads = g.nb_frags; r.nb_input_reads = g.nb_reads; r.on_device = 1; r.fixed_len = 0;
    if (g.nb_bases) {
        const ull end = g.nb_bases;
        SIMKA_LAUNCH(k_store_u64, dim3(1), dim3(1), 0, ctx->copy_stream, (ull *)(ctx->d_offsets[li] + g.nb_frags), end);
        HIPCHK(hipStreamSynchronize(ctx->copy_stream));
        r.packed = ctx->d_reads[li]; r.offsets = ctx->d_offsets[li];
    }
    // (the geometry may have to be set up first: it fixes the number of lanes; a context whose lane count differs from the guess of
    // simka_ingest_begin -- only the first sample can see that -- still finds its buffers: the lane index is kept with the state)
    const int rc = simka_count_sample(ctx, sample, &r);
    return rc;
}

SIMKA_EXPORT int simka_get_sample_totals(simka_ctx *ctx, uint32_t sample, simka_sample_totals *out) {
    if (!ctx || !out) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples;
    if (sample >= N || !ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "simka_get_sample_totals: sample %u not counted", sample);
    int rc = resolve_pending(ctx);
    if (rc) return rc;
    rc = check_device_error(ctx);
    if (rc) return rc;
    uint64_t t[SIMKA_NB_TOTALS];
    for (int i = 0; i < SIMKA_NB_TOTALS; i++)
        HIPCHK(hipMemcpyAsync(&t[i], ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, i) + sample, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->nb_reads = ctx->nb_reads[sample];
    out->nb_distinct = t[SIMKA_TOT_D]; out->nb_kmers = t[SIMKA_TOT_N]; out->sum_sq = t[SIMKA_TOT_Q];
    out->kmer_occurrences = t[SIMKA_TOT_KOCC]; out->distinct_all = t[SIMKA_TOT_DALL];
    return SIMKA_OK;
}

// ---- -keep-tmp: spectra out of / into the context --------------------------------------------
static int spectrum_rows(simka_ctx *ctx, uint32_t sample, const char *who, std::vector<uint32_t> &foff, std::vector<uint32_t> &fcnt) {
    const uint32_t N = ctx->cfg.nb_samples;
    if (sample >= N || !ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "%s: sample %u not counted", who, sample);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = resolve_pending(ctx);
    if (rc) return rc;
    rc = check_device_error(ctx);
    if (rc) return rc;
    foff.assign(ctx->nparts, 0); fcnt.assign(ctx->nparts, 0);
    if (!ctx->geometry_ready) return SIMKA_OK;        // only empty samples so far
    HIPCHK(hipMemcpyAsync(foff.data(), ctx->d_foff + (uint64_t)sample * ctx->nparts, ctx->nparts * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(fcnt.data(), ctx->d_fcnt + (uint64_t)sample * ctx->nparts, ctx->nparts * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}

// wide-k spectra are sorted by k-mer; their "partitions" are key-prefix ranges (cfg.log2_partitions bits, 12 by default)
static uint32_t wide_log2_parts(const simka_ctx *ctx) {
    const uint32_t lp = ctx->cfg.log2_partitions ? ctx->cfg.log2_partitions : 12u;
    return std::min<uint32_t>(std::min<uint32_t>(lp, 16u), 2u * ctx->cfg.kmer_size - 2u);
}

static int wide_check_counted(simka_ctx *ctx, uint32_t sample, const char *who) {
    if (sample >= ctx->cfg.nb_samples || !ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "%s: sample %u not counted", who, sample);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_sample_spectrum_info(simka_ctx *ctx, uint32_t sample, simka_spectrum_info *out) {
    if (!ctx || !out) return SIMKA_ERR_INVALID;
    if (ctx->wide) {
        const int rc = wide_check_counted(ctx, sample, "simka_sample_spectrum_info"); if (rc) return rc;
        out->nb_records = simka_wide_sample_records(ctx->wide, sample); out->nb_partitions = (uint64_t)1 << wide_log2_parts(ctx); out->key_words = 2;
        return SIMKA_OK;
    }
    out->key_words = 1;
    std::vector<uint32_t> foff, fcnt;
    const int rc = spectrum_rows(ctx, sample, "simka_sample_spectrum_info", foff, fcnt);
    if (rc) return rc;
    uint64_t n = 0;
    for (uint32_t c : fcnt) n += c;
    out->nb_records = n; out->nb_partitions = ctx->nparts;
    return SIMKA_OK;
}

static int export_sample(simka_ctx *ctx, uint32_t sample, uint32_t *part_counts, void *keys, void *counts, bool on_device, const char *who) {
    if (!ctx || !part_counts) return SIMKA_ERR_INVALID;
    if (ctx->wide) {
        int rc = wide_check_counted(ctx, sample, who); if (rc) return rc;
        int wrc = simka_wide_part_counts(ctx->wide, sample, wide_log2_parts(ctx), part_counts);
        if (wrc) return wide_fail(ctx, wrc);
        if (simka_wide_sample_records(ctx->wide, sample) == 0) return SIMKA_OK;
        if (!keys || !counts) return ctx->fail(SIMKA_ERR_INVALID, "%s: keys / counts are NULL", who);
        wrc = simka_wide_export(ctx->wide, sample, keys, counts, on_device ? 1 : 0);
        return wrc ? wide_fail(ctx, wrc) : SIMKA_OK;
    }
    std::vector<uint32_t> foff, fcnt;
    int rc = spectrum_rows(ctx, sample, who, foff, fcnt);
    if (rc) return rc;
    std::vector<ull> off(ctx->nparts + 1, 0);
    for (uint64_t p = 0; p < ctx->nparts; p++) { part_counts[p] = fcnt[p]; off[p + 1] = off[p] + fcnt[p]; }
    const uint64_t n = off[ctx->nparts];
    if (n == 0) return SIMKA_OK;
    if (!keys || !counts) return ctx->fail(SIMKA_ERR_INVALID, "%s: keys / counts are NULL", who);
    // gather the partition segments (slab-reserved, so with gaps) into one partition-major run on the device
    ull *d_off = nullptr, *d_keys = on_device ? (ull *)keys : nullptr; uint32_t *d_counts = on_device ? (uint32_t *)counts : nullptr;
    auto cleanup = [&] { if (d_off) (void)hipFree(d_off); if (!on_device) { if (d_keys) (void)hipFree(d_keys); if (d_counts) (void)hipFree(d_counts); } };
    if (dev_alloc(&d_off, ctx->nparts + 1) != hipSuccess || (!on_device && (dev_alloc(&d_keys, n) != hipSuccess || dev_alloc(&d_counts, n) != hipSuccess))) {
        cleanup();
        return ctx->fail(SIMKA_ERR_NOMEM, "%s: cannot allocate %llu records", who, (unsigned long long)n);
    }
    hipError_t e = hipMemcpyAsync(d_off, off.data(), (ctx->nparts + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        SIMKA_LAUNCH(k_gather_sample, dim3((uint32_t)std::min<uint64_t>(ctx->nparts, (uint64_t)ctx->num_cus * 8)), dim3(256), 0, ctx->stream,
                           ctx->d_solid_keys, ctx->d_solid_counts, ctx->d_sample_base + sample, ctx->d_foff + (uint64_t)sample * ctx->nparts,
                           ctx->d_fcnt + (uint64_t)sample * ctx->nparts, d_off, (uint32_t)ctx->nparts, d_keys, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !on_device) e = hipMemcpyAsync(keys, d_keys, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && !on_device) e = hipMemcpyAsync(counts, d_counts, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(SIMKA_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_export_sample(simka_ctx *ctx, uint32_t sample, uint32_t *part_counts, uint64_t *keys, uint32_t *counts) {
    return export_sample(ctx, sample, part_counts, keys, counts, false, "simka_export_sample");
}

SIMKA_EXPORT int simka_export_sample_device(simka_ctx *ctx, uint32_t sample, uint32_t *part_counts, void *d_keys, void *d_counts) {
    return export_sample(ctx, sample, part_counts, d_keys, d_counts, true, "simka_export_sample_device");
}

static int import_sample(simka_ctx *ctx, uint32_t sample, const simka_sample_totals *totals, const uint32_t *part_counts,
                         uint64_t nb_partitions, const void *keys, const void *counts_any, uint64_t nb_records, bool on_device) {
    if (!ctx || !totals || !part_counts) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    if (sample >= N) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: sample index %u out of range", sample);
    if (ctx->counted[sample]) return ctx->fail(SIMKA_ERR_STATE, "simka_import_sample: sample %u was already counted", sample);
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_import_sample: merge already ran");
    if (nb_records && (!keys || !counts_any)) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: keys / counts are NULL");
    if (nb_partitions == 0 || (nb_partitions & (nb_partitions - 1))) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: nb_partitions must be a power of two");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc;
    if (ctx->wide) {      // sorted two-word keys [hi x n][lo x n]; the partition counts only have to add up
        uint64_t sum = 0;
        for (uint64_t p = 0; p < nb_partitions; p++) sum += part_counts[p];
        if (sum != nb_records) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: part_counts sum to %llu, nb_records is %llu", (unsigned long long)sum, (unsigned long long)nb_records);
        const int wrc = simka_wide_import(ctx->wide, sample, keys, counts_any, nb_records, on_device ? 1 : 0);
        if (wrc) return wide_fail(ctx, wrc);
        ull t[SIMKA_NB_TOTALS];
        t[SIMKA_TOT_D] = totals->nb_distinct; t[SIMKA_TOT_N] = totals->nb_kmers; t[SIMKA_TOT_Q] = totals->sum_sq;
        t[SIMKA_TOT_DALL] = totals->distinct_all; t[SIMKA_TOT_KOCC] = totals->kmer_occurrences;
        for (int i = 0; i < SIMKA_NB_TOTALS; i++)
            HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, i) + sample, &t[i], 8, hipMemcpyHostToDevice, ctx->stream));
        if (ctx->d_hist) {
            std::vector<uint32_t> hc;
            const uint32_t *cnts = (const uint32_t *)counts_any;
            if (on_device && nb_records) { hc.resize(nb_records); HIPCHK(hipMemcpyAsync(hc.data(), counts_any, nb_records * 4, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); cnts = hc.data(); }
            std::vector<ull> hist(SIMKA_HIST_MAX, 0);
            std::vector<uint32_t> ovf;
            for (uint64_t i = 0; i < nb_records; i++) { const uint32_t c = cnts[i]; if (c < SIMKA_HIST_MAX) hist[c]++; else { ovf.push_back(sample); ovf.push_back(c); } }
            HIPCHK(hipMemcpyAsync(ctx->d_hist + (uint64_t)sample * SIMKA_HIST_MAX, hist.data(), SIMKA_HIST_MAX * 8, hipMemcpyHostToDevice, ctx->stream));
            if (!ovf.empty()) {
                ull novf = 0;
                HIPCHK(hipMemcpyAsync(&novf, ctx->d_ovf_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                const ull add = ovf.size() / 2;
                if (novf + add <= ctx->ovf_cap) HIPCHK(hipMemcpyAsync(ctx->d_ovf_list + 2 * novf, ovf.data(), ovf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                novf += add;
                HIPCHK(hipMemcpyAsync(ctx->d_ovf_cursor, &novf, 8, hipMemcpyHostToDevice, ctx->stream));
            }
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->nb_reads[sample] = totals->nb_reads;
        ctx->counted[sample] = 1;
        ctx->seg_dirty = true;      // (its segments carry no index of key-hash blocks: k_segment_rows builds it at the merge)
        return SIMKA_OK;
    }
    if (!ctx->geometry_ready) {
        ctx->cfg.log2_partitions = ceil_log2_u64(nb_partitions);     // the spectrum fixes the partition count of this run
        rc = setup_geometry(ctx, std::max<uint64_t>({ ctx->cfg.max_kmers_per_sample, totals->kmer_occurrences, nb_records, 1 }));
        if (rc) return rc;
    }
    if (ctx->nparts != nb_partitions)
        return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: spectrum has %llu partitions, this run %llu (pass log2_partitions of the exporting run)",
                         (unsigned long long)nb_partitions, (unsigned long long)ctx->nparts);
    std::vector<uint32_t> foff(ctx->nparts);
    uint64_t run = 0;
    for (uint64_t p = 0; p < ctx->nparts; p++) { foff[p] = (uint32_t)run; run += part_counts[p]; }
    if (run != nb_records || nb_records > 0xffffffffull) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_sample: part_counts sum to %llu, nb_records is %llu", (unsigned long long)run, (unsigned long long)nb_records);
    rc = resolve_pending(ctx);          // the arena cursor is only advanced by kernels: settle them first
    if (rc) return rc;
    ull cursor = 0;
    HIPCHK(hipMemcpyAsync(&cursor, ctx->d_arena_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (cursor + nb_records > ctx->arena_cap)
        return ctx->fail(SIMKA_ERR_NOMEM, "solid-spectrum arena exhausted (%llu records): raise solid_capacity", (unsigned long long)ctx->arena_cap);
    const ull next = cursor + nb_records;
    rc = arena_ensure(ctx, next); if (rc) return rc;
    ctx->arena_hi = std::max<uint64_t>(ctx->arena_hi, next);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (nb_records) {
        HIPCHK(hipMemcpyAsync(ctx->d_solid_keys + cursor, keys, nb_records * 8, kind, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_solid_counts + cursor, counts_any, nb_records * 4, kind, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(ctx->d_sample_base + sample, &cursor, 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_arena_cursor, &next, 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_foff + (uint64_t)sample * ctx->nparts, foff.data(), ctx->nparts * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_fcnt + (uint64_t)sample * ctx->nparts, part_counts, ctx->nparts * 4, hipMemcpyHostToDevice, ctx->stream));
    ull t[SIMKA_NB_TOTALS];
    t[SIMKA_TOT_D] = totals->nb_distinct; t[SIMKA_TOT_N] = totals->nb_kmers; t[SIMKA_TOT_Q] = totals->sum_sq;
    t[SIMKA_TOT_DALL] = totals->distinct_all; t[SIMKA_TOT_KOCC] = totals->kmer_occurrences;
    for (int i = 0; i < SIMKA_NB_TOTALS; i++)
        HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, i) + sample, &t[i], 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<ull> hist;
    std::vector<uint32_t> ovf;
    ull novf = 0;
    std::vector<uint32_t> counts_host;
    const uint32_t *counts = on_device ? nullptr : (const uint32_t *)counts_any;
    if (ctx->d_hist && on_device && nb_records) {
        counts_host.resize(nb_records);
        HIPCHK(hipMemcpyAsync(counts_host.data(), counts_any, nb_records * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        counts = counts_host.data();
    }
    if (ctx->d_hist) {   // -complex-dist: the histogram of solid counts (Whittaker's one-sided terms) is a function of `counts`
        hist.assign(SIMKA_HIST_MAX, 0);
        for (uint64_t i = 0; i < nb_records; i++) { const uint32_t c = counts[i]; if (c < SIMKA_HIST_MAX) hist[c]++; else { ovf.push_back(sample); ovf.push_back(c); } }
        HIPCHK(hipMemcpyAsync(ctx->d_hist + (uint64_t)sample * SIMKA_HIST_MAX, hist.data(), SIMKA_HIST_MAX * 8, hipMemcpyHostToDevice, ctx->stream));
        if (!ovf.empty()) {
            HIPCHK(hipMemcpyAsync(&novf, ctx->d_ovf_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            const ull add = ovf.size() / 2;
            if (novf + add <= ctx->ovf_cap) HIPCHK(hipMemcpyAsync(ctx->d_ovf_list + 2 * novf, ovf.data(), ovf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            novf += add;      // past the capacity simka_merge reports the overflow, as for counted samples
            HIPCHK(hipMemcpyAsync(ctx->d_ovf_cursor, &novf, 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));       // host buffers (caller's and ours) may go away
    ctx->nb_reads[sample] = totals->nb_reads;
    ctx->counted[sample] = 1;
    ctx->seg_dirty = true;      // (its segments carry no index of key-hash blocks: k_segment_rows builds it at the merge)
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_import_sample(simka_ctx *ctx, uint32_t sample, const simka_sample_totals *totals, const uint32_t *part_counts,
                                     uint64_t nb_partitions, const uint64_t *keys, const uint32_t *counts, uint64_t nb_records) {
    return import_sample(ctx, sample, totals, part_counts, nb_partitions, keys, counts, nb_records, false);
}

SIMKA_EXPORT int simka_import_sample_device(simka_ctx *ctx, uint32_t sample, const simka_sample_totals *totals, const uint32_t *part_counts,
                                            uint64_t nb_partitions, const void *d_keys, const void *d_counts, uint64_t nb_records) {
    return import_sample(ctx, sample, totals, part_counts, nb_partitions, d_keys, d_counts, nb_records, true);
}

// ---- batch forms (multi-GPU exchange) ---------------------------------------------------------
SIMKA_EXPORT int simka_samples_spectrum_info(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, uint32_t *part_counts, simka_sample_totals *totals) {
    if (!ctx || (nb && (!samples || !part_counts || !totals))) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    for (uint32_t j = 0; j < nb; j++)
        if (samples[j] >= N || !ctx->counted[samples[j]]) return ctx->fail(SIMKA_ERR_STATE, "simka_samples_spectrum_info: sample %u not counted", samples[j]);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = resolve_pending(ctx);
    if (rc) return rc;
    rc = check_device_error(ctx);
    if (rc) return rc;
    std::vector<ull> tot((size_t)SIMKA_NB_TOTALS * N, 0);
    HIPCHK(hipMemcpyAsync(tot.data(), ctx->d_stats + stats_off_tot(N, fl, 0), tot.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    uint32_t *pc_host = part_counts;          // where the rows are downloaded to (a pinned slot when part_counts is pageable)
    if (!ctx->wide && ctx->geometry_ready) pc_host = (uint32_t *)stage_d2h(ctx, 0, part_counts, (size_t)nb * ctx->nparts * 4);
    for (uint32_t j = 0; j < nb; j++) {
        if (ctx->wide) {      // sorted two-word spectra: partitions are key-prefix ranges
            const int wrc = simka_wide_part_counts(ctx->wide, samples[j], wide_log2_parts(ctx), part_counts + ((size_t)j << wide_log2_parts(ctx)));
            if (wrc) return wide_fail(ctx, wrc);
        } else if (ctx->geometry_ready) HIPCHK(hipMemcpyAsync(pc_host + (size_t)j * ctx->nparts, ctx->d_fcnt + (uint64_t)samples[j] * ctx->nparts, ctx->nparts * 4, hipMemcpyDeviceToHost, ctx->stream));
        else memset(part_counts + (size_t)j * ctx->nparts, 0, ctx->nparts * 4);
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (pc_host != part_counts) memcpy(part_counts, pc_host, (size_t)nb * ctx->nparts * 4);
    for (uint32_t j = 0; j < nb; j++) {
        const uint32_t s = samples[j];
        totals[j].nb_reads = ctx->nb_reads[s];
        totals[j].nb_distinct = tot[(size_t)SIMKA_TOT_D * N + s]; totals[j].nb_kmers = tot[(size_t)SIMKA_TOT_N * N + s];
        totals[j].sum_sq = tot[(size_t)SIMKA_TOT_Q * N + s]; totals[j].kmer_occurrences = tot[(size_t)SIMKA_TOT_KOCC * N + s];
        totals[j].distinct_all = tot[(size_t)SIMKA_TOT_DALL * N + s];
    }
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_gather_samples_device(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const uint64_t *out_offsets, void *d_keys, void *d_counts) {
    if (!ctx) return SIMKA_ERR_INVALID;
    if (ctx->wide) {      // d_keys: [high words of every run][low words], the halves `total records` apart (given by the caller through nb_total)
        return ctx->fail(SIMKA_ERR_INVALID, "simka_gather_samples_device: use simka_gather_samples_device_wide for kmer_size >= 32");
    }
    if (nb == 0 || !ctx->geometry_ready) return SIMKA_OK;
    if (!samples || !out_offsets || !d_keys || !d_counts) return ctx->fail(SIMKA_ERR_INVALID, "simka_gather_samples_device: NULL argument");
    const uint32_t N = ctx->cfg.nb_samples;
    for (uint32_t j = 0; j < nb; j++)
        if (samples[j] >= N || !ctx->counted[samples[j]]) return ctx->fail(SIMKA_ERR_STATE, "simka_gather_samples_device: sample %u not counted", samples[j]);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = resolve_pending(ctx);
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_xoff, &ctx->xoff_cap, (uint64_t)nb * ctx->nparts + (nb + 1) / 2 + 1); if (rc) return rc;
    uint32_t *d_samples = (uint32_t *)(ctx->d_xoff + (uint64_t)nb * ctx->nparts);
    HIPCHK(hipMemcpyAsync(ctx->d_xoff, stage_h2d(ctx, 1, out_offsets, (size_t)nb * ctx->nparts * 8), (uint64_t)nb * ctx->nparts * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_samples, samples, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
    SIMKA_LAUNCH(k_gather_samples, dim3((uint32_t)std::min<uint64_t>(ctx->nparts, (uint64_t)ctx->num_cus * 4), nb), dim3(256), 0, ctx->stream,
                       ctx->d_solid_keys, ctx->d_solid_counts, ctx->d_sample_base, ctx->d_foff, ctx->d_fcnt, d_samples, ctx->d_xoff,
                       (uint32_t)ctx->nparts, (ull *)d_keys, (uint32_t *)d_counts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));       // the caller hands the buffers to RCCL on another stream
    return SIMKA_OK;
}

// ---- the same exchange with its tables computed on the device (ABI 8) ---------------------------------------------------------------
SIMKA_EXPORT int simka_pack_plan(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, uint32_t nb_ranges, uint64_t *send_records) {
    if (!ctx || !send_records || nb_ranges == 0 || (nb && !samples)) return SIMKA_ERR_INVALID;
    if (ctx->wide) return ctx->fail(SIMKA_ERR_INVALID, "simka_pack_plan: kmer_size >= 32 exchanges whole sorted runs (simka_gather_samples_device_wide)");
    const uint32_t N = ctx->cfg.nb_samples;
    for (uint32_t j = 0; j < nb; j++)
        if (samples[j] >= N || !ctx->counted[samples[j]]) return ctx->fail(SIMKA_ERR_STATE, "simka_pack_plan: sample %u not counted", samples[j]);
    for (uint32_t g = 0; g < nb_ranges; g++) send_records[g] = 0;
    ctx->drop_plan();            // (a plan exists only once every check below has passed)
    if (ctx->geometry_ready && nb_ranges > ctx->nparts) return ctx->fail(SIMKA_ERR_INVALID, "simka_pack_plan: more ranges (%u) than partitions", nb_ranges);
    if (nb == 0 || !ctx->geometry_ready) { ctx->plan_samples.assign(samples, samples + nb); ctx->plan_ranges = nb_ranges; ctx->plan_valid = true; return SIMKA_OK; }
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = resolve_pending(ctx); if (rc) return rc;
    rc = check_device_error(ctx); if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_xrows, &ctx->xrows_cap, (uint64_t)nb * nb_ranges); if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_xsamples, &ctx->xsamples_cap, (uint64_t)nb); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_xsamples, samples, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
    SIMKA_LAUNCH(k_range_rowsum, dim3(nb_ranges, nb), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_fcnt, (const uint32_t *)ctx->d_xsamples, (uint32_t)ctx->nparts, nb_ranges, ctx->d_xrows);
    std::vector<ull> rows((size_t)nb * nb_ranges), starts((size_t)nb * nb_ranges);
    HIPCHK(hipMemcpyAsync(rows.data(), ctx->d_xrows, rows.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ull pos = 0;
    for (uint32_t g = 0; g < nb_ranges; g++) {          // destination-major: [g][slot j][partitions of g]
        for (uint32_t j = 0; j < nb; j++) { starts[(size_t)j * nb_ranges + g] = pos; pos += rows[(size_t)j * nb_ranges + g]; send_records[g] += rows[(size_t)j * nb_ranges + g]; }
    }
    HIPCHK(hipMemcpyAsync(ctx->d_xrows, starts.data(), starts.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // (starts is a local)
    ctx->plan_samples.assign(samples, samples + nb); ctx->plan_ranges = nb_ranges; ctx->plan_total = pos; ctx->plan_valid = true;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_pack_run(simka_ctx *ctx, void *d_keys, void *d_counts, int32_t *d_meta, uint32_t nb_slots, uint32_t width) {
    if (!ctx) return SIMKA_ERR_INVALID;
    // the plan lives in d_xrows / d_xsamples: simka_reset, an import or a failed simka_pack_plan drop it (stale starts and sample ids would
    // send the gather out of bounds)
    if (!ctx->plan_valid || ctx->plan_ranges == 0) return ctx->fail(SIMKA_ERR_STATE, "simka_pack_run: no valid plan (simka_pack_plan first; simka_reset and the imports drop it)");
    const uint32_t nb = (uint32_t)ctx->plan_samples.size(), G = ctx->plan_ranges;
    if (nb == 0 || !ctx->geometry_ready) return SIMKA_OK;
    if (ctx->plan_total && (!d_keys || !d_counts)) return ctx->fail(SIMKA_ERR_INVALID, "simka_pack_run: NULL buffers");
    const uint32_t maxw = (uint32_t)((ctx->nparts + G - 1) / G);
    if (d_meta && (nb_slots < nb || width < maxw)) return ctx->fail(SIMKA_ERR_INVALID, "simka_pack_run: meta of %u slots x %u partitions, the plan needs %u x %u", nb_slots, width, nb, maxw);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = ensure_cap(ctx, &ctx->d_xoff, &ctx->xoff_cap, (uint64_t)nb * ctx->nparts + 1); if (rc) return rc;
    SIMKA_LAUNCH(k_range_offsets, dim3(nb), dim3(1024), 0, ctx->stream, (const uint32_t *)ctx->d_fcnt, (const uint32_t *)ctx->d_xsamples, (uint32_t)ctx->nparts, G,
                       (const ull *)ctx->d_xrows, ctx->d_xoff, d_meta, nb_slots, width);
    SIMKA_LAUNCH(k_gather_samples, dim3((uint32_t)std::min<uint64_t>(ctx->nparts, (uint64_t)ctx->num_cus * 4), nb), dim3(256), 0, ctx->stream,
                       ctx->d_solid_keys, ctx->d_solid_counts, ctx->d_sample_base, ctx->d_foff, ctx->d_fcnt, ctx->d_xsamples, ctx->d_xoff,
                       (uint32_t)ctx->nparts, (ull *)d_keys, (uint32_t *)d_counts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));       // the caller hands the buffers to RCCL on another stream
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_import_block_device(simka_ctx *ctx, const uint32_t *slot_samples, uint32_t nb_slots_total, const simka_sample_totals *totals,
                                           uint64_t part_lo, uint64_t part_width, uint32_t width, const int32_t *d_meta, uint64_t nb_partitions,
                                           const void *d_keys, const void *d_counts, uint64_t nb_records) {
    if (ctx && ctx->wide) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: use simka_import_samples_device_wide for kmer_size >= 32");
    if (!ctx || (nb_slots_total && (!slot_samples || !totals || !d_meta))) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_import_block_device: merge already ran");
    std::vector<uint8_t> seen(N, 0);
    for (uint32_t q = 0; q < nb_slots_total; q++) {
        const uint32_t s = slot_samples[q];
        if (s == 0xffffffffu) continue;
        if (s >= N) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: sample index %u out of range", s);
        if (ctx->counted[s] || seen[s]) return ctx->fail(SIMKA_ERR_STATE, "simka_import_block_device: sample %u was already counted", s);
        seen[s] = 1;
    }
    if (nb_records && (!d_keys || !d_counts)) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: keys / counts are NULL");
    if (nb_partitions == 0 || (nb_partitions & (nb_partitions - 1))) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: nb_partitions must be a power of two");
    if (part_lo + part_width > nb_partitions || part_width > width) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: partition range outside [0, nb_partitions) or wider than the meta rows");
    if (nb_slots_total == 0) return SIMKA_OK;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc;
    if (!ctx->geometry_ready) {
        ctx->cfg.log2_partitions = ceil_log2_u64(nb_partitions);
        uint64_t hint = std::max<uint64_t>(ctx->cfg.max_kmers_per_sample, 1);
        for (uint32_t q = 0; q < nb_slots_total; q++) if (slot_samples[q] != 0xffffffffu) hint = std::max<uint64_t>(hint, totals[q].kmer_occurrences);
        rc = setup_geometry(ctx, hint); if (rc) return rc;
    }
    if (ctx->nparts != nb_partitions)
        return ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: spectra have %llu partitions, this run %llu", (unsigned long long)nb_partitions, (unsigned long long)ctx->nparts);
    rc = resolve_pending(ctx); if (rc) return rc;
    const uint64_t P = ctx->nparts;
    ctx->drop_plan();            // (d_xrows / d_xsamples hold a pack plan until here)
    rc = ensure_cap(ctx, &ctx->d_xrows, &ctx->xrows_cap, (uint64_t)nb_slots_total); if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->d_xsamples, &ctx->xsamples_cap, (uint64_t)nb_slots_total); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_xsamples, slot_samples, (size_t)nb_slots_total * 4, hipMemcpyHostToDevice, ctx->stream));
    SIMKA_LAUNCH(k_import_tables, dim3(nb_slots_total), dim3(1024), 0, ctx->stream, d_meta, (const uint32_t *)ctx->d_xsamples, width, (uint32_t)part_width, (uint32_t)P, (uint32_t)part_lo,
                       ctx->d_foff, ctx->d_fcnt, ctx->d_xrows);
    std::vector<ull> slot_tot(nb_slots_total);
    ull cursor = 0;
    // k_import_tables has written the foff / fcnt rows of the target samples (the slot totals come out of the same scan): a block that
    // is rejected below must not leave them behind -- the samples stay "not counted", and a later merge would walk rows that point
    // anywhere.  The rows of a sample that is not counted are all zero, so clearing the partition range restores them.
    auto rollback = [&](int code) {
        for (uint32_t q = 0; q < nb_slots_total; q++) {
            const uint32_t s = slot_samples[q];
            if (s == 0xffffffffu || part_width == 0) continue;
            (void)hipMemsetAsync(ctx->d_foff + (uint64_t)s * P + part_lo, 0, (size_t)part_width * 4, ctx->stream);
            (void)hipMemsetAsync(ctx->d_fcnt + (uint64_t)s * P + part_lo, 0, (size_t)part_width * 4, ctx->stream);
        }
        (void)hipStreamSynchronize(ctx->stream);
        return code;
    };
    if (hipMemcpyAsync(slot_tot.data(), ctx->d_xrows, (size_t)nb_slots_total * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(&cursor, ctx->d_arena_cursor, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return rollback(ctx->fail(SIMKA_ERR_HIP, "simka_import_block_device: reading the slot totals failed: %s", hipGetErrorString(hipGetLastError())));
    ull sum = 0;
    for (uint32_t q = 0; q < nb_slots_total; q++) {
        if (slot_tot[q] > 0xffffffffull) return rollback(ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: slot %u holds more than 2^32 records", q));
        sum += slot_tot[q];
    }
    if (sum != nb_records) return rollback(ctx->fail(SIMKA_ERR_INVALID, "simka_import_block_device: the meta rows sum to %llu records, nb_records is %llu", (unsigned long long)sum, (unsigned long long)nb_records));
    if (cursor + nb_records > ctx->arena_cap)
        return rollback(ctx->fail(SIMKA_ERR_NOMEM, "solid-spectrum arena exhausted (%llu records): raise solid_capacity", (unsigned long long)ctx->arena_cap));
    const ull next = cursor + nb_records;
    rc = arena_ensure(ctx, next); if (rc) return rollback(rc);
    ctx->arena_hi = std::max<uint64_t>(ctx->arena_hi, next);
    if (nb_records) {
        HIPCHK(hipMemcpyAsync(ctx->d_solid_keys + cursor, d_keys, nb_records * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_solid_counts + cursor, d_counts, nb_records * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(ctx->d_arena_cursor, &next, 8, hipMemcpyHostToDevice, ctx->stream));
    // sample bases and totals: read-modify-write of two small arrays
    std::vector<ull> bases(N + 1), tot((size_t)SIMKA_NB_TOTALS * N);
    HIPCHK(hipMemcpyAsync(bases.data(), ctx->d_sample_base, (size_t)(N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(tot.data(), ctx->d_stats + stats_off_tot(N, fl, 0), tot.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ull pos = cursor;
    for (uint32_t q = 0; q < nb_slots_total; q++) {
        const uint32_t s = slot_samples[q];
        if (s == 0xffffffffu) continue;
        bases[s] = pos; pos += slot_tot[q];
        tot[(size_t)SIMKA_TOT_D * N + s] = totals[q].nb_distinct; tot[(size_t)SIMKA_TOT_N * N + s] = totals[q].nb_kmers;
        tot[(size_t)SIMKA_TOT_Q * N + s] = totals[q].sum_sq; tot[(size_t)SIMKA_TOT_KOCC * N + s] = totals[q].kmer_occurrences;
        tot[(size_t)SIMKA_TOT_DALL * N + s] = totals[q].distinct_all;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_sample_base, bases.data(), (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, 0), tot.data(), tot.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->d_hist && part_width) {   // -complex-dist: per-sample histogram of the imported solid counts
        // one clear + one launch per RUN of consecutive sample indices among the slots (blockIdx.y = sample of the run), as
        // simka_import_samples_device does: rank r's slots hold the samples r, r + world, ... -- runs of one -- but a caller with contiguous
        // sample blocks gets one launch per block instead of one per slot
        const uint32_t gx = (uint32_t)std::min<uint64_t>(part_width, 1024);
        std::vector<uint32_t> ss;
        for (uint32_t q = 0; q < nb_slots_total; q++) if (slot_samples[q] != 0xffffffffu) ss.push_back(slot_samples[q]);
        std::sort(ss.begin(), ss.end());
        for (size_t a = 0; a < ss.size(); ) {
            size_t b = a + 1;
            while (b < ss.size() && ss[b] == ss[b - 1] + 1u && b - a < 65535u) b++;
            const uint32_t s0 = ss[a], ns = (uint32_t)(b - a);
            HIPCHK(hipMemsetAsync(ctx->d_hist + (uint64_t)s0 * SIMKA_HIST_MAX, 0, (size_t)ns * SIMKA_HIST_MAX * 8, ctx->stream));
            SIMKA_LAUNCH(k_import_hist, dim3(gx, ns), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_solid_counts, (const ull *)ctx->d_sample_base,
                               (const uint32_t *)ctx->d_foff, (const uint32_t *)ctx->d_fcnt, (uint32_t)P, (uint32_t)part_lo, (uint32_t)part_width, s0, (ull *)ctx->d_hist,
                               ctx->d_ovf_list, (ull *)ctx->d_ovf_cursor, (ull)ctx->ovf_cap);
            a = b;
        }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t q = 0; q < nb_slots_total; q++) { const uint32_t s = slot_samples[q]; if (s != 0xffffffffu) { ctx->nb_reads[s] = totals[q].nb_reads; ctx->counted[s] = 1; } }
    ctx->seg_dirty = true;      // (its segments carry no index of key-hash blocks: k_segment_rows builds it at the merge)
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_import_samples_device(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const simka_sample_totals *totals,
                                             uint64_t part_lo, uint64_t part_width, const uint32_t *part_counts, const uint64_t *in_offsets,
                                             uint64_t nb_partitions, const void *d_keys, const void *d_counts, uint64_t nb_records) {
    if (ctx && ctx->wide) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: use simka_import_samples_device_wide for kmer_size >= 32");
    if (!ctx || (nb && (!samples || !totals || !part_counts || !in_offsets))) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_import_samples_device: merge already ran");
    for (uint32_t j = 0; j < nb; j++) {
        if (samples[j] >= N) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: sample index %u out of range", samples[j]);
        if (ctx->counted[samples[j]]) return ctx->fail(SIMKA_ERR_STATE, "simka_import_samples_device: sample %u was already counted", samples[j]);
    }
    if (nb_records && (!d_keys || !d_counts)) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: keys / counts are NULL");
    if (nb_partitions == 0 || (nb_partitions & (nb_partitions - 1))) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: nb_partitions must be a power of two");
    if (part_lo + part_width > nb_partitions) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: partition range outside [0, nb_partitions)");
    if (nb == 0) return SIMKA_OK;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc;
    if (!ctx->geometry_ready) {
        ctx->cfg.log2_partitions = ceil_log2_u64(nb_partitions);
        uint64_t hint = std::max<uint64_t>(ctx->cfg.max_kmers_per_sample, 1);
        for (uint32_t j = 0; j < nb; j++) hint = std::max<uint64_t>(hint, totals[j].kmer_occurrences);
        rc = setup_geometry(ctx, hint); if (rc) return rc;
    }
    if (ctx->nparts != nb_partitions)
        return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: spectra have %llu partitions, this run %llu", (unsigned long long)nb_partitions, (unsigned long long)ctx->nparts);
    const uint64_t P = ctx->nparts, w = part_width, pmin = part_lo;
    uint64_t sum = 0;
    for (uint32_t j = 0; j < nb; j++)
        for (uint64_t p = 0; p < w; p++) {
            const uint32_t c = part_counts[(size_t)j * w + p];
            if (!c) continue;
            if (in_offsets[(size_t)j * w + p] + c > nb_records) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: run (%u,%llu) lies outside the block", j, (unsigned long long)(pmin + p));
            sum += c;
        }
    if (sum != nb_records) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: part_counts sum to %llu, nb_records is %llu", (unsigned long long)sum, (unsigned long long)nb_records);
    rc = resolve_pending(ctx); if (rc) return rc;
    ull cursor = 0;
    HIPCHK(hipMemcpyAsync(&cursor, ctx->d_arena_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (cursor + nb_records > ctx->arena_cap)
        return ctx->fail(SIMKA_ERR_NOMEM, "solid-spectrum arena exhausted (%llu records): raise solid_capacity", (unsigned long long)ctx->arena_cap);
    const ull next = cursor + nb_records;
    rc = arena_ensure(ctx, next); if (rc) return rc;
    ctx->arena_hi = std::max<uint64_t>(ctx->arena_hi, next);
    if (nb_records) {
        HIPCHK(hipMemcpyAsync(ctx->d_solid_keys + cursor, d_keys, nb_records * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_solid_counts + cursor, d_counts, nb_records * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(ctx->d_arena_cursor, &next, 8, hipMemcpyHostToDevice, ctx->stream));
    // sample_base of a sample = its first run in the block; foff (u32) is a run's offset from there, so only the runs of ONE
    // sample must lie within 2^32 records of each other -- the block itself may be larger (C3 on two ranks: 3.7e9 records)
    bool consecutive = true;
    for (uint32_t j = 1; j < nb; j++) if (samples[j] != samples[0] + j) consecutive = false;
    // (both tables go up as nb x w words: from pinned memory -- the offsets are built in a slot, the caller's counts staged in another)
    std::vector<uint32_t> hfo_v;
    const size_t tab_bytes = (size_t)nb * std::max<uint64_t>(w, 1) * 4;
    uint32_t *hfo = tab_bytes >= stage_min() ? (uint32_t *)pin_slot(ctx, 2, tab_bytes) : nullptr;
    if (!hfo) { hfo_v.resize((size_t)nb * std::max<uint64_t>(w, 1)); hfo = hfo_v.data(); }
    const uint32_t *hfc_p = (const uint32_t *)stage_h2d(ctx, 3, part_counts, (size_t)nb * w * 4);
    std::vector<ull> bases(nb, cursor);
    for (uint32_t j = 0; j < nb; j++) {
        uint64_t lo = ~0ull, hi = 0;
        for (uint64_t p = 0; p < w; p++)
            if (part_counts[(size_t)j * w + p]) { lo = std::min(lo, in_offsets[(size_t)j * w + p]); hi = std::max(hi, in_offsets[(size_t)j * w + p]); }
        if (lo == ~0ull) lo = 0;
        if (hi - lo > 0xffffffffull) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device: the runs of sample %u span more than 2^32 records of the block", samples[j]);
        bases[j] = cursor + lo;
        for (uint64_t p = 0; p < w; p++)
            hfo[(size_t)j * w + p] = part_counts[(size_t)j * w + p] ? (uint32_t)(in_offsets[(size_t)j * w + p] - lo) : 0u;
    }
    if (consecutive) {
        if (w) {
            HIPCHK(hipMemcpy2DAsync(ctx->d_foff + (uint64_t)samples[0] * P + pmin, P * 4, hfo, w * 4, w * 4, nb, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpy2DAsync(ctx->d_fcnt + (uint64_t)samples[0] * P + pmin, P * 4, hfc_p, w * 4, w * 4, nb, hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(ctx->d_sample_base + samples[0], bases.data(), (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (uint32_t j = 0; j < nb; j++) {
            if (w) {
                HIPCHK(hipMemcpyAsync(ctx->d_foff + (uint64_t)samples[j] * P + pmin, hfo + (size_t)j * w, w * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(hipMemcpyAsync(ctx->d_fcnt + (uint64_t)samples[j] * P + pmin, hfc_p + (size_t)j * w, w * 4, hipMemcpyHostToDevice, ctx->stream));
            }
            HIPCHK(hipMemcpyAsync(ctx->d_sample_base + samples[j], &bases[j], 8, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    // totals: rows of N values; read-modify-write the whole block once
    std::vector<ull> tot((size_t)SIMKA_NB_TOTALS * N);
    HIPCHK(hipMemcpyAsync(tot.data(), ctx->d_stats + stats_off_tot(N, fl, 0), tot.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t j = 0; j < nb; j++) {
        const uint32_t s = samples[j];
        tot[(size_t)SIMKA_TOT_D * N + s] = totals[j].nb_distinct; tot[(size_t)SIMKA_TOT_N * N + s] = totals[j].nb_kmers;
        tot[(size_t)SIMKA_TOT_Q * N + s] = totals[j].sum_sq; tot[(size_t)SIMKA_TOT_KOCC * N + s] = totals[j].kmer_occurrences;
        tot[(size_t)SIMKA_TOT_DALL * N + s] = totals[j].distinct_all;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, 0), tot.data(), tot.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->d_hist && w) {   // -complex-dist: per-sample histogram of the imported solid counts (Whittaker's one-sided terms), on the device:
        // the imported runs are described by the tables just uploaded (C5 on eight ranks imports 2.4e9 records per rank -- not a host loop)
        const uint32_t gx = (uint32_t)std::min<uint64_t>(w, 1024);
        bool hist_ok = true;
        auto launch = [&](uint32_t s0, uint32_t ns) {
            if (hipMemsetAsync(ctx->d_hist + (uint64_t)s0 * SIMKA_HIST_MAX, 0, (size_t)ns * SIMKA_HIST_MAX * 8, ctx->stream) != hipSuccess) { hist_ok = false; return; }
            SIMKA_LAUNCH(k_import_hist, dim3(gx, ns), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_solid_counts, (const ull *)ctx->d_sample_base,
                               (const uint32_t *)ctx->d_foff, (const uint32_t *)ctx->d_fcnt, (uint32_t)P, (uint32_t)pmin, (uint32_t)w, s0, (ull *)ctx->d_hist,
                               ctx->d_ovf_list, (ull *)ctx->d_ovf_cursor, (ull)ctx->ovf_cap);
        };
        if (consecutive) launch(samples[0], nb);
        else for (uint32_t j = 0; j < nb; j++) launch(samples[j], 1);
        if (!hist_ok) { ctx->err = "simka_import_samples_device: clearing the count histogram failed"; return SIMKA_ERR_HIP; }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t j = 0; j < nb; j++) { ctx->nb_reads[samples[j]] = totals[j].nb_reads; ctx->counted[samples[j]] = 1; }
    ctx->seg_dirty = true;      // (its segments carry no index of key-hash blocks: k_segment_rows builds it at the merge)
    return SIMKA_OK;
}

// ---- pair accumulation: shared by the hash merge (k_group's CSR) and the sort merge of the wide-k path -------------------
#ifdef SIMKA_PHASE_PROF
static void group_phase_report() {
    ull h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_group_phase), 64) != hipSuccess) return;
    ull t_ = 0; for (int i_ = 0; i_ < 8; i_++) t_ += h[i_];
    if (t_) fprintf(stderr, "k_group phases %%: sizing %.1f clear %.1f gather+hash %.1f geometry %.1f scan %.1f slab bookkeeping %.1f groups+entries %.1f\n", 100.0 * h[0] / t_, 100.0 * h[1] / t_, 100.0 * h[2] / t_,
                    100.0 * h[3] / t_, 100.0 * h[4] / t_, 100.0 * h[5] / t_, 100.0 * h[6] / t_);
    memset(h, 0, 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_group_phase), h, 64);
}
static void pairs_phase_report() {
    ull h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pairs_phase), 64) != hipSuccess) return;
    ull t_ = 0; for (int i_ = 0; i_ < 8; i_++) t_ += h[i_];
    if (t_) fprintf(stderr, "k_pairs phases %%: loop-top %.1f stage %.1f sync+flush %.1f compaction %.1f scan %.1f search %.1f pairs %.1f\n", 100.0 * h[0] / t_, 100.0 * h[1] / t_, 100.0 * h[2] / t_,
                    100.0 * h[3] / t_, 100.0 * h[4] / t_, 100.0 * h[5] / t_, 100.0 * h[6] / t_);
    memset(h, 0, 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pairs_phase), h, 64);
}
#endif

struct PairLaunch { SimkaPairCfg pc; size_t lds_pairs = 0; uint32_t ntp = 1, nblk = 1; bool small_block = false; };

// is the tile-major pair kernel usable (SIMKA_PAIRS_LEGACY=1 keeps the scan-and-compact kernel: tests / A-B; read at every merge)
static bool tile_major_enabled() { return simka_test_knob("SIMKA_PAIRS_LEGACY") == nullptr; }

static void pair_setup(simka_ctx *ctx, PairLaunch &pl, bool legacy_layout = false, uint32_t force_span_cap = 0) {
    // pair-accumulator tiling: all N(N-1)/2 cells in LDS when they fit, else T x T sample tiles
    const uint32_t flags = ctx->cfg.dist_flags;
    const uint32_t N = ctx->cfg.nb_samples;
    SimkaPairCfg &pc = pl.pc;
    pc.nb_samples = N; pc.nacc32 = stats_nacc32(flags); pc.nacc64 = stats_nacc64(flags); pc.nacc = pc.nacc32 + pc.nacc64;
    pc.simple = (flags & SIMKA_DIST_SIMPLE) ? 1u : 0u;
    pc.nb_pairs = (uint64_t)N * (N - 1) / 2;
    pc.tot_n = (const ull *)ctx->d_stats + stats_off_tot(N, flags, SIMKA_TOT_N);   // GLOBAL N_i: all-reduced by the caller when sharded
    // LDS of k_pairs: head | packed cells | ent | (complex: p, p ln p, tile N table) | gdesc | gpref | tmp | (tiled: gdescB, epre, idxA, idxB)
    // The span capacity EC (entries staged per iteration) takes what the cells leave: longer spans amortise the per-span cost.
    const bool cplx = pc.nacc64 != 0;
    auto lds_single = [&](size_t ec) { return SIMKA_LDS_HEAD + ec * 8 + (cplx ? ec * 16 + SIMKA_PAIR_TN * 16 + SIMKA_LNTAB * 16 : 0) + (ec / 2) * 4 + (ec / 2 + 2) * 4 + 32 * 4 + 64; };
    // tiled: + gdescB + (scan-and-compact kernel: flag scan, two index lists | tile-major kernel: the two run tables)
    const bool tm_layout = !legacy_layout && tile_major_enabled();
    auto lds_tiled = [&](size_t ec) { return lds_single(ec) + (ec / 2) * 4 + (tm_layout ? (ec / 2) * 8 + 2 * sizeof(KtmRange) + 64 /* range tables */ : (ec + 2) * 4 + ec * 4); };
    const size_t lds_max = 160 * 1024;
    const size_t cell_bytes = 4 * pc.nacc32 + 8 * pc.nacc64;
    size_t lds_fixed;
    if (lds_single(K3_CAP) + (pc.nb_pairs + 4) * cell_bytes <= lds_max && (!cplx || N <= SIMKA_PAIR_TN)) {
        pc.tile = N; pc.ntiles = 1; pc.ncell = (uint32_t)pc.nb_pairs;
        pc.span_cap = K3_CAP;
        const uint32_t span_max = N <= 32 ? 2 * K3_CAP : SIMKA_SPAN_MAX;      // few samples: small blocks, several per CU
        while (pc.span_cap + K3_CAP <= span_max && lds_single(pc.span_cap + K3_CAP) + (pc.nb_pairs + 4) * cell_bytes <= lds_max) pc.span_cap += K3_CAP;
        lds_fixed = lds_single(pc.span_cap);
    } else {
        pc.span_cap = 2 * K3_CAP;       // tiled: larger spans cost tile edge (more tile pairs replaying the spans)
        if (tm_layout && simka_exp_knob("SIMKA_TM_SPAN")) pc.span_cap = std::min<uint32_t>(SIMKA_SPAN_MAX, std::max<uint32_t>(K3_CAP, (uint32_t)atoi(simka_exp_knob("SIMKA_TM_SPAN")) / K3_CAP * K3_CAP));   // experiments
        if (force_span_cap) pc.span_cap = force_span_cap;                // the spans already exist
        lds_fixed = lds_tiled(pc.span_cap);
        const uint64_t max_cells = (lds_max - lds_fixed) / cell_bytes - 4;     // ncell_pad rounds up to a multiple of 4
        uint32_t T = 1; while ((uint64_t)(T + 1) * (T + 1) <= max_cells) T++;
        if (cplx && T > SIMKA_PAIR_TN / 2) T = SIMKA_PAIR_TN / 2;
        pc.tile = T; pc.ntiles = (N + T - 1) / T; pc.ncell = T * T;
    }
    pc.ncell_pad = (pc.ncell + 3u) & ~3u;
    pl.ntp = pc.ntiles * (pc.ntiles + 1) / 2;
    pl.lds_pairs = lds_fixed + (size_t)pc.ncell_pad * cell_bytes;
    pl.small_block = pc.ntiles == 1 && N <= 32;     // few pairs per span: more, smaller blocks
    const uint32_t per_cu = (uint32_t)std::min<size_t>(pl.small_block ? 6 : 2, std::max<size_t>(1, 128 / ((pl.lds_pairs + 1279) / 1280)));      // (LDS granules of 1280 bytes, 128 per CU)
    pl.nblk = (uint32_t)ctx->num_cus * per_cu;
}

// grow-only device buffer of the tile-major path; false (and the buffer left as it was) when the allocation fails
template <typename T>
static bool tm_reserve(simka_ctx *ctx, T **p, uint64_t *cap, uint64_t need) {
    if (*cap >= need && *p) return true;
    if (*p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const uint64_t n = need + need / 4 + 64;
    if (dev_alloc(p, n) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
    *cap = n;
    return true;
}

// nb_entries / nb_spans: what the CSR holds if the caller knows (sort-based merge); 0: read from the device cursors
static void pair_launch(simka_ctx *ctx, const PairLaunch &pl, const SimkaSpan *spans, const ull *cursors, const ull *entries, const uint32_t *groups,
                        const SimkaSpan *huge, ull *acc, bool have_spans = true, uint64_t nb_entries = 0, uint64_t nb_spans = 0) {
    const SimkaPairCfg &pc = pl.pc;
    // N beyond one LDS tile: reorder the spans tile-major once (k_tile_major), then every tile pair stages only its two segments
    const char *nt_env = simka_test_knob("SIMKA_TM_MAX_TILES");          // tests: force the fallback below a smaller tile count
    const uint32_t nt_max = nt_env ? std::min<uint32_t>(KTM_NT_MAX, (uint32_t)atoi(nt_env)) : (uint32_t)KTM_NT_MAX;
    bool tile_major = have_spans && pc.ntiles > 1 && pc.ntiles <= nt_max && tile_major_enabled();
    if (tile_major) {
        if (!nb_spans) {      // (one small download per merge batch; the batches are large)
            ull cur[4] = { 0, 0, 0, 0 };
            if (hipMemcpyAsync(cur, cursors, 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) tile_major = false;
            nb_entries = cur[0]; nb_spans = cur[2];
            if (spans == ctx->d_spans && nb_spans > ctx->span_cap) { have_spans = false; tile_major = false; }      // (k_group ran out of span slots and flagged it: nothing to pair up)
        }
        if (tile_major && nb_spans == 0) have_spans = false;
        const bool cplx = pc.nacc64 != 0;
        if (tile_major && have_spans)
            tile_major = tm_reserve(ctx, &ctx->d_tm_ent, &ctx->tm_ent_cap, nb_entries + 16) &&
                         (!cplx || tm_reserve(ctx, &ctx->d_tm_p, &ctx->tm_p_cap, nb_entries + 16)) &&
                         tm_reserve(ctx, &ctx->d_tm_off, &ctx->tm_off_cap, nb_spans * (uint64_t)(pc.ntiles + 1) + 16);
    }
    ull *work = (have_spans && pl.ntp == 1 && (pl.small_block || pc.ntiles == 1)) ? ctx->d_work : nullptr;      // one tile: spans handed out dynamically
    if (work && hipMemsetAsync(work, 0, 8, ctx->stream) != hipSuccess) work = nullptr;
    if (have_spans) launch_timed(ctx, KID_PAIRS, [&] {
        if (pl.small_block)
            SIMKA_LAUNCH((k_pairs<false, K4_BLOCK_SMALL>), dim3(pl.nblk, pl.ntp), dim3(K4_BLOCK_SMALL), pl.lds_pairs, ctx->stream, spans, cursors, entries, groups, pc, acc, work);
#ifdef SIMKA_DEBUG_KNOBS
        else if (pc.ntiles == 1 && simka_exp_knob("SIMKA_PAIRS_BLOCK512"))      // experiments: eight waves per block instead of sixteen
            SIMKA_LAUNCH((k_pairs<false, 512>), dim3(pl.nblk, pl.ntp), dim3(512), pl.lds_pairs, ctx->stream, spans, cursors, entries, groups, pc, acc, work);
#endif
        else if (pc.ntiles == 1)
            SIMKA_LAUNCH((k_pairs<false, K4_BLOCK_BIG>), dim3(pl.nblk, pl.ntp), dim3(K4_BLOCK_BIG), pl.lds_pairs, ctx->stream, spans, cursors, entries, groups, pc, acc, work);
        else if (tile_major) {
            const uint32_t grid_tm = (uint32_t)std::min<uint64_t>((nb_spans + KTM_WAVES - 1) / KTM_WAVES, (uint64_t)ctx->num_cus * 4);
            SIMKA_LAUNCH(k_tile_major, dim3(grid_tm), dim3(64 * KTM_WAVES), 0, ctx->stream, spans, cursors, entries, groups, pc, ctx->d_tm_ent, ctx->d_tm_p, ctx->d_tm_off);
            if (pc.nacc64) SIMKA_LAUNCH(k_pairs_tm<true>, dim3(pl.nblk, pl.ntp), dim3(K4_BLOCK_BIG), pl.lds_pairs, ctx->stream, spans, cursors, (const ull *)ctx->d_tm_ent,
                                              (const ktm_p_t *)ctx->d_tm_p, (const uint32_t *)ctx->d_tm_off, pc, acc);
            else SIMKA_LAUNCH(k_pairs_tm<false>, dim3(pl.nblk, pl.ntp), dim3(K4_BLOCK_BIG), pl.lds_pairs, ctx->stream, spans, cursors, (const ull *)ctx->d_tm_ent,
                                    (const ktm_p_t *)ctx->d_tm_p, (const uint32_t *)ctx->d_tm_off, pc, acc);
        } else {
            // (the tile-major buffers could not be had, or too many tiles: the scan-and-compact kernel with its own LDS layout;
            // the spans were built for pc.span_cap entries, which its tile geometry keeps)
            PairLaunch lg = pl;
            if (tile_major_enabled()) pair_setup(ctx, lg, true, pl.pc.span_cap);
            SIMKA_LAUNCH((k_pairs<true, K4_BLOCK_BIG>), dim3(lg.nblk, lg.ntp), dim3(K4_BLOCK_BIG), lg.lds_pairs, ctx->stream, spans, cursors, entries, groups, lg.pc, acc, (ull *)nullptr);
        }
    });
#ifdef SIMKA_PHASE_PROF
    group_phase_report();
    pairs_phase_report();
#endif
    if (huge)
        launch_timed(ctx, KID_PAIRS_GLOBAL, [&] {
            SIMKA_LAUNCH(k_pairs_global, dim3(64, 64), dim3(256), 0, ctx->stream, huge, cursors, entries, pc, acc);
        });
}

static int complex_finish(simka_ctx *ctx, const SimkaPairCfg &pc);


// the two-word forms of the batch gather / import (kmer_size >= 32): high and low words travel in separate buffers
SIMKA_EXPORT int simka_gather_samples_device_wide(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const uint64_t *out_offsets, void *d_keys_hi,
                                                  void *d_keys_lo, void *d_counts) {
    if (!ctx || !ctx->wide) return SIMKA_ERR_INVALID;
    if (nb == 0) return SIMKA_OK;
    if (!samples || !out_offsets || !d_keys_hi || !d_keys_lo || !d_counts) return ctx->fail(SIMKA_ERR_INVALID, "simka_gather_samples_device_wide: NULL argument");
    for (uint32_t j = 0; j < nb; j++)
        if (samples[j] >= ctx->cfg.nb_samples || !ctx->counted[samples[j]]) return ctx->fail(SIMKA_ERR_STATE, "simka_gather_samples_device_wide: sample %u not counted", samples[j]);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const int wrc = simka_wide_gather(ctx->wide, samples, nb, wide_log2_parts(ctx), out_offsets, d_keys_hi, d_keys_lo, d_counts);
    return wrc ? wide_fail(ctx, wrc) : SIMKA_OK;
}

SIMKA_EXPORT int simka_import_samples_device_wide(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const simka_sample_totals *totals,
                                                  const uint64_t *sample_offsets, const uint64_t *sample_records, const void *d_keys_hi, const void *d_keys_lo,
                                                  const void *d_counts) {
    if (!ctx || !ctx->wide || (nb && (!samples || !totals || !sample_offsets || !sample_records))) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_import_samples_device_wide: merge already ran");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    std::vector<ull> tot((size_t)SIMKA_NB_TOTALS * N);
    HIPCHK(hipMemcpyAsync(tot.data(), ctx->d_stats + stats_off_tot(N, fl, 0), tot.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t j = 0; j < nb; j++) {
        const uint32_t s = samples[j];
        if (s >= N) return ctx->fail(SIMKA_ERR_INVALID, "simka_import_samples_device_wide: sample index %u out of range", s);
        if (ctx->counted[s]) return ctx->fail(SIMKA_ERR_STATE, "simka_import_samples_device_wide: sample %u was already counted", s);
        const uint64_t o = sample_offsets[j], n = sample_records[j];
        const int wrc = simka_wide_import_words(ctx->wide, s, (const ull *)d_keys_hi + o, (const ull *)d_keys_lo + o, (const uint32_t *)d_counts + o, n);
        if (wrc) return wide_fail(ctx, wrc);
        tot[(size_t)SIMKA_TOT_D * N + s] = totals[j].nb_distinct; tot[(size_t)SIMKA_TOT_N * N + s] = totals[j].nb_kmers;
        tot[(size_t)SIMKA_TOT_Q * N + s] = totals[j].sum_sq; tot[(size_t)SIMKA_TOT_KOCC * N + s] = totals[j].kmer_occurrences;
        tot[(size_t)SIMKA_TOT_DALL * N + s] = totals[j].distinct_all;
        if (ctx->d_hist && n) {      // -complex-dist: histogram of the imported solid counts
            std::vector<uint32_t> hc(n);
            HIPCHK(hipMemcpyAsync(hc.data(), (const uint32_t *)d_counts + o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<ull> hist(SIMKA_HIST_MAX, 0);
            std::vector<uint32_t> ovf;
            for (uint32_t c : hc) { if (c < SIMKA_HIST_MAX) hist[c]++; else { ovf.push_back(s); ovf.push_back(c); } }
            HIPCHK(hipMemcpyAsync(ctx->d_hist + (uint64_t)s * SIMKA_HIST_MAX, hist.data(), SIMKA_HIST_MAX * 8, hipMemcpyHostToDevice, ctx->stream));
            if (!ovf.empty()) {
                ull novf = 0;
                HIPCHK(hipMemcpyAsync(&novf, ctx->d_ovf_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                const ull add = ovf.size() / 2;
                if (novf + add <= ctx->ovf_cap) HIPCHK(hipMemcpyAsync(ctx->d_ovf_list + 2 * novf, ovf.data(), ovf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                novf += add;
                HIPCHK(hipMemcpyAsync(ctx->d_ovf_cursor, &novf, 8, hipMemcpyHostToDevice, ctx->stream));
            }
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        ctx->nb_reads[s] = totals[j].nb_reads;
        ctx->counted[s] = 1;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, fl, 0), tot.data(), tot.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}

// ---- merge side ---------------------------------------------------------------------------
SIMKA_EXPORT int simka_merge(simka_ctx *ctx) {
    if (!ctx) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples;
    for (uint32_t s = 0; s < N; s++) if (!ctx->counted[s]) return ctx->fail(SIMKA_ERR_STATE, "simka_merge: sample %u has not been counted", s);
    if (ctx->merged) return ctx->fail(SIMKA_ERR_STATE, "simka_merge: already merged");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    simka_trace::set_arena(ctx->trace_id, 0xffffffffu, ctx->arena_mapped, ctx->arena_hi, ctx->arena_cap);      // (sample ~0: the merge)
    int rc = resolve_pending(ctx);
    if (rc) return rc;
    rc = check_device_error(ctx);
    if (rc) return rc;
    ctx->merged = true;
    if (ctx->wide) {
        if (N < 2) {
            HIPCHK(hipMemcpyAsync(ctx->d_stats, ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, SIMKA_TOT_D), 8, hipMemcpyDeviceToDevice, ctx->stream));
            return SIMKA_OK;
        }
        PairLaunch pl;
        pair_setup(ctx, pl);
        SimkaWideCsr csr;
        const int wrc = simka_wide_merge(ctx->wide, pl.pc.span_cap, &csr);          // the N-way merge by sorting: CSR of the shared k-mers
        if (wrc) return wide_fail(ctx, wrc);
        const ull head[2] = { csr.nb_distinct, csr.nb_shared };
        HIPCHK(hipMemcpyAsync(ctx->d_stats, head, 16, hipMemcpyHostToDevice, ctx->stream));
        if (csr.nb_spans || csr.nb_huge) pair_launch(ctx, pl, csr.spans, csr.cursors, csr.entries, csr.groups, csr.nb_huge ? csr.huge : nullptr, (ull *)ctx->d_stats + stats_off_acc(N, 0), csr.nb_spans != 0,
                                                         csr.nb_entries, csr.nb_spans);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->cfg.dist_flags & SIMKA_DIST_COMPLEX) return complex_finish(ctx, pl.pc);
        return SIMKA_OK;
    }
    if (!ctx->geometry_ready) return SIMKA_OK;            // only empty samples
    if (N < 2) {   // nothing to pair up: the union of k-mers is the sample's own solid spectrum, none of it shared
        HIPCHK(hipMemcpyAsync(ctx->d_stats, ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, SIMKA_TOT_D), 8, hipMemcpyDeviceToDevice, ctx->stream));
        return SIMKA_OK;
    }
    const uint64_t nparts = ctx->nparts;

    // records per partition over all samples -> host scan (also drives the batching)
    HIPCHK(hipMemsetAsync(ctx->d_part_total + nparts, 0, 8, ctx->stream));
    launch_timed(ctx, KID_PART_TOTALS, [&] {
        SIMKA_LAUNCH(k_part_totals, dim3((uint32_t)((nparts + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_fcnt, N,
                           nparts, ctx->d_part_total);
    });
    if (ctx->h_part_n < 2 * (nparts + 1)) {
        if (ctx->h_part) (void)hipHostFree(ctx->h_part);
        ctx->h_part = nullptr; ctx->h_part_n = 0;
        if (hipHostMalloc((void **)&ctx->h_part, 2 * (nparts + 1) * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return ctx->fail(SIMKA_ERR_NOMEM, "simka_merge: cannot allocate %llu bytes of pinned host memory", (unsigned long long)(2 * (nparts + 1) * 8)); }
        ctx->h_part_n = 2 * (nparts + 1);
    }
    ull *const ptot = ctx->h_part, *const poff = ctx->h_part + nparts + 1;
    HIPCHK(hipMemcpyAsync(ptot, ctx->d_part_total, (nparts + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // A (sample, partition) segment beyond 65535 records (a small user-set log2_partitions; a hot partition): the 16-bit rows the count
    // kernels left cannot index it -- every batch gets 32-bit rows from k_segment_rows<true> and k_group reads those (slower, rare).
    const bool rows32 = ptot[nparts] > 0xffffull;
    ull total = 0, maxpart = 0, nonempty = 0;
    for (uint64_t p = 0; p < nparts; p++) { poff[p] = total; total += ptot[p]; maxpart = std::max(maxpart, ptot[p]); nonempty += ptot[p] ? 1 : 0; }
    poff[nparts] = total;
    if (total == 0) return SIMKA_OK;
    HIPCHK(hipMemcpyAsync(ctx->d_part_off, poff, (nparts + 1) * 8, hipMemcpyHostToDevice, ctx->stream));      // (pinned: really asynchronous -- poff is not written again before the merge's last synchronize)

    // sub-range bits: at most K3_TARGET records per k_group round on average (a round hashes up to K3_CAP)
    SimkaKeyCfg key = ctx->key;
    uint32_t t = ctx->cfg.log2_subranges;
    // (more than K3_BLOCK samples: k_group<512> -- twice the records per round, every sample in one tile)
    const bool group_big = N > (uint32_t)K3_BLOCK || (simka_exp_knob("SIMKA_GROUP_BIG") && atoi(simka_exp_knob("SIMKA_GROUP_BIG")) > 0);
    const uint32_t g_mul = group_big ? 2u : 1u;
    if (t == 0) t = ceil_log2_u64((total / std::max<ull>(nonempty, 1) + K3_TARGET * g_mul - 1) / (K3_TARGET * g_mul));
    t = std::min<uint32_t>(t, SIMKA_SEG_BITS);            // (the segments are ordered by that many key bits; k_group splits larger sub-ranges on further bits)
    t = std::min<uint32_t>(t, key.W);
    key.t = t; ctx->key.t = t;
    const uint32_t nsub = 1u << t;

    // bounded CSR buffers, processed in batches of consecutive partitions
    // batch size: up to 2^29 records (6 GB of CSR buffers; larger batches = fewer launches and shorter tails: C3 merge side 420 ->
    // 398 ms against 2^27), never more than half of what is free now (on top of the buffers a previous merge left)
    uint64_t cap_mem = (uint64_t)1 << 27;
    if (total > cap_mem && !ctx->cfg.csr_capacity) { size_t fr = 0, tot_ = 0; if (hipMemGetInfo(&fr, &tot_) == hipSuccess) cap_mem = std::max<uint64_t>((uint64_t)1 << 24, ctx->merge_cap + (uint64_t)(fr / 2) / 32); }
    uint64_t cap = ctx->cfg.csr_capacity ? ctx->cfg.csr_capacity : std::min<uint64_t>(total, std::min<uint64_t>((uint64_t)1 << 29, cap_mem));
    cap = std::max<uint64_t>(cap, maxpart);
    if (cap >= ((uint64_t)1 << 32)) cap = ((uint64_t)1 << 32) - 1;
    if (maxpart > cap) return ctx->fail(SIMKA_ERR_NOMEM, "simka_merge: one partition holds %llu records, more than the merge buffer", maxpart);
    const uint64_t max_parts_batch = std::min<uint64_t>(nparts, std::min<uint64_t>((uint64_t)1 << 16, std::max<uint64_t>(1, ((uint64_t)1 << (rows32 ? 22 : 26)) / N)));
    const uint64_t fb_cap = max_parts_batch * nsub;
    const uint32_t grid_group = (uint32_t)ctx->num_cus * (simka_exp_knob("SIMKA_GROUP_BPC") ? (uint32_t)atoi(simka_exp_knob("SIMKA_GROUP_BPC")) : (group_big ? 2 : 5));
    // spans: one per work item and open-span break, plus the rounds of the sub-ranges that k_group has to split further (a round holds
    // K3_PRESPLIT / 2 .. K3_PRESPLIT records unless the key bits are skewed; beyond the capacity the merge fails cleanly)
    const uint64_t span_cap = fb_cap * 2 + 4096 + (uint64_t)grid_group * K3_SLAB_SPAN + cap / (K3_PRESPLIT / 4);
    const uint64_t csr_cap = cap + (uint64_t)grid_group * K3_SLAB_ENT;     // slab reservation leaves unused tails
    if (ctx->merge_cap < cap) {
        void *old[] = { ctx->d_entries, ctx->d_groups };
        for (void *p : old) if (p) HIPCHK(hipFree(p));
        ctx->d_entries = nullptr; ctx->d_groups = nullptr;
        if (dev_alloc(&ctx->d_entries, csr_cap) != hipSuccess || dev_alloc(&ctx->d_groups, csr_cap) != hipSuccess)
            return ctx->fail(SIMKA_ERR_NOMEM, "simka_merge: cannot allocate merge buffers for %llu records", (unsigned long long)cap);
        ctx->merge_cap = cap;
    }
    const uint64_t seg_cap = max_parts_batch * N;
    struct Rows32 { ull *abs = nullptr; uint4 *rows = nullptr; ~Rows32() { if (abs) (void)hipFree(abs); if (rows) (void)hipFree(rows); } } r32;
    if (rows32) {
        if (dev_alloc(&r32.abs, seg_cap) != hipSuccess || dev_alloc(&r32.rows, seg_cap * 4) != hipSuccess)
            return ctx->fail(SIMKA_ERR_NOMEM, "simka_merge: cannot allocate the 32-bit segment rows (%llu segments)", (unsigned long long)seg_cap);
    } else
    if (!ctx->seg_all && ctx->seg_cap < seg_cap) {
        if (ctx->d_seg_abs) HIPCHK(hipFree(ctx->d_seg_abs)); if (ctx->d_seg_rows) HIPCHK(hipFree(ctx->d_seg_rows));
        ctx->d_seg_abs = nullptr; ctx->d_seg_rows = nullptr;
        HIPCHK(dev_alloc(&ctx->d_seg_abs, seg_cap)); HIPCHK(dev_alloc(&ctx->d_seg_rows, seg_cap * 2)); ctx->seg_cap = seg_cap;
    }
    if (ctx->span_cap < span_cap) { if (ctx->d_spans) HIPCHK(hipFree(ctx->d_spans)); ctx->d_spans = nullptr; HIPCHK(dev_alloc(&ctx->d_spans, span_cap)); ctx->span_cap = span_cap; }

    PairLaunch pl;
    pair_setup(ctx, pl);
    const SimkaPairCfg &pc = pl.pc;
    const uint32_t flags = ctx->cfg.dist_flags;
    // k-mers shared by more than K3_CAP samples (possible only when N > K3_CAP) leave k_group on a list of their own
    const uint64_t huge_cap = N > K3_CAP ? cap / K3_CAP + 16 : 1;
    if (ctx->huge_cap < huge_cap) { if (ctx->d_huge) HIPCHK(hipFree(ctx->d_huge)); ctx->d_huge = nullptr; HIPCHK(dev_alloc(&ctx->d_huge, huge_cap)); ctx->huge_cap = huge_cap; }

    SimkaMergeIn in;
    in.solid_keys = ctx->d_solid_keys; in.solid_counts = ctx->d_solid_counts; in.sample_base = ctx->d_sample_base;
    in.foff = ctx->d_foff; in.fcnt = ctx->d_fcnt; in.nb_samples = N; in.nparts = nparts;
    SimkaCsrOut co;
    co.entries = ctx->d_entries; co.groups = ctx->d_groups; co.spans = ctx->d_spans; co.cursors = ctx->d_cursors;
    co.cap_entries = csr_cap; co.cap_groups = csr_cap; co.cap_spans = span_cap; co.span_cap = pc.span_cap; co.huge = ctx->d_huge; co.cap_huge = huge_cap; co.glob = (ull *)ctx->d_stats; co.err = ctx->d_err;
    const uint32_t min_share = 2;   // -complex-dist would need 1 (ref: src/SimkaMerge.cpp:1317)
    const size_t lds_group = K3_LDS_BYTES(K3_BLOCK * g_mul);        // 31 688 bytes (five blocks per CU: the LDS granule is 1280 bytes) / 64 456
    ull *acc = (ull *)ctx->d_stats + stats_off_acc(N, 0);

    // A partition shard owns the partitions p = g + G i; the others are empty.  With the index of every partition in place the merge
    // walks the owned ones only (work item i of a batch = partition pb + i * stride): k_group spent as long on the empty partitions as
    // on the owned ones (C3 on eight ranks: 110 ms of k_group per rank whatever G).
    const bool strided = ctx->cfg.shard_count > 1 && ctx->seg_all && !rows32 && !ctx->seg_dirty;
    const uint64_t stride = strided ? ctx->cfg.shard_count : 1, first = strided ? ctx->cfg.shard_index : 0;
    const uint64_t nwork = nparts > first ? (nparts - first + stride - 1) / stride : 0;
    uint64_t ib = 0;
    while (ib < nwork) {
        uint64_t ie = ib; ull recs = 0;
        while (ie < nwork && ie - ib < max_parts_batch && recs + ptot[first + ie * stride] <= cap) { recs += ptot[first + ie * stride]; ie++; }
        if (ie == ib) return ctx->fail(SIMKA_ERR_NOMEM, "simka_merge: batching failed at partition %llu", (unsigned long long)(first + ib * stride));
        const uint64_t pb = first + ib * stride;
        if (recs) {
            const uint32_t np = (uint32_t)(ie - ib);
            const uint32_t nfb = np * nsub;
            HIPCHK(hipMemsetAsync(ctx->d_cursors, 0, 32, ctx->stream));
            // the index of the batch's segments: written by the count kernels, or built here (imported spectra; too many segments to keep all)
            ull *b_abs = rows32 ? r32.abs : ctx->d_seg_abs + (ctx->seg_all ? pb * N : 0);
            uint4 *b_rows = rows32 ? r32.rows : ctx->d_seg_rows + (ctx->seg_all ? pb * N * 2 : 0);
            if (rows32 || !ctx->seg_all || ctx->seg_dirty)
                launch_timed(ctx, KID_SEG_ROWS, [&] {
                    SIMKA_LAUNCH(rows32 ? k_segment_rows<true> : k_segment_rows<false>, dim3((uint32_t)std::min<uint64_t>(((uint64_t)np * N + 3) / 4, (uint64_t)ctx->num_cus * 8)), dim3(256), 0, ctx->stream, in, key, pb, np,
                                       b_abs, b_rows, ctx->d_err);
                });
            launch_timed(ctx, KID_GROUP, [&] {
                auto kg = group_big ? (rows32 ? k_group<2 * K3_BLOCK, true> : k_group<2 * K3_BLOCK, false>) : (rows32 ? k_group<K3_BLOCK, true> : k_group<K3_BLOCK, false>);
                SIMKA_LAUNCH(kg, dim3(std::max<uint32_t>(32u, std::min<uint32_t>((np * 4u + 31u) / 32u * 32u, grid_group / 32u * 32u))),
                                   dim3(group_big ? 2 * K3_BLOCK : K3_BLOCK), lds_group, ctx->stream, in, (const ull *)b_abs, (const uint16_t *)b_rows, np, key, min_share, co, (uint32_t)stride);
            });
            if (simka_exp_knob("SIMKA_DEBUG_MERGE")) {
                ull cur[4]; HIPCHK(hipMemcpyAsync(cur, ctx->d_cursors, 32, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream));
                std::vector<SimkaSpan> hs(cur[2]); HIPCHK(hipMemcpy(hs.data(), ctx->d_spans, cur[2] * sizeof(SimkaSpan), hipMemcpyDeviceToHost));
                ull empty = 0, ent = 0, grp = 0, full = 0; for (auto &sp : hs) { if (!sp.ngrp) empty++; ent += sp.nent; grp += sp.ngrp; if (sp.nent > 3500) full++; }
                fprintf(stderr, "merge batch: %llu records, span slots %llu (empty %llu, > 3500 entries %llu), entries %llu (slab cursor %llu), groups %llu, entries per real span %.0f\n",
                        (unsigned long long)recs, cur[2], empty, full, ent, cur[0], grp, (double)ent / std::max<ull>(1, cur[2] - empty));
            }
            pair_launch(ctx, pl, ctx->d_spans, ctx->d_cursors, ctx->d_entries, ctx->d_groups, N > K3_CAP ? ctx->d_huge : nullptr, acc);
        }
        ib = ie;
    }
    HIPCHK(hipGetLastError());
    int rcd = check_device_error(ctx);
    if (rcd) return rcd;
    if (flags & SIMKA_DIST_COMPLEX) return complex_finish(ctx, pc);
    return SIMKA_OK;
}

static int complex_finish(simka_ctx *ctx, const SimkaPairCfg &pc) {
    const uint32_t N = ctx->cfg.nb_samples;
    {
        // Whittaker's one-sided terms  sum_{k-mers of i} g(c, N_j),  g(c,M) = |(int)(u64)(c*M)|  (ref: src/core/SimkaAlgorithm.hpp:481,512),
        // from this shard's histogram of solid counts and the GLOBAL N_j; k_pairs already subtracted g for the both-present pairs.
        std::vector<ull> hist((size_t)N * SIMKA_HIST_MAX), totn(N), whit(pc.nb_pairs);
        ull novf = 0;
        HIPCHK(hipMemcpyAsync(hist.data(), ctx->d_hist, hist.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(totn.data(), pc.tot_n, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&novf, ctx->d_ovf_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
        ull *d_whit = (ull *)ctx->d_stats + stats_off_acc(N, pc.nacc32 + 0);
        HIPCHK(hipMemcpyAsync(whit.data(), d_whit, pc.nb_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> ovf(2 * novf);
        if (novf > ctx->ovf_cap) {
            // the fixed-size list overflowed (the cursor kept counting): rebuild it at its exact size from the resident spectra
            if (ctx->wide) return ctx->fail(SIMKA_ERR_OVERFLOW, "more than %llu k-mers with a count >= %d: complex-dist histogram list exhausted (k >= 32)", (unsigned long long)ctx->ovf_cap, SIMKA_HIST_MAX);
            uint32_t *d_list = nullptr; ull *d_cur = nullptr;
            if (dev_alloc(&d_list, 2 * novf) != hipSuccess || dev_alloc(&d_cur, 1) != hipSuccess) { if (d_list) (void)hipFree(d_list); return ctx->fail(SIMKA_ERR_NOMEM, "cannot allocate the list of %llu large counts", (unsigned long long)novf); }
            HIPCHK(hipMemsetAsync(d_cur, 0, 8, ctx->stream));
            SIMKA_LAUNCH(k_big_counts, dim3((uint32_t)std::min<uint64_t>(ctx->nparts, 1024), N), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_solid_counts,
                               (const ull *)ctx->d_sample_base, (const uint32_t *)ctx->d_foff, (const uint32_t *)ctx->d_fcnt, (uint32_t)ctx->nparts, d_list, d_cur, novf);
            ull got = 0;
            HIPCHK(hipMemcpyAsync(&got, d_cur, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(ovf.data(), d_list, novf * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_list); (void)hipFree(d_cur);
            if (got != novf) return ctx->fail(SIMKA_ERR_STATE, "complex-dist: %llu large counts on the list, %llu in the spectra", (unsigned long long)novf, (unsigned long long)got);
        } else if (novf) HIPCHK(hipMemcpy(ovf.data(), ctx->d_ovf_list, novf * 8, hipMemcpyDeviceToHost));
        std::vector<std::vector<std::pair<uint32_t, ull>>> cnts(N);        // per sample: (count, #k-mers)
        for (uint32_t i = 0; i < N; i++)
            for (uint32_t cc = 0; cc < SIMKA_HIST_MAX; cc++) if (hist[(size_t)i * SIMKA_HIST_MAX + cc]) cnts[i].push_back({cc, hist[(size_t)i * SIMKA_HIST_MAX + cc]});
        for (ull w = 0; w < novf; w++) cnts[ovf[2 * w]].push_back({ovf[2 * w + 1], 1});
        auto gsum = [&](uint32_t i, uint32_t j) {
            ull acc_ = 0;
            const double M = (double)totn[j];
            for (auto &pr : cnts[i]) {
                const int t_ = (int)(ull)((double)pr.first * M);
                const int r_ = (t_ == (int)0x80000000) ? t_ : (t_ < 0 ? -t_ : t_);
                acc_ += (ull)(long long)r_ * pr.second;
            }
            return acc_;
        };
        uint64_t cell = 0;
        for (uint32_t i = 0; i < N; i++) for (uint32_t j = i + 1; j < N; j++, cell++) whit[cell] += gsum(i, j) + gsum(j, i);
        HIPCHK(hipMemcpyAsync(d_whit, whit.data(), pc.nb_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return SIMKA_OK;
}

// ---- statistics ---------------------------------------------------------------------------
SIMKA_EXPORT int simka_stats_device_buffer(simka_ctx *ctx, void **p, uint64_t *n) {
    if (!ctx || !p || !n) return SIMKA_ERR_INVALID;
    *p = ctx->d_stats; *n = stats_off_derived(ctx->cfg.nb_samples, ctx->cfg.dist_flags);   // pair arrays + totals (derived tail is host-only)
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_stats_device_ranges(simka_ctx *ctx, void **head, uint64_t *nb_head, void **totals, uint64_t *nb_totals) {
    if (!ctx || !head || !nb_head || !totals || !nb_totals) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    *head = ctx->d_stats; *nb_head = stats_off_tot(N, fl, 0);
    *totals = ctx->d_stats + stats_off_tot(N, fl, 0); *nb_totals = (uint64_t)SIMKA_NB_TOTALS * N;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_totals_download(simka_ctx *ctx, uint64_t *out) {
    if (!ctx || !out) return SIMKA_ERR_INVALID;
    { int rcp = resolve_pending(ctx); if (rcp) return rcp; }
    const uint32_t N = ctx->cfg.nb_samples;
    HIPCHK(hipMemcpyAsync(out, ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, 0), (size_t)SIMKA_NB_TOTALS * N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_totals_upload(simka_ctx *ctx, const uint64_t *in) {
    if (!ctx || !in) return SIMKA_ERR_INVALID;
    const uint32_t N = ctx->cfg.nb_samples;
    HIPCHK(hipMemcpyAsync(ctx->d_stats + stats_off_tot(N, ctx->cfg.dist_flags, 0), in, (size_t)SIMKA_NB_TOTALS * N * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_stats_download(simka_ctx *ctx, uint64_t *h, uint64_t n, simka_stats_view *view) {
    if (!ctx || !h) return SIMKA_ERR_INVALID;
    if (n < ctx->stats_n) return ctx->fail(SIMKA_ERR_INVALID, "simka_stats_download: buffer too small (%llu < %llu)", (unsigned long long)n, (unsigned long long)ctx->stats_n);
    HIPCHK(hipMemcpyAsync(h, ctx->d_stats, ctx->stats_n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (view) return simka_stats_describe(ctx->cfg.nb_samples, ctx->cfg.dist_flags, h, n, view);
    return SIMKA_OK;
}

// ---- RCCL: the cross-GPU reduction of SimkaStatistics (operator+=, ref: src/core/SimkaDistance.cpp:156-213) ------------
// One communicator per GPU (one process per GPU, or one host thread per GPU inside `simka -nb-gpus`).  The accumulators of a
// context are ONE flat u64 buffer, so the reduction is a single ncclAllReduce(sum, uint64) on the context's stream.
// RCCL is loaded at the first simka_comm_* call (dlopen): a single-GPU installation needs neither the library nor its header, and
// a context never touches it.  The handful of ABI constants below are those of RCCL's nccl.h.
typedef void *ncclComm_t;
typedef struct { char internal[SIMKA_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclUint64 = 5 };       // ncclDataType_t
enum { ncclSum = 0 };                         // ncclRedOp_t
struct RcclApi {
    void *lib = nullptr;
    std::string err, how, path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static const RcclApi *rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // ONE RCCL per process: a copy that is already loaded (a Python caller's torch brings its own librccl.so and has bootstrapped the
        // process group with it) serves the data path as well -- RTLD_NOLOAD finds it by its soname whatever directory it came from.
        // SIMKA_RCCL_PATH names a file explicitly; otherwise the usual search.
        for (const char *name : { "librccl.so.1", "librccl.so" }) { api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (api.lib) { api.how = "already loaded in the process"; break; } }
        if (!api.lib) { const char *pth = getenv("SIMKA_RCCL_PATH"); if (pth && *pth) { api.lib = dlopen(pth, RTLD_NOW | RTLD_LOCAL); if (api.lib) api.how = "SIMKA_RCCL_PATH"; } }
        if (!api.lib)
            for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) { api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (api.lib) { api.how = "loaded by the library"; break; } }
        if (!api.lib) { api.err = "RCCL is not installed (librccl.so not found): multi-GPU collectives are unavailable"; return; }
        bool ok = true;
        auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p) { ok = false; api.err = std::string("librccl.so lacks ") + n; } return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy"); api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.Send = (decltype(api.Send))sym("ncclSend"); api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart"); api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.lib); api.lib = nullptr; return; }
        Dl_info di;
        if (dladdr((void *)api.AllReduce, &di) && di.dli_fname) api.path = di.dli_fname;
    });
    return &api;
}

struct simka_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nb_ranks = 1, device = 0;
    std::string err;
};
static thread_local std::string g_comm_error;

#define NCCLCHK(c, call)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) { (c)->err = std::string(#call) + " failed: " + rccl()->GetErrorString(r_); return SIMKA_ERR_HIP; } \
    } while (0)

SIMKA_EXPORT int simka_comm_library(char *path, uint64_t cap) {
    const RcclApi *R = rccl();
    if (!R->lib) { g_comm_error = R->err; return SIMKA_ERR_UNSUPPORTED; }
    if (!path || cap == 0) return SIMKA_ERR_INVALID;
    snprintf(path, (size_t)cap, "%s (%s)", R->path.empty() ? "?" : R->path.c_str(), R->how.c_str());
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_comm_unique_id(uint8_t *id) {
    if (!id) return SIMKA_ERR_INVALID;
    const RcclApi *R = rccl();
    if (!R->lib) { g_comm_error = R->err; return SIMKA_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    const ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId failed: ") + R->GetErrorString(r); return SIMKA_ERR_HIP; }
    memcpy(id, &u, sizeof u);
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_comm_create(const uint8_t *id, int nb_ranks, int rank, int device, simka_comm **out) {
    if (!id || !out || nb_ranks < 1 || rank < 0 || rank >= nb_ranks) { g_comm_error = "simka_comm_create: bad argument"; return SIMKA_ERR_INVALID; }
    const RcclApi *R = rccl();
    if (!R->lib) { g_comm_error = R->err; return SIMKA_ERR_UNSUPPORTED; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_error = "simka_comm_create: hipSetDevice failed"; return SIMKA_ERR_HIP; }
    simka_comm *c = new simka_comm();
    c->rank = rank; c->nb_ranks = nb_ranks; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = R->CommInitRank(&c->comm, nb_ranks, u, rank);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank failed: ") + R->GetErrorString(r); delete c; return SIMKA_ERR_HIP; }
    *out = c;
    return SIMKA_OK;
}

SIMKA_EXPORT void simka_comm_destroy(simka_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

SIMKA_EXPORT const char *simka_comm_last_error(const simka_comm *c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

SIMKA_EXPORT int simka_comm_info(const simka_comm *c, int *rank, int *nb_ranks) {
    if (!c) return SIMKA_ERR_INVALID;
    if (rank) *rank = c->rank;
    if (nb_ranks) *nb_ranks = c->nb_ranks;
    return SIMKA_OK;
}

// in-place sum of n u64 words of device memory over the ranks, on `stream`
SIMKA_EXPORT int simka_comm_allreduce_u64(simka_comm *c, void *d_buf, uint64_t n, void *stream) {
    if (!c || (!d_buf && n)) return SIMKA_ERR_INVALID;
    if (n == 0 || c->nb_ranks == 1) return SIMKA_OK;
    NCCLCHK(c, rccl()->AllReduce(d_buf, d_buf, (size_t)n, ncclUint64, ncclSum, c->comm, (hipStream_t)stream));
    return SIMKA_OK;
}

// all-to-all with uneven splits (elements of elem_bytes bytes; counts and displacements in elements, one per rank):
// grouped ncclSend / ncclRecv, i.e. point-to-point transfers over the xGMI links, all in flight at once
SIMKA_EXPORT int simka_comm_alltoallv(simka_comm *c, const void *d_send, const uint64_t *send_counts, const uint64_t *send_displs, void *d_recv,
                                      const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, void *stream) {
    if (!c || !send_counts || !send_displs || !recv_counts || !recv_displs || !elem_bytes) return SIMKA_ERR_INVALID;
    const hipStream_t st = (hipStream_t)stream;
    const char *sb = (const char *)d_send; char *rb = (char *)d_recv;
    const int me = c->rank;
    if (send_counts[me] != recv_counts[me]) { c->err = "simka_comm_alltoallv: local send and receive counts differ"; return SIMKA_ERR_INVALID; }
    const RcclApi *R = rccl();
    NCCLCHK(c, R->GroupStart());
    ncclResult_t bad = ncclSuccess;       // (a failure inside the group must not leave it open: close it first, report then)
    for (int p = 0; p < c->nb_ranks && bad == ncclSuccess; p++) {
        // the local block moves with a device copy; RCCL handles the others
        if (p == me) continue;
        if (send_counts[p]) bad = R->Send(sb + send_displs[p] * elem_bytes, (size_t)(send_counts[p] * elem_bytes), ncclUint8, p, c->comm, st);
        if (recv_counts[p] && bad == ncclSuccess) bad = R->Recv(rb + recv_displs[p] * elem_bytes, (size_t)(recv_counts[p] * elem_bytes), ncclUint8, p, c->comm, st);
    }
    const ncclResult_t ended = R->GroupEnd();
    if (bad != ncclSuccess || ended != ncclSuccess) { c->err = std::string("simka_comm_alltoallv: ncclSend / ncclRecv group failed: ") + R->GetErrorString(bad != ncclSuccess ? bad : ended); return SIMKA_ERR_HIP; }
    if (send_counts[me]) {
        const hipError_t e = hipMemcpyAsync(rb + recv_displs[me] * elem_bytes, sb + send_displs[me] * elem_bytes, (size_t)(send_counts[me] * elem_bytes), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { c->err = std::string("simka_comm_alltoallv: local copy failed: ") + hipGetErrorString(e); return SIMKA_ERR_HIP; }
    }
    return SIMKA_OK;
}

// which: 0 = the whole buffer (head + totals), 1 = head only (the totals are already global), 2 = the totals rows only
static int stats_allreduce(simka_ctx *ctx, simka_comm *c, int which) {
    if (!ctx || !c) return SIMKA_ERR_INVALID;
    { int rcp = resolve_pending(ctx); if (rcp) return rcp; }          // every lane's work must be ordered before the collective
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const uint32_t N = ctx->cfg.nb_samples, fl = ctx->cfg.dist_flags;
    uint64_t *p = ctx->d_stats; uint64_t n = stats_off_derived(N, fl);
    if (which == 1) n = stats_off_tot(N, fl, 0);
    if (which == 2) { p = ctx->d_stats + stats_off_tot(N, fl, 0); n = (uint64_t)SIMKA_NB_TOTALS * N; }
    if (ctx->wide) HIPCHK(hipStreamSynchronize(ctx->stream));
    const int rc = simka_comm_allreduce_u64(c, p, n, ctx->stream);
    if (rc) return ctx->fail(rc, "%s", c->err.c_str());
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_stats_allreduce(simka_ctx *ctx, simka_comm *c) { return stats_allreduce(ctx, c, 0); }
SIMKA_EXPORT int simka_stats_allreduce_head(simka_ctx *ctx, simka_comm *c) { return stats_allreduce(ctx, c, 1); }
SIMKA_EXPORT int simka_totals_allreduce(simka_ctx *ctx, simka_comm *c) { return stats_allreduce(ctx, c, 2); }

// ---- profiling / introspection ------------------------------------------------------------
SIMKA_EXPORT int simka_profile_enable(simka_ctx *ctx, int on) {
    if (!ctx) return SIMKA_ERR_INVALID;
    ctx->profiling = on != 0;
    ctx->prof_mask = (on == 0 || on == 1) ? ~0u : ((uint32_t)on >> 1);      // on >= 2: bit (i + 1) selects kernel i of simka_profile_get
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_profile_reset(simka_ctx *ctx) {
    if (!ctx) return SIMKA_ERR_INVALID;
    profile_collect(ctx);
    for (int i = 0; i < KID_NB; i++) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_profile_nb_kernels(simka_ctx *) { return KID_NB; }
SIMKA_EXPORT int simka_profile_get(simka_ctx *ctx, int which, const char **name, uint64_t *n, double *ms) {
    if (!ctx || which < 0 || which >= KID_NB) return SIMKA_ERR_INVALID;
    profile_collect(ctx);
    if (name) *name = (ctx->wide_hash && which == KID_SKM_COUNT) ? "k_skm_count_wide_fast" : (ctx->wide_hash && which == KID_COUNT) ? "k_skm_count_wide" :
                      (ctx->used_gather && which == KID_SKM_SPLIT) ? "k_skm_chunksort" : KID_NAMES[which];
    if (n) *n = ctx->prof_n[which];
    if (ms) *ms = ctx->prof_ms[which];
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_host_alloc(uint64_t nb_bytes, void **p) {
    if (!p) return SIMKA_ERR_INVALID;
    *p = nullptr;
    if (nb_bytes == 0) return SIMKA_OK;
    if (hipHostMalloc(p, (size_t)nb_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return SIMKA_ERR_NOMEM; }
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_host_free(void *p) {
    if (!p) return SIMKA_OK;
    return hipHostFree(p) == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}

SIMKA_EXPORT int simka_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t fr = 0, tot = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return SIMKA_ERR_HIP;
    if (free_bytes) *free_bytes = fr;
    if (total_bytes) *total_bytes = tot;
    return SIMKA_OK;
}

// device buffers of a caller that moves spectra between GPUs itself (simka_gather_samples_device on one, simka_import_samples_device
// on another): no host bounce, no HIP in the caller
SIMKA_EXPORT int simka_device_alloc(int device, uint64_t nb_bytes, void **p) {
    if (!p) return SIMKA_ERR_INVALID;
    *p = nullptr;
    if (hipSetDevice(device) != hipSuccess) return SIMKA_ERR_HIP;
    if (hipMalloc(p, nb_bytes ? nb_bytes : 1) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return SIMKA_ERR_NOMEM; }
    return SIMKA_OK;
}
SIMKA_EXPORT int simka_device_free(int device, void *p) {
    if (!p) return SIMKA_OK;
    if (hipSetDevice(device) != hipSuccess) return SIMKA_ERR_HIP;
    return hipFree(p) == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}
// the CPUs next to a device ("0-63,128-191": the kernel's local_cpulist of its PCI function), for a host that wants its loader threads and
// its pinned staging memory on the GPU's NUMA node -- on a two-socket box the copies of threads on the far socket cross the
// inter-socket link (the `simka` driver binds itself with this).  Empty string when the kernel does not say.
SIMKA_EXPORT int simka_device_cpulist(int device, char *buf, uint64_t buf_bytes) {
    if (!buf || buf_bytes < 2) return SIMKA_ERR_INVALID;
    buf[0] = 0;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return SIMKA_ERR_HIP; }
    for (char *q = bdf; *q; q++) if (*q >= 'A' && *q <= 'Z') *q = (char)(*q - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return SIMKA_OK;
    if (fgets(buf, (int)std::min<uint64_t>(buf_bytes, 1u << 20), f)) { size_t n = strlen(buf); while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' ')) buf[--n] = 0; } else buf[0] = 0;
    fclose(f);
    return SIMKA_OK;
}

// host -> device, synchronous, on a stream private to the calling thread (several loader threads upload at once)
SIMKA_EXPORT int simka_device_upload(int device, void *dst, const void *src_host, uint64_t nb_bytes) {
    if (!nb_bytes) return SIMKA_OK;
    if (!dst || !src_host) return SIMKA_ERR_INVALID;
    static thread_local hipStream_t tl_stream = nullptr;
    static thread_local int tl_device = -1;
    if (hipSetDevice(device) != hipSuccess) return SIMKA_ERR_HIP;
    if (!tl_stream || tl_device != device) {
        if (hipStreamCreateWithFlags(&tl_stream, hipStreamNonBlocking) != hipSuccess) { tl_stream = nullptr; return SIMKA_ERR_HIP; }
        tl_device = device;      // (a thread that changes device leaks one stream: the loader threads never do)
    }
    if (hipMemcpyAsync(dst, src_host, nb_bytes, hipMemcpyHostToDevice, tl_stream) != hipSuccess) return SIMKA_ERR_HIP;
    return hipStreamSynchronize(tl_stream) == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}

SIMKA_EXPORT int simka_device_copy(int dst_device, void *dst, int src_device, const void *src, uint64_t nb_bytes) {
    if (nb_bytes == 0) return SIMKA_OK;
    if (!dst || !src) return SIMKA_ERR_INVALID;
    if (hipSetDevice(dst_device) != hipSuccess) return SIMKA_ERR_HIP;
    // (distinct devices: a peer copy over xGMI; the same device: an ordinary device-to-device copy).  Synchronous.
    const hipError_t e = dst_device == src_device ? hipMemcpy(dst, src, nb_bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, dst_device, src, src_device, nb_bytes);
    return e == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}

SIMKA_EXPORT int simka_get_geometry(simka_ctx *ctx, uint32_t *l1, uint32_t *l2, uint32_t *t, uint64_t *arena, uint64_t *csr) {
    if (!ctx) return SIMKA_ERR_INVALID;
    if (l1) *l1 = ctx->wide ? ctx->key.l1 : ctx->skm.l1; if (l2) *l2 = ctx->wide ? ctx->key.l2 : ctx->skm.l2; if (t) *t = ctx->key.t;
    if (arena) *arena = ctx->arena_cap; if (csr) *csr = ctx->merge_cap;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_arena_info(simka_ctx *ctx, uint64_t *mapped_mode, uint64_t *reserved_records, uint64_t *mapped_records, uint64_t *retired_va_bytes) {
    if (!ctx) return SIMKA_ERR_INVALID;
    std::lock_guard<std::mutex> g_(g_vmm_lock);
    if (mapped_mode) *mapped_mode = ctx->arena_vmm ? 1 : 0;
    if (reserved_records) *reserved_records = ctx->arena_vmm ? ctx->arena_reserved : ctx->arena_cap;
    if (mapped_records) *mapped_records = ctx->arena_mapped;
    if (retired_va_bytes) *retired_va_bytes = g_vmm_retired_bytes;
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_count_paths(simka_ctx *ctx, uint64_t *nb_partitioned, uint64_t *nb_sorted, uint64_t *nb_exact_redone, uint64_t *nb_full_sorts) {
    if (!ctx) return SIMKA_ERR_INVALID;
    if (nb_full_sorts) *nb_full_sorts = ctx->wide ? simka_wide_full_sorts(ctx->wide) : 0;
    if (nb_partitioned) *nb_partitioned = ctx->wide ? ctx->nb_wide_hash : ctx->nb_counted_this_run;
    if (nb_sorted) *nb_sorted = ctx->nb_wide_sort;
    if (nb_exact_redone) *nb_exact_redone = ctx->nb_exact_fallbacks;
    return SIMKA_OK;
}

// ---- synthetic reads ----------------------------------------------------------------------
SIMKA_EXPORT int simka_synth_genomes(void *stream, uint64_t *d_pool, uint32_t nb_genomes, uint64_t genome_words, uint64_t seed) {
    const uint64_t n = (uint64_t)nb_genomes * genome_words;
    if (!d_pool || n == 0) return SIMKA_ERR_INVALID;
    SIMKA_LAUNCH(k_synth_genomes, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_pool, nb_genomes,
                       genome_words, seed);
    return hipGetLastError() == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}
SIMKA_EXPORT int simka_synth_reads(void *stream, uint64_t *d_packed, uint64_t nb_reads, uint32_t read_len, const uint64_t *d_pool,
                                   uint64_t genome_words, uint64_t genome_len, const uint32_t *d_ids, const uint32_t *d_cdf,
                                   uint32_t nb_sel, uint64_t seed, uint32_t err_thr) {
    if (!d_packed || !d_pool || !d_ids || !d_cdf || nb_sel == 0 || read_len == 0 || genome_len < read_len) return SIMKA_ERR_INVALID;
    const uint64_t nw = (nb_reads * (uint64_t)read_len + 31) / 32;
    if (nw == 0) return SIMKA_OK;
    SIMKA_LAUNCH(k_synth_reads, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_packed, nb_reads,
                       read_len, d_pool, genome_words, genome_len, d_ids, d_cdf, nb_sel, seed, err_thr);
    return hipGetLastError() == hipSuccess ? SIMKA_OK : SIMKA_ERR_HIP;
}
