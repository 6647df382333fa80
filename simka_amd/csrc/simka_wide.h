// simka_wide.h -- internal interface of the sort-based path for 32 <= k <= 63 (simka_wide.hip), used by simka_ctx.hip.
#pragma once
#include <stdint.h>
#include "simka_kernels.h"

struct SimkaWide;

enum { SIMKA_WIDE_OK = 0, SIMKA_WIDE_ERR_HIP = 1, SIMKA_WIDE_ERR_NOMEM = 2, SIMKA_WIDE_ERR_LIMIT = 3 };

// the CSR of the groups shared by >= 2 samples, in the layout k_pairs consumes (owned by the wide state)
struct SimkaWideCsr {
    unsigned long long *entries; uint32_t *groups; SimkaSpan *spans; unsigned long long *cursors;
    uint32_t nb_spans;
    uint64_t nb_entries;                   // entries the spans cover (the huge groups' entries follow them)
    SimkaSpan *huge; uint32_t nb_huge;     // groups larger than a span (k_pairs_global)
    uint64_t nb_distinct, nb_shared;       // union of the samples' solid k-mers / those in >= 2 samples
};

int simka_wide_create(SimkaWide **out, int device, uint32_t nb_samples, uint32_t k, void *stream);
void simka_wide_destroy(SimkaWide *w);
// partition shard of the k-mer space this state keeps (default: everything)
void simka_wide_set_shard(SimkaWide *w, uint32_t shard_index, uint32_t shard_count);
int simka_wide_reset(SimkaWide *w);
const char *simka_wide_error(SimkaWide *w);
// packed / offsets: DEVICE pointers.  totals5: D N Q D_all K_occ (SIMKA_TOT_* order), host.  d_hist_row etc.: -complex-dist (or NULL).
int simka_wide_count_sample(SimkaWide *w, uint32_t sample, const void *packed, uint64_t nb_bases, uint64_t nb_words, const void *offsets, uint64_t nb_reads,
                            uint32_t fixed_len, uint32_t amin, uint32_t amax, unsigned long long totals5[5], void *d_hist_row, void *d_ovf_list,
                            void *d_ovf_cursor, uint64_t ovf_cap);
// solid records counted by the partitioned pipeline (k <= 51): unordered device triples, n of them in nb_slots slots (count 0: unused) ->
// the sample's run of the arena (sorted on demand)
int simka_wide_adopt(SimkaWide *w, uint32_t sample, const void *d_hi, const void *d_lo, const void *d_cnt, uint64_t nb_slots, uint64_t n);
int simka_wide_merge(SimkaWide *w, uint32_t span_cap, SimkaWideCsr *out);

// spectra out of / into the wide arena: partition p of a (sorted) sample = the records whose top log2_parts key bits equal p
int simka_wide_part_counts(SimkaWide *w, uint32_t sample, uint32_t log2_parts, uint32_t *host_counts);
uint64_t simka_wide_sample_records(SimkaWide *w, uint32_t sample);
uint64_t simka_wide_full_sorts(SimkaWide *w);      // counts / merges that sorted by k-mer (fallback of the bucket routes)
int simka_wide_export(SimkaWide *w, uint32_t sample, void *keys_hi_then_lo, void *counts, int on_device);
int simka_wide_import(SimkaWide *w, uint32_t sample, const void *keys_hi_then_lo, const void *counts, uint64_t n, int on_device);
int simka_wide_gather(SimkaWide *w, const uint32_t *samples, uint32_t nb, uint32_t log2_parts, const uint64_t *out_offsets, void *d_hi, void *d_lo, void *d_counts);
int simka_wide_import_words(SimkaWide *w, uint32_t sample, const void *d_hi, const void *d_lo, const void *d_counts, uint64_t n);
