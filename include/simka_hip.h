/*
 * simka_hip.h -- C ABI of libsimka_hip.so, the MI355X (gfx950) implementation of Simka's
 * multiset k-mer counting + pairwise ecological-distance hot path.
 *
 * Plain C, plain pointers and sizes; no C++/torch types cross this boundary.  Every entry
 * point names the reference interface it replaces ("ref:", paths relative to the GATB/simka
 * tree).  All functions return SIMKA_OK (0) or a SIMKA_ERR_* code and never throw; the
 * message of the last failure is available from simka_last_error() -- this mirrors the
 * reference's "catch gatb Exception -> print EXCEPTION: msg -> EXIT_FAILURE" convention
 * (ref: src/SimkaPotara.cpp:149-160, src/SimkaCount.cpp:382-390, src/SimkaMerge.cpp:1582-1590).
 *
 * Threading: a simka_ctx is single-owner (one host thread, one GPU), like one simkaCount /
 * simkaMerge process in the reference.  The device work of a ctx runs on streams the ctx owns:
 * consecutive samples alternate between two "lanes" (a stream and scratch buffers each, so the
 * scan of one sample overlaps the count of another), host-provided reads arrive through a copy
 * stream, merge / statistics on the ctx stream; calls are asynchronous unless stated.
 */
#ifndef SIMKA_HIP_H
#define SIMKA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMKA_ABI_VERSION 8

enum {
    SIMKA_OK = 0,
    SIMKA_ERR_INVALID = 1,     /* bad argument (the reference: "ERROR: ..." + exit(1)) */
    SIMKA_ERR_HIP = 2,         /* HIP runtime failure */
    SIMKA_ERR_NOMEM = 3,       /* device or host allocation failed / arena exhausted */
    SIMKA_ERR_OVERFLOW = 4,    /* a partition exceeded its LDS table: re-create with more partitions */
    SIMKA_ERR_STATE = 5,       /* call order violated (e.g. merge before every sample was counted) */
    SIMKA_ERR_IO = 6,          /* file could not be read / written */
    SIMKA_ERR_UNSUPPORTED = 7  /* feature not available here (e.g. a collective without RCCL installed) */
};

/* -simple-dist / -complex-dist   ref: src/core/Simka.cpp:25-117, src/core/SimkaAlgorithm.cpp:178-179 */
enum { SIMKA_DIST_SIMPLE = 1u, SIMKA_DIST_COMPLEX = 2u };

/* simka_config.flags */
enum {
    SIMKA_CFG_ARENA_PLAIN = 1u   /* allocate the solid-spectrum arena with one plain device allocation instead of a reserved virtual range
                                  * that is mapped as the samples arrive.  The mode is fixed in simka_create: a context created while another
                                  * one is alive on the same device takes a plain arena by itself; the EARLIER context keeps its mapped
                                  * range, so a caller that creates several contexts on one device up front sets this flag on ALL of them.
                                  * (Background: a virtual range that is unmapped and mapped again is accessed through stale translations on
                                  * ROCm 7.0 / gfx950, scripts/ubench/vmm_two_contexts.hip; the library retires the ranges of destroyed
                                  * contexts instead of reusing them, the plain mode is the second line of defence.)  Any other bit set in
                                  * simka_config.flags is rejected with SIMKA_ERR_INVALID. */
};

typedef struct simka_ctx simka_ctx;

/* Job description: what `simka` passes down to simkaCount / simkaMerge on their command lines
 * (ref: src/SimkaPotara.hpp:847-864, 1013-1024) plus the GPU-side sizing knobs. */
typedef struct simka_config {
    uint32_t struct_size;        /* = sizeof(simka_config) */
    uint32_t nb_samples;         /* N, SimkaStatistics::_nbBanks (1..65535) */
    uint32_t kmer_size;          /* -kmer-size, 1..127 (the reference's spans 32 / 64 / 96 / 128, ref: CMakeLists.txt:66-71).  64..127: the k-mer is rolled
                                  * in four words and counted / merged as a 126-bit fingerprint on the two-word path (simka_wide.hip: wfinger).
                                  * k <= 31 (one 64-bit word, Kmer<span=32>): the hash pipeline.  32..63 (two words,
                                  * Kmer<span=64>): counted on the same minimizer-partitioned pipeline up to k = 51, occurrence by
                                  * occurrence in hash buckets from 52 on (~2x slower); merged by hash buckets + LDS grouping; a partition shard keeps the k-mers that hash to it */
    uint32_t abundance_min;      /* -abundance-min (ref: src/minikc/MiniKC.hpp:56) */
    uint32_t abundance_max;      /* -abundance-max, clamped to 999999999 (ref: src/core/SimkaAlgorithm.cpp:188) */
    uint32_t dist_flags;         /* SIMKA_DIST_* */
    int32_t  device;             /* HIP device ordinal */
    uint32_t shard_index;        /* this GPU's shard of the partition space ... */
    uint32_t shard_count;        /* ... of shard_count (1 = everything); partitions p with p % count == index */
    uint32_t log2_partitions;    /* 0 = derive from max_kmers_per_sample */
    uint32_t log2_subranges;     /* 0 = derive from nb_samples */
    uint32_t flags;              /* SIMKA_CFG_* (0 = defaults; was reserved0 up to ABI 5) */
    uint64_t max_kmers_per_sample; /* upper bound of k-mer occurrences of the largest sample (sizing) */
    uint64_t solid_capacity;     /* capacity (records) of the solid-spectrum arena, 0 = from free memory */
    uint64_t csr_capacity;       /* capacity (records) of the merged-group buffer, 0 = auto */
    void    *stream;             /* hipStream_t to run on; NULL = private stream */
} simka_config;

/* One sample's reads, 2-bit packed: base b lives in bits [2*(b%32), 2*(b%32)+2) of word b/32;
 * code A=0 C=1 T=2 G=3 (= (ascii>>1)&3, the tree's own convention, ref: src/core/SimkaCommons.hpp:400-411).
 * Reads are concatenated without padding.  Non-ACGT letters never reach the device: the host
 * packer splits a read at them (a k-mer window containing one is skipped, gatb Kmer model).
 * `packed` must be readable for 16 bytes past the last word.  Host buffers may be reused as soon as
 * simka_count_sample returns; DEVICE buffers must stay valid until the next synchronising call on the
 * context (simka_get_sample_totals / simka_merge / simka_sync) has returned. */
typedef struct simka_reads {
    const uint64_t *packed;
    uint64_t nb_bases;           /* total bases over all fragments */
    uint64_t nb_reads;           /* fragments in `packed` */
    const uint64_t *offsets;     /* [nb_reads+1] first base of each fragment; ignored if fixed_len>0 */
    uint32_t fixed_len;          /* >0: every fragment has exactly this many bases */
    uint32_t on_device;          /* 1: pointers are device memory, 0: host memory (copied) */
    uint64_t nb_input_reads;     /* reads before splitting (the .ok file's nbReads line) */
} simka_reads;

/* ---- device-side ingest ---------------------------------------------------------
 * Replaces gatb's Bank layer + the 2-bit packing for plain-text inputs (ref: formats README.md:165; the reference's
 * IBank / Sequence iteration is in the absent gatb-core): the bytes of a FASTA / FASTQ file are copied to the GPU as they are
 * and parsed there (line table, sequence lines, ACGT runs -> packed bases + fragment offsets; simka_ingest.hip).
 *   simka_ingest_begin(ctx, sample)
 *   simka_ingest_text(ctx, sample, text, nb_bytes, format, &nb_reads, &irregular)   once per file, in order
 *   simka_ingest_count(ctx, sample, &nb_bases, &nb_reads)                           = simka_count_sample on what was appended
 * `text` is host memory (pinned: simka_host_alloc, or pageable); format 0 = FASTA (multi-line sequences allowed), 1 = FASTQ
 * (4-line records).  Anything the kernels do not parse exactly as the host path does (blank lines inside a file, other FASTQ
 * layouts, >= 4 GB of text) sets *irregular = 1 and appends NOTHING: parse that sample on the host (simka_pack_read +
 * simka_count_sample).  Read-level policies (-max-reads, read filters) are host-side: use this path only without them. */
int simka_ingest_begin(simka_ctx *ctx, uint32_t sample);
int simka_ingest_text(simka_ctx *ctx, uint32_t sample, const char *text, uint64_t nb_bytes, int format, uint64_t *nb_reads, int *irregular);
int simka_ingest_count(simka_ctx *ctx, uint32_t sample, uint64_t *nb_bases, uint64_t *nb_reads);
/* The same with the text ALREADY in device memory (16-byte aligned, readable for nb_bytes; e.g. uploaded piecewise by loader threads with
 * simka_device_upload while the kernels of other samples run -- the `simka` driver: the main thread then only launches kernels, the
 * copies run on the DMA engines from small pinned staging buffers).  The buffer is read where it is and may be reused when the call
 * returns.  (ABI 7) */
int simka_ingest_text_device(simka_ctx *ctx, uint32_t sample, const void *d_text, uint64_t nb_bytes, int format, uint64_t *nb_reads, int *irregular);

/* The 4 lines of count_synchro/<ID>.ok (ref: src/SimkaCount.cpp:303-317,355-368) + pre-filter counts. */
typedef struct simka_sample_totals {
    uint64_t nb_reads;           /* line 1 */
    uint64_t nb_distinct;        /* line 2: distinct k-mers kept by the abundance filter (D_i) */
    uint64_t nb_kmers;           /* line 3: sum of their counts (N_i) */
    uint64_t sum_sq;             /* line 4: sum of count^2 (Q_i, chord norm) */
    uint64_t kmer_occurrences;   /* k-mer occurrences before the filter */
    uint64_t distinct_all;       /* distinct canonical k-mers before the filter */
} simka_sample_totals;

/* Flat view of the accumulators == SimkaStatistics (ref: src/core/SimkaDistance.hpp:68-139).
 * Pair arrays have nb_pairs = N(N-1)/2 cells, cell(i<j) = i*N - i*(i+1)/2 + (j-i-1).
 * All are exact integers; `kl` (f64) exists only with SIMKA_DIST_COMPLEX. */
typedef struct simka_stats_view {
    uint32_t nb_samples, dist_flags;
    uint64_t nb_pairs;
    const uint64_t *nb_distinct;     /* [N] _nbSolidDistinctKmersPerBank */
    const uint64_t *nb_kmers;        /* [N] _nbSolidKmersPerBank */
    const uint64_t *sum_sq;          /* [N] squares of _chord_sqrt_N2 */
    const uint64_t *shared_ij;       /* [pairs] _matrixNbSharedKmers[i][j] */
    const uint64_t *shared_ji;       /* [pairs] _matrixNbSharedKmers[j][i] */
    const uint64_t *distinct_shared; /* [pairs] _matrixNbDistinctSharedKmers */
    const uint64_t *bray_curtis;     /* [pairs] _brayCurtisNumerator (== _kulczynski_minNiNj[i][j]) */
    const uint64_t *chord;           /* [pairs] _chord_NiNj, simple */
    const uint64_t *hellinger;       /* [pairs] _hellinger_SqrtNiNj, simple */
    const uint64_t *whittaker;       /* [pairs] _whittaker_minNiNj, complex */
    const uint64_t *canberra;        /* [pairs] _canberra, complex */
    const double   *kl;              /* [pairs] _kullbackLeibler, complex */
    uint64_t nb_distinct_kmers;      /* _nbDistinctKmers (union) */
    uint64_t nb_shared_kmers;        /* _nbSharedKmers */
} simka_stats_view;

/* ---- lifecycle -------------------------------------------------------------------------- */
int  simka_abi_version(void);
int  simka_create(const simka_config *cfg, simka_ctx **out);
void simka_destroy(simka_ctx *ctx);
/* message of the last error on this ctx (ctx==NULL: last simka_create failure) */
const char *simka_last_error(const simka_ctx *ctx);
/* block until everything issued on the ctx's stream has finished */
int  simka_sync(simka_ctx *ctx);
/* forget all samples and zero the accumulators, keeping every device allocation (a new job with the
 * same configuration; the reference's equivalent is wiping -out-tmp, ref: src/SimkaPotara.hpp:288-325) */
int  simka_reset(simka_ctx *ctx);

/* ---- count side ---------------------------------------------------------------------------
 * Replaces one `simkaCount` job: gatb SortingCountAlgorithm (k-mer extraction, canonical 2-bit
 * coding, partitioning, counting; ref: src/SimkaCount.cpp:291-297) with the
 * SimkaCompressedProcessor::process plugin fused in (abundance filter + per-partition totals,
 * ref: src/minikc/MiniKC.hpp:54-79).  The solid spectrum stays resident in HBM, partitioned,
 * where the reference writes solid/part_<p>/__p__<i>.gz.  Each sample index exactly once. */
int simka_count_sample(simka_ctx *ctx, uint32_t sample_index, const simka_reads *reads);
/* totals of a counted sample (synchronises). Replaces reading count_synchro/<ID>.ok. */
int simka_get_sample_totals(simka_ctx *ctx, uint32_t sample_index, simka_sample_totals *out);

/* ---- -keep-tmp: a counted sample's solid spectrum, out of the context and back ---------------
 * The reference keeps solid/part_<p>/__p__<i>.gz + count_synchro/<ID>.ok in the temp dir and skips the samples whose
 * .ok file exists when run again with more samples (ref: src/SimkaPotara.hpp:837-842, src/SimkaCount.cpp:279-317,
 * README.md:205-206).  Here the spectrum of sample i (this shard's partitions, partition-major) is exported to caller
 * buffers and imported into a later context with the SAME kmer_size, abundance filter, shard and partition count
 * instead of calling simka_count_sample. */
typedef struct simka_spectrum_info {
    uint64_t nb_records;         /* solid k-mers of the sample on this shard */
    uint64_t nb_partitions;      /* 2^log2_partitions */
    uint64_t key_words;          /* 64-bit words per key: 1 (kmer_size <= 31), 2 (32..63: keys[] holds nb_records high words, then
                                  * nb_records low words; partition = top bits of the k-mer, the records are sorted by k-mer) */
} simka_spectrum_info;
int simka_sample_spectrum_info(simka_ctx *ctx, uint32_t sample_index, simka_spectrum_info *out);
/* part_counts[nb_partitions], keys[nb_records * key_words] (the library's internal key of each k-mer), counts[nb_records] */
int simka_export_sample(simka_ctx *ctx, uint32_t sample_index, uint32_t *part_counts, uint64_t *keys, uint32_t *counts);
/* totals: what simka_get_sample_totals returned for the exported sample.  A context that has not counted anything yet
 * takes its partition count from nb_partitions. */
int simka_import_sample(simka_ctx *ctx, uint32_t sample_index, const simka_sample_totals *totals, const uint32_t *part_counts,
                        uint64_t nb_partitions, const uint64_t *keys, const uint32_t *counts, uint64_t nb_records);

/* The same with keys / counts in DEVICE memory of the context's GPU (part_counts stays a host array): the multi-GPU exchange
 * -- samples counted on one rank, merged by partition range on another; the reference's counterpart is every simkaMerge job
 * reading partition p of every sample's solid/ directory (ref: src/SimkaMerge.cpp:1164-1264) -- moves spectra between
 * GPUs with RCCL without touching the host.  The buffers must hold nb_records (simka_sample_spectrum_info) elements. */
int simka_export_sample_device(simka_ctx *ctx, uint32_t sample_index, uint32_t *part_counts, void *d_keys, void *d_counts);
int simka_import_sample_device(simka_ctx *ctx, uint32_t sample_index, const simka_sample_totals *totals, const uint32_t *part_counts,
                               uint64_t nb_partitions, const void *d_keys, const void *d_counts, uint64_t nb_records);

/* Batch forms (one synchronisation for many samples; what simka_amd/dist.py's exchange uses):
 *  - info:   part_counts[nb][nb_partitions] and totals[nb] of counted samples;
 *  - gather: the caller chooses where each (sample, partition) run goes in its device buffers through
 *            out_offsets[nb][nb_partitions] (e.g. destination-rank-major, ready to be sent);
 *  - import: d_keys / d_counts hold nb_records records (< 2^32) of the partitions [part_lo, part_lo + part_width): the run of
 *            partition part_lo + p of samples[i] starts at in_offsets[i][p] and has part_counts[i][p] records (both matrices
 *            are [nb][part_width]); the block is copied into the context once. */
int simka_samples_spectrum_info(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, uint32_t *part_counts, simka_sample_totals *totals);
int simka_gather_samples_device(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const uint64_t *out_offsets, void *d_keys, void *d_counts);
int simka_import_samples_device(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const simka_sample_totals *totals,
                                uint64_t part_lo, uint64_t part_width, const uint32_t *part_counts, const uint64_t *in_offsets,
                                uint64_t nb_partitions, const void *d_keys, const void *d_counts, uint64_t nb_records);

/* The same exchange with its per-(sample, partition) tables computed ON THE DEVICE (ABI 8): with 2^19 partitions those tables are tens of
 * megabytes per rank, and their prefix sums on the host cost ten times what the device needs to move the records.
 *   simka_pack_plan(ctx, samples, nb, nb_ranges, send_records): the counted samples[nb] (slot j = samples[j]) are to be sent to nb_ranges
 *     ranks, rank g receiving the partitions [P g / nb_ranges, P (g + 1) / nb_ranges); send_records[g] (host) = records bound for rank g.
 *   simka_pack_run(ctx, d_keys, d_counts, d_meta, nb_slots, width): gathers the planned runs destination-major -- [g][slot j][partitions
 *     of g] -- into d_keys (u64) / d_counts (u32) of sum(send_records) records, and writes the run lengths to d_meta, device int32
 *     [nb_ranges][nb_slots][width] (nb_slots >= nb, width >= the widest range; cells it does not own are left as they are: clear it first).
 *   simka_import_block_device: the receive side.  The block holds nb_slots_total slots one after the other (source rank major), slot q =
 *     the runs of sample slot_samples[q] (0xffffffff: an empty slot) in the partitions [part_lo, part_lo + part_width), their lengths in
 *     row q of d_meta (device int32 [nb_slots_total][width]); totals[q] as for simka_import_samples_device.
 * (kmer_size >= 32 exchanges whole sorted runs: simka_gather_samples_device_wide / simka_import_samples_device_wide.)
 * Stream ordering of caller buffers: every entry point that takes a DEVICE pointer reads / writes it on the context's stream
 *   (simka_config.stream, or a stream of its own) and returns after that work has completed.  Work the caller has queued on ANOTHER stream for
 *   the same buffer (the zero-fill of d_meta, an asynchronous copy or a collective that produces d_keys / d_counts / d_meta) must have
 *   completed -- or be ordered by an event the context's stream waits on -- before the call; only the legacy null stream orders implicitly.
 * State: a plan lives until the next simka_pack_plan, simka_reset or import on the context (those drop it: simka_pack_run then fails with
 *   SIMKA_ERR_STATE); a simka_import_block_device that fails leaves the context as it was (the tables of the target samples are rolled back). */
int simka_pack_plan(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, uint32_t nb_ranges, uint64_t *send_records);
int simka_pack_run(simka_ctx *ctx, void *d_keys, void *d_counts, int32_t *d_meta, uint32_t nb_slots, uint32_t width);
int simka_import_block_device(simka_ctx *ctx, const uint32_t *slot_samples, uint32_t nb_slots_total, const simka_sample_totals *totals,
                              uint64_t part_lo, uint64_t part_width, uint32_t width, const int32_t *d_meta, uint64_t nb_partitions,
                              const void *d_keys, const void *d_counts, uint64_t nb_records);

/* kmer_size >= 32 (key_words == 2): the batch gather / import with the high and the low key words in separate device buffers.
 * A received block holds each sample's records contiguously and sorted (a sample comes from ONE rank): sample_offsets /
 * sample_records say where.  simka_samples_spectrum_info is shared (its partitions are key-prefix ranges). */
int simka_gather_samples_device_wide(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const uint64_t *out_offsets, void *d_keys_hi, void *d_keys_lo,
                                     void *d_counts);
int simka_import_samples_device_wide(simka_ctx *ctx, const uint32_t *samples, uint32_t nb, const simka_sample_totals *totals, const uint64_t *sample_offsets,
                                     const uint64_t *sample_records, const void *d_keys_hi, const void *d_keys_lo, const void *d_counts);

/* ---- merge side ---------------------------------------------------------------------------
 * Replaces every `simkaMerge` job: the N-way k-mer merge (ref: src/SimkaMerge.cpp:1164-1264),
 * its gate (ref: :1307-1326) and SimkaCountProcessorSimple::process -> updateDistance*
 * (ref: src/core/SimkaAlgorithm.hpp:200,341-516), over this shard's partitions, accumulating
 * into the ctx-owned device SimkaStatistics.  Requires all nb_samples counted. */
int simka_merge(simka_ctx *ctx);

/* ---- statistics ---------------------------------------------------------------------------
 * The accumulators live in ONE flat device buffer of nb_u64 64-bit words so that the
 * cross-shard reduction (SimkaStatistics::operator+=, ref: src/core/SimkaDistance.cpp:156-213)
 * is a single all-reduce(sum, uint64): simka_stats_allreduce() below (RCCL), or the caller's own
 * collective on this buffer. */
int simka_stats_device_buffer(simka_ctx *ctx, void **device_ptr, uint64_t *nb_u64);
/* download (synchronises) into a host buffer of nb_u64 words and describe it */
int simka_stats_download(simka_ctx *ctx, uint64_t *host_buf, uint64_t nb_u64, simka_stats_view *view);
/* describe an arbitrary host copy of the flat buffer (e.g. after the caller's all-reduce) */
int simka_stats_describe(uint32_t nb_samples, uint32_t dist_flags, uint64_t *host_buf, uint64_t nb_u64,
                         simka_stats_view *view);   /* with SIMKA_DIST_COMPLEX it also fills the derived tail (canberra, kl) */
uint64_t simka_stats_nb_u64(uint32_t nb_samples, uint32_t dist_flags);
/* word offsets of the flat buffer: out[8] = { #pair arrays, first pair array, first totals row, derived tail,
 * nb_pairs, words of the head (header + pair arrays), total words, 0 }.  Layout:
 * [8 header | S_ij S_ji a bc (chord hell) (whittaker klfix) x nb_pairs | D N Q D_all K_occ x N | canberra, kl(f64) x nb_pairs] */
int simka_stats_layout(uint32_t nb_samples, uint32_t dist_flags, uint64_t out[8]);
/* the two device ranges of the buffer.  Sharded (multi-GPU) protocol: with SIMKA_DIST_COMPLEX all-reduce `totals`
 * BEFORE simka_merge (the per-k-mer complex terms need the global N_i; the reference reads them from
 * every count_synchro/<ID>.ok in SimkaStatistics' constructor, ref: src/core/SimkaDistance.cpp:116-151) and `head`
 * after it; without it one all-reduce of simka_stats_device_buffer() after simka_merge is enough. */
int simka_stats_device_ranges(simka_ctx *ctx, void **head, uint64_t *nb_head, void **totals, uint64_t *nb_totals);
/* host round trip of the 5 x N per-sample totals rows (D, N, Q, D_all, K_occ): lets a driver without a collective
 * library make the totals global across its contexts before simka_merge (synchronise) */
int simka_totals_download(simka_ctx *ctx, uint64_t *out_5n);
int simka_totals_upload(simka_ctx *ctx, const uint64_t *in_5n);

/* ---- cross-GPU reduction (RCCL over xGMI) -----------------------------------------------------
 * The reference combines the per-partition SimkaStatistics of its simkaMerge jobs with operator+= after reading
 * stats/part_<p>.gz (ref: src/core/SimkaDistance.cpp:156-213, src/SimkaPotara.hpp:1130-1187).  Here every GPU holds the
 * accumulators of its shard in one flat u64 buffer and ONE ncclAllReduce(sum, uint64) on the context's stream combines
 * them.  A simka_comm wraps one RCCL communicator: rank 0 calls simka_comm_unique_id, hands the 128 bytes to the other
 * ranks by any means (MPI, torch.distributed, a file, shared memory between the threads of one process), every rank calls
 * simka_comm_create.  One communicator per GPU; its calls must come from the thread that owns the context. */
#define SIMKA_COMM_ID_BYTES 128
typedef struct simka_comm simka_comm;
/* Which RCCL serves the collectives: "<file> (<how it was found>)".  A copy that is already loaded in the process -- e.g. the one a Python
 * caller's torch has bootstrapped its process group with -- is reused (one RCCL per process); SIMKA_RCCL_PATH names a file explicitly;
 * otherwise librccl.so.1 from the loader's search path.  SIMKA_ERR_UNSUPPORTED without RCCL.  (ABI 8) */
int  simka_comm_library(char *path, uint64_t capacity);
int  simka_comm_unique_id(uint8_t id[SIMKA_COMM_ID_BYTES]);
int  simka_comm_create(const uint8_t id[SIMKA_COMM_ID_BYTES], int nb_ranks, int rank, int device, simka_comm **out);
void simka_comm_destroy(simka_comm *comm);
const char *simka_comm_last_error(const simka_comm *comm);   /* comm==NULL: last simka_comm_create / _unique_id failure */
int  simka_comm_info(const simka_comm *comm, int *rank, int *nb_ranks);
/* sum the whole accumulator buffer (pair arrays + per-sample totals) of the context over the ranks, in place, asynchronously
 * on the context's stream: SimkaStatistics::operator+= across GPUs.  Call after simka_merge. */
int simka_stats_allreduce(simka_ctx *ctx, simka_comm *comm);
/* the two-step protocol of SIMKA_DIST_COMPLEX (see simka_stats_device_ranges): totals BEFORE simka_merge, head after it */
int simka_totals_allreduce(simka_ctx *ctx, simka_comm *comm);
int simka_stats_allreduce_head(simka_ctx *ctx, simka_comm *comm);
/* building blocks for callers that exchange solid spectra between GPUs (samples counted on one rank, merged by partition
 * range on another; the reference's counterpart is every simkaMerge job reading solid/part_<p>/ of every sample,
 * ref: src/SimkaMerge.cpp:1082-1103): an in-place u64 sum and an all-to-all with uneven splits over device buffers
 * (grouped ncclSend/ncclRecv: point-to-point transfers, one per xGMI peer).  Counts / displacements in elements. */
int simka_comm_allreduce_u64(simka_comm *comm, void *d_buf, uint64_t nb_u64, void *stream);
int simka_comm_alltoallv(simka_comm *comm, const void *d_send, const uint64_t *send_counts, const uint64_t *send_displs, void *d_recv,
                         const uint64_t *recv_counts, const uint64_t *recv_displs, uint32_t elem_bytes, void *stream);

/* ---- finalisation (host) ------------------------------------------------------------------
 * SimkaDistance: the 21 distance matrices as float32 cells (ref: src/core/SimkaDistance.cpp:920-1226,
 * src/core/SimkaDistance.hpp:155-475) and the CSV writer (ref: src/core/SimkaDistance.cpp:603-699). */
int         simka_nb_matrices(void);
const char *simka_matrix_name(int which);                    /* "mat_abundance_braycurtis", ... reference order */
int         simka_matrix_enabled(int which, uint32_t dist_flags);
int simka_compute_matrix(const simka_stats_view *view, int which, float *out_nxn);
/* writes <dir>/<name>.csv.gz (gz=1) or <dir>/<name>.csv exactly as dumpMatrix does */
int simka_write_matrix_csv(const char *dir, const char *name, const char *const *sample_ids, uint32_t nb_samples,
                           const float *matrix_nxn, int gz);

/* ---- host-side ingest helper --------------------------------------------------------------
 * 2-bit packer for ASCII reads (gatb Bank + Kmer model coding).  Appends to a growing packed
 * buffer owned by the caller: *nb_bases is the running base count, `offsets` receives one entry
 * per produced fragment (reads are split at non-ACGT letters).  Returns the number of fragments
 * appended, or <0 on error.  Caller guarantees capacity for `len` more bases and len+1 offsets. */
int64_t simka_pack_read(const char *seq, uint64_t len, uint64_t *packed, uint64_t *nb_bases, uint64_t *offsets_out);
/* Page-locked host memory for the packed reads / offsets handed to simka_count_sample(on_device = 0): the copy to the GPU is
 * then one DMA at PCIe speed into the staging buffer of the sample's lane (two buffers: the copy of sample i + 1 overlaps the
 * kernels of sample i), where pageable memory goes through the runtime's bounce buffer.  Where the reference's count job reads
 * its bank through gatb's buffered BankFasta iterator (ref: src/SimkaCount.cpp:283-299, src/core/SimkaCommons.hpp:159-314) the
 * `simka` driver parses into these buffers.  *p = NULL and SIMKA_ERR_NOMEM when the allocation fails (callers may fall back to
 * malloc: the ABI accepts any host pointer). */
int simka_host_alloc(uint64_t nb_bytes, void **p);
int simka_host_free(void *p);

/* ---- profiling ----------------------------------------------------------------------------
 * HIP-event timing of the kernels launched by the ctx, on the stream they are launched on.  on = 0: off; 1: every kernel;
 * >= 2: only the kernels i (index of simka_profile_get) whose bit (i + 1) is set -- two event records per launch are not
 * free, a caller timing a whole job can restrict them to the kernel it reports. */
int simka_profile_enable(simka_ctx *ctx, int on);
int simka_profile_reset(simka_ctx *ctx);
int simka_profile_nb_kernels(simka_ctx *ctx);
/* synchronises; name is a static string */
int simka_profile_get(simka_ctx *ctx, int which, const char **name, uint64_t *nb_launches, double *total_ms);
/* log2 of the partition count a context would pick for samples of up to max_kmers_per_sample k-mer occurrences (what
 * simka_config::log2_partitions == 0 resolves to): contexts that exchange spectra must be created with the SAME explicit value */
uint32_t simka_default_log2_partitions(uint64_t max_kmers_per_sample, uint32_t kmer_size);
/* free / total memory of a device in bytes (hipMemGetInfo), for callers that plan how many partition ranges to merge at once */
int simka_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* Device buffers for a caller that moves spectra between the GPUs of one process itself -- simka_gather_samples_device on the GPU
 * that counted the samples, simka_device_copy (a peer copy over xGMI when the devices differ, synchronous), simka_import_samples_device
 * on the GPU that merges the partition range -- the `simka -nb-gpus` driver: what the reference moves through solid/part_<p>/ files
 * on a shared disk (ref: src/SimkaPotara.hpp:813-1124) never leaves device memory.  SIMKA_ERR_NOMEM when the allocation fails. */
int simka_device_alloc(int device, uint64_t nb_bytes, void **p);
int simka_device_free(int device, void *p);
int simka_device_copy(int dst_device, void *dst, int src_device, const void *src, uint64_t nb_bytes);
/* the CPUs on the device's NUMA node as the kernel lists them ("0-63,128-191"; "" if unknown): where a host should run the threads
 * that feed the device and allocate their pinned staging memory (ABI 7) */
int simka_device_cpulist(int device, char *buf, uint64_t buf_bytes);
/* host -> device copy, synchronous, on a stream private to the calling thread: safe from several threads at once (ABI 7) */
int simka_device_upload(int device, void *dst, const void *src_host, uint64_t nb_bytes);
/* sizes chosen by the ctx (partition bits etc.), for DESIGN/bench reporting */
int simka_get_geometry(simka_ctx *ctx, uint32_t *log2_level1, uint32_t *log2_level2, uint32_t *log2_subranges,
                       uint64_t *arena_capacity, uint64_t *csr_capacity);
/* the solid-spectrum arena of the context: 1 = a reserved virtual range mapped as the samples arrive, 0 = one plain allocation
 * (SIMKA_CFG_ARENA_PLAIN, or another context alive on the device at creation); records reserved / backed by memory now; and the
 * address space (bytes) of the ranges of DESTROYED contexts of this process, which are retired instead of reused (a remapped range is
 * read through stale translations on ROCm 7.0 / gfx950) -- past 2^45 bytes new contexts take plain arenas.  (ABI 8) */
int simka_arena_info(simka_ctx *ctx, uint64_t *mapped_mode, uint64_t *reserved_records, uint64_t *mapped_records, uint64_t *retired_va_bytes);
/* how the samples counted so far were counted: on the minimizer-partitioned pipeline (every sample for kmer_size <= 31, and for
 * 32 <= kmer_size <= 51 unless a partition outgrew its table) or k-mer occurrence by occurrence (kmer_size >= 52, the fallback,
 * SIMKA_SORT_PATH); how many had their level-1 buckets sized exactly after a capacity-sized attempt overflowed; and how many counts /
 * merges of two-word k-mers sorted every record by k-mer because a hash bucket held too many distinct k-mers (0 in normal runs) */
int simka_count_paths(simka_ctx *ctx, uint64_t *nb_partitioned, uint64_t *nb_sorted, uint64_t *nb_exact_redone, uint64_t *nb_full_sorts);

/* ---- synthetic reads (bench / test utility, not part of the reference path) ---------------
 * Seeded generator of SURVEY.md section 8(d): a pool of random genomes and reads sampled from
 * it with substitution errors, written 2-bit packed straight into device memory.  Integer-only,
 * so simka_amd/synth.py reproduces it bit-for-bit on the CPU for the parity tests. */
int simka_synth_genomes(void *stream, uint64_t *d_pool, uint32_t nb_genomes, uint64_t genome_words, uint64_t seed);
int simka_synth_reads(void *stream, uint64_t *d_packed, uint64_t nb_reads, uint32_t read_len,
                      const uint64_t *d_pool, uint64_t genome_words, uint64_t genome_len,
                      const uint32_t *d_genome_ids, const uint32_t *d_cdf, uint32_t nb_sel,
                      uint64_t seed, uint32_t err_threshold16);

#ifdef __cplusplus
}
#endif
#endif /* SIMKA_HIP_H */
