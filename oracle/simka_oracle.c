/*
 * oracle/simka_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of Simka's hot path
 *     simkaCount -> simkaMerge -> SimkaStatistics / SimkaDistance
 * used ONLY as the checker for the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under simka_amd/ may include, link or
 * call this file.
 *
 * Pinning: the reference binary cannot be built here (thirdparty/gatb-core is an
 * empty, un-vendored submodule, /root/reference/.gitmodules:1-3), so this
 * restatement is pinned against the reference's own golden matrices
 * (tests/truth/results_k{21,31}_t{0,2}/ *.csv, compared byte-for-byte exactly as
 * /root/reference/tests/simple_test.py:29-68 does) -- see tests/test_oracle_golden.py.
 *
 * All "ref:" citations are relative to /root/reference/.
 *
 * Stages (SURVEY.md section 8a row names in brackets):
 *   [a1] or_parse_input        ref: src/core/SimkaAlgorithm.cpp:245-351
 *   [a2] or_iter_* / or_filter ref: src/core/SimkaCommons.hpp:159-314 (SimkaInputIterator: -max-reads per paired part, the read m+1
 *                                   that is fetched and overwritten, files-per-part = composition / nbPaired) and :317-436
 *                                   (SimkaSequenceFilter: -min-read-size, -min-shannon-index), restated method by method
 *   [a3] or_count_sample       ref: src/SimkaCount.cpp:291-297 (gatb SortingCountAlgorithm: canonical k-mers,
 *                                   emitted in increasing order with their count; gatb-core 1.x, absent)
 *   [a4] or_filter_totals      ref: src/minikc/MiniKC.hpp:54-79
 *   [a5] totals                ref: src/SimkaCount.cpp:303-317
 *   [a6] or_merge_partition    ref: src/SimkaMerge.cpp:1164-1264 (N-way min-heap merge)
 *   [a7] or_insert             ref: src/SimkaMerge.cpp:1307-1326
 *   [a8] or_update_*           ref: src/core/SimkaAlgorithm.hpp:341-516
 *   [a9] or_stats_add          ref: src/core/SimkaDistance.cpp:156-213
 *   [a10] or_dist_*            ref: src/core/SimkaDistance.cpp:920-1226, src/core/SimkaDistance.hpp:155-475
 *   [a11] or_dump_matrix       ref: src/core/SimkaDistance.cpp:653-699
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* small utilities                                                            */
/* ------------------------------------------------------------------------- */
/* a k-mer of up to 63 bases: 126 bits.  (k <= 31 fits 64 bits, as Kmer<span=32> of the reference; k >= 32 is its
 * span=64 instantiation, KSIZE_LIST of the reference's CMakeLists.txt) */
typedef unsigned __int128 kmer_t;
typedef struct { kmer_t *v; size_t n, cap; } u64vec;

static void u64vec_push(u64vec *a, kmer_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 1024;
        a->v = (kmer_t *)realloc(a->v, a->cap * sizeof(kmer_t));
        if (!a->v) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    }
    a->v[a->n++] = x;
}

/* k <= 32: the keys fit 64 bits (Kmer<span=32> of the reference) -- sort 8-byte words, as a CPU implementation would */
static void radix_sort_words(uint64_t *a, size_t n, int bits) {
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *src = a, *dst = tmp;
    for (int shift = 0; shift < bits; shift += 8) {
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (size_t i = 0; i < n; i++) hist[(size_t)((src[i] >> shift) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (size_t i = 0; i < n; i++) dst[hist[(size_t)((src[i] >> shift) & 0xff)]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
    free(tmp);
}

/* LSD radix sort of k-mers, `bits` significant bits. */
static void radix_sort_u64(kmer_t *a, size_t n, int bits) {
    if (n < 2) return;
    if (bits <= 64) {      /* (a range of wide k-mers is sorted on its low bits only: the words must hold the whole k-mer) */
        uint64_t *w = (uint64_t *)malloc(n * sizeof(uint64_t));
        uint64_t high = 0;
        for (size_t i = 0; i < n; i++) { w[i] = (uint64_t)a[i]; high |= (uint64_t)(a[i] >> 64); }
        if (high == 0) {
            radix_sort_words(w, n, bits);
            for (size_t i = 0; i < n; i++) a[i] = (kmer_t)w[i];
            free(w);
            return;
        }
        free(w);
    }
    kmer_t *tmp = (kmer_t *)malloc(n * sizeof(kmer_t));
    kmer_t *src = a, *dst = tmp;
    for (int shift = 0; shift < bits; shift += 8) {
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (size_t i = 0; i < n; i++) hist[(size_t)((src[i] >> shift) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (size_t i = 0; i < n; i++) dst[hist[(size_t)((src[i] >> shift) & 0xff)]++] = src[i];
        kmer_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(kmer_t));
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* [a3] canonical 2-bit k-mers                                                */
/* ------------------------------------------------------------------------- */
/* Nucleotide code: A=0 C=1 T=2 G=3, i.e. (ascii>>1)&3 -- the tree's own table
 * (ref: src/core/SimkaCommons.hpp:400-411).  Complement is code^2.  Any fixed
 * total order gives the same distances. */
static inline int or_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'T': case 't': return 2;
        case 'G': case 'g': return 3;
        default: return -1; /* N / IUPAC: every window containing it is skipped */
    }
}

/* ---- 64 <= k <= 127 (the reference's Kmer<span=96/128>, ref: CMakeLists.txt:66-71, src/SimkaPotara.cpp:132-141): the canonical k-mer
 * is kept WHOLE, four 64-bit words (w[3] most significant), and every distinct one gets its RANK in the sorted dictionary of the run --
 * distances depend only on which k-mers are equal, so everything behind the extraction works on the ranks with the
 * machinery of k <= 63.  Two passes over the reads: collect (mode 1), then map (mode 2). */
typedef struct { uint64_t w[4]; } k256;
/* the dictionary belongs to ONE oracle instance (it lives in the oracle struct: two instances, or two threads running two oracles, do not
 * share state); mode 0: unused, 1: collect, 2: map */
typedef struct { int mode; k256 *v; size_t n, cap; int failed; } k256_dict;
static int k256_cmp(const void *a, const void *b) {
    const k256 *x = (const k256 *)a, *y = (const k256 *)b;
    for (int q = 3; q >= 0; q--) if (x->w[q] != y->w[q]) return x->w[q] < y->w[q] ? -1 : 1;
    return 0;
}
static void k256_collect(k256_dict *d, const k256 *x) {
#ifdef _OPENMP
#pragma omp critical(k256_collect)
#endif
    {
        if (d->n == d->cap && !d->failed) {
            const size_t cap = d->cap ? d->cap * 2 : 4096;
            k256 *nv = (k256 *)realloc(d->v, cap * sizeof(k256));
            if (nv) { d->v = nv; d->cap = cap; } else d->failed = 1;        /* (reported by oracle_run_shard) */
        }
        if (d->n < d->cap) d->v[d->n++] = *x;
    }
}
static void k256_build_dictionary(k256_dict *d) {
    qsort(d->v, d->n, sizeof(k256), k256_cmp);
    size_t w = 0;
    for (size_t i = 0; i < d->n; i++) if (w == 0 || k256_cmp(&d->v[w - 1], &d->v[i]) != 0) d->v[w++] = d->v[i];
    d->n = w;
}
static uint64_t k256_rank(const k256_dict *d, const k256 *x) {
    const k256 *f = (const k256 *)bsearch(x, d->v, d->n, sizeof(k256), k256_cmp);
    return f ? (uint64_t)(f - d->v) : ~0ull;       /* (every k-mer of the map pass was collected) */
}
static size_t or_kmers_of_read_256(const char *seq, size_t len, int k, u64vec *out, k256_dict *dict) {
    const int W = 2 * k, tw = (W - 1) / 64, top = 2 * (k - 1);
    const uint64_t mtop = (W % 64) ? ((1ull << (W % 64)) - 1ull) : ~0ull;
    k256 f, r; memset(&f, 0, sizeof f); memset(&r, 0, sizeof r);
    size_t valid = 0, emitted = 0;
    for (size_t i = 0; i < len; i++) {
        int c = or_code((unsigned char)seq[i]);
        if (c < 0) { valid = 0; memset(&f, 0, sizeof f); memset(&r, 0, sizeof r); continue; }
        for (int q = 3; q > 0; q--) f.w[q] = (f.w[q] << 2) | (f.w[q - 1] >> 62);
        f.w[0] = (f.w[0] << 2) | (uint64_t)c;
        f.w[tw] &= mtop; for (int q = tw + 1; q < 4; q++) f.w[q] = 0;
        for (int q = 0; q < 3; q++) r.w[q] = (r.w[q] >> 2) | (r.w[q + 1] << 62);
        r.w[3] >>= 2;
        r.w[top / 64] |= (uint64_t)(c ^ 2) << (top % 64);
        if (++valid >= (size_t)k) {
            const k256 *canon = k256_cmp(&f, &r) < 0 ? &f : &r;
            if (dict->mode == 1) { k256_collect(dict, canon); u64vec_push(out, (kmer_t)0); }
            else u64vec_push(out, (kmer_t)k256_rank(dict, canon));
            emitted++;
        }
    }
    return emitted;
}

/* Append the canonical k-mers of one read to `out`; returns #k-mers appended. */
static size_t or_kmers_of_read(const char *seq, size_t len, int k, u64vec *out, k256_dict *dict) {
    if (k >= 64) return or_kmers_of_read_256(seq, len, k, out, dict);
    if (k <= 31) {      /* 64-bit rolling words (the reference's Kmer<span=32>) */
        const uint64_t mask = (1ull << (2 * k)) - 1ull;
        uint64_t fwd = 0, rev = 0;
        size_t valid = 0, emitted = 0;
        for (size_t i = 0; i < len; i++) {
            int c = or_code((unsigned char)seq[i]);
            if (c < 0) { valid = 0; fwd = rev = 0; continue; }
            fwd = ((fwd << 2) | (uint64_t)c) & mask;
            rev = (rev >> 2) | ((uint64_t)(c ^ 2) << (2 * (k - 1)));
            if (++valid >= (size_t)k) { u64vec_push(out, (kmer_t)(fwd < rev ? fwd : rev)); emitted++; }
        }
        return emitted;
    }
    const kmer_t mask = (((kmer_t)1) << (2 * k)) - 1;       /* k <= 63 */
    kmer_t fwd = 0, rev = 0;
    size_t valid = 0, emitted = 0;
    for (size_t i = 0; i < len; i++) {
        int c = or_code((unsigned char)seq[i]);
        if (c < 0) { valid = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (kmer_t)c) & mask;
        rev = (rev >> 2) | ((kmer_t)(c ^ 2) << (2 * (k - 1)));
        if (++valid >= (size_t)k) {
            u64vec_push(out, fwd < rev ? fwd : rev);
            emitted++;
        }
    }
    return emitted;
}

/* ------------------------------------------------------------------------- */
/* samples                                                                    */
/* ------------------------------------------------------------------------- */
typedef struct {
    char *id;
    char **files; int nfiles; int nb_paired;   /* [a1] */
    /* in-memory reads (alternative to files): concatenated ASCII + offsets */
    const char *bases; const uint64_t *offsets; uint64_t nreads_mem;
    /* [a3] results */
    uint64_t nb_reads;      /* reads seen */
    uint64_t k_occ;         /* k-mer occurrences */
    uint64_t d_all;         /* distinct canonical k-mers before the abundance filter */
    /* [a4] solid spectrum, sorted by k-mer */
    kmer_t *kmer; uint32_t *count; size_t nsolid;
    /* [a5] totals after filter */
    uint64_t D, N, Q;
} or_sample;

typedef struct {
    int n, simple, complex_;
    size_t symsize;
    /* ref: src/core/SimkaDistance.hpp:79-129 */
    uint64_t *D, *Nk;                 /* _nbSolidDistinctKmersPerBank, _nbSolidKmersPerBank */
    long double *sqrtN2;              /* _chord_sqrt_N2 */
    uint64_t *a, *bc;                 /* _matrixNbDistinctSharedKmers, _brayCurtisNumerator (sym) */
    uint64_t *S;                      /* _matrixNbSharedKmers [n][n] */
    long double *chord;               /* _chord_NiNj [n][n] */
    uint64_t *hell, *kul;             /* _hellinger_SqrtNiNj, _kulczynski_minNiNj [n][n] */
    uint64_t *whit, *canb;            /* _whittaker_minNiNj, _canberra [n][n] */
    long double *kl;                  /* _kullbackLeibler [n][n] */
    uint64_t nb_distinct, nb_shared;  /* _nbDistinctKmers, _nbSharedKmers */
} or_stats;

typedef struct {
    or_sample *s; int n;
    int k; uint32_t amin, amax;
    uint64_t max_reads; uint64_t min_read_size; double min_shannon;   /* [a2] read policies (0 = off) */
    or_stats *stats;
    k256_dict dict;                   /* 64 <= k <= 127: the run's dictionary of whole k-mers */
    char err[512];
} oracle;

/* ------------------------------------------------------------------------- */
/* [a1] input grammar    ref: src/core/SimkaAlgorithm.cpp:245-351             */
/* ------------------------------------------------------------------------- */
static char *or_strndup(const char *s, size_t n) {
    char *r = (char *)malloc(n + 1); memcpy(r, s, n); r[n] = 0; return r;
}

static int or_parse_input(oracle *o, const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { snprintf(o->err, sizeof o->err, "ERROR: Input filename does not exist"); return -1; }
    char *rp = realpath(path, NULL);
    char *slash = strrchr(rp, '/');
    size_t dirlen = slash ? (size_t)(slash - rp) : 0;
    char *line = NULL; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        /* remove ALL spaces (:269); also drop the newline getline keeps (std::getline strips it) */
        size_t w = 0;
        for (ssize_t i = 0; i < len; i++) if (line[i] != ' ' && line[i] != '\n' && line[i] != '\r') line[w++] = line[i];
        line[w] = 0;
        if (w == 0) continue;                                   /* :270 */
        char *colon = strchr(line, ':');                        /* :278-283 */
        if (!colon) { snprintf(o->err, sizeof o->err, "Syntax error in input file"); fclose(f); free(line); free(rp); return -1; }
        o->s = (or_sample *)realloc(o->s, (o->n + 1) * sizeof(or_sample));
        or_sample *s = &o->s[o->n++];
        memset(s, 0, sizeof *s);
        s->id = or_strndup(line, (size_t)(colon - line));
        char *rest = colon + 1;
        char *c2 = strchr(rest, ':'); if (c2) *c2 = 0;          /* only lineIdDatasets[1] is used (:283) */
        /* split on ';' -> paired parts (:286-294), each on ',' -> files (:301-320) */
        char *save1 = NULL;
        for (char *part = strtok_r(rest, ";", &save1); part; part = strtok_r(NULL, ";", &save1)) {
            s->nb_paired++;
            char *save2 = NULL;
            for (char *fn = strtok_r(part, ",", &save2); fn; fn = strtok_r(NULL, ",", &save2)) {
                char *full;
                if (fn[0] == '/') full = strdup(fn);            /* :312-314 */
                else {                                          /* :315-319 relative to the input file's dir */
                    full = (char *)malloc(dirlen + strlen(fn) + 2);
                    memcpy(full, rp, dirlen); full[dirlen] = '/'; strcpy(full + dirlen + 1, fn);
                }
                s->files = (char **)realloc(s->files, (s->nfiles + 1) * sizeof(char *));
                s->files[s->nfiles++] = full;
            }
        }
    }
    free(line); free(rp); fclose(f);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* [a2] read iteration (default policy: every read of every listed file, in   */
/* order; -max-reads / read filters are "next" rows)            */
/* FASTA (multi-line) and FASTQ (4-line), plain or gz (zlib reads both).      */
/* ------------------------------------------------------------------------- */
typedef struct { char *b; size_t n, cap; } strbuf;
static void sb_app(strbuf *s, const char *p, size_t n) {
    if (s->n + n + 1 > s->cap) { s->cap = (s->n + n + 1) * 2; s->b = (char *)realloc(s->b, s->cap); }
    memcpy(s->b + s->n, p, n); s->n += n; s->b[s->n] = 0;
}

/* one sequence file as gatb's Bank iterator: first() / next() / isDone() / item() */
typedef struct {
    const char *path; gzFile g;
    strbuf seq, cur;            /* record being read / the current item */
    int state;                  /* 0 none, 1 fasta seq, 2 fastq seq line next, 3 fastq '+' seen (skip quals) */
    size_t qual_left;
    int done;
} or_bank;

/* next record of the file into b->cur; 0 at the end of the file */
static int or_bank_read(or_bank *b) {
    static __thread char buf[1 << 16];
    while (gzgets(b->g, buf, sizeof buf)) {
        size_t l = strlen(buf);
        int full_line = (l > 0 && buf[l - 1] == '\n');
        while (l > 0 && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) buf[--l] = 0;
        if (b->state == 3) {           /* quality lines: consume as many chars as the sequence had */
            if (l >= b->qual_left) { b->qual_left = 0; b->state = 0; } else b->qual_left -= l;
            continue;
        }
        if (b->state != 2 && buf[0] == '>') {
            const int had = b->state == 1;
            if (had) { b->cur.n = 0; sb_app(&b->cur, b->seq.b ? b->seq.b : "", b->seq.n); }
            b->seq.n = 0; b->state = 1;
            while (!full_line && gzgets(b->g, buf, sizeof buf)) { size_t m = strlen(buf); full_line = (m > 0 && buf[m - 1] == '\n'); }
            if (had) return 1;          /* the record before this header (possibly empty) */
            continue;
        }
        if (b->state == 0 && buf[0] == '@') {
            b->state = 2; b->seq.n = 0;
            while (!full_line && gzgets(b->g, buf, sizeof buf)) { size_t m = strlen(buf); full_line = (m > 0 && buf[m - 1] == '\n'); }
            continue;
        }
        if (b->state == 2) {
            if (buf[0] == '+') {
                b->cur.n = 0; sb_app(&b->cur, b->seq.b ? b->seq.b : "", b->seq.n);
                b->qual_left = b->seq.n; b->seq.n = 0; b->state = b->qual_left ? 3 : 0;
                return 1;
            }
            sb_app(&b->seq, buf, l);
            continue;
        }
        if (b->state == 1) sb_app(&b->seq, buf, l);
    }
    if (b->state == 1) { b->cur.n = 0; sb_app(&b->cur, b->seq.b ? b->seq.b : "", b->seq.n); b->seq.n = 0; b->state = 0; return 1; }
    return 0;
}
static int or_bank_first(or_bank *b) {
    if (b->g) gzclose(b->g);
    b->g = gzopen(b->path, "rb");
    if (!b->g) { b->done = 1; return -1; }
    gzbuffer(b->g, 1 << 20);
    b->state = 0; b->qual_left = 0; b->seq.n = 0;
    b->done = !or_bank_read(b);
    return 0;
}
static void or_bank_next(or_bank *b) { if (!b->done) b->done = !or_bank_read(b); }
static void or_bank_close(or_bank *b) { if (b->g) gzclose(b->g); free(b->seq.b); free(b->cur.b); memset(b, 0, sizeof *b); }

/* SimkaSequenceFilter::operator()   ref: src/core/SimkaCommons.hpp:359-436 */
static float or_shannon_index(const char *seq, size_t n) {          /* getShannonIndex, :392-432 */
    static const char nt2bin[128] = { ['C'] = 1, ['T'] = 2, ['G'] = 3, ['N'] = 4 };      /* nt2binTab :394-405: every other letter is 0 */
    float freqs[5] = { 0, 0, 0, 0, 0 };
    float index = 0;
    for (size_t i = 0; i < n; i++) freqs[(int)nt2bin[(unsigned char)seq[i] & 127]] += 1.0f;
    for (int i = 0; i < 5; i++) {
        freqs[i] /= (float)n;
        if (freqs[i] != 0) index += freqs[i] * log(freqs[i]) / log(2);
    }
    return fabsf(index);
}
static int or_filter(const oracle *o, const strbuf *seq) {
    if (o->min_read_size != 0 && !(seq->n >= o->min_read_size)) return 0;                               /* isReadSizeValid :378-381 */
    if (o->min_shannon != 0 && !(or_shannon_index(seq->b ? seq->b : "", seq->n) >= o->min_shannon)) return 0;   /* isShannonIndexValid :383-386 */
    return 1;
}

/* SimkaInputIterator   ref: src/core/SimkaCommons.hpp:159-314, member for member */
typedef struct {
    const oracle *o;
    or_bank *comp; size_t ncomp;        /* _mainref->getComposition(): every file of the sample, in the order listed */
    or_bank *ref;                       /* _ref */
    int isDone;
    size_t currentBank, nbBanks, currentInternalBank, currentDataset, nbDatasets;
    uint64_t maxReads, nbReadProcessed;
    const strbuf *item;
    int io_error;
} or_iter;

static int or_iter_is_finished(or_iter *it) {                       /* :183-189 */
    if (it->currentDataset == it->nbDatasets) { it->isDone = 1; return 1; }
    return 0;
}
static void or_iter_first(or_iter *it) {                            /* :224-236 */
    if (or_bank_first(it->ref) != 0) it->io_error = 1;
    while (!it->ref->done && !or_filter(it->o, &it->ref->cur)) or_bank_next(it->ref);
    it->isDone = it->ref->done;
    if (!it->isDone) it->item = &it->ref->cur;
}
static void or_iter_next_dataset(or_iter *it) {                     /* :191-208 */
    it->currentDataset += 1;
    if (or_iter_is_finished(it)) return;
    it->currentBank = it->currentDataset * it->nbBanks;
    it->currentInternalBank = 0;
    it->nbReadProcessed = 0;
    if (or_iter_is_finished(it)) return;
    it->ref = &it->comp[it->currentBank];
    it->isDone = 0;
    or_iter_first(it);
}
static void or_iter_next_bank(or_iter *it) {                        /* :210-222 */
    it->currentInternalBank += 1;
    if (it->currentInternalBank == it->nbBanks) or_iter_next_dataset(it);
    else {
        it->isDone = 0;
        it->currentBank += 1;
        it->ref = &it->comp[it->currentBank];
        or_iter_first(it);
    }
}
static void or_iter_next(or_iter *it) {                             /* :238-288 */
    if (or_iter_is_finished(it)) { it->isDone = 1; return; }
    or_bank_next(it->ref);
    while (!it->ref->done && !or_filter(it->o, &it->ref->cur)) or_bank_next(it->ref);
    it->isDone = it->ref->done;
    if (it->isDone) {
        if (or_iter_is_finished(it)) return;
        else {
            or_iter_next_bank(it);
            if (or_iter_is_finished(it)) return;
        }
    } else {
        it->item = &it->ref->cur;
        it->nbReadProcessed += 1;
    }
    if (it->maxReads && it->nbReadProcessed >= it->maxReads) {
        if (or_iter_is_finished(it)) return;
        else or_iter_next_dataset(it);
    }
}

/* the reads a simkaCount job hands to the counter: SimkaPotaraBankFiltered over the sample's files (ref: src/SimkaCount.cpp:44-70,267-268) */
static int or_read_sample_files(const oracle *o, or_sample *s, u64vec *out) {
    or_iter it; memset(&it, 0, sizeof it);
    it.o = o;
    it.ncomp = (size_t)s->nfiles;
    it.comp = (or_bank *)calloc(it.ncomp ? it.ncomp : 1, sizeof(or_bank));
    for (size_t f = 0; f < it.ncomp; f++) it.comp[f].path = s->files[f];
    /* constructor :165-181 */
    it.ref = &it.comp[0];
    it.isDone = 0;
    it.nbDatasets = (size_t)(s->nb_paired > 0 ? s->nb_paired : 1);
    it.nbBanks = it.ncomp / it.nbDatasets;          /* (integer division: every paired part is ASSUMED to list as many files) */
    it.maxReads = o->max_reads;
    int rc = 0;
    if (it.ncomp == 0 || it.nbBanks == 0) rc = -1;
    else
        for (or_iter_first(&it); !it.isDone; or_iter_next(&it)) {
            s->k_occ += or_kmers_of_read(it.item->b ? it.item->b : "", it.item->n, o->k, out, (k256_dict *)&o->dict);
            s->nb_reads++;
        }
    if (it.io_error) rc = -1;
    for (size_t f = 0; f < it.ncomp; f++) or_bank_close(&it.comp[f]);
    free(it.comp);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* [a3]+[a4]+[a5] per-sample counting, abundance filter, totals               */
/* ------------------------------------------------------------------------- */
/* `threads` > 1: the k-mer space is cut into 4096 prefix ranges, sorted and counted by a team of threads (the reference's
 * counterpart: gatb counts the partitions of one sample on all its cores).  The result is the same sorted spectrum. */
#define OR_NB_RANGES 4096
typedef struct { size_t nd, ns; uint64_t D, N, Q; } or_range_out;

static void or_count_range(const oracle *o, kmer_t *v, size_t n, int sort_bits, or_range_out *r, kmer_t *ok, uint32_t *oc) {
    radix_sort_u64(v, n, sort_bits);
    memset(r, 0, sizeof *r);
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && v[j] == v[i]) j++;
        uint64_t c = j - i;
        r->nd++;
        /* SimkaCompressedProcessor::process, ref: src/minikc/MiniKC.hpp:54-62 */
        if (!(c < o->amin || c > o->amax)) {
            if (ok) { ok[r->ns] = v[i]; oc[r->ns] = (uint32_t)c; }
            r->ns++;
            r->D += 1;                  /* _nbDistinctKmerPerParts[partId] += 1 */
            r->N += c;                  /* _nbKmerPerParts[partId] += count[0]  */
            r->Q += (uint64_t)pow((double)c, 2);  /* _chordPerParts[partId] += pow(count[0], 2) */
        }
        i = j;
    }
}

static int or_count_sample(oracle *o, or_sample *s, int threads) {
    s->nb_reads = 0; s->k_occ = 0;
    if (threads < 1) threads = 1;
    const int W = o->k >= 64 ? 64 : 2 * o->k;              /* (k >= 64: the "k-mers" are dictionary ranks, or_kmers_of_read_256) */
    const int rb = (W >= 12 && o->k < 64) ? 12 : 0;        /* range = top 12 bits of the k-mer (one range for tiny k and for ranks) */
    const size_t NR = (size_t)1 << rb;
    const int T = (s->bases && s->nreads_mem >= 64) ? threads : 1;        /* extraction threads (files are read by one iterator) */
    u64vec *loc = (u64vec *)calloc((size_t)T, sizeof(u64vec));
    uint64_t *tk = (uint64_t *)calloc((size_t)T, sizeof(uint64_t));
    if (s->bases) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(T)
#endif
        for (int t = 0; t < T; t++) {
            const uint64_t r0 = s->nreads_mem * (uint64_t)t / (uint64_t)T, r1 = s->nreads_mem * (uint64_t)(t + 1) / (uint64_t)T;
            for (uint64_t r = r0; r < r1; r++)
                tk[t] += or_kmers_of_read(s->bases + s->offsets[r], (size_t)(s->offsets[r + 1] - s->offsets[r]), o->k, &loc[t], &o->dict);
        }
        s->nb_reads = s->nreads_mem;
        for (int t = 0; t < T; t++) s->k_occ += tk[t];
    } else {
        if (or_read_sample_files(o, s, &loc[0]) != 0) {
            snprintf(o->err, sizeof o->err, "ERROR: Can't open dataset: %s", s->id);
            free(loc[0].v); free(loc); free(tk); return -1;
        }
    }
    size_t total = 0;
    for (int t = 0; t < T; t++) total += loc[t].n;
    /* gatb DSK emits each distinct canonical k-mer once, in increasing order, with its count (the merge's min-heap assumes
     * sorted streams, ref: src/SimkaMerge.cpp:1198-1263): scatter by range, sort and count every range, concatenate */
    size_t *hist = (size_t *)calloc((size_t)T * NR + 1, sizeof(size_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(T)
#endif
    for (int t = 0; t < T; t++)
        for (size_t i = 0; i < loc[t].n; i++) hist[(size_t)t * NR + (rb ? (size_t)(loc[t].v[i] >> (W - rb)) : 0)]++;
    size_t *rstart = (size_t *)malloc((NR + 1) * sizeof(size_t));
    {
        size_t run = 0;
        for (size_t r = 0; r < NR; r++) {
            rstart[r] = run;
            for (int t = 0; t < T; t++) { const size_t c = hist[(size_t)t * NR + r]; hist[(size_t)t * NR + r] = run; run += c; }
        }
        rstart[NR] = run;
    }
    kmer_t *all = (kmer_t *)malloc((total ? total : 1) * sizeof(kmer_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(T)
#endif
    for (int t = 0; t < T; t++) {
        size_t *cur = hist + (size_t)t * NR;
        for (size_t i = 0; i < loc[t].n; i++) all[cur[rb ? (size_t)(loc[t].v[i] >> (W - rb)) : 0]++] = loc[t].v[i];
        free(loc[t].v); loc[t].v = NULL;
    }
    free(loc); free(tk); free(hist);
    or_range_out *ro = (or_range_out *)calloc(NR, sizeof(or_range_out));
    /* pass 1: sort + count every range; pass 2: the solid records at their final place */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (size_t r = 0; r < NR; r++) or_count_range(o, all + rstart[r], rstart[r + 1] - rstart[r], W - rb, &ro[r], NULL, NULL);
    size_t nd = 0, ns = 0;
    size_t *sstart = (size_t *)malloc((NR + 1) * sizeof(size_t));
    s->D = s->N = s->Q = 0;
    for (size_t r = 0; r < NR; r++) { sstart[r] = ns; nd += ro[r].nd; ns += ro[r].ns; s->D += ro[r].D; s->N += ro[r].N; s->Q += ro[r].Q; }
    s->kmer = (kmer_t *)malloc((ns ? ns : 1) * sizeof(kmer_t));
    s->count = (uint32_t *)malloc((ns ? ns : 1) * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (size_t r = 0; r < NR; r++) {
        const kmer_t *v = all + rstart[r]; const size_t n = rstart[r + 1] - rstart[r];
        size_t w = sstart[r];
        for (size_t i = 0; i < n;) {            /* (the range is sorted already) */
            size_t j = i + 1;
            while (j < n && v[j] == v[i]) j++;
            const uint64_t c = j - i;
            if (!(c < o->amin || c > o->amax)) { s->kmer[w] = v[i]; s->count[w] = (uint32_t)c; w++; }
            i = j;
        }
    }
    s->d_all = nd; s->nsolid = ns;
    free(all); free(ro); free(rstart); free(sstart);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* SimkaStatistics   ref: src/core/SimkaDistance.cpp:27-153                   */
/* ------------------------------------------------------------------------- */
static or_stats *or_stats_new(const oracle *o, int simple, int complex_) {
    int n = o->n;
    or_stats *st = (or_stats *)calloc(1, sizeof *st);
    st->n = n; st->simple = simple; st->complex_ = complex_;
    st->symsize = ((size_t)n * (n + 1)) / 2;                      /* :31 */
    st->D = (uint64_t *)calloc(n, 8); st->Nk = (uint64_t *)calloc(n, 8);
    st->sqrtN2 = (long double *)calloc(n, sizeof(long double));
    st->a = (uint64_t *)calloc(st->symsize, 8); st->bc = (uint64_t *)calloc(st->symsize, 8);
    st->S = (uint64_t *)calloc((size_t)n * n, 8);
    st->chord = (long double *)calloc((size_t)n * n, sizeof(long double));
    st->hell = (uint64_t *)calloc((size_t)n * n, 8); st->kul = (uint64_t *)calloc((size_t)n * n, 8);
    st->whit = (uint64_t *)calloc((size_t)n * n, 8); st->canb = (uint64_t *)calloc((size_t)n * n, 8);
    st->kl = (long double *)calloc((size_t)n * n, sizeof(long double));
    for (int i = 0; i < n; i++) {                                 /* totals come from the .ok files, :116-151 */
        st->D[i] = o->s[i].D; st->Nk[i] = o->s[i].N;
        st->sqrtN2[i] = sqrt((double)o->s[i].Q);                  /* :139  sqrt(strtoull(..)) -> double overload */
    }
    return st;
}
static void or_stats_free(or_stats *st) {
    if (!st) return;
    free(st->D); free(st->Nk); free(st->sqrtN2); free(st->a); free(st->bc); free(st->S); free(st->chord);
    free(st->hell); free(st->kul); free(st->whit); free(st->canb); free(st->kl); free(st);
}
/* [a9] operator+=   ref: src/core/SimkaDistance.cpp:156-213 (per-sample vectors are NOT summed) */
static void or_stats_add(or_stats *d, const or_stats *s) {
    size_t nn = (size_t)d->n * d->n;
    d->nb_distinct += s->nb_distinct; d->nb_shared += s->nb_shared;
    for (size_t i = 0; i < d->symsize; i++) { d->bc[i] += s->bc[i]; d->a[i] += s->a[i]; }
    for (size_t i = 0; i < nn; i++) d->S[i] += s->S[i];
    if (d->simple) for (size_t i = 0; i < nn; i++) { d->chord[i] += s->chord[i]; d->hell[i] += s->hell[i]; d->kul[i] += s->kul[i]; }
    if (d->complex_) for (size_t i = 0; i < nn; i++) { d->canb[i] += s->canb[i]; d->whit[i] += s->whit[i]; d->kl[i] += s->kl[i]; }
}

/* ------------------------------------------------------------------------- */
/* [a8] accumulators   ref: src/core/SimkaAlgorithm.hpp:341-516               */
/* counts[] is the dense CountVector of one k-mer; shared[] = {i : counts[i]}  */
/* ------------------------------------------------------------------------- */
static inline size_t or_sym(size_t n, size_t i, size_t j) { return j + ((n - 1) * i) - (i * (i - 1) / 2); } /* :364 */

static void or_update_default(or_stats *st, const int32_t *counts, const uint16_t *shared, size_t ns) {
    size_t n = st->n;
    for (size_t ii = 0; ii < ns; ii++)
        for (size_t jj = ii + 1; jj < ns; jj++) {
            size_t i = shared[ii], j = shared[jj];
            size_t sym = or_sym(n, i, j);
            uint64_t ai = (uint64_t)counts[i], aj = (uint64_t)counts[j];
            st->S[i * n + j] += counts[i];                       /* :369 */
            st->S[j * n + i] += counts[j];                       /* :370 */
            st->a[sym] += 1;                                     /* :371 */
            st->bc[sym] += ai < aj ? ai : aj;                    /* :374 */
        }
}
static void or_update_simple(or_stats *st, const int32_t *counts, const uint16_t *shared, size_t ns) {
    size_t n = st->n;
    for (size_t ii = 0; ii < ns; ii++)
        for (size_t jj = ii + 1; jj < ns; jj++) {
            size_t i = shared[ii], j = shared[jj];
            uint64_t ai = (uint64_t)counts[i], aj = (uint64_t)counts[j];
            st->chord[i * n + j] += ai * aj;                                        /* :396 long double += u64 */
            /* :397  u64 += double  ==  (u64)((double)u64 + sqrt((double)(ai*aj))) */
            st->hell[i * n + j] = (uint64_t)((double)st->hell[i * n + j] + sqrt((double)(ai * aj)));
            st->kul[i * n + j] += ai < aj ? ai : aj;                                /* :398 only [i][j], i<j */
        }
}
static inline uint64_t or_whit_term(double ai, double aj, uint64_t Ni, uint64_t Nj) {
    /* :481  abs((int)((u_int64_t)(ai*N_j) - (u_int64_t)(aj*N_i)))  then u64 += int */
    uint64_t d = (uint64_t)(ai * (double)Nj) - (uint64_t)(aj * (double)Ni);
    int t = (int)d;
    int r = (t == INT_MIN) ? INT_MIN : abs(t);
    return (uint64_t)(int64_t)r;
}
static void or_update_complex(or_stats *st, const int32_t *counts, const uint16_t *shared, size_t ns) {
    size_t n = st->n;
    const uint64_t *Nk = st->Nk;
    for (size_t i = 0; i < n; i++) {
        if (counts[i]) {                                                        /* :418 */
            for (size_t j = i + 1; j < n; j++) {
                double ai = counts[i], aj = counts[j];
                double d1, d2;
                double yX = aj * (double)Nk[i], xY = ai * (double)Nk[j];
                double xi = ai / (double)Nk[i];
                d1 = xi * log((2 * xY) / (xY + yX));                            /* :440 / :453 */
                if (aj) { double xj = aj / (double)Nk[j]; d2 = xj * log((2 * yX) / (xY + yX)); }   /* :445-446 */
                else d2 = 0;
                st->kl[i * n + j] += d1 + d2;                                   /* :477 */
                st->canb[i * n + j] = (uint64_t)((double)st->canb[i * n + j] + fabs(ai - aj) / (ai + aj)); /* :479 */
                st->whit[i * n + j] += or_whit_term(ai, aj, Nk[i], Nk[j]);      /* :481 */
            }
        } else {                                                                /* :488-515, counts[i]==0 */
            for (size_t jj = 0; jj < ns; jj++) {
                size_t j = shared[jj];
                if (i > j) continue;
                double ai = counts[i], aj = counts[j];
                double xY = ai * (double)Nk[j], yX = aj * (double)Nk[i];
                double xj = aj / (double)Nk[j];
                double d2 = xj * log((2 * yX) / (xY + yX));                     /* :504 */
                st->kl[i * n + j] += 0 + d2;                                    /* :506 */
                st->canb[i * n + j] = (uint64_t)((double)st->canb[i * n + j] + fabs(ai - aj) / (ai + aj));
                st->whit[i * n + j] += or_whit_term(ai, aj, Nk[i], Nk[j]);      /* :512 */
            }
        }
    }
}
/* updateDistance :341-354, gated by SimkaMergeAlgorithm::insert, ref: src/SimkaMerge.cpp:1307-1326 */
static void or_insert(or_stats *st, const int32_t *counts, size_t nb_having, uint16_t *shared) {
    st->nb_distinct += 1;                                                       /* :1315 */
    if (st->complex_ || nb_having > 1) {                                        /* :1317 */
        if (nb_having > 1) st->nb_shared += 1;                                  /* :1319-1321 */
        size_t ns = 0;
        for (int i = 0; i < st->n; i++) if (counts[i]) shared[ns++] = (uint16_t)i;
        or_update_default(st, counts, shared, ns);
        if (st->simple) or_update_simple(st, counts, shared, ns);
        if (st->complex_) or_update_complex(st, counts, shared, ns);
    }
}

/* ------------------------------------------------------------------------- */
/* [a6] N-way merge of one partition   ref: src/SimkaMerge.cpp:1164-1264      */
/* The reference's partition = f(minimizer); distances do not depend on the   */
/* choice (ref: tests/simple_test.py:125-133), here partition = mix(kmer)%P.  */
/* ------------------------------------------------------------------------- */
static inline uint64_t or_mix(kmer_t km) {
    uint64_t x = (uint64_t)km ^ ((uint64_t)(km >> 64) * 0x9E3779B97F4A7C15ULL);      /* fold the high word in (k >= 33) */
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
typedef struct { kmer_t kmer; int bank; } heap_item;
static void heap_sift_down(heap_item *h, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && (h[l].kmer < h[m].kmer || (h[l].kmer == h[m].kmer && h[l].bank < h[m].bank))) m = l;
        if (r < n && (h[r].kmer < h[m].kmer || (h[r].kmer == h[m].kmer && h[r].bank < h[m].bank))) m = r;
        if (m == i) return;
        heap_item t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}
static void or_merge_partition(const oracle *o, or_stats *st, unsigned part, unsigned nparts) {
    int n = o->n;
    size_t *pos = (size_t *)calloc(n, sizeof(size_t));
    heap_item *heap = (heap_item *)malloc(n * sizeof(heap_item));
    int32_t *counts = (int32_t *)calloc(n, sizeof(int32_t));     /* CountVector (CountNumber = int32) */
    uint16_t *shared = (uint16_t *)malloc(n * sizeof(uint16_t));
    size_t hn = 0;
#define OR_ADVANCE(b) while (pos[b] < o->s[b].nsolid && nparts > 1 && or_mix(o->s[b].kmer[pos[b]]) % nparts != part) pos[b]++
    for (int b = 0; b < n; b++) {
        OR_ADVANCE(b);
        if (pos[b] < o->s[b].nsolid) { heap[hn].kmer = o->s[b].kmer[pos[b]]; heap[hn].bank = b; hn++; }
    }
    for (size_t i = hn; i-- > 0;) heap_sift_down(heap, hn, i);
    while (hn) {
        kmer_t cur = heap[0].kmer;
        size_t nb_having = 0;
        memset(counts, 0, n * sizeof(int32_t));                   /* SimkaCounterBuilderMerge::init :293-297 */
        while (hn && heap[0].kmer == cur) {
            int b = heap[0].bank;
            counts[b] += (int32_t)o->s[b].count[pos[b]];          /* increase :301 */
            nb_having++;
            pos[b]++;
            OR_ADVANCE(b);
            if (pos[b] < o->s[b].nsolid) heap[0].kmer = o->s[b].kmer[pos[b]];
            else heap[0] = heap[--hn];
            heap_sift_down(heap, hn, 0);
        }
        or_insert(st, counts, nb_having, shared);                 /* :1240 / :1263 */
    }
#undef OR_ADVANCE
    free(pos); free(heap); free(counts); free(shared);
}

/* ------------------------------------------------------------------------- */
/* [a10] final distances   ref: src/core/SimkaDistance.cpp:920-1226           */
/* ------------------------------------------------------------------------- */
static void or_abc(const or_stats *st, size_t i, size_t j, size_t sym, uint64_t *a, uint64_t *b, uint64_t *c) {
    *a = st->a[sym]; *b = st->D[i] - *a; *c = st->D[j] - *a;                     /* :920-926 */
}
static double d_ab_braycurtis(const or_stats *st, size_t i, size_t j, size_t sym) {     /* :928-939 */
    double union_ = st->Nk[i] + st->Nk[j];
    if (union_ == 0) return 1;
    double intersection = 2 * st->bc[sym];
    return 1 - intersection / union_;
}
static double d_ab_chord(const or_stats *st, size_t i, size_t j) {                      /* :942-959 */
    double den = st->sqrtN2[i] * st->sqrtN2[j];
    if (den == 0) return sqrt(2);
    long double r = sqrtl(2 - 2 * st->chord[i * st->n + j] / den);
    return r;
}
static double d_ab_hellinger(const or_stats *st, size_t i, size_t j) {                  /* :962-972 */
    double union_ = sqrt((double)st->Nk[i]) * sqrt((double)st->Nk[j]);
    if (union_ == 0) return sqrt(2);
    double intersection = 2 * st->hell[i * st->n + j];
    return sqrt(2 - (intersection / union_));
}
static double d_ab_whittaker(const or_stats *st, size_t i, size_t j) {                  /* :988-998 */
    long double union_ = st->Nk[i] * st->Nk[j];
    if (union_ == 0) return 1;
    long double intersection = st->whit[i * st->n + j];
    double w = 0.5 * (intersection / union_);
    return w;
}
static double d_ab_kl(const or_stats *st, size_t i, size_t j) {                         /* :1001-1007 */
    if (st->kl[i * st->n + j] == 0) return 1;
    return sqrtl(0.5 * st->kl[i * st->n + j]);
}
static double d_ab_canberra(const or_stats *st, size_t i, size_t j, uint64_t ua, uint64_t ub, uint64_t uc) { /* :1010-1021 */
    double a = (double)ua, b = (double)ub, c = (double)uc;
    if ((a + b + c) == 0) return 1;
    return (1 / (a + b + c)) * st->canb[i * st->n + j];
}
static double d_ab_kulczynski(const or_stats *st, size_t i, size_t j) {                 /* :1024-1038 */
    if (st->Nk[i] == 0 || st->Nk[j] == 0) return 1;
    long double n1 = (double)st->kul[i * st->n + j] / (double)st->Nk[i];
    long double n2 = (double)st->kul[j * st->n + i] / (double)st->Nk[j];   /* never written for j>i: 0 */
    double r = 1 - 0.5 * (n1 + n2);
    return r;
}
static double d_ab_jaccard_simka(const or_stats *st, size_t i, size_t j, int asym) {    /* :1041-1065 */
    double A1 = st->S[i * st->n + j], B1 = st->S[j * st->n + i], A0 = st->Nk[i], B0 = st->Nk[j];
    double num, den;
    if (!asym) { num = A1 + B1; den = A0 + B0; } else { num = A1; den = A0; }
    if (den == 0) return 1;
    return 1 - num / den;
}
static double d_ab_ochiai(const or_stats *st, size_t i, size_t j) {                     /* :1068-1078 */
    double A1 = st->S[i * st->n + j], B1 = st->S[j * st->n + i], A0 = st->Nk[i], B0 = st->Nk[j];
    if (A0 == 0 || B0 == 0) return 1;
    return 1 - sqrt(A1 / A0) * sqrt(B1 / B0);
}
static double d_ab_sorensen(const or_stats *st, size_t i, size_t j) {                   /* :1081-1096 */
    double A1 = st->S[i * st->n + j], B1 = st->S[j * st->n + i], A0 = st->Nk[i], B0 = st->Nk[j];
    double num = 2 * A1 * B1, den = A0 * B1 + A1 * B0;
    if (den == 0) return 1;
    return 1 - num / den;
}
static double d_ab_jaccard(const or_stats *st, size_t i, size_t j) {                    /* :1099-1115 */
    double A1 = st->S[i * st->n + j], B1 = st->S[j * st->n + i], A0 = st->Nk[i], B0 = st->Nk[j];
    double num = A1 * B1, den = A0 * B1 + A1 * B0 - A1 * B1;
    if (den == 0) return 1;
    return 1 - num / den;
}
static double d_pa_chord(uint64_t ua, uint64_t ub, uint64_t uc) {                       /* :1117-1127 */
    double a = ua, b = ub, c = uc;
    double p1 = sqrt((a + b) * (a + c));
    if (p1 == 0) return sqrt(2);
    return sqrt(2 * (1 - a / p1));
}
static double d_pa_whittaker(uint64_t ua, uint64_t ub, uint64_t uc) {                   /* :1129-1145 */
    double a = ua, b = ub, c = uc;
    if (a + b == 0 || a + c == 0) return 1;
    double p1 = b / (a + b), p2 = c / (a + c), p3 = a / (a + b), p4 = a / (a + c);
    return 0.5 * (p1 + p2 + fabs(p3 - p4));
}
static double d_pa_kulczynski(uint64_t ua, uint64_t ub, uint64_t uc) {                  /* :1156-1170 */
    double a = ua, b = ub, c = uc;
    if (a + b == 0 || a + c == 0) return 1;
    return 1 - 0.5 * (a / (a + b) + a / (a + c));
}
static double d_pa_braycurtis(uint64_t ua, uint64_t ub, uint64_t uc) {                  /* :1172-1183 */
    double a = ua, b = ub, c = uc;
    if ((2 * a + b + c) == 0) return 1;
    return (b + c) / (2 * a + b + c);
}
static double d_pa_ochiai(uint64_t ua, uint64_t ub, uint64_t uc) {                      /* :1188-1198 */
    double a = ua, b = ub, c = uc;
    float val = sqrt((a + b) * (a + c));            /* float32 temporary, as in the reference */
    if (val == 0) return 1;
    return 1 - (a / val);
}
static double d_pa_jaccard(uint64_t ua, uint64_t ub, uint64_t uc) {                     /* :1200-1208 */
    double a = ua, b = ub, c = uc;
    if ((a + b + c) == 0) return 1;
    return (b + c) / (a + b + c);
}
static double d_pa_jaccard_simka(const or_stats *st, size_t i, size_t j, size_t sym, int asym) { /* :1210-1226 */
    double num, den;
    (void)j;
    if (!asym) { num = 2 * st->a[sym]; den = st->D[i] + st->D[j]; }
    else { num = st->a[sym]; den = st->D[i]; }
    if (den == 0) return 1;
    return 1 - num / den;
}

/* matrix names in reference output order, ref: src/core/SimkaDistance.cpp:617-647 */
static const char *OR_MATRIX_NAMES[] = {
    "mat_presenceAbsence_chord", "mat_presenceAbsence_whittaker", "mat_presenceAbsence_kulczynski",
    "mat_presenceAbsence_braycurtis", "mat_presenceAbsence_jaccard", "mat_presenceAbsence_simka-jaccard",
    "mat_presenceAbsence_simka-jaccard_asym", "mat_presenceAbsence_ochiai",
    "mat_abundance_simka-jaccard", "mat_abundance_simka-jaccard_asym", "mat_abundance_ab-ochiai",
    "mat_abundance_ab-sorensen", "mat_abundance_ab-jaccard", "mat_abundance_braycurtis", "mat_abundance_jaccard",
    "mat_abundance_chord", "mat_abundance_hellinger", "mat_abundance_kulczynski",          /* -simple-dist  */
    "mat_abundance_whittaker", "mat_abundance_jensenshannon", "mat_abundance_canberra" };  /* -complex-dist */
#define OR_NB_MATRICES 21

/* Build matrix `which` (index into OR_MATRIX_NAMES) as float32 cells,
 * ref: src/core/SimkaDistance.hpp:155-475 (zero-filled, loops over i<j only). */
static void or_matrix(const or_stats *st, int which, float *m) {
    size_t n = st->n;
    memset(m, 0, n * n * sizeof(float));
    if (which == 14) {   /* computeJaccardDistanceFromBrayCurtis, .hpp:463-475: every cell, from the FLOAT BC matrix */
        float *bcm = (float *)malloc(n * n * sizeof(float));
        or_matrix(st, 13, bcm);
        for (size_t i = 0; i < n * n; i++) { double B = bcm[i]; double J = (2 * B) / (1 + B); m[i] = J; }
        free(bcm);
        return;
    }
    for (size_t i = 0; i < n; i++)
        for (size_t j = i + 1; j < n; j++) {
            size_t sym = or_sym(n, i, j);
            uint64_t a, b, c;
            or_abc(st, i, j, sym, &a, &b, &c);
            double dij = 0, dji = 0; int asym = 0;
            switch (which) {
                case 0: dij = d_pa_chord(a, b, c); break;
                case 1: dij = d_pa_whittaker(a, b, c); break;
                case 2: dij = d_pa_kulczynski(a, b, c); break;
                case 3: dij = d_pa_braycurtis(a, b, c); break;
                case 4: dij = d_pa_jaccard(a, b, c); break;
                case 5: dij = d_pa_jaccard_simka(st, i, j, sym, 0); break;
                case 6: dij = d_pa_jaccard_simka(st, i, j, sym, 1); dji = d_pa_jaccard_simka(st, j, i, sym, 1); asym = 1; break;
                case 7: dij = d_pa_ochiai(a, b, c); break;
                case 8: dij = d_ab_jaccard_simka(st, i, j, 0); break;
                case 9: dij = d_ab_jaccard_simka(st, i, j, 1); dji = d_ab_jaccard_simka(st, j, i, 1); asym = 1; break;
                case 10: dij = d_ab_ochiai(st, i, j); break;
                case 11: dij = d_ab_sorensen(st, i, j); break;
                case 12: dij = d_ab_jaccard(st, i, j); break;
                case 13: dij = d_ab_braycurtis(st, i, j, sym); break;
                case 15: dij = d_ab_chord(st, i, j); break;
                case 16: dij = d_ab_hellinger(st, i, j); break;
                case 17: dij = d_ab_kulczynski(st, i, j); break;
                case 18: dij = d_ab_whittaker(st, i, j); break;
                case 19: dij = d_ab_kl(st, i, j); break;
                case 20: dij = d_ab_canberra(st, i, j, a, b, c); break;
            }
            m[i * n + j] = (float)dij;
            m[j * n + i] = (float)(asym ? dji : dij);
        }
}

/* [a11] CSV   ref: src/core/SimkaDistance.cpp:653-699 */
static int or_dump_matrix(const oracle *o, const char *dir, const char *name, const float *m, int gz) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s.csv%s", dir, name, gz ? ".gz" : "");
    strbuf s = {0};
    char cell[64];
    for (int i = 0; i < o->n; i++) { sb_app(&s, ";", 1); sb_app(&s, o->s[i].id, strlen(o->s[i].id)); }
    sb_app(&s, "\n", 1);
    for (int i = 0; i < o->n; i++) {
        sb_app(&s, o->s[i].id, strlen(o->s[i].id));
        for (int j = 0; j < o->n; j++) {
            int l = snprintf(cell, sizeof cell, ";%f", (double)m[(size_t)i * o->n + j]);
            sb_app(&s, cell, (size_t)l);
        }
        sb_app(&s, "\n", 1);
    }
    int rc = 0;
    if (gz) { gzFile g = gzopen(path, "wb"); if (!g) rc = -1; else { gzwrite(g, s.b, (unsigned)s.n); gzclose(g); } }
    else { FILE *f = fopen(path, "wb"); if (!f) rc = -1; else { fwrite(s.b, 1, s.n, f); fclose(f); } }
    free(s.b);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* public (test-only) API, used through ctypes                                */
/* ------------------------------------------------------------------------- */
OR_API oracle *oracle_new(void) { return (oracle *)calloc(1, sizeof(oracle)); }

OR_API void oracle_free(oracle *o) {
    if (!o) return;
    for (int i = 0; i < o->n; i++) {
        free(o->s[i].id);
        for (int f = 0; f < o->s[i].nfiles; f++) free(o->s[i].files[f]);
        free(o->s[i].files); free(o->s[i].kmer); free(o->s[i].count);
    }
    free(o->s); or_stats_free(o->stats); free(o->dict.v); free(o);
}
OR_API const char *oracle_error(const oracle *o) { return o->err; }
OR_API int oracle_load_input(oracle *o, const char *input_txt) { return or_parse_input(o, input_txt); }
OR_API int oracle_nb_samples(const oracle *o) { return o->n; }
OR_API const char *oracle_sample_id(const oracle *o, int i) { return o->s[i].id; }
OR_API int oracle_sample_nb_files(const oracle *o, int i) { return o->s[i].nfiles; }
OR_API const char *oracle_sample_file(const oracle *o, int i, int f) { return o->s[i].files[f]; }
OR_API int oracle_sample_nb_paired(const oracle *o, int i) { return o->s[i].nb_paired; }

/* [a2] read policies of the run (apply to samples read from files): -max-reads m (0 = all), -min-read-size, -min-shannon-index */
OR_API void oracle_set_read_policy(oracle *o, uint64_t max_reads, uint64_t min_read_size, double min_shannon) {
    o->max_reads = max_reads; o->min_read_size = min_read_size; o->min_shannon = min_shannon;
}
/* -max-reads 0: (min + mean) / 2 of the samples' read counts divided by their paired parts (computeMaxReads, ref:
 * src/core/SimkaAlgorithm.cpp:377-445).  The reference takes gatb's estimateNbItems() (an ESTIMATE from the head of the
 * file, gatb-core absent); exact counts are used here. */
OR_API uint64_t oracle_auto_max_reads(oracle *o) {
    uint64_t total = 0, mn = (uint64_t)-1;
    oracle tmp = *o; tmp.max_reads = 0; tmp.min_read_size = 0; tmp.min_shannon = 0;
    for (int i = 0; i < o->n; i++) {
        uint64_t n = 0;
        for (int f = 0; f < o->s[i].nfiles; f++) {
            or_bank b; memset(&b, 0, sizeof b); b.path = o->s[i].files[f];
            for (or_bank_first(&b); !b.done; or_bank_next(&b)) n++;
            or_bank_close(&b);
        }
        n /= (uint64_t)(o->s[i].nb_paired > 0 ? o->s[i].nb_paired : 1);
        total += n; if (n < mn) mn = n;
    }
    (void)tmp;
    return o->n ? (mn + total / (uint64_t)o->n) / 2 : 0;
}

/* in-memory sample: `bases` = concatenated ASCII reads, offsets[nreads+1]; pointers must outlive oracle_run */
OR_API int oracle_add_sample_mem(oracle *o, const char *id, const char *bases, const uint64_t *offsets, uint64_t nreads) {
    o->s = (or_sample *)realloc(o->s, (o->n + 1) * sizeof(or_sample));
    or_sample *s = &o->s[o->n++];
    memset(s, 0, sizeof *s);
    s->id = strdup(id); s->bases = bases; s->offsets = offsets; s->nreads_mem = nreads; s->nb_paired = 1;
    return o->n - 1;
}

/* count every sample, merge the partitions p of `nparts` with p % shard_count == shard_index, reduce.
 * (shard_count=1: everything.)  threads<=1 -> serial. */
OR_API int oracle_run_shard(oracle *o, int k, uint32_t amin, uint32_t amax, int simple, int complex_, unsigned nparts, int threads,
                            unsigned shard_index, unsigned shard_count) {
    if (k < 1 || k > 127) { snprintf(o->err, sizeof o->err, "oracle: k must be in [1,127]"); return -1; }
    o->k = k; o->amin = amin; o->amax = amax > 999999999u ? 999999999u : amax;  /* ref: src/core/SimkaAlgorithm.cpp:188 */
    if (nparts < 1) nparts = 1;
    int rc = 0;
    for (int i = 0; i < o->n; i++) { free(o->s[i].kmer); free(o->s[i].count); o->s[i].kmer = NULL; o->s[i].count = NULL; }
    if (threads < 1) threads = 1;
    o->dict.mode = 0;
    if (k >= 64) {      /* pass 1: the dictionary of the run's canonical k-mers (the counts of this pass are thrown away) */
        o->dict.n = 0; o->dict.failed = 0; o->dict.mode = 1;
        for (int i = 0; i < o->n; i++) {
            if (or_count_sample(o, &o->s[i], threads) != 0) { o->dict.mode = 0; return -1; }
            free(o->s[i].kmer); free(o->s[i].count); o->s[i].kmer = NULL; o->s[i].count = NULL;
        }
        if (o->dict.failed) { o->dict.mode = 0; snprintf(o->err, sizeof o->err, "oracle: out of memory for the dictionary of %d-base k-mers", k); return -1; }
        k256_build_dictionary(&o->dict);
        o->dict.mode = 2;
    }
    /* at least as many samples as threads: one sample per thread (as the reference runs one simkaCount job per sample);
     * fewer: the samples one after the other, each on all threads */
    if (o->n >= threads || threads == 1) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
        for (int i = 0; i < o->n; i++) { if (or_count_sample(o, &o->s[i], 1) != 0) rc = -1; }
    } else {
        for (int i = 0; i < o->n; i++) { if (or_count_sample(o, &o->s[i], threads) != 0) rc = -1; }
    }
    if (rc) return rc;
    or_stats_free(o->stats);
    o->stats = or_stats_new(o, simple, complex_);
    or_stats **ps = (or_stats **)calloc(nparts, sizeof(or_stats *));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (unsigned p = 0; p < nparts; p++) {
        if (shard_count > 1 && p % shard_count != shard_index) continue;
        ps[p] = or_stats_new(o, simple, complex_); or_merge_partition(o, ps[p], p, nparts);
    }
    for (unsigned p = 0; p < nparts; p++) if (ps[p]) { or_stats_add(o->stats, ps[p]); or_stats_free(ps[p]); }  /* stats(), ref: src/SimkaPotara.hpp:1152-1187 */
    free(ps);
    (void)threads;
    return 0;
}
OR_API int oracle_run(oracle *o, int k, uint32_t amin, uint32_t amax, int simple, int complex_, unsigned nparts, int threads) {
    return oracle_run_shard(o, k, amin, amax, simple, complex_, nparts, threads, 0, 1);
}
/* D,N,Q restricted to the k-mers of one shard's partitions: out[i*3 + {0:D 1:N 2:Q}] (they add up over shards) */
OR_API void oracle_get_shard_totals(const oracle *o, unsigned nparts, unsigned shard_index, unsigned shard_count, uint64_t *out) {
    for (int i = 0; i < o->n; i++) {
        const or_sample *s = &o->s[i];
        uint64_t D = 0, N = 0, Q = 0;
        for (size_t j = 0; j < s->nsolid; j++) {
            unsigned p = nparts > 1 ? (unsigned)(or_mix(s->kmer[j]) % nparts) : 0;
            if (shard_count > 1 && p % shard_count != shard_index) continue;
            uint64_t c = s->count[j];
            D += 1; N += c; Q += c * c;
        }
        out[i * 3 + 0] = D; out[i * 3 + 1] = N; out[i * 3 + 2] = Q;
    }
}

/* per-sample totals: out[i*6 + {0:nbReads 1:K_occ 2:D_all 3:D 4:N 5:Q}] */
OR_API void oracle_get_totals(const oracle *o, uint64_t *out) {
    for (int i = 0; i < o->n; i++) {
        const or_sample *s = &o->s[i];
        out[i * 6 + 0] = s->nb_reads; out[i * 6 + 1] = s->k_occ; out[i * 6 + 2] = s->d_all;
        out[i * 6 + 3] = s->D; out[i * 6 + 4] = s->N; out[i * 6 + 5] = s->Q;
    }
}
OR_API void oracle_get_global(const oracle *o, uint64_t *out) { out[0] = o->stats->nb_distinct; out[1] = o->stats->nb_shared; }
/* accumulators as dense N x N u64 ([i][j], i<j carries the sym ones too); which:
 * 0 S, 1 a (sym->[i][j]), 2 bc (sym->[i][j]), 3 chord (long double -> u64, exact below 2^64),
 * 4 hell, 5 kul, 6 whit, 7 canb */
OR_API void oracle_get_acc_u64(const oracle *o, int which, uint64_t *out) {
    const or_stats *st = o->stats; size_t n = st->n;
    memset(out, 0, n * n * 8);
    for (size_t i = 0; i < n; i++) for (size_t j = 0; j < n; j++) {
        uint64_t v = 0;
        switch (which) {
            case 0: v = st->S[i * n + j]; break;
            case 1: v = (i < j) ? st->a[or_sym(n, i, j)] : 0; break;
            case 2: v = (i < j) ? st->bc[or_sym(n, i, j)] : 0; break;
            case 3: v = (uint64_t)st->chord[i * n + j]; break;
            case 4: v = st->hell[i * n + j]; break;
            case 5: v = st->kul[i * n + j]; break;
            case 6: v = st->whit[i * n + j]; break;
            case 7: v = st->canb[i * n + j]; break;
        }
        out[i * n + j] = v;
    }
}
OR_API void oracle_get_kl(const oracle *o, double *out) {
    size_t n = o->stats->n;
    for (size_t i = 0; i < n * n; i++) out[i] = (double)o->stats->kl[i];
}
OR_API int oracle_nb_matrices(void) { return OR_NB_MATRICES; }
OR_API const char *oracle_matrix_name(int which) { return OR_MATRIX_NAMES[which]; }
OR_API void oracle_get_matrix(const oracle *o, int which, float *out) { or_matrix(o->stats, which, out); }

/* solid spectrum of one sample (sorted canonical k-mers, A0C1T2G3 code) -- for kernel-level tests */
OR_API uint64_t oracle_sample_nsolid(const oracle *o, int i) { return o->s[i].nsolid; }
OR_API void oracle_get_sample_solid(const oracle *o, int i, uint64_t *kmers, uint32_t *counts) {
    for (size_t j = 0; j < o->s[i].nsolid; j++) kmers[j] = (uint64_t)o->s[i].kmer[j];      /* low 64 bits: the whole k-mer for k <= 32 */
    memcpy(counts, o->s[i].count, o->s[i].nsolid * 4);
}

/* outputMatrix, ref: src/core/SimkaDistance.cpp:603-649 */
OR_API int oracle_write_matrices(const oracle *o, const char *dir, int gz) {
    size_t n = o->n;
    float *m = (float *)malloc(n * n * sizeof(float));
    int rc = 0;
    for (int w = 0; w < OR_NB_MATRICES; w++) {
        if (w >= 15 && w <= 17 && !o->stats->simple) continue;
        if (w >= 18 && !o->stats->complex_) continue;
        or_matrix(o->stats, w, m);
        if (or_dump_matrix(o, dir, OR_MATRIX_NAMES[w], m, gz) != 0) rc = -1;
    }
    free(m);
    return rc;
}

#ifdef ORACLE_MAIN
/* minimal CLI: simka_oracle -in X -out DIR [-kmer-size K] [-abundance-min M] [-abundance-max M]
 *              [-simple-dist] [-complex-dist] [-nb-partitions P] [-nb-cores T] [-gz] */
int main(int argc, char **argv) {
    const char *in = NULL, *out = "./simka_results";
    int k = 21, simple = 0, complex_ = 0, gz = 0, threads = 1; unsigned P = 1; uint32_t amin = 2, amax = 999999999u;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-in") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "-out") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-kmer-size") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-abundance-min") && i + 1 < argc) amin = (uint32_t)strtoul(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "-abundance-max") && i + 1 < argc) amax = (uint32_t)strtoul(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "-nb-partitions") && i + 1 < argc) P = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-nb-cores") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-simple-dist")) simple = 1;
        else if (!strcmp(argv[i], "-complex-dist")) complex_ = 1;
        else if (!strcmp(argv[i], "-gz")) gz = 1;
    }
    if (!in) { fprintf(stderr, "usage: simka_oracle -in input.txt -out dir ...\n"); return 1; }
    oracle *o = oracle_new();
    if (oracle_load_input(o, in) != 0 || oracle_run(o, k, amin, amax, simple, complex_, P, threads) != 0) {
        fprintf(stderr, "%s\n", oracle_error(o)); return 1;
    }
    if (oracle_write_matrices(o, out, gz) != 0) { fprintf(stderr, "cannot write to %s\n", out); return 1; }
    oracle_free(o);
    return 0;
}
#endif
