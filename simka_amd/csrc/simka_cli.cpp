// simka_cli.cpp -- `simka`: drop-in host driver over the C ABI of libsimka_hip.so.
//
// Same command line, -in grammar and mat_*.csv.gz outputs as the reference driver
// (ref: src/SimkaPotara.cpp:29-53,147-163, src/core/Simka.cpp:25-117, src/SimkaPotara.hpp:259-326).
// Where the reference forks one simkaCount process per sample and one simkaMerge process per
// partition and synchronises through files, this driver calls simka_count_sample() per sample and
// simka_merge() once; on G GPUs the samples are spread over the GPUs for counting, their spectra
// exchanged by partition range, and the GPUs' pair accumulators summed.
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <chrono>
#include <thread>
#include <iostream>
#include <sstream>
#include <string>
#include <memory>
#include <vector>

#include "../../include/simka_hip.h"

namespace {

struct Options {
    std::string in, out = "./simka_results", out_tmp;
    bool keep_tmp = false, data_info = false, simple = false, complex_ = false;
    int kmer_size = 21;
    long long abundance_min = 2, abundance_max = 999999999LL;
    double kmer_shannon = 0;            // parsed, no-op in the reference too (ref: src/core/SimkaAlgorithm.hpp:226-232)
    long long max_reads = -1;           // -1 all, 0 estimate, m>0 first m reads per paired part
    long long min_read_size = 0;
    double min_shannon = 0;
    int nb_cores = 0;
    long long max_memory = 5000;
    int verbose = 1;
    int nb_gpus = 1, first_gpu = 0;     // new: GPUs to spread the samples (count) and the partition ranges (merge) over
    bool same_gpu = false;              // new (tests): all -nb-gpus contexts on GPU -gpu
    bool mapped_arenas = false;         // new: -nb-gpus on distinct devices with lazily mapped arenas (default: plain allocations until a multi-GPU box has run the mapped path)
    int ingest_window = 0;              // new: samples whose text is in flight between the reader threads and the GPU (0 = auto)
    long long ingest_chunk = 0;         // new: bytes of a file handed to the device-side parser at a time (cut at record boundaries); 0 = 3 GiB when the
                                        // loader threads upload the text themselves (one GPU), 1 GiB of pinned memory per piece otherwise
    bool numa_bind = true;              // new: run on the CPUs of the GPU's NUMA node (the loader threads' staging memory is then local to it)
    bool host_upload = false;           // new: the text goes through whole-file pinned buffers and the MAIN thread copies it (the round-3 route; -nb-gpus uses it)
    bool host_parse = false;            // new: parse + pack every input on the host (default: plain-text inputs without read policies are parsed on the GPU)
    bool host_spectra = false;          // new: -nb-gpus keeps the spectra in host memory between count and merge (the round-2 route)
    bool gpu_partition = false;         // new: -gpu-shards partition -- BASELINE north_star's split: every GPU scans all reads and keeps the minimizer partitions p % G == g
    bool gpu_host_sum = false;          // new: -gpu-host-sum -- partition shards: add the heads on the host even where RCCL could
    bool gpu_allreduce = false;         // new: -nb-gpus combines the merges' accumulators with one RCCL all-reduce instead of summing them on the host
    long long solid_capacity = 0;       // new (tests): records of the solid-spectrum arena of every context (0: from the free memory)
    int merge_ranges = 0;               // new: >0 keeps the spectra in host memory and merges in that many partition ranges per GPU
    bool parse_only = false;            // new: stop after reading + packing the inputs (ingest benchmark, no GPU needed)
};

struct Sample {
    std::string id;
    std::vector<std::vector<std::string>> parts;   // ';'-separated paired parts, each a ','-list of files
};

[[noreturn]] void die(const std::string &msg, int code = 1) {
    std::cerr << msg << std::endl;
    exit(code);
}

bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

void mkdir_p(const std::string &p) {
    if (p.empty() || exists(p)) return;
    const size_t slash = p.find_last_of('/');
    if (slash != std::string::npos && slash > 0) mkdir_p(p.substr(0, slash));
    if (mkdir(p.c_str(), 0777) != 0 && errno != EEXIST) die("Error: can't create output directory (" + p + ")");
}

void usage() {
    std::cout <<
        "[Simka options]\n"
        "       -in               (1 arg) :    input file of samples. One sample per line: id1: filename1...\n"
        "       -out              (1 arg) :    output directory for result files (distance matrices)  [default './simka_results']\n"
        "       -out-tmp          (1 arg) :    output directory for temporary files\n"
        "       -keep-tmp         (0 arg) :    keep temporary files\n"
        "       -data-info        (0 arg) :    compute (and display) information before running Simka, such as the number of reads per dataset\n"
        "   [distance options]\n"
        "       -simple-dist      (0 arg) :    compute all simple distances (Chord, Hellinger...)\n"
        "       -complex-dist     (0 arg) :    compute all complex distances (Jensen-Shannon...)\n"
        "   [kmer options]\n"
        "       -kmer-size        (1 arg) :    size of a kmer  [default '21']  (1..127; up to 63 the k-mers are compared word for word, from 64 on by a\n"
        "                                       126-bit fingerprint of the four-word k-mer: two distinct k-mers collide with probability < D^2 / 2^127)\n"
        "       -abundance-min    (1 arg) :    min abundance a kmer need to be considered  [default '2']\n"
        "       -abundance-max    (1 arg) :    max abundance a kmer can have to be considered  [default '999999999']\n"
        "       -kmer-shannon-index (1 arg) :    minimal Shannon index a kmer should have to be kept. Float in [0,2]  [default '0']\n"
        "   [read options]\n"
        "       -max-reads        (1 arg) :    maximum number of reads per sample to process. Can be -1: use all reads. Can be 0: estimate it  [default '-1']\n"
        "       -min-read-size    (1 arg) :    minimal size a read should have to be kept  [default '0']\n"
        "       -min-shannon-index (1 arg) :    minimal Shannon index a read should have to be kept. Float in [0,2]  [default '0']\n"
        "   [core options]\n"
        "       -nb-cores         (1 arg) :    number of cores  [default '0']\n"
        "       -max-memory       (1 arg) :    max memory (MB)  [default '5000']\n"
        "       -max-count        (1 arg) :    accepted for compatibility (no job processes here)\n"
        "       -max-merge        (1 arg) :    accepted for compatibility\n"
        "   [gpu options]\n"
        "       -nb-gpus          (1 arg) :    MI355X devices: samples are counted on GPU i % n, partition ranges merged per GPU, spectra moved between the GPUs  [default '1']\n"
        "       -gpu-shards       (1 arg) :    with -nb-gpus: 'sample' (a GPU counts the samples i % n, the spectra are exchanged by partition range) or 'partition' (every GPU scans all reads and keeps the minimizer partitions p % n; nothing is exchanged but ONE all-reduce of the N x N accumulators: RCCL over xGMI on distinct devices, the host with -gpu-shared / -gpu-host-sum)  [default 'sample']\n"
        "       -gpu-host-sum     (0 arg) :    with -gpu-shards partition: add the GPUs' accumulators on the host instead of the RCCL all-reduce\n"
        "       -host-spectra     (0 arg) :    with -nb-gpus: keep the k-mer spectra in host memory between count and merge\n"
        "       -merge-ranges     (1 arg) :    keep the k-mer spectra in host memory and merge in this many partition ranges per GPU (0: only when GPU memory requires it)  [default '0']\n"
        "       -gpu              (1 arg) :    first device ordinal  [default '0']\n"
        "       -verbose          (1 arg) :    verbosity level  [default '1']\n";
}

Options parse_args(int argc, char **argv) {
    Options o;
    auto need = [&](int &i) -> std::string {
        if (i + 1 >= argc) { std::cout << "ERROR: option " << argv[i] << " needs an argument" << std::endl; usage(); exit(1); }
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-in") o.in = need(i);
        else if (a == "-out") o.out = need(i);
        else if (a == "-out-tmp") o.out_tmp = need(i);
        else if (a == "-keep-tmp") o.keep_tmp = true;
        else if (a == "-data-info") o.data_info = true;
        else if (a == "-simple-dist") o.simple = true;
        else if (a == "-complex-dist") o.complex_ = true;
        else if (a == "-kmer-size") o.kmer_size = atoi(need(i).c_str());
        else if (a == "-abundance-min") o.abundance_min = atoll(need(i).c_str());
        else if (a == "-abundance-max") o.abundance_max = atoll(need(i).c_str());
        else if (a == "-kmer-shannon-index") o.kmer_shannon = atof(need(i).c_str());
        else if (a == "-max-reads") o.max_reads = atoll(need(i).c_str());
        else if (a == "-min-read-size") o.min_read_size = atoll(need(i).c_str());
        else if (a == "-min-shannon-index") o.min_shannon = atof(need(i).c_str());
        else if (a == "-nb-cores") o.nb_cores = atoi(need(i).c_str());
        else if (a == "-max-memory") o.max_memory = atoll(need(i).c_str());
        else if (a == "-verbose") o.verbose = atoi(need(i).c_str());
        else if (a == "-nb-gpus") o.nb_gpus = atoi(need(i).c_str());
        else if (a == "-gpu") o.first_gpu = atoi(need(i).c_str());
        else if (a == "-gpu-shared") o.same_gpu = true;
        else if (a == "-gpu-mapped-arenas") o.mapped_arenas = true;
        else if (a == "-host-spectra") o.host_spectra = true;
        else if (a == "-gpu-allreduce") o.gpu_allreduce = true;
        else if (a == "-gpu-host-sum") o.gpu_host_sum = true;
        else if (a == "-gpu-shards") { const std::string v = need(i); if (v == "partition") o.gpu_partition = true; else if (v == "sample") o.gpu_partition = false; else die("ERROR: -gpu-shards takes sample or partition"); }
        else if (a == "-host-parse") o.host_parse = true;
        else if (a == "-ingest-window") o.ingest_window = atoi(need(i).c_str());
        else if (a == "-ingest-host-upload") o.host_upload = true;
        else if (a == "-no-numa-bind") o.numa_bind = false;
        else if (a == "-ingest-chunk") o.ingest_chunk = std::min<long long>(std::max<long long>(64, atoll(need(i).c_str())), 0xf0000000ll);
        else if (a == "-merge-ranges") o.merge_ranges = atoi(need(i).c_str());
        else if (a == "-solid-capacity") o.solid_capacity = atoll(need(i).c_str());
        else if (a == "-parse-only") o.parse_only = true;
        else if (a == "-max-count" || a == "-max-merge" || a == "-count-cmd" || a == "-merge-cmd" || a == "-count-file" ||
                 a == "-merge-file" || a == "-minimizer-size" || a == "-solidity-kind" || a == "-max-disk" ||
                 a == "-minimizer-type" || a == "-repartition-type" || a == "-storage-type" || a == "-histo-max")
            (void)need(i);   // cluster / DSK knobs: accepted, meaningless without job processes or disk partitions
        else if (a == "-help" || a == "-h" || a == "--help") { usage(); exit(0); }
        else if (a == "-version") { std::cout << "simka (MI355X) 0.1.0" << std::endl; exit(0); }
        else { std::cout << "ERROR: Unknown parameter '" << a << "'" << std::endl; usage(); exit(1); }
    }
    if (o.in.empty()) { std::cout << "ERROR: Option '-in' is mandatory" << std::endl; usage(); exit(1); }
    if (o.out_tmp.empty()) { std::cout << "ERROR: Option '-out-tmp' is mandatory" << std::endl; usage(); exit(1); }
    return o;
}

// ---- -in grammar (ref: src/core/SimkaAlgorithm.cpp:245-351) ---------------------------------
std::vector<Sample> parse_input(const std::string &path) {
    std::ifstream f(path.c_str());
    if (!f) die("ERROR: Input filename does not exist");
    char *rp = realpath(path.c_str(), nullptr);
    std::string dir = rp ? rp : path;
    free(rp);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    std::vector<Sample> samples;
    std::string line;
    while (std::getline(f, line)) {
        line.erase(std::remove(line.begin(), line.end(), ' '), line.end());
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());
        if (line.empty()) continue;
        std::vector<std::string> fields;
        { std::stringstream ss(line); std::string p; while (std::getline(ss, p, ':')) fields.push_back(p); }
        if (fields.size() < 2) { std::cout << "Syntax error in input file" << std::endl; exit(1); }
        Sample s; s.id = fields[0];
        std::stringstream ps(fields[1]); std::string part;
        while (std::getline(ps, part, ';')) {
            std::vector<std::string> files;
            std::stringstream fs(part); std::string fn;
            while (std::getline(fs, fn, ',')) { if (fn.empty()) continue; files.push_back(fn[0] == '/' ? fn : dir + "/" + fn); }
            s.parts.push_back(files);
        }
        samples.push_back(s);
    }
    return samples;
}

// ---- sequence files: FASTA (multi-line) / FASTQ (4-line), plain or gz -------------------------
class SeqReader {
public:
    explicit SeqReader(const std::string &path) : g_(gzopen(path.c_str(), "rb")), path_(path) {
        if (g_) gzbuffer(g_, 1 << 20);
    }
    ~SeqReader() { if (g_) gzclose(g_); }
    bool ok() const { return g_ != nullptr; }
    // next sequence into `seq`; false at end of file
    bool next(std::string &seq) {
        seq.clear();
        std::string line;
        if (!have_pending_ && !getline(pending_)) return false;
        have_pending_ = false;
        while (pending_.empty()) if (!getline(pending_)) return false;
        if (pending_[0] == '>') {
            while (getline(line)) {
                if (!line.empty() && line[0] == '>') { pending_ = line; have_pending_ = true; break; }
                seq += line;
            }
            return true;
        }
        if (pending_[0] == '@') {
            size_t qual = 0;
            while (getline(line)) { if (!line.empty() && line[0] == '+') break; seq += line; }
            while (qual < seq.size() && getline(line)) qual += line.size();
            return true;
        }
        die("ERROR: unrecognised sequence file: " + path_);
    }

private:
    bool getline(std::string &out) {
        out.clear();
        char buf[1 << 16];
        bool any = false;
        while (gzgets(g_, buf, sizeof buf)) {
            any = true;
            size_t l = strlen(buf);
            const bool full = l > 0 && buf[l - 1] == '\n';
            while (l > 0 && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) l--;
            out.append(buf, l);
            if (full) break;
        }
        return any;
    }
    gzFile g_;
    std::string path_, pending_;
    bool have_pending_ = false;
};

// read filters (ref: src/core/SimkaCommons.hpp:317-436)
float shannon_index(const std::string &s) {
    static int tab[128];
    static bool init = false;
    if (!init) { memset(tab, 0, sizeof tab); tab['C'] = 1; tab['T'] = 2; tab['G'] = 3; tab['N'] = 4; init = true; }
    std::vector<float> freq(5, 0.f);
    for (unsigned char c : s) freq[c < 128 ? tab[c] : 0] += 1.0f;
    float index = 0;
    for (float &f : freq) { f /= (float)s.size(); if (f != 0) index += f * log(f) / log(2); }
    return fabsf(index);
}
bool read_passes(const std::string &s, const Options &o) {
    if (o.min_read_size && (long long)s.size() < o.min_read_size) return false;
    if (o.min_shannon != 0 && !(shannon_index(s) >= o.min_shannon)) return false;
    return true;
}

// number of filter-passing reads of a sample divided by its paired parts (computeMaxReads, ref: src/core/SimkaAlgorithm.cpp:377-445)
uint64_t count_reads(const Sample &s) {
    uint64_t n = 0; std::string seq;
    for (auto &part : s.parts) for (auto &fn : part) { SeqReader r(fn); if (!r.ok()) return 0; while (r.next(seq)) n++; }
    return s.parts.empty() ? 0 : n / s.parts.size();
}

// Page-locked host buffers for what goes to the GPU (simka_host_alloc: the H2D copy is then one DMA into the staging buffer of
// the sample's lane; plain malloc when pinning fails).  Blocks are recycled through a pool: pinning costs about as much as
// first-touching the pages, and the loader keeps only a window of samples alive.  The pool holds at most kMaxIdleBytes of idle blocks,
// and a request is never served by a block more than four times its size (a deep sample's buffer is not pinned forever behind small ones).
class PinnedPool {
public:
    static PinnedPool &get() { static PinnedPool p; return p; }
    void *take(size_t bytes, size_t &cap, bool &pinned) {
        {
            std::lock_guard<std::mutex> g(m_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++) if (free_[i].cap >= bytes && free_[i].cap / 4 <= bytes + (1u << 20) && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best != free_.size()) { Block b = free_[best]; free_.erase(free_.begin() + (long)best); idle_ -= b.cap; cap = b.cap; pinned = b.pinned; return b.p; }
        }
        void *p = nullptr;
        cap = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
        pinned = simka_host_alloc(cap, &p) == SIMKA_OK && p;
        if (!pinned) p = malloc(cap);
        return p;
    }
    void give(void *p, size_t cap, bool pinned) {
        if (!p) return;
        std::lock_guard<std::mutex> g(m_);
        if (free_.size() >= 256 || idle_ + cap > max_idle_) { if (pinned) simka_host_free(p); else free(p); return; }
        free_.push_back(Block{p, cap, pinned});
        idle_ += cap;
    }
private:
    struct Block { void *p; size_t cap; bool pinned; };
    // idle blocks kept for reuse: pinning a fresh buffer costs as much as filling it, so the pool should hold what the reader window
    // cycles through (window x file size); set_max_idle() is called by the driver once it knows both
    size_t max_idle_ = (size_t)8 << 30;
public:
    void set_max_idle(size_t b) { std::lock_guard<std::mutex> g(m_); max_idle_ = std::max(max_idle_, b); }
private:
    std::vector<Block> free_;
    size_t idle_ = 0;
    std::mutex m_;
};

template <class T>
class PinnedBuf {
public:
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    PinnedBuf(PinnedBuf &&o) noexcept { steal(o); }
    PinnedBuf &operator=(PinnedBuf &&o) noexcept { if (this != &o) { release(); steal(o); } return *this; }
    ~PinnedBuf() { release(); }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    // grows geometrically and keeps the content; the new words are NOT cleared (simka_pack_read clears a word when it starts it)
    void resize(size_t n) {
        if (n * sizeof(T) > cap_) {
            size_t cap = 0; bool pinned = false;
            T *q = (T *)PinnedPool::get().take(std::max(n * sizeof(T), cap_ * 2), cap, pinned);
            if (!q) throw std::bad_alloc();
            if (n_) memcpy(q, p_, n_ * sizeof(T));
            PinnedPool::get().give(p_, cap_, pinned_);
            p_ = q; cap_ = cap; pinned_ = pinned;
        }
        n_ = n;
    }
    void reserve(size_t n) { const size_t keep = n_; if (n > n_) { resize(n); n_ = keep; } }
private:
    void release() { PinnedPool::get().give(p_, cap_, pinned_); p_ = nullptr; n_ = 0; cap_ = 0; }
    void steal(PinnedBuf &o) { p_ = o.p_; n_ = o.n_; cap_ = o.cap_; pinned_ = o.pinned_; o.p_ = nullptr; o.n_ = 0; o.cap_ = 0; }
    T *p_ = nullptr; size_t n_ = 0, cap_ = 0; bool pinned_ = false;
};

typedef PinnedBuf<uint64_t> PinnedWords;

// Device buffers for the text of the files (one GPU): recycled like the pinned host buffers -- a hipMalloc / hipFree per piece would
// synchronise the device under the kernels of the samples before.
class DevPool {
public:
    static DevPool &get() { static DevPool p; return p; }
    void *take(int device, size_t bytes, size_t &cap) {
        {
            std::lock_guard<std::mutex> g(m_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].device == device && free_[i].cap >= bytes && free_[i].cap / 4 <= bytes + (1u << 20) && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best != free_.size()) { Block b = free_[best]; free_.erase(free_.begin() + (long)best); idle_ -= b.cap; cap = b.cap; return b.p; }
        }
        void *p = nullptr;
        cap = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
        if (simka_device_alloc(device, cap, &p) != SIMKA_OK) { cap = 0; return nullptr; }
        return p;
    }
    void give(int device, void *p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(m_);
            if (free_.size() < 256 && idle_ + cap <= max_idle_) { free_.push_back(Block{p, cap, device}); idle_ += cap; return; }
        }
        simka_device_free(device, p);
    }
    void set_max_idle(size_t b) { std::lock_guard<std::mutex> g(m_); max_idle_ = std::max(max_idle_, b); }
private:
    struct Block { void *p; size_t cap; int device; };
    std::vector<Block> free_;
    size_t idle_ = 0, max_idle_ = (size_t)2 << 30;
    std::mutex m_;
};
struct DevText {
    void *p = nullptr; size_t cap = 0, n = 0; int device = -1;
    DevText() {}
    DevText(const DevText &) = delete;
    DevText &operator=(const DevText &) = delete;
    DevText(DevText &&o) noexcept { p = o.p; cap = o.cap; n = o.n; device = o.device; o.p = nullptr; o.cap = 0; }
    DevText &operator=(DevText &&o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; n = o.n; device = o.device; o.p = nullptr; o.cap = 0; } return *this; }
    ~DevText() { release(); }
    void release() { if (p) DevPool::get().give(device, p, cap); p = nullptr; cap = 0; }
};

struct Packed {
    PinnedWords words, offsets;
    uint64_t nb_bases = 0, nb_frag = 0, nb_reads = 0;
    // device-side ingest (simka_ingest_*): the files' bytes as they are, parsed on the GPU
    bool raw = false;
    std::vector<PinnedBuf<char>> texts;
    std::vector<DevText> dtexts;          // ... or already in DEVICE memory, uploaded by the loader thread (load_sample_raw_device)
    std::vector<int> formats;             // 0 FASTA, 1 FASTQ
    std::vector<uint32_t> file_of;        // which file a piece belongs to (a FILE that delivers no read ends the sample)
};

// Which reads of a sample are counted (-max-reads m, paired parts, read filters).  The reference walks the sample's files with
// SimkaInputIterator (ref: src/core/SimkaCommons.hpp:159-314); what that iterator DOES, stated as loops (the oracle keeps the
// literal member-for-member restatement, tests/ compare the two):
//   * the files of all parts form one flat list; part d owns files [d F, (d + 1) F) with F = #files / #parts (:174) -- every part
//     is ASSUMED to list as many files; with unequal parts the reference reads the wrong files, and so does this drop-in (the
//     driver warns); files beyond #parts * F are never read; F = 0 (more parts than files): no reads at all;
//   * only reads that pass the filters (-min-read-size, -min-shannon-index) exist for what follows;
//   * inside a part a counter runs across the files; the FIRST read a file delivers does not advance it (:224-236), every other
//     read does; the read that brings the counter to m is fetched and dropped, and the part ends (:277-287): a part delivers
//     its files' first reads plus at most m - 1 others;
//   * a file that delivers no read at all (empty, or every read filtered) ends the SAMPLE: the iterator reports "done" when the
//     bank it just opened is done (:192-208);
//   * m = 0: no limit.
template <class Consume>
static bool for_counted_reads(const Sample &s, const Options &o, uint64_t max_reads, Consume &&consume) {
    std::vector<const std::string *> files;
    for (auto &part : s.parts) for (auto &fn : part) files.push_back(&fn);
    const size_t nparts = std::max<size_t>(1, s.parts.size());
    const size_t per_part = files.size() / nparts;
    if (files.empty() || per_part == 0) return false;
    std::string seq;
    for (size_t d = 0; d < nparts; d++) {
        uint64_t counted = 0;
        bool part_full = false;
        for (size_t f = 0; f < per_part && !part_full; f++) {
            SeqReader r(*files[d * per_part + f]);
            if (!r.ok()) return false;
            bool delivered = false;
            while (r.next(seq)) {
                if (!read_passes(seq, o)) continue;
                if (!delivered) { delivered = true; if (!consume(seq)) return false; continue; }      // a file's first read is free
                if (max_reads && ++counted >= max_reads) { part_full = true; break; }                 // read m + 1: fetched, never seen
                if (!consume(seq)) return false;
            }
            if (!delivered) return true;          // nothing came out of this file: the sample ends here
        }
    }
    return true;
}

// The files of a sample as raw text for the device-side parser -- only where it parses exactly what the host path would deliver: every
// read of every listed file (no -max-reads limit, no read filter), plain text (not gzip), below 4 GB per file.  false: use load_sample.
bool load_sample_raw(const Sample &s, const Options &o, uint64_t max_reads, Packed &out) {
    if (max_reads || o.min_read_size || o.min_shannon != 0) return false;
    std::vector<const std::string *> files;
    for (auto &part : s.parts) for (auto &fn : part) files.push_back(&fn);
    const size_t nparts = std::max<size_t>(1, s.parts.size());
    const size_t per_part = files.size() / nparts;
    if (files.empty() || per_part == 0) return false;
    out = Packed();
    out.raw = true;
    for (size_t f = 0; f < nparts * per_part; f++) {       // (files beyond #parts * files-per-part are never read: for_counted_reads)
        FILE *fp = fopen(files[f]->c_str(), "rb");
        if (!fp) return false;
        struct stat st;
        if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode)) { fclose(fp); return false; }
        // The file goes over in pieces of at most -ingest-chunk bytes (1 GiB), each ending at a record boundary: the parser takes one piece
        // as one text (32-bit offsets), and a pinned buffer of the whole file would be as large as the file.  FASTA: a piece ends before
        // the last line that starts with '>'; FASTQ (4-line records, which the device parser verifies): before the last line whose
        // number in the piece is a multiple of 4.  A record longer than a piece: the host parser takes the sample.
        const size_t fsize = (size_t)st.st_size, chunk = (size_t)(o.ingest_chunk > 0 ? o.ingest_chunk : (1ll << 30));
        size_t done = 0, carry = 0;
        int fmt = -1;
        PinnedBuf<char> cur;
        bool first = true;
        while (first || done < fsize) {
            if (carry + std::max<size_t>(64, chunk / 16) > chunk) { fclose(fp); return false; }       // a record about as long as a piece: the host parser takes the sample (reading on in ever smaller steps would copy the carry every time)
            const size_t want = std::min(chunk - carry, fsize - done);
            PinnedBuf<char> buf;
            buf.resize(carry + want + 1);
            if (carry) memcpy(buf.data(), cur.data() + (cur.size() - carry), carry);
            size_t got = 0;
            while (got < want) { const size_t r = fread(buf.data() + carry + got, 1, want - got, fp); if (r == 0) break; got += r; }
            if (got != want) { fclose(fp); return false; }
            done += got;
            size_t len = carry + got;
            if (first) {
                first = false;
                if (len >= 2 && (unsigned char)buf[0] == 0x1f && (unsigned char)buf[1] == 0x8b) { fclose(fp); return false; }       // gzip: the host inflates
                size_t p = 0;
                while (p < len && (buf[p] == '\n' || buf[p] == '\r')) p++;
                if (p == len && done == fsize) fmt = 0;     // an empty file: delivers no read (which ends the sample)
                else if (p == 0 && buf[0] == '>') fmt = 0;
                else if (p == 0 && buf[0] == '@') fmt = 1;
                if (fmt < 0) { fclose(fp); return false; }
            }
            size_t cut = len;                               // the last piece takes everything
            if (done < fsize) {
                cut = 0;
                if (fmt == 0) { for (size_t p = len; p-- > 1; ) if (buf[p] == '>' && buf[p - 1] == '\n') { cut = p; break; } }
                else {
                    size_t nl = 0;
                    for (const char *q = buf.data(), *e = q + len; (q = (const char *)memchr(q, '\n', (size_t)(e - q))) != nullptr; q++) nl++;
                    size_t skip = nl % 4 + 1;               // the (nl mod 4 + 1)-th newline from the end closes the last whole record
                    if (nl >= 4) for (size_t p = len; p-- > 0; ) if (buf[p] == '\n' && --skip == 0) { cut = p + 1; break; }
                }
                if (cut == 0) { fclose(fp); return false; }      // no whole record in the piece
            }
            carry = len - cut;
            cur = std::move(buf);
            PinnedBuf<char> piece;
            if (carry == 0) { cur.resize(cut); piece = std::move(cur); cur = PinnedBuf<char>(); }
            else { piece.resize(cut); memcpy(piece.data(), cur.data(), cut); cur.resize(len); }
            out.texts.push_back(std::move(piece));
            out.formats.push_back(fmt);
            out.file_of.push_back((uint32_t)f);
        }
        fclose(fp);
    }
    return true;
}

// The same for ONE GPU, without the whole-file pinned buffers: the loader thread reads the file 16 MiB at a time into a small pinned
// staging buffer of its own and uploads every block straight into a device buffer (simka_device_upload: a stream per thread), so
// the copies of all loader threads run on the DMA engines under the kernels of the samples before, the main thread only launches
// kernels (simka_ingest_text_device), and the pinned memory of the process is a few staging buffers -- pinning runs at 4 - 6 GB/s for
// the whole process and a piece in flight had to be pinned first (scripts/ubench/read_rate.hip).  Pieces of at most -ingest-chunk
// bytes (3 GiB) end at a record boundary, looked for in the LAST staging block of the piece (a longer record: the host parser).
// -verbose 2: where the loader threads of the upload route spend their time, summed over the threads (microseconds)
static std::atomic<long long> g_ld_stage_us(0), g_ld_dev_us(0), g_ld_read_us(0), g_ld_up_us(0);
static inline long long ld_now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool load_sample_raw_device(const Sample &s, const Options &o, uint64_t max_reads, int device, Packed &out) {
    if (max_reads || o.min_read_size || o.min_shannon != 0) return false;
    std::vector<const std::string *> files;
    for (auto &part : s.parts) for (auto &fn : part) files.push_back(&fn);
    const size_t nparts = std::max<size_t>(1, s.parts.size());
    const size_t per_part = files.size() / nparts;
    if (files.empty() || per_part == 0) return false;
    out = Packed();
    out.raw = true;
    const size_t chunk = (size_t)(o.ingest_chunk > 0 ? o.ingest_chunk : (3ll << 30));
    const size_t S = std::min<size_t>((size_t)16 << 20, chunk);      // (small blocks: whatever else crosses the bus waits for at most one of them)
    static thread_local PinnedBuf<char> stage;
    { const long long t0 = ld_now_us(); if (stage.size() < S + 16) stage.resize(S + 16); g_ld_stage_us += ld_now_us() - t0; }
    auto count_nl = [](const char *q, size_t n) { size_t c = 0; for (const char *e = q + n; (q = (const char *)memchr(q, '\n', (size_t)(e - q))) != nullptr; q++) c++; return c; };
    for (size_t f = 0; f < nparts * per_part; f++) {
        const int fd = open(files[f]->c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return false; }
        const size_t fsize = (size_t)st.st_size;
        // gzip: this thread inflates into its staging block and uploads the TEXT -- the GPU parses it like any other (the host parser
        // behind load_sample saturates at ~16 threads and 4.8 GB/s of text; zlib scales with the threads)
        unsigned char magic[2] = { 0, 0 };
        const bool gz = fsize >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        gzFile gzf = nullptr;
        if (gz) { gzf = gzopen(files[f]->c_str(), "rb"); if (!gzf) { close(fd); return false; } gzbuffer(gzf, 1 << 20); }
        auto done_with = [&]() { if (gzf) gzclose(gzf); close(fd); };
        size_t done = 0;                        // plain: bytes of the file read so far
        bool eof = fsize == 0;
        auto read_block = [&](char *dst, size_t want) -> long long {      // < 0: error
            size_t got = 0;
            if (gz) {
                while (got < want) { const int r = gzread(gzf, dst + got, (unsigned)std::min<size_t>(want - got, (size_t)1 << 30)); if (r < 0) return -1; if (r == 0) { eof = true; break; } got += (size_t)r; }
            } else {
                const size_t lim = std::min(want, fsize - done);
                while (got < lim) { const ssize_t r = pread(fd, dst + got, lim - got, (off_t)(done + got)); if (r <= 0) return -1; got += (size_t)r; }
                done += got;
                if (done == fsize) eof = true;
            }
            return (long long)got;
        };
        int fmt = -1;
        bool first = true;
        std::vector<char> carry;                 // what the piece before left behind its last whole record
        for (size_t piece = 0;; piece++) {
            if (carry.size() + std::max<size_t>(64, chunk / 16) > chunk) { done_with(); return false; }       // a record about as long as a piece: the host parser
            // capacity of the piece: the rest of a plain file; for gzip an estimate of the text (6 x the compressed bytes) -- what does
            // not fit goes to a further piece
            const size_t cap = gz ? std::min(chunk, std::max<size_t>((size_t)64 << 20, 6 * fsize + ((size_t)16 << 20))) : std::min(chunk, carry.size() + (fsize - done));
            const bool known_last = !gz && carry.size() + (fsize - done) <= chunk;
            DevText dt;
            dt.device = device;
            { const long long t0 = ld_now_us(); dt.p = DevPool::get().take(device, cap + 64, dt.cap); g_ld_dev_us += ld_now_us() - t0; }
            if (!dt.p) { done_with(); return false; }
            size_t at = 0, nl = 0, tail_off = 0, tail_len = 0;
            if (!carry.empty()) {
                memcpy(stage.data(), carry.data(), carry.size());
                if (simka_device_upload(device, dt.p, stage.data(), carry.size()) != SIMKA_OK) { done_with(); return false; }
                if (fmt == 1 && !known_last) nl += count_nl(stage.data(), carry.size());
                at = carry.size(); tail_off = 0; tail_len = carry.size();
            }
            while (at < cap && !eof) {
                const long long tr0 = ld_now_us();
                const long long got_ = read_block(stage.data(), std::min(S, cap - at));
                g_ld_read_us += ld_now_us() - tr0;
                if (got_ < 0) { done_with(); return false; }
                const size_t got = (size_t)got_;
                if (got == 0) break;
                if (first) {
                    first = false;
                    size_t p = 0;
                    while (p < got && (stage[p] == '\n' || stage[p] == '\r')) p++;
                    if (p == got && eof) fmt = 0;                 // only line ends: delivers no read (which ends the sample)
                    else if (p == 0 && stage[0] == '>') fmt = 0;
                    else if (p == 0 && stage[0] == '@') fmt = 1;
                    if (fmt < 0) { done_with(); return false; }
                }
                if (fmt == 1 && !known_last) nl += count_nl(stage.data(), got);
                { const long long t0 = ld_now_us(); const int urc = simka_device_upload(device, (char *)dt.p + at, stage.data(), got); g_ld_up_us += ld_now_us() - t0; if (urc != SIMKA_OK) { done_with(); return false; } }
                tail_off = at; tail_len = got; at += got;
            }
            if (first) { first = false; fmt = 0; }       // an empty file
            const bool last_piece = eof;
            const size_t len = at;
            if (len == 0 && piece > 0) break;            // (gzip: the text ended exactly where the piece before was full)
            size_t cut = len;
            carry.clear();
            if (!last_piece) {
                cut = 0;
                if (fmt == 0) { for (size_t p = tail_len; p-- > 1; ) if (stage[p] == '>' && stage[p - 1] == '\n') { cut = tail_off + p; break; } }
                else {
                    size_t skip = nl % 4 + 1;           // the (nl mod 4 + 1)-th newline from the end closes the last whole record
                    if (nl >= 4) for (size_t p = tail_len; p-- > 0; ) if (stage[p] == '\n' && --skip == 0) { cut = tail_off + p + 1; break; }
                }
                if (cut == 0) { done_with(); return false; }      // no record boundary in the last block of the piece
                carry.assign(stage.data() + (cut - tail_off), stage.data() + tail_len);
            }
            dt.n = cut;
            out.dtexts.push_back(std::move(dt));
            out.formats.push_back(fmt);
            out.file_of.push_back((uint32_t)f);
            if (last_piece) break;
        }
        done_with();
    }
    return true;
}

bool load_sample(const Sample &s, const Options &o, uint64_t max_reads, Packed &out) {
    out = Packed();
    if (max_reads == 0) {   // size the buffers once from the files (2-bit bases <= file bytes, x5 for gz): growing would copy pinned memory around
        uint64_t bytes = 0;
        for (auto &part : s.parts) for (auto &fn : part) { struct stat st; if (stat(fn.c_str(), &st) == 0) bytes += (uint64_t)st.st_size * (fn.size() > 3 && fn.substr(fn.size() - 3) == ".gz" ? 5 : 1); }
        out.words.reserve((size_t)(bytes / 32 + 16));
        out.offsets.reserve((size_t)(bytes / 64 + 16));
    }
    const bool ok = for_counted_reads(s, o, max_reads, [&](const std::string &seq) {
        out.nb_reads++;
        const uint64_t need_words = (out.nb_bases + seq.size()) / 32 + 3;
        if (out.words.size() < need_words) out.words.resize(need_words * 2);
        if (out.offsets.size() < out.nb_frag + seq.size() / 2 + 4) out.offsets.resize((out.nb_frag + seq.size() / 2 + 4) * 2);
        const int64_t nf = simka_pack_read(seq.data(), seq.size(), out.words.data(), &out.nb_bases, out.offsets.data() + out.nb_frag);
        if (nf < 0) return false;
        out.nb_frag += (uint64_t)nf;
        return true;
    });
    if (!ok) return false;
    if (out.offsets.size() < out.nb_frag + 1) out.offsets.resize(out.nb_frag + 1);
    out.offsets[out.nb_frag] = out.nb_bases;
    if (out.words.size() < out.nb_bases / 32 + 3) out.words.resize(out.nb_bases / 32 + 3);
    return true;
}

// Host ingest: worker threads parse + 2-bit pack upcoming samples (one sample per thread, gz
// decompression included) while the main thread hands finished samples to the GPU in input order.  At most `window`
// samples are in flight, which bounds host memory.
class SampleLoader {
public:
    // skip[i] != 0: sample i is not read at all (-keep-tmp found its spectrum); get() returns immediately with an empty slot
    // raw: hand over the files' text where the device-side parser can take it (load_sample_raw), else parse + pack here
    SampleLoader(const std::vector<Sample> &samples, const Options &o, uint64_t max_reads, unsigned threads, unsigned window,
                 const std::vector<char> &skip = std::vector<char>(), bool raw = false, int raw_device = -1)
        : samples_(samples), o_(o), max_reads_(max_reads), window_(std::max(1u, window)), slots_(samples.size()), state_(samples.size(), 0), skip_(skip), raw_(raw), raw_device_(raw_device) {
        // raw text: a file is only READ here (page cache -> pinned memory, ~3 GB/s per thread) -- eight samples in flight keep the GPU
        // fed, and their pinned buffers are recycled (pinning a buffer costs as much as filling it: 66 fresh 150-MB buffers cost seconds)
        // How many samples are in flight.  Measured on the MI355X box (scripts/ubench/read_rate.hip): threads read the page cache into
        // WARM pinned buffers at 85 GB/s (8 threads) -- but pinning a fresh buffer runs at 4 - 6 GB/s for the whole process, and every
        // byte in flight has to be pinned once (and unpinned when the process ends).  So the window is sized in BYTES: about 6 GB of text
        // in flight, between 3 and 8 samples (C3 at full depth, 1.54-GB files: 4; a tenth of that depth: 8); -ingest-window overrides.
        // (.gz inputs: with one GPU the loader threads inflate them and upload the text -- zlib at ~0.3 GB/s of text per thread, so the
        //  window stays wide, bounded by ~32 GB of device buffers; without that route they are inflated AND parsed here, one sample per
        //  thread: the window stays as wide as the thread pool)
        size_t ngz = 0, nfiles_ = 0;
        uint64_t bytes = 0, est_text = 0;
        for (auto &sm : samples) for (auto &part : sm.parts) for (auto &fn : part) {
            nfiles_++;
            struct stat st_;
            const uint64_t sz = stat(fn.c_str(), &st_) == 0 ? (uint64_t)st_.st_size : 0;
            bytes += sz;
            if (fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0) { ngz++; est_text += 6 * sz + ((uint64_t)16 << 20); } else est_text += sz;
        }
        const bool mostly_gz = ngz * 2 > nfiles_;
        if (raw_ && !mostly_gz) {
            const uint64_t per = std::max<uint64_t>(1, bytes / std::max<size_t>(1, samples.size()));
            const size_t by_bytes = (size_t)std::min<uint64_t>(8, std::max<uint64_t>(3, ((uint64_t)6 << 30) / per));
            window_ = std::min<size_t>(window_, o.ingest_window > 0 ? (size_t)o.ingest_window : by_bytes);
            // the buffers of the window (+ 2 being handed over) stay in the pool between samples
            if (raw_device_ >= 0) DevPool::get().set_max_idle((size_t)std::min<uint64_t>((uint64_t)64 << 30, per * (window_ + 2) + ((uint64_t)1 << 30)));
            else PinnedPool::get().set_max_idle((size_t)std::min<uint64_t>((uint64_t)64 << 30, per * (window_ + 2) + ((uint64_t)1 << 30)));
        } else if (raw_ && raw_device_ >= 0) {
            const uint64_t per = std::max<uint64_t>(1, est_text / std::max<size_t>(1, samples.size()));
            const size_t by_bytes = (size_t)std::max<uint64_t>(4, ((uint64_t)32 << 30) / per);
            window_ = std::min<size_t>(window_, o.ingest_window > 0 ? (size_t)o.ingest_window : by_bytes);
            DevPool::get().set_max_idle((size_t)std::min<uint64_t>((uint64_t)64 << 30, per * (window_ + 2) + ((uint64_t)1 << 30)));
            // (measured on the 128-CPU node of the MI355X box, 100 x 57 MB of .fastq.gz: inflating takes 0.5 s per sample alone, 1.0 s with
            //  32 threads, 2.4 s with 100 -- 4.3 s against 5.4 s end to end; -nb-cores overrides)
            if (o.nb_cores <= 0) threads = std::min(threads, 32u);
        }
        threads = std::max(1u, std::min<unsigned>(threads, (unsigned)std::min<size_t>(samples.size(), window_)));
        for (unsigned t = 0; t < threads; t++) workers_.emplace_back([this] { run(); });
    }
    ~SampleLoader() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    // blocks until sample i is packed; false if it could not be read
    bool get(size_t i, Packed *&out) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return state_[i] != 0; });
        out = &slots_[i];
        return state_[i] > 0;
    }
    void release(size_t i) {
        { std::lock_guard<std::mutex> g(m_); slots_[i] = Packed(); consumed_ = i + 1; }
        cv_.notify_all();
    }

private:
    void run() {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || next_ >= samples_.size() || next_ < consumed_ + window_; });
                if (stop_ || next_ >= samples_.size()) return;
                i = next_++;
            }
            Packed pk;
            const bool ok = (i < skip_.size() && skip_[i]) ? true : ((raw_ && (raw_device_ >= 0 ? load_sample_raw_device(samples_[i], o_, max_reads_, raw_device_, pk) : load_sample_raw(samples_[i], o_, max_reads_, pk))) || load_sample(samples_[i], o_, max_reads_, pk));
            { std::lock_guard<std::mutex> g(m_); slots_[i] = std::move(pk); state_[i] = ok ? 1 : -1; }
            cv_.notify_all();
        }
    }
    const std::vector<Sample> &samples_;
    const Options &o_;
    uint64_t max_reads_;
    size_t window_;
    std::vector<Packed> slots_;
    std::vector<int> state_;
    std::vector<char> skip_;
    bool raw_ = false;
    int raw_device_ = -1;               // >= 0: the loader threads upload the text to this device themselves
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    size_t next_ = 0, consumed_ = 0;
    bool stop_ = false;
};

// ---- -keep-tmp: <tmp>/solid/<ID>.g<g>of<G>.spec, one per (sample, GPU shard) -----------------------------------------------
// The reference keeps solid/part_<p>/__p__<i>.gz + count_synchro/<ID>.ok and skips the samples whose .ok exists
// (ref: src/SimkaPotara.hpp:837-842).  Here: the sample's solid spectrum as simka_export_sample() returns it, behind a
// header that pins everything the spectrum depends on (k, abundance filter, read policies, the input files, the shard).
struct SpecHeader {
    char magic[8];                       // "SIMKSPC3" (3: segments ordered by key prefix)
    uint64_t abi, kmer_size, abundance_min, abundance_max, shard_index, shard_count, nb_partitions, nb_records, signature;
    uint64_t key_words;                  // 64-bit words per key: 1, or 2 for -kmer-size >= 32 (high words, then low words)
    simka_sample_totals totals;
};

uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

// everything besides (k, abundance filter, shard) that decides a sample's spectrum: read policies and the files themselves
uint64_t sample_signature(const Sample &s, const Options &o, uint64_t max_reads) {
    uint64_t h = 0xcbf29ce484222325ULL;
    const uint64_t pol[3] = { max_reads, (uint64_t)o.min_read_size, 0 };
    h = fnv1a(h, pol, 16);
    h = fnv1a(h, &o.min_shannon, sizeof o.min_shannon);
    for (auto &p : s.parts) {
        h = fnv1a(h, ";", 1);
        for (auto &fn : p) {
            struct stat st;
            uint64_t meta[2] = { 0, 0 };
            if (stat(fn.c_str(), &st) == 0) { meta[0] = (uint64_t)st.st_size; meta[1] = (uint64_t)st.st_mtime; }
            h = fnv1a(h, fn.data(), fn.size());
            h = fnv1a(h, meta, sizeof meta);
        }
    }
    return h;
}

std::string spec_path(const std::string &tmp, const Sample &s, uint32_t g, uint32_t G) {
    std::ostringstream os;
    os << tmp << "/solid/" << s.id << ".g" << g << "of" << G << ".spec";
    return os.str();
}

bool read_spec_header(const std::string &path, SpecHeader &h) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "SIMKSPC3", 8) == 0;
    fclose(f);
    return ok;
}

bool spec_matches(const SpecHeader &h, const Options &o, uint32_t g, uint32_t G, uint64_t sig) {
    return h.abi == (uint64_t)simka_abi_version() && h.kmer_size == (uint64_t)o.kmer_size && h.abundance_min == (uint64_t)o.abundance_min &&
           h.abundance_max == (uint64_t)o.abundance_max && h.shard_index == g && h.shard_count == G && h.signature == sig &&
           h.nb_partitions && !(h.nb_partitions & (h.nb_partitions - 1));
}

// (keys / counts are page-locked: they travel host <-> GPU whole with -nb-gpus, -merge-ranges and -keep-tmp)
struct Spectrum { SpecHeader h; std::vector<uint32_t> part_counts; PinnedBuf<uint32_t> counts; PinnedBuf<uint64_t> keys; };

bool read_spec(const std::string &path, Spectrum &sp) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = fread(&sp.h, sizeof sp.h, 1, f) == 1 && memcmp(sp.h.magic, "SIMKSPC3", 8) == 0;
    if (ok) {
        if (sp.h.key_words < 1 || sp.h.key_words > 2) { fclose(f); return false; }
        sp.part_counts.resize(sp.h.nb_partitions); sp.keys.resize(sp.h.nb_records * sp.h.key_words); sp.counts.resize(sp.h.nb_records);
        ok = fread(sp.part_counts.data(), 4, sp.part_counts.size(), f) == sp.part_counts.size() &&
             fread(sp.keys.data(), 8, sp.keys.size(), f) == sp.keys.size() && fread(sp.counts.data(), 4, sp.counts.size(), f) == sp.counts.size();
    }
    fclose(f);
    return ok;
}

bool write_spec(const std::string &path, const Spectrum &sp) {
    const std::string part = path + ".part";          // complete files only: written aside, then renamed
    FILE *f = fopen(part.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(&sp.h, sizeof sp.h, 1, f) == 1 && fwrite(sp.part_counts.data(), 4, sp.part_counts.size(), f) == sp.part_counts.size() &&
              fwrite(sp.keys.data(), 8, sp.keys.size(), f) == sp.keys.size() && fwrite(sp.counts.data(), 4, sp.counts.size(), f) == sp.counts.size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(part.c_str(), path.c_str()) == 0;
    if (!ok) unlink(part.c_str());
    return ok;
}

void check(simka_ctx *ctx, int rc, const char *what) {
    if (rc == SIMKA_OK) return;
    std::cout << "EXCEPTION: " << what << ": " << simka_last_error(ctx) << std::endl;
    exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, char **argv) {
    const auto t_main = std::chrono::steady_clock::now();
    auto since_main = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(); };
    Options o = parse_args(argc, argv);
    // ref: src/SimkaPotara.hpp:376-387
    if (o.max_memory < 2000) std::cout << "WARNING: running Simka with low memory is risky. Simka may hang because of that. Consider running with -max-memory X where X > 2000" << std::endl;
    if (o.max_memory < 500) { std::cout << "Please run Simka with higher memory usage than 500 MB" << std::endl; return 1; }
    if (!exists(o.in)) die("ERROR: Input filename does not exist");
    if (o.kmer_size < 1 || o.kmer_size > 127) die("ERROR: -kmer-size must be in [1,127]");       // (the reference's largest span, ref: CMakeLists.txt:66-71)
    if (o.nb_gpus < 1) die("ERROR: -nb-gpus must be >= 1");
    if (o.abundance_min < 0) o.abundance_min = 0;
    o.abundance_max = std::min<long long>(std::max<long long>(o.abundance_max, 0), 999999999LL);
    o.min_shannon = std::min(std::max(o.min_shannon, 0.0), 2.0);

    mkdir_p(o.out);
    mkdir_p(o.out_tmp);
    const std::string tmp = o.out_tmp + "/simka_output_temp";     // ref: src/core/SimkaAlgorithm.cpp:234-239
    mkdir_p(tmp);

    if (o.verbose) std::cout << std::endl << "Creating input" << std::endl;
    std::vector<Sample> samples = parse_input(o.in);
    const uint32_t N = (uint32_t)samples.size();
    if (N == 0) die("ERROR: no sample in the input file");
    if (o.verbose) std::cout << "\tNb input datasets: " << N << std::endl << std::endl;
    for (auto &s : samples) for (auto &p : s.parts) for (auto &fn : p)
        if (!exists(fn)) die("ERROR: Can't open dataset: " + s.id);
    for (auto &s : samples) {       // (ref: src/core/SimkaCommons.hpp:174 divides the file list evenly over the paired parts)
        bool even = true;
        for (auto &p : s.parts) if (p.size() != s.parts[0].size()) even = false;
        if (!even) std::cerr << "WARNING: sample " << s.id << " lists a different number of files in its paired parts; like the reference, simka then takes "
                             << "files-per-part = files / parts and reads the file list in that grouping (some files may be skipped)" << std::endl;
    }
    {   // datasetIds, as the reference leaves it in the temp dir (ref: src/SimkaPotara.hpp:445-456)
        std::ofstream ids((tmp + "/datasetIds").c_str());
        for (auto &s : samples) ids << s.id << "\n";
    }

    // -max-reads (ref: src/core/SimkaAlgorithm.cpp:377-445)
    uint64_t max_reads = 0;
    if (o.max_reads == 0 || o.data_info) {
        uint64_t total = 0, mn = ~0ull, mx = 0;
        for (auto &s : samples) { const uint64_t n = count_reads(s); total += n; mn = std::min(mn, n); mx = std::max(mx, n); }
        const uint64_t mean = total / N;
        if (o.verbose) {
            std::cout << "Smaller sample contains: " << mn << " reads" << std::endl;
            std::cout << "Larger sample contains: " << mx << " reads" << std::endl;
            std::cout << "Whole dataset contains a mean of: " << mean << " reads" << std::endl << std::endl;
        }
        if (o.max_reads == 0) max_reads = (mn + mean) / 2;
    }
    if (o.max_reads > 0) max_reads = (uint64_t)o.max_reads;
    if (o.verbose) {
        if (max_reads) std::cout << "Reads per sample used up to: " << max_reads << std::endl << std::endl;
        else std::cout << "Reads per sample used: all" << std::endl << std::endl;
    }

    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nthreads = o.nb_cores > 0 ? (unsigned)o.nb_cores : hw;      // -nb-cores 0 = all (ref: src/core/Simka.cpp:88)
    if (o.parse_only) {
        SampleLoader loader(samples, o, max_reads, nthreads, nthreads + 2);
        uint64_t bases = 0, reads = 0;
        for (uint32_t i = 0; i < N; i++) {
            Packed *pk0;
            if (!loader.get(i, pk0)) die("ERROR: Can't open dataset: " + samples[i].id);
            bases += pk0->nb_bases; reads += pk0->nb_reads;
            if (o.verbose) std::cout << "sample " << samples[i].id << ": " << pk0->nb_reads << " reads, " << pk0->nb_bases << " bases" << std::endl;
            loader.release(i);
        }
        std::cout << "parsed " << reads << " reads, " << bases << " bases with " << std::min<unsigned>(nthreads, N) << " threads" << std::endl;
        return EXIT_SUCCESS;
    }
    // Contexts.  One GPU: a single context counts and merges.  G GPUs: sample i is counted over the WHOLE key space by a
    // one-sample context on GPU i % G (the reference: one simkaCount job per sample), its solid spectrum is exported, and
    // the slice of partition range [P*g/G, P*(g+1)/G) is imported into the merge context of GPU g (the reference: every
    // simkaMerge job reads partition p of every sample's solid/ directory, ref: src/SimkaMerge.cpp:1164-1264).
    const uint32_t G = (uint32_t)o.nb_gpus;
    // Several contexts of this process on ONE device (-gpu-shared: the tests of the -nb-gpus routes on a one-GPU box): their arenas are
    // plain allocations.  With lazily mapped ranges (hipMemMap / hipMemSetAccess by one worker thread while another context's kernels
    // run on the same device) about one run in a hundred ended in a GPU memory access fault at an arena's base on ROCm 7.0; 300 runs
    // with plain allocations: none.  Page tables are per device, so contexts on DISTINCT devices should be safe with mapped ranges --
    // but the worker threads of -nb-gpus map from several threads of one process all the same, and no multi-GPU box has run that
    // path yet: every -nb-gpus run takes plain arenas until one has (-gpu-mapped-arenas restores the lazily mapped ranges).
    if (G > 1 && !(o.mapped_arenas && !o.same_gpu)) setenv("SIMKA_ARENA_MALLOC", "1", 1);
    const uint32_t flags = (o.simple ? SIMKA_DIST_SIMPLE : 0u) | (o.complex_ ? SIMKA_DIST_COMPLEX : 0u);
    // -keep-tmp: samples whose spectrum is in the temp dir and still valid are not read again
    std::vector<char> reuse(N, 0);
    std::vector<uint64_t> sig(N, 0);
    uint64_t kept_partitions = 0;
    if (o.keep_tmp) {
        mkdir_p(tmp + "/solid");
        for (uint32_t i = 0; i < N; i++) {
            sig[i] = sample_signature(samples[i], o, max_reads);
            SpecHeader h;
            bool ok = read_spec_header(spec_path(tmp, samples[i], 0, 1), h) && spec_matches(h, o, 0, 1, sig[i]);
            if (ok && kept_partitions && h.nb_partitions != kept_partitions) ok = false;      // all samples of a run share one partitioning
            if (ok) { reuse[i] = 1; kept_partitions = h.nb_partitions; }
        }
    }
    // partition geometry from the largest input (2-bit bases <= file bytes); fixed up front: every context must agree
    uint64_t biggest = 1;
    for (auto &s : samples) { uint64_t b = 0; for (auto &p : s.parts) for (auto &fn : p) { struct stat st; if (stat(fn.c_str(), &st) == 0) b += (uint64_t)st.st_size * (fn.size() > 3 && fn.substr(fn.size() - 3) == ".gz" ? 5 : 1); } biggest = std::max(biggest, b); }
    uint32_t log2_parts = simka_default_log2_partitions(biggest, (uint32_t)o.kmer_size);
    if (kept_partitions) { log2_parts = 0; while (((uint64_t)1 << log2_parts) < kept_partitions) log2_parts++; }      // new samples join the kept partitioning
    const uint64_t P = (uint64_t)1 << log2_parts;
    auto make_ctx = [&](uint32_t nb_samples, int device, uint32_t shard_index = 0, uint32_t shard_count = 1) {
        simka_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.nb_samples = nb_samples; cfg.kmer_size = (uint32_t)o.kmer_size;
        cfg.abundance_min = (uint32_t)std::min<long long>(o.abundance_min, 0xffffffffLL);
        cfg.abundance_max = (uint32_t)o.abundance_max;
        cfg.dist_flags = flags; cfg.device = device; cfg.shard_index = shard_index; cfg.shard_count = shard_count;
        cfg.max_kmers_per_sample = biggest; cfg.log2_partitions = log2_parts;
        cfg.solid_capacity = (uint64_t)std::max<long long>(0, o.solid_capacity);
        simka_ctx *c = nullptr;
        if (simka_create(&cfg, &c) != SIMKA_OK) { std::cout << "EXCEPTION: " << simka_last_error(nullptr) << std::endl; exit(EXIT_FAILURE); }
        return c;
    };
    auto device_of = [&](uint32_t g) { return o.first_gpu + (o.same_gpu ? 0 : (int)g); };
    const uint64_t nw = simka_stats_nb_u64(N, flags);
    uint64_t lay[8];
    simka_stats_layout(N, flags, lay);
    std::vector<uint64_t> flat(nw, 0);
    std::vector<simka_sample_totals> totals(N);
    std::mutex out_lock;
    auto fatal = [&](simka_ctx *c, const char *what) {
        std::lock_guard<std::mutex> lk(out_lock);
        std::cout << "EXCEPTION: " << what << ": " << simka_last_error(c) << std::endl;
        exit(EXIT_FAILURE);
    };
    auto export_from = [&](simka_ctx *c, uint32_t index, uint32_t i, Spectrum &sp) -> int {
        simka_spectrum_info info;
        int rc = simka_sample_spectrum_info(c, index, &info);
        if (rc != SIMKA_OK) return rc;
        sp.part_counts.resize(info.nb_partitions); sp.keys.resize(info.nb_records * info.key_words); sp.counts.resize(info.nb_records);
        rc = simka_export_sample(c, index, sp.part_counts.data(), sp.keys.data(), sp.counts.data());
        if (rc != SIMKA_OK) return rc;
        memset(&sp.h, 0, sizeof sp.h);
        memcpy(sp.h.magic, "SIMKSPC3", 8);
        sp.h.abi = (uint64_t)simka_abi_version(); sp.h.kmer_size = (uint64_t)o.kmer_size;
        sp.h.abundance_min = (uint64_t)o.abundance_min; sp.h.abundance_max = (uint64_t)o.abundance_max;
        sp.h.shard_index = 0; sp.h.shard_count = 1; sp.h.nb_partitions = info.nb_partitions; sp.h.nb_records = info.nb_records;
        sp.h.signature = sig[i]; sp.h.key_words = info.key_words;
        return simka_get_sample_totals(c, index, &sp.h.totals);
    };
    auto fill_reads = [](const Packed &pk, simka_reads &r) {
        memset(&r, 0, sizeof r);
        r.packed = pk.words.data(); r.nb_bases = pk.nb_bases; r.nb_reads = pk.nb_frag; r.offsets = pk.offsets.data();
        r.fixed_len = 0; r.on_device = 0; r.nb_input_reads = pk.nb_reads;
    };
    auto say_reused = [&](uint32_t i) {
        if (!o.verbose) return;
        std::lock_guard<std::mutex> lk(out_lock);
        std::cout << "\t" << samples[i].id << ": k-mer spectrum reused from " << tmp << "/solid" << std::endl;
    };
    if (o.verbose) std::cout << "Counting k-mers... (log files are " << tmp << "/log/count_*)" << std::endl;

    // one sample into slot `index` of context c: its files' text parsed on the GPU (simka_ingest.hip), or the host-packed reads.
    // Returns the library's code (SIMKA_ERR_NOMEM: the caller may take another route); anything else than that and OK is fatal.
    auto count_into = [&](simka_ctx *c, uint32_t index, uint32_t i, Packed *pkp, uint64_t &n_dev_parsed, uint64_t &n_pieces) -> int {
        auto chk = [&](int r, const char *what) { if (r != SIMKA_OK && r != SIMKA_ERR_NOMEM) fatal(c, what); return r; };
        int rc;
        if (pkp->raw) {
            if ((rc = chk(simka_ingest_begin(c, index), "simka_ingest_begin")) != SIMKA_OK) return rc;
            bool irregular = false;
            const bool on_dev = !pkp->dtexts.empty();
            const size_t npieces = on_dev ? pkp->dtexts.size() : pkp->texts.size();
            for (size_t f = 0; !irregular && f < npieces; f++) {
                uint64_t nr = 0; int irr = 0;
                if (on_dev) rc = chk(simka_ingest_text_device(c, index, pkp->dtexts[f].p, pkp->dtexts[f].n, pkp->formats[f], &nr, &irr), "simka_ingest_text_device");
                else rc = chk(simka_ingest_text(c, index, pkp->texts[f].data(), pkp->texts[f].size(), pkp->formats[f], &nr, &irr), "simka_ingest_text");
                if (rc != SIMKA_OK) return rc;
                irregular = irr != 0;
                if (!irregular && nr == 0 && (f + 1 == npieces || pkp->file_of[f + 1] != pkp->file_of[f]) && (f == 0 || pkp->file_of[f - 1] != pkp->file_of[f]))
                    break;       // a FILE that delivers no read ends the sample (a file in several pieces has reads in every piece)
            }
            if (!irregular) {
                rc = chk(simka_ingest_count(c, index, nullptr, nullptr), "simka_ingest_count");
                if (rc == SIMKA_OK) { n_dev_parsed++; n_pieces += npieces; }
                return rc;
            }
            // something the device parser does not take (blank lines inside a file, multi-line FASTQ, ...): the host parser decides
            Packed hp;
            if (!load_sample(samples[i], o, max_reads, hp)) die("ERROR: Can't open dataset: " + samples[i].id);
            simka_reads r;
            fill_reads(hp, r);
            return chk(simka_count_sample(c, index, &r), "simka_count_sample");
        }
        simka_reads r;
        fill_reads(*pkp, r);
        return chk(simka_count_sample(c, index, &r), "simka_count_sample");
    };

    // ---- DIRECT: one GPU, every solid spectrum stays in its HBM arena; count, merge, download.
    // Returns SIMKA_ERR_NOMEM when the spectra do not fit the arena: the caller then takes the host-spectra path below.
    auto direct_run = [&]() -> int {
        if (o.numa_bind) {
            // The process moves to the CPUs next to the GPU before it starts its loader threads and pins their staging buffers: on the
            // two-socket MI355X box a run whose threads happened to sit on the far socket took 7.6 s where a near one took 5 s.
            char cl[4096];
            cpu_set_t set; CPU_ZERO(&set);
            int ncpu = 0;
            if (simka_device_cpulist(device_of(0), cl, sizeof cl) == SIMKA_OK && cl[0]) {
                for (const char *q = cl; *q; ) {
                    char *e; const long a = strtol(q, &e, 10); long b = a;
                    if (e == q) break;
                    if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
                    for (long x = a; x <= b && x < CPU_SETSIZE; x++) { CPU_SET((int)x, &set); ncpu++; }
                    q = *e == ',' ? e + 1 : e;
                    if (*e != ',' ) break;
                }
                if (ncpu >= 4 && sched_setaffinity(0, sizeof set, &set) == 0 && o.verbose >= 2) std::cout << "process: bound to the " << ncpu << " CPUs next to GPU " << device_of(0) << " (" << cl << ")" << std::endl;
            }
        }
        simka_ctx *c = make_ctx(N, device_of(0));
        SampleLoader loader(samples, o, max_reads, nthreads, nthreads + 2, reuse, !o.host_parse, o.host_upload ? -1 : device_of(0));
        int rc = SIMKA_OK;
        auto soft = [&](int r, const char *what) { if (r == SIMKA_OK) return true; if (r != SIMKA_ERR_NOMEM) fatal(c, what); rc = r; return false; };
        double t_wait = 0, t_count = 0;        // -verbose 2: where the main thread spends its time
        uint64_t n_dev_parsed = 0, n_pieces = 0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_begin = now();
        for (uint32_t i = 0; i < N && rc == SIMKA_OK; i++) {
            Packed *pkp;
            const double t0 = now();
            if (!loader.get(i, pkp)) die("ERROR: Can't open dataset: " + samples[i].id);
            const double t1 = now();
            t_wait += t1 - t0;
            if (reuse[i]) {     // ref: src/SimkaPotara.hpp:837-842 (count_synchro/<ID>.ok exists -> the sample is not recounted)
                Spectrum sp;
                const std::string path = spec_path(tmp, samples[i], 0, 1);
                if (!read_spec(path, sp)) die("ERROR: cannot read " + path + " (remove it to recount the sample)");
                if (sp.h.nb_partitions != P) die("ERROR: spectrum of " + samples[i].id + " has another partition count (remove " + tmp + "/solid to recount)");
                if (soft(simka_import_sample(c, i, &sp.h.totals, sp.part_counts.data(), P, sp.keys.data(), sp.counts.data(), sp.h.nb_records), "simka_import_sample")) say_reused(i);
                loader.release(i);
                continue;
            }
            const bool ok = soft(count_into(c, i, i, pkp, n_dev_parsed, n_pieces), "counting a sample");
            loader.release(i);     // host buffers may be reused as soon as the count call returns
            t_count += now() - t1;
            if (ok && o.keep_tmp) {      // persist the spectrum so that a later run with more samples skips this one
                Spectrum sp;
                if (soft(export_from(c, i, i, sp), "simka_export_sample") && !write_spec(spec_path(tmp, samples[i], 0, 1), sp))
                    die("ERROR: cannot write " + spec_path(tmp, samples[i], 0, 1));
            }
        }
        const double t2 = now();
        for (uint32_t i = 0; i < N && rc == SIMKA_OK; i++) soft(simka_get_sample_totals(c, i, &totals[i]), "simka_get_sample_totals");
        const double t3 = now();
        if (rc == SIMKA_OK) soft(simka_merge(c), "simka_merge");
        if (rc == SIMKA_OK) soft(simka_stats_download(c, flat.data(), nw, nullptr), "simka_stats_download");
        if (o.verbose >= 2) std::cout << "ingest: " << n_dev_parsed << " samples parsed on the GPU (" << n_pieces << " pieces of text), " << N - n_dev_parsed << " on the host" << std::endl;
        if (o.verbose >= 2 && g_ld_read_us.load()) std::cout << "loader threads (summed): staging buffers " << g_ld_stage_us.load() * 1e-6 << " s, device buffers " << g_ld_dev_us.load() * 1e-6
                                                             << " s, reading / inflating " << g_ld_read_us.load() * 1e-6 << " s, uploading " << g_ld_up_us.load() * 1e-6 << " s" << std::endl;
        if (o.verbose >= 2) std::cout << "main thread: waiting for the loader " << t_wait << " s, ingest / count calls " << t_count << " s, draining the count kernels " << t3 - t2
                                      << " s, merge + download " << now() - t3 << " s (since the context: " << now() - t_begin << " s)" << std::endl;
        simka_destroy(c);
        return rc;
    };

    // ---- PARTITION SHARDS (-gpu-shards partition): BASELINE.json north_star's decomposition.  GPU g holds ONE context over all N samples
    // with shard (g, G): it scans every sample's reads and keeps the minimizer partitions p with p % G == g, so no k-mer ever moves
    // between GPUs.  With -complex-dist the per-sample totals N_i are made global first (the per-k-mer terms need them; the
    // reference reads them from every count_synchro/<ID>.ok, ref: src/core/SimkaDistance.cpp:116-151), every GPU merges its partitions
    // (the reference: one simkaMerge job per partition, ref: src/SimkaPotara.hpp:974-1124), and ONE all-reduce(sum, u64) of the
    // N x N numerators / denominators combines them (SimkaStatistics::operator+=, ref: src/core/SimkaDistance.cpp:156-213): RCCL over
    // xGMI when the G contexts sit on G distinct devices and the library finds RCCL, the host otherwise (-gpu-shared, -gpu-host-sum).
    // Returns SIMKA_ERR_NOMEM when a GPU cannot hold its share of the spectra.
    auto partition_run = [&]() -> int {
        std::vector<simka_ctx *> pctx(G, nullptr);
        for (uint32_t g = 0; g < G; g++) pctx[g] = make_ctx(N, device_of(g), g, G);
        std::atomic<int> worst(SIMKA_OK);
        auto note = [&](int r) { if (r != SIMKA_OK) worst.store(r); return r == SIMKA_OK; };
        uint64_t n_dev_parsed = 0, n_pieces = 0;
        std::mutex cnt_lock;
        {
            // the text of a sample goes to every GPU: the loader leaves it in pinned host memory and each context uploads its copy
            SampleLoader loader(samples, o, max_reads, nthreads, nthreads + 2, std::vector<char>(), !o.host_parse, -1);
            for (uint32_t i = 0; i < N; i++) {
                Packed *pkp;
                if (!loader.get(i, pkp)) die("ERROR: Can't open dataset: " + samples[i].id);
                std::vector<std::thread> th;
                for (uint32_t g = 0; g < G; g++)
                    th.emplace_back([&, g] {
                        if (worst.load() != SIMKA_OK) return;
                        uint64_t ndp = 0, npc = 0;
                        note(count_into(pctx[g], i, i, pkp, ndp, npc));
                        if (g == 0) { std::lock_guard<std::mutex> lk(cnt_lock); n_dev_parsed += ndp; n_pieces += npc; }
                    });
                for (auto &t : th) t.join();
                loader.release(i);
            }
        }
        auto destroy_all = [&] { for (uint32_t g = 0; g < G; g++) if (pctx[g]) { simka_destroy(pctx[g]); pctx[g] = nullptr; } };
        if (worst.load() != SIMKA_OK) { destroy_all(); return worst.load(); }
        if (o.verbose >= 2) std::cout << "ingest: " << n_dev_parsed << " samples parsed on the GPUs (" << n_pieces << " pieces of text per GPU), " << N - n_dev_parsed << " on the host" << std::endl;
        // the shards' per-sample totals (5 rows of N) add up to the samples' totals
        std::vector<uint64_t> tsum((size_t)5 * N, 0), tpart((size_t)5 * N);
        for (uint32_t g = 0; g < G; g++) {
            const int rc = simka_totals_download(pctx[g], tpart.data());
            if (rc != SIMKA_OK && rc != SIMKA_ERR_NOMEM) fatal(pctx[g], "simka_totals_download");
            if (!note(rc)) { destroy_all(); return rc; }
            for (size_t w = 0; w < tsum.size(); w++) tsum[w] += tpart[w];
        }
        for (uint32_t i = 0; i < N; i++) {
            simka_sample_totals t0;
            if (simka_get_sample_totals(pctx[0], i, &t0) != SIMKA_OK) fatal(pctx[0], "simka_get_sample_totals");
            totals[i].nb_reads = t0.nb_reads;           // (every shard saw every read)
            totals[i].nb_distinct = tsum[0 * (size_t)N + i]; totals[i].nb_kmers = tsum[1 * (size_t)N + i]; totals[i].sum_sq = tsum[2 * (size_t)N + i];
            totals[i].distinct_all = tsum[3 * (size_t)N + i]; totals[i].kmer_occurrences = tsum[4 * (size_t)N + i];
        }
        bool use_rccl = G > 1 && !o.same_gpu && !o.gpu_host_sum;
        uint8_t comm_id[SIMKA_COMM_ID_BYTES];
        if (use_rccl && simka_comm_unique_id(comm_id) != SIMKA_OK) {
            if (o.verbose) std::cout << "partition shards: " << simka_comm_last_error(nullptr) << "; the accumulators are added on the host" << std::endl;
            use_rccl = false;
        }
        if (o.verbose) std::cout << std::endl << "Merging k-mer counts and computing distances... (" << G << " partition shards, one all-reduce "
                                 << (use_rccl ? "over RCCL" : "on the host") << ")" << std::endl;
        std::mutex acc_lock;
        auto merger = [&](uint32_t g) {
            simka_ctx *c = pctx[g];
            simka_comm *comm = nullptr;
            if (use_rccl && simka_comm_create(comm_id, (int)G, (int)g, device_of(g), &comm) != SIMKA_OK)
                die(std::string("EXCEPTION: simka_comm_create: ") + simka_comm_last_error(nullptr));
            if (o.complex_) {       // N_i global BEFORE the merge
                if (use_rccl) { if (simka_totals_allreduce(c, comm) != SIMKA_OK) fatal(c, "simka_totals_allreduce"); }
                else if (simka_totals_upload(c, tsum.data()) != SIMKA_OK) fatal(c, "simka_totals_upload");
            }
            const int rc = simka_merge(c);
            if (rc != SIMKA_OK && rc != SIMKA_ERR_NOMEM) fatal(c, "simka_merge");
            const bool ok = note(rc);
            if (use_rccl) {        // (every rank must reach the collective)
                if (!ok) die("EXCEPTION: a GPU ran out of memory in a partition-shard merge; run with -gpu-host-sum or fewer samples per GPU");
                if (o.complex_) { if (simka_stats_allreduce_head(c, comm) != SIMKA_OK) fatal(c, "simka_stats_allreduce_head"); }
                else if (simka_stats_allreduce(c, comm) != SIMKA_OK) fatal(c, "simka_stats_allreduce");
                if (g == 0) { if (simka_stats_download(c, flat.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download"); }
                else if (simka_sync(c) != SIMKA_OK) fatal(c, "simka_sync");
            } else if (ok) {
                std::vector<uint64_t> shard(nw);
                if (simka_stats_download(c, shard.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download");
                std::lock_guard<std::mutex> lk(acc_lock);          // SimkaStatistics::operator+= over the shards
                for (uint64_t w = 0; w < lay[5]; w++) flat[w] += shard[w];
            }
            if (comm) simka_comm_destroy(comm);
        };
        std::vector<std::thread> th;
        for (uint32_t g = 0; g < G; g++) th.emplace_back(merger, g);
        for (auto &t : th) t.join();
        if (!use_rccl && worst.load() == SIMKA_OK) {      // the rows of the totals behind the head: global by construction
            for (uint32_t i = 0; i < N; i++)
                for (uint32_t r = 0; r < 5; r++) flat[lay[2] + (size_t)r * N + i] = tsum[(size_t)r * N + i];
        }
        destroy_all();
        return worst.load();
    };

    // ---- DEVICE SPECTRA: G GPUs, nothing leaves device memory.  Phase 1: GPU g counts the samples i = g, g + G, ... in ONE context
    // (text parsed on the GPU like the direct path).  Phase 2: every GPU gathers its samples' records destination-major (the runs of
    // partition range h of all its samples back to back) into a send buffer and lets its count context go; GPU h copies its
    // block from every GPU (simka_device_copy: xGMI peer copies), imports the blocks into a merge context, merges its range, and the
    // heads are added (host, or one RCCL all-reduce with -gpu-allreduce).  The reference moves the same data through the
    // solid/part_<p>/ files of a shared disk (ref: src/SimkaPotara.hpp:813-1124, src/SimkaMerge.cpp:1164-1264).
    // Returns SIMKA_ERR_NOMEM when a GPU cannot hold its share: the caller then takes the host-spectra path.
    auto device_run = [&]() -> int {
        std::vector<std::vector<uint32_t>> mine(G);
        for (uint32_t i = 0; i < N; i++) mine[i % G].push_back(i);
        std::vector<simka_ctx *> cctx(G, nullptr);
        std::vector<std::vector<uint32_t>> pc(G);                   // [n_g][P] records per (local sample, partition)
        std::vector<std::vector<simka_sample_totals>> tot(G);
        const bool two = o.kmer_size > 31;          // two-word k-mers: high and low words in separate buffers (simka_*_samples_device_wide)
        std::vector<void *> send_k(G, nullptr), send_k2(G, nullptr), send_c(G, nullptr);
        std::vector<std::vector<uint64_t>> blk(G, std::vector<uint64_t>(G + 1, 0));      // blk[g][h]: where the block for GPU h starts in g's send buffers
        std::atomic<int> worst(SIMKA_OK);
        auto note = [&](int r) { if (r != SIMKA_OK) worst.store(r); return r == SIMKA_OK; };
        auto lo_of = [&](uint32_t h) { return P * h / G; };
        uint64_t n_dev_parsed = 0, n_pieces = 0;
        std::mutex cnt_lock;
        {
            SampleLoader loader(samples, o, max_reads, nthreads, nthreads + 2 * G, reuse, !o.host_parse);
            for (uint32_t g = 0; g < G; g++) if (!mine[g].empty()) cctx[g] = make_ctx((uint32_t)mine[g].size(), device_of(g));
            auto counter = [&](uint32_t g) {
                const uint32_t n = (uint32_t)mine[g].size();
                if (n == 0) return;
                simka_ctx *c = cctx[g];
                uint64_t ndp = 0, npc = 0;
                bool ok = true;
                for (uint32_t j = 0; j < n; j++) {
                    Packed *pkp;
                    if (!loader.get(mine[g][j], pkp)) die("ERROR: Can't open dataset: " + samples[mine[g][j]].id);
                    if (ok) ok = note(count_into(c, j, mine[g][j], pkp, ndp, npc));        // (after a failure: keep the loader's window moving)
                    loader.release(mine[g][j]);
                }
                { std::lock_guard<std::mutex> lk(cnt_lock); n_dev_parsed += ndp; n_pieces += npc; }
                if (!ok) return;
                std::vector<uint32_t> idx(n);
                for (uint32_t j = 0; j < n; j++) idx[j] = j;
                pc[g].assign((size_t)n * P, 0); tot[g].resize(n);
                int rc = simka_samples_spectrum_info(c, idx.data(), n, pc[g].data(), tot[g].data());
                if (rc != SIMKA_OK && rc != SIMKA_ERR_NOMEM) fatal(c, "simka_samples_spectrum_info");
                if (!note(rc)) return;
                // destination-major layout of the send buffers
                std::vector<uint64_t> off((size_t)n * P);
                uint64_t run = 0;
                for (uint32_t h = 0; h < G; h++) {
                    blk[g][h] = run;
                    for (uint32_t j = 0; j < n; j++)
                        for (uint64_t p = lo_of(h); p < lo_of(h + 1); p++) { off[(size_t)j * P + p] = run; run += pc[g][(size_t)j * P + p]; }
                }
                blk[g][G] = run;
                if (!note(simka_device_alloc(device_of(g), run * 8 + 8, &send_k[g])) || !note(simka_device_alloc(device_of(g), run * 4 + 8, &send_c[g])) ||
                    (two && !note(simka_device_alloc(device_of(g), run * 8 + 8, &send_k2[g])))) return;
                rc = two ? simka_gather_samples_device_wide(c, idx.data(), n, off.data(), send_k[g], send_k2[g], send_c[g])
                         : simka_gather_samples_device(c, idx.data(), n, off.data(), send_k[g], send_c[g]);
                if (rc != SIMKA_OK) fatal(c, "simka_gather_samples_device");
            };
            std::vector<std::thread> th;
            for (uint32_t g = 0; g < G; g++) th.emplace_back(counter, g);
            for (auto &t : th) t.join();
        }
        for (uint32_t g = 0; g < G; g++) if (cctx[g]) simka_destroy(cctx[g]);         // the arenas make room for the merge contexts
        auto drop_send = [&] { for (uint32_t g = 0; g < G; g++) { simka_device_free(device_of(g), send_k[g]); simka_device_free(device_of(g), send_k2[g]); simka_device_free(device_of(g), send_c[g]); send_k[g] = send_k2[g] = send_c[g] = nullptr; } };
        if (worst.load() != SIMKA_OK) { drop_send(); return worst.load(); }
        for (uint32_t g = 0; g < G; g++) for (size_t j = 0; j < mine[g].size(); j++) totals[mine[g][j]] = tot[g][j];
        if (o.verbose >= 2) std::cout << "ingest: " << n_dev_parsed << " samples parsed on the GPUs (" << n_pieces << " pieces of text), " << N - n_dev_parsed << " on the host" << std::endl;
        if (o.verbose) std::cout << std::endl << "Merging k-mer counts and computing distances... (spectra exchanged between the GPUs: "
                                 << [&] { uint64_t t = 0; for (uint32_t g = 0; g < G; g++) t += blk[g][G]; return t; }() << " solid k-mers)" << std::endl;
        std::mutex acc_lock;
        bool have_tail = false;
        bool use_rccl = o.gpu_allreduce && !o.same_gpu;
        uint8_t comm_id[SIMKA_COMM_ID_BYTES];
        if (use_rccl && simka_comm_unique_id(comm_id) != SIMKA_OK) {
            std::cerr << "-gpu-allreduce: " << simka_comm_last_error(nullptr) << "; summing on the host" << std::endl;
            use_rccl = false;
        }
        // (the merge contexts are created before and destroyed after the worker threads: with several contexts on ONE device -- -gpu-shared,
        // the tests -- a context's virtual ranges being reserved or given back while another context's kernels run ended in a memory
        // access fault once in some dozen runs)
        std::vector<simka_ctx *> mctx(G, nullptr);
        for (uint32_t h = 0; h < G; h++) mctx[h] = make_ctx(N, device_of(h));
        auto merger = [&](uint32_t h) {
            simka_ctx *c = mctx[h];
            simka_comm *comm = nullptr;
            if (use_rccl && simka_comm_create(comm_id, (int)G, (int)h, device_of(h), &comm) != SIMKA_OK)
                die(std::string("EXCEPTION: simka_comm_create: ") + simka_comm_last_error(nullptr));
            const uint64_t lo = lo_of(h), width = lo_of(h + 1) - lo;
            uint64_t nrec = 0;
            for (uint32_t g = 0; g < G; g++) nrec += blk[g][h + 1] - blk[g][h];
            void *rk = nullptr, *rk2 = nullptr, *rc_ = nullptr;
            bool ok = note(simka_device_alloc(device_of(h), nrec * 8 + 8, &rk)) && note(simka_device_alloc(device_of(h), nrec * 4 + 8, &rc_)) &&
                      (!two || note(simka_device_alloc(device_of(h), nrec * 8 + 8, &rk2)));
            uint64_t at = 0;
            for (uint32_t g = 0; ok && g < G; g++) {
                const uint32_t n = (uint32_t)mine[g].size();
                const uint64_t cnt = blk[g][h + 1] - blk[g][h];
                if (n == 0) continue;
                if (simka_device_copy(device_of(h), (char *)rk + at * 8, device_of(g), (const char *)send_k[g] + blk[g][h] * 8, cnt * 8) != SIMKA_OK ||
                    simka_device_copy(device_of(h), (char *)rc_ + at * 4, device_of(g), (const char *)send_c[g] + blk[g][h] * 4, cnt * 4) != SIMKA_OK ||
                    (two && simka_device_copy(device_of(h), (char *)rk2 + at * 8, device_of(g), (const char *)send_k2[g] + blk[g][h] * 8, cnt * 8) != SIMKA_OK))
                    die("EXCEPTION: simka_device_copy between GPUs failed");
                std::vector<uint32_t> pcs((size_t)n * width);
                std::vector<uint64_t> ino((size_t)n * width);
                uint64_t run = 0;
                for (uint32_t j = 0; j < n; j++)
                    for (uint64_t p = 0; p < width; p++) { pcs[(size_t)j * width + p] = pc[g][(size_t)j * P + lo + p]; ino[(size_t)j * width + p] = run; run += pcs[(size_t)j * width + p]; }
                int rc;
                if (two) {      // every sample's run of the block is contiguous and sorted (partitions = key-prefix ranges, ascending)
                    std::vector<uint64_t> soff(n), srec(n);
                    uint64_t at_ = 0;
                    for (uint32_t j = 0; j < n; j++) { soff[j] = at_; for (uint64_t p = 0; p < width; p++) at_ += pcs[(size_t)j * width + p]; srec[j] = at_ - soff[j]; }
                    rc = simka_import_samples_device_wide(c, mine[g].data(), n, tot[g].data(), soff.data(), srec.data(), (const char *)rk + at * 8, (const char *)rk2 + at * 8, (const char *)rc_ + at * 4);
                } else
                    rc = simka_import_samples_device(c, mine[g].data(), n, tot[g].data(), lo, width, pcs.data(), ino.data(), P, (const char *)rk + at * 8, (const char *)rc_ + at * 4, cnt);
                if (rc != SIMKA_OK && rc != SIMKA_ERR_NOMEM) fatal(c, "simka_import_samples_device");
                ok = note(rc);
                at += cnt;
            }
            simka_device_free(device_of(h), rk); simka_device_free(device_of(h), rk2); simka_device_free(device_of(h), rc_);
            if (ok) {
                const int rc = simka_merge(c);
                if (rc != SIMKA_OK && rc != SIMKA_ERR_NOMEM) fatal(c, "simka_merge");
                ok = note(rc);
            }
            if (use_rccl) {        // (every rank must reach the collective: a failed rank contributes zeros... it cannot: all or nothing)
                if (!ok) die("EXCEPTION: a GPU ran out of memory under -gpu-allreduce; run without it");
                if (simka_stats_allreduce_head(c, comm) != SIMKA_OK) fatal(c, "simka_stats_allreduce_head");
                if (h == 0 && simka_stats_download(c, flat.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download");
                else if (h != 0 && simka_sync(c) != SIMKA_OK) fatal(c, "simka_sync");
            } else if (ok) {
                std::vector<uint64_t> shard(nw);
                if (simka_stats_download(c, shard.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download");
                std::lock_guard<std::mutex> lk(acc_lock);          // SimkaStatistics::operator+= over the ranges (imported totals are global)
                for (uint64_t w = 0; w < lay[5]; w++) flat[w] += shard[w];
                if (!have_tail) { for (uint64_t w = lay[5]; w < nw; w++) flat[w] = shard[w]; have_tail = true; }
            }
            if (comm) simka_comm_destroy(comm);
        };
        std::vector<std::thread> th;
        for (uint32_t h = 0; h < G; h++) th.emplace_back(merger, h);
        for (auto &t : th) t.join();
        for (uint32_t h = 0; h < G; h++) simka_destroy(mctx[h]);
        drop_send();
        return worst.load();
    };

    // ---- HOST SPECTRA: spectra larger than the GPUs' memory, -keep-tmp or two-word k-mers with several GPUs.  Phase 1: sample i is counted over the whole key space by
    // a one-sample context on GPU i % G and its spectrum is taken to host memory.  Phase 2: the partition space is cut into
    // V = G * R ranges; GPU g imports the slice of range v = g, g + G, ... of every sample, merges it and adds its pair
    // accumulators to the total (the reference: one simkaMerge job per partition, summed by SimkaStatistics::operator+=).
    auto host_run = [&](uint32_t want_ranges) {
        std::vector<Spectrum> spectra(N);
        {
            std::vector<simka_ctx *> cctx(G);
            for (uint32_t g = 0; g < G; g++) cctx[g] = make_ctx(1, device_of(g));
            SampleLoader loader(samples, o, max_reads, nthreads, nthreads + 2 * G, reuse);
            auto worker = [&](uint32_t g) {
                for (uint32_t i = g; i < N; i += G) {
                    Packed *pkp;
                    if (!loader.get(i, pkp)) die("ERROR: Can't open dataset: " + samples[i].id);
                    Spectrum &sp = spectra[i];
                    if (reuse[i]) {
                        const std::string path = spec_path(tmp, samples[i], 0, 1);
                        if (!read_spec(path, sp)) die("ERROR: cannot read " + path + " (remove it to recount the sample)");
                        say_reused(i);
                        loader.release(i);
                    } else {
                        simka_reads r;
                        fill_reads(*pkp, r);
                        if (simka_count_sample(cctx[g], 0, &r) != SIMKA_OK) fatal(cctx[g], "simka_count_sample");
                        loader.release(i);
                        if (export_from(cctx[g], 0, i, sp) != SIMKA_OK) fatal(cctx[g], "simka_export_sample");
                        if (simka_reset(cctx[g]) != SIMKA_OK) fatal(cctx[g], "simka_reset");
                        if (o.keep_tmp && !write_spec(spec_path(tmp, samples[i], 0, 1), sp)) die("ERROR: cannot write " + spec_path(tmp, samples[i], 0, 1));
                    }
                    if (sp.h.nb_partitions != P) die("ERROR: spectrum of " + samples[i].id + " has another partition count (remove " + tmp + "/solid to recount)");
                    totals[i] = sp.h.totals;
                }
            };
            std::vector<std::thread> th;
            for (uint32_t g = 0; g < G; g++) th.emplace_back(worker, g);
            for (auto &t : th) t.join();
            for (uint32_t g = 0; g < G; g++) simka_destroy(cctx[g]);
        }
        // ranges per GPU: as many as it takes for one range of all samples to fit the arena (12 B per solid k-mer in 40 % of HBM)
        uint64_t total_records = 0;
        for (auto &sp : spectra) total_records += sp.h.nb_records;
        uint64_t fr = 0, tot = 0;
        simka_device_memory(device_of(0), &fr, &tot);
        uint64_t cap_records = std::max<uint64_t>(1, (uint64_t)(0.40 * (double)fr) / 12);
        if (o.solid_capacity > 0) cap_records = std::min<uint64_t>(cap_records, (uint64_t)o.solid_capacity * 8 / 10);
        uint64_t R = want_ranges ? want_ranges : std::max<uint64_t>(1, ((total_records + total_records / 8) / G + cap_records - 1) / cap_records);
        uint64_t V = std::min<uint64_t>(P, (uint64_t)G * R);
        if (o.verbose && V > G) std::cout << "Merging in " << V << " partition ranges (" << total_records << " solid k-mers)" << std::endl;
        if (o.verbose) std::cout << std::endl << "Merging k-mer counts and computing distances..." << std::endl;
        std::mutex acc_lock;
        bool have_tail = false;
        // One range per GPU on distinct devices: the heads of the G merges are combined by ONE RCCL all-reduce over xGMI
        // (simka_stats_allreduce_head: SimkaStatistics::operator+= across GPUs, ref: src/core/SimkaDistance.cpp:156-213); the
        // imported per-sample totals are global on every GPU already.  Opt-in (-gpu-allreduce: no multi-GPU node has run it yet); the
        // default, several ranges per GPU, -gpu-shared, or no RCCL on the machine: summed on the host (0.7 MB per GPU at N = 100).
        bool use_rccl = o.gpu_allreduce && G > 1 && V == G && !o.same_gpu;
        uint8_t comm_id[SIMKA_COMM_ID_BYTES];
        if (use_rccl && simka_comm_unique_id(comm_id) != SIMKA_OK) {
            std::cerr << "-gpu-allreduce: " << simka_comm_last_error(nullptr) << "; summing on the host" << std::endl;
            use_rccl = false;
        }
        std::vector<simka_ctx *> mctx(G, nullptr);          // (created before, destroyed after the worker threads: see device_run)
        for (uint32_t g = 0; g < G; g++) mctx[g] = make_ctx(N, device_of(g));
        auto merger = [&](uint32_t g) {
            simka_ctx *c = mctx[g];
            simka_comm *comm = nullptr;
            if (use_rccl && simka_comm_create(comm_id, (int)G, (int)g, device_of(g), &comm) != SIMKA_OK)
                die(std::string("EXCEPTION: simka_comm_create: ") + simka_comm_last_error(nullptr));
            std::vector<uint64_t> shard(nw), off(P + 1), kslice;
            std::vector<uint32_t> pc(P);
            bool first = true;
            for (uint64_t v = g; v < V; v += G) {
                if (!first && simka_reset(c) != SIMKA_OK) fatal(c, "simka_reset");
                first = false;
                const uint64_t lo = P * v / V, hi = P * (v + 1) / V;
                for (uint32_t i = 0; i < N; i++) {
                    const Spectrum &sp = spectra[i];
                    uint64_t before = 0, n = 0;
                    for (uint64_t p = 0; p < lo; p++) before += sp.part_counts[p];
                    std::fill(pc.begin(), pc.end(), 0u);
                    for (uint64_t p = lo; p < hi; p++) { pc[p] = sp.part_counts[p]; n += pc[p]; }
                    const uint64_t *kp = n ? sp.keys.data() + before : nullptr;
                    if (n && sp.h.key_words == 2) {        // [hi x records][lo x records]: the slice of both halves, back to back
                        kslice.resize(2 * n);
                        std::copy(sp.keys.data() + before, sp.keys.data() + before + n, kslice.begin());
                        std::copy(sp.keys.data() + sp.h.nb_records + before, sp.keys.data() + sp.h.nb_records + before + n, kslice.begin() + n);
                        kp = kslice.data();
                    }
                    if (simka_import_sample(c, i, &sp.h.totals, pc.data(), P, kp, n ? sp.counts.data() + before : nullptr, n) != SIMKA_OK)
                        fatal(c, "simka_import_sample");
                }
                if (simka_merge(c) != SIMKA_OK) fatal(c, "simka_merge");
                if (use_rccl) {
                    if (simka_stats_allreduce_head(c, comm) != SIMKA_OK) fatal(c, "simka_stats_allreduce_head");
                    if (g == 0 && simka_stats_download(c, flat.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download");
                    else if (g != 0 && simka_sync(c) != SIMKA_OK) fatal(c, "simka_sync");
                    continue;
                }
                if (simka_stats_download(c, shard.data(), nw, nullptr) != SIMKA_OK) fatal(c, "simka_stats_download");
                std::lock_guard<std::mutex> lk(acc_lock);          // SimkaStatistics::operator+= over the ranges (imported totals are global)
                for (uint64_t w = 0; w < lay[5]; w++) flat[w] += shard[w];
                if (!have_tail) { for (uint64_t w = lay[5]; w < nw; w++) flat[w] = shard[w]; have_tail = true; }
            }
            if (comm) simka_comm_destroy(comm);
        };
        std::vector<std::thread> th;
        for (uint32_t g = 0; g < G; g++) th.emplace_back(merger, g);
        for (auto &t : th) t.join();
        for (uint32_t g = 0; g < G; g++) simka_destroy(mctx[g]);
    };

    bool host_mode = G > 1 || o.merge_ranges > 0;
    bool partition_done = false;
    if (G > 1 && o.gpu_partition) {
        if (o.keep_tmp || o.merge_ranges > 0 || o.host_spectra || o.kmer_size > 31)
            std::cout << "-gpu-shards partition: not combined with -keep-tmp / -merge-ranges / -gpu-host-spectra / -kmer-size >= 32; sample shards instead" << std::endl;
        else {
            const int rc = partition_run();
            if (rc == SIMKA_OK) { partition_done = true; host_mode = false; }
            else {
                if (o.verbose) std::cout << "The partition shards do not fit the GPUs' memory: recounting with the spectra in host memory and merging by partition ranges" << std::endl;
                std::fill(flat.begin(), flat.end(), 0);
            }
        }
    }
    if (partition_done) {
    } else if (G > 1 && o.merge_ranges <= 0 && !o.keep_tmp && !o.host_spectra && N >= G && !o.gpu_partition) {
        const int rc = device_run();
        if (rc == SIMKA_OK) host_mode = false;
        else {
            if (o.verbose) std::cout << "The solid k-mer spectra do not fit the GPUs' memory at once: recounting with the spectra in host memory and merging by partition ranges" << std::endl;
            std::fill(flat.begin(), flat.end(), 0);
        }
    } else if (!host_mode) {
        if (o.verbose >= 2) std::cout << "process: " << since_main() << " s before the first context" << std::endl;
        const int rc = direct_run();
        if (o.verbose >= 2) std::cout << "process: " << since_main() << " s after count + merge (context destroyed)" << std::endl;
        if (rc == SIMKA_ERR_NOMEM) {
            if (o.verbose) std::cout << "The solid k-mer spectra do not fit the GPU memory at once: recounting with the spectra in host memory and merging by partition ranges" << std::endl;
            std::fill(flat.begin(), flat.end(), 0);
            if (o.keep_tmp) for (uint32_t i = 0; i < N; i++) { SpecHeader h; if (read_spec_header(spec_path(tmp, samples[i], 0, 1), h) && spec_matches(h, o, 0, 1, sig[i]) && h.nb_partitions == P) reuse[i] = 1; }
            host_mode = true;
        }
    }
    if (host_mode) host_run((uint32_t)std::max(0, o.merge_ranges));
    if (o.keep_tmp) {   // count_synchro/<ID>.ok with the reference's 4 lines (ref: src/SimkaCount.cpp:303-317)
        mkdir_p(tmp + "/count_synchro");
        for (uint32_t i = 0; i < N; i++) {
            std::ofstream ok((tmp + "/count_synchro/" + samples[i].id + ".ok").c_str());
            ok << totals[i].nb_reads << "\n" << totals[i].nb_distinct << "\n" << totals[i].nb_kmers << "\n" << totals[i].sum_sq << "\n";
        }
    }
    if (o.verbose) {
        std::cout << std::endl << "Nb reads / distinct k-mers / k-mers per sample (after the abundance filter):" << std::endl;
        for (uint32_t i = 0; i < N; i++)
            std::cout << "\t" << samples[i].id << ": " << totals[i].nb_reads << " / " << totals[i].nb_distinct << " / " << totals[i].nb_kmers << std::endl;
    }
    simka_stats_view view;
    if (simka_stats_describe(N, flags, flat.data(), nw, &view) != SIMKA_OK) die("EXCEPTION: simka_stats_describe");

    // outputMatrix (ref: src/core/SimkaDistance.cpp:603-649)
    std::vector<const char *> ids(N);
    for (uint32_t i = 0; i < N; i++) ids[i] = samples[i].id.c_str();
    std::vector<float> m((size_t)N * N);
    for (int w = 0; w < simka_nb_matrices(); w++) {
        if (!simka_matrix_enabled(w, flags)) continue;
        if (simka_compute_matrix(&view, w, m.data()) != SIMKA_OK) die("EXCEPTION: simka_compute_matrix");
        if (simka_write_matrix_csv(o.out.c_str(), simka_matrix_name(w), ids.data(), N, m.data(), 1) != SIMKA_OK)
            die(std::string("EXCEPTION: cannot write ") + simka_matrix_name(w) + " to " + o.out);
    }
    if (o.verbose) {
        std::cout << std::endl << "Stats" << std::endl;
        std::cout << "\tDistinct k-mers (union of samples): " << view.nb_distinct_kmers << std::endl;
        std::cout << "\tShared distinct k-mers: " << view.nb_shared_kmers << std::endl;
        std::cout << std::endl << "Output dir: " << o.out << std::endl << std::endl;
    }
    if (!o.keep_tmp) {
        unlink((tmp + "/datasetIds").c_str());
        rmdir(tmp.c_str());
    }
    if (o.verbose >= 2) std::cout << "process: " << since_main() << " s at the end of main" << std::endl;
    // Everything is on disk.  Unpinning the recycled host buffers and tearing the HIP runtime down in static destructors costs 0.3 s
    // (a fifth of a 15-GB run): the process ends here, the operating system and the driver reclaim what is left.
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    _Exit(EXIT_SUCCESS);
}
