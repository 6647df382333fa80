// simka_ingest.hip -- K1 of SURVEY 2.5: FASTA / FASTQ text -> 2-bit packed reads ON THE DEVICE (gfx950, wave64).
//
// The reference reads its inputs through gatb's Bank layer (absent from the tree; formats per README.md:165); the host path of this
// framework (SeqReader + simka_pack_read, simka_cli.cpp / simka_host.cpp) parses ~280 MB/s per thread, far below what the count
// kernels consume.  Here the file's bytes are copied to the GPU as they are and parsed there:
//
//   k_ing_nl_count / k_ing_nl_fill   the line table: start offset of every line (newline flags -> tile counts -> prefix sum -> fill)
//   k_ing_lines                      one thread per line: header / sequence line (FASTA: first character '>'; FASTQ: line number
//                                    modulo 4), its ACGT bases and the fragments that START in it (a fragment = maximal run of
//                                    ACGT letters of one read; every other letter ends it, as simka_pack_read does; a FASTA
//                                    sequence may continue over several lines)
//   (prefix sums of bases / fragments per line)
//   k_ing_pack                       one thread per sequence line: the bases into the packed stream (atomicOr of whole words),
//                                    the fragment starts into the offsets array
//
// What the kernels do not take -- anything but 4-line FASTQ records, blank lines inside a file, a file of 4 GB or more -- is FLAGGED
// (irregular), never guessed: the caller parses such a file on the host.  Results are bit-identical to the host path
// (tests/test_gpu_parity.py::test_device_ingest_*).
#ifndef SIMKA_INGEST_HIP
#define SIMKA_INGEST_HIP
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ING_TILE 4096            // bytes per block of the newline kernels (256 threads x 16)
#define ING_BLOCK 256
#define ING_FASTA 0
#define ING_FASTQ 1
// line info word: bases of the line (bits 0..27) | flags
#define ING_SEQ (1u << 28)
#define ING_HDR (1u << 29)

__device__ __forceinline__ bool ing_is_base(unsigned char c) { const unsigned char u = c & 0xDFu; return u == 'A' || u == 'C' || u == 'G' || u == 'T'; }
__device__ __forceinline__ uint32_t ing_code(unsigned char c) { return ((uint32_t)c >> 1) & 3u; }      // A 0, C 1, T 2, G 3 (either case)

// newlines per tile
__global__ void __launch_bounds__(ING_BLOCK)
k_ing_nl_count(const unsigned char *text, uint64_t n, uint32_t *cnt) {
    __shared__ uint32_t s_w[ING_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * 16u;
    uint32_t c = 0;
    if (base + 16u <= n) {
        const uint4 v = *(const uint4 *)(text + base);          // (the text buffer is 16-byte aligned, tiles are multiples of 16)
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b = 0; b < 4; b++) c += ((w[q] >> (8 * b)) & 0xffu) == (uint32_t)'\n' ? 1u : 0u;
    } else for (uint64_t i = base; i < n && i < base + 16u; i++) c += text[i] == '\n' ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// lines[0] = 0, lines[1 + r] = position after the r-th newline; the caller sets lines[nlines] = n + 1 (sentinel: the last line ends at n)
__global__ void __launch_bounds__(ING_BLOCK)
k_ing_nl_fill(const unsigned char *text, uint64_t n, const uint32_t *tile_off, uint32_t *lines) {
    __shared__ uint32_t s_w[ING_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * 16u;
    unsigned char b[16];
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { b[q] = (base + q < n) ? text[base + q] : (unsigned char)0; c += b[q] == '\n' ? 1u : 0u; }
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((threadIdx.x & 63u) >= (uint32_t)o) inc += t; }
    if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t r = tile_off[blockIdx.x] + inc - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) r += s_w[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) lines[0] = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) if (b[q] == '\n') { lines[1u + r] = (uint32_t)(base + q + 1u); r++; }
}

// totals: [0] reads, [1] irregular flag
// end of line i without its line terminator(s)
__device__ __forceinline__ uint32_t ing_line_end(const unsigned char *text, const uint32_t *lines, uint32_t i) {
    const uint32_t st = lines[i];
    uint32_t en = lines[i + 1u] - 1u;                   // position of the '\n' (or n for the last line)
    while (en > st && text[en - 1u] == '\r') en--;     // every trailing CR, as the host's SeqReader::getline does (one left behind would end a FASTA fragment that continues on the next line)
    return en;
}

__global__ void __launch_bounds__(ING_BLOCK)
k_ing_lines(const unsigned char *text, const uint32_t *lines, uint32_t nlines, int format, uint32_t *lbases, uint32_t *lfrags, unsigned long long *totals) {
    const uint32_t i = blockIdx.x * ING_BLOCK + threadIdx.x;
    uint32_t reads = 0, bad = 0;
    if (i < nlines) {
        const uint32_t st = lines[i], en = ing_line_end(text, lines, i);
        const bool empty = en == st;
        bool seq = false;
        if (empty) {
            // a blank line is only accepted at the end of the file (every line after it blank as well: checked pairwise)
            if (i + 1u < nlines && ing_line_end(text, lines, i + 1u) != lines[i + 1u]) bad = 1;
        } else if (format == ING_FASTA) {
            if (text[st] == '>') reads = 1; else { seq = true; if (i == 0) bad = 1; }       // (a sequence line before any header)
        } else {
            const uint32_t ph = i & 3u;
            if (ph == 0) { if (text[st] == '@') reads = 1; else bad = 1; }
            else if (ph == 1) seq = true;
            else if (ph == 2) { if (text[st] != '+') bad = 1; }
        }
        if (format == ING_FASTQ && empty && (i & 3u) == 1u && i + 1u < nlines) bad = 1;
        uint32_t nb = 0, nf = 0;
        if (seq) {
            // does the first fragment continue the last one of the line before (FASTA sequences over several lines)?
            bool open = false;
            if (format == ING_FASTA && i > 0) {
                const uint32_t pst = lines[i - 1u], pen = ing_line_end(text, lines, i - 1u);
                open = pen > pst && text[pst] != '>' && ing_is_base(text[pen - 1u]);
            }
            for (uint32_t p = st; p < en; p++) {
                if (ing_is_base(text[p])) { nb++; if (!open) { nf++; open = true; } }
                else open = false;
            }
        }
        lbases[i] = nb; lfrags[i] = nf;
    }
    const unsigned long long rm = __ballot(reads != 0u);
    if (__ballot(bad != 0u) && (threadIdx.x & 63u) == 0) atomicOr(&totals[1], 1ull);
    if (rm && (threadIdx.x & 63u) == 0) atomicAdd(&totals[0], (unsigned long long)__popcll(rm));
}

// the bases of every sequence line into packed[] (2 bits each, base `base0 + lbo[i] + j` at bit 2 * (that % 32) of word that / 32; the
// words from base0 on are zero), the starts of its fragments into offsets[frag0 + lfo[i] ...]
__global__ void __launch_bounds__(ING_BLOCK)
k_ing_pack(const unsigned char *text, const uint32_t *lines, uint32_t nlines, int format, const uint32_t *lbases, const uint32_t *lbo, const uint32_t *lfo,
           unsigned long long base0, unsigned long long frag0, unsigned long long *packed, unsigned long long *offsets) {
    const uint32_t i = blockIdx.x * ING_BLOCK + threadIdx.x;
    if (i >= nlines || lbases[i] == 0u) return;
    const uint32_t st = lines[i], en = ing_line_end(text, lines, i);
    bool open = false;
    if (format == ING_FASTA && i > 0) {
        const uint32_t pst = lines[i - 1u], pen = ing_line_end(text, lines, i - 1u);
        open = pen > pst && text[pst] != '>' && ing_is_base(text[pen - 1u]);
    }
    unsigned long long pos = base0 + lbo[i], f = frag0 + lfo[i];
    unsigned long long acc = 0, w = pos >> 5;
    for (uint32_t p = st; p < en; p++) {
        const unsigned char c = text[p];
        if (!ing_is_base(c)) { open = false; continue; }
        if (!open) { offsets[f++] = pos; open = true; }
        if ((pos >> 5) != w) { if (acc) atomicOr(&packed[w], acc); acc = 0; w = pos >> 5; }
        acc |= (unsigned long long)ing_code(c) << ((pos & 31ull) * 2ull);
        pos++;
    }
    if (acc) atomicOr(&packed[w], acc);
}

#endif
