// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the merge side (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel touches a KNOWN set of
// 128-byte lines of a buffer far larger than the caches (no reuse); run under  rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE  and compare
// (scripts/fetch_calib.sh prints counter / known bytes per kernel).
//   k_stream16     every lane reads 16 consecutive bytes (global_load_dwordx4), whole lines
//   k_runs8        runs of 9 consecutive u64 (72 B) at random 8-byte-aligned starts, one lane per element -- a slice of key records as
//                  k_group gathers it; the lines a run touches (1 or 2) are counted on the host
//   k_runs4        the same with u32 (36-byte runs) -- the count slices
//   k_write8       coalesced 8-byte stores of whole lines (the CSR entries k_group writes)
//   k_write_runs16 runs of three 16-byte records at random 16-byte-aligned starts (partial-line stores, the scan's bucket runs)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_stream16(const uint4 *in, uint64_t n16, uint64_t *out) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}
template <typename T, int RUN>
__global__ void k_runs(const T *in, const uint64_t *starts, uint64_t nruns, uint64_t *out) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nruns * RUN; i += (uint64_t)gridDim.x * blockDim.x) acc += (uint64_t)in[starts[i / RUN] + i % RUN];
    if (acc == 0x123456789abcdefull) out[0] = acc;
}
__global__ void k_write8(uint64_t *o, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) o[i] = i;
}
__global__ void k_write_runs16(uint4 *o, const uint64_t *starts, uint64_t nruns) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nruns * 3; i += (uint64_t)gridDim.x * blockDim.x) o[starts[i / 3] + i % 3] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const uint64_t BYTES = 8ull << 30;              // 8 GiB buffer: 32 x the Infinity Cache
    void *buf; CHK(hipMalloc(&buf, BYTES)); CHK(hipMemset(buf, 1, BYTES));
    uint64_t *out; CHK(hipMalloc(&out, 64));
    const uint64_t NR = 32ull << 20;                // runs per gather kernel
    std::mt19937_64 rng(12345);
    auto make_starts = [&](uint64_t elem, uint64_t run, uint64_t align_elems, uint64_t &lines) {
        // disjoint random runs: run r lives in its own window of 64 elements-worth of lines, at a random aligned offset
        std::vector<uint64_t> s(NR);
        const uint64_t window = BYTES / elem / NR;
        lines = 0;
        for (uint64_t r = 0; r < NR; r++) {
            const uint64_t off = (rng() % (window - run)) / align_elems * align_elems;
            s[r] = r * window + off;
            const uint64_t b0 = s[r] * elem, b1 = b0 + run * elem - 1;
            lines += b1 / 128 - b0 / 128 + 1;
        }
        return s;
    };
    uint64_t l8, l4, l16;
    std::vector<uint64_t> s8 = make_starts(8, 9, 1, l8), s4 = make_starts(4, 9, 1, l4), s16 = make_starts(16, 3, 1, l16);
    uint64_t *d8, *d4, *d16;
    CHK(hipMalloc(&d8, NR * 8)); CHK(hipMalloc(&d4, NR * 8)); CHK(hipMalloc(&d16, NR * 8));
    CHK(hipMemcpy(d8, s8.data(), NR * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(d4, s4.data(), NR * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(d16, s16.data(), NR * 8, hipMemcpyHostToDevice));
    CHK(hipDeviceSynchronize());
    const int grid = 256 * 8, block = 256;
    k_stream16<<<grid, block>>>((const uint4 *)buf, BYTES / 16, out);
    k_runs<uint64_t, 9><<<grid, block>>>((const uint64_t *)buf, d8, NR, out);
    k_runs<uint32_t, 9><<<grid, block>>>((const uint32_t *)buf, d4, NR, out);
    k_write8<<<grid, block>>>((uint64_t *)buf, BYTES / 8);
    k_write_runs16<<<grid, block>>>((uint4 *)buf, d16, NR);
    CHK(hipDeviceSynchronize());
    // known bytes: whole lines the kernels must move (+ the start tables of the gather kernels, read coalesced: 8 B per run, each word by 9 / 3 lanes of one wave)
    printf("known k_stream16 read %llu write 0\n", (unsigned long long)BYTES);
    printf("known k_runs<unsigned long, 9> read %llu write 0 payload %llu\n", (unsigned long long)(l8 * 128 + NR * 8), (unsigned long long)(NR * 72));
    printf("known k_runs<unsigned int, 9> read %llu write 0 payload %llu\n", (unsigned long long)(l4 * 128 + NR * 8), (unsigned long long)(NR * 36));
    printf("known k_write8 read 0 write %llu\n", (unsigned long long)BYTES);
    printf("known k_write_runs16 read %llu write %llu payload %llu\n", (unsigned long long)(NR * 8), (unsigned long long)(l16 * 128), (unsigned long long)(NR * 48));
    return 0;
}
