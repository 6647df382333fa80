// lds_occupancy.hip -- how many 256-thread blocks with B bytes of dynamic LDS does a CU of this GPU really hold?
// The occupancy API divides 160 KB by B; the dispatcher may allocate in larger granules.  Measured: a grid of (#CUs x n) blocks that each
// spin for a fixed time finishes in one spin if n blocks are co-resident per CU, in two if not.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occupancy scripts/ubench/lds_occupancy.hip && /tmp/lds_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_spin(unsigned *out, unsigned long long ticks) {
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) out[blockIdx.x] = s[255];
}
int main() {
    (void)hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned *out; (void)hipMalloc(&out, (size_t)cus * 16 * 4);
    const unsigned long long ticks = 20000;      // 200 us at the 100 MHz wall clock
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int sizes[] = { 40576, 33280, 32768, 32712, 32256, 32000, 31744, 30720, 29696, 28672, 27136, 26624, 25600 };
    for (int b : sizes) {
        int api = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void *)k_spin, 256, (size_t)b);
        printf("%6d bytes: API says %d;", b, api);
        for (int n = 4; n <= 7; n++) {
            hipLaunchKernelGGL(k_spin, dim3(cus * n), dim3(256), (size_t)b, 0, out, 100ull);      // warm
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_spin, dim3(cus * n), dim3(256), (size_t)b, 0, out, ticks);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %d per CU: %.2f ms", n, ms);
        }
        printf("\n");
    }
    return 0;
}
