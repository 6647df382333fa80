// how long hipMalloc takes by size, and a virtual range mapped chunk by chunk (hipMemAddressReserve / hipMemCreate / hipMemMap)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n, size_t stride) { size_t i = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main() {
    hipFree(0);
    for (size_t gib : {1, 4, 8, 16, 32, 64, 100}) {
        void *p = nullptr; double t = now();
        hipError_t e = hipMalloc(&p, gib << 30); double t1 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (char *)p, gib << 30, (gib << 30) / (4096 * 256)); hipDeviceSynchronize(); double t2 = now();
        hipFree(p); double t3 = now();
        printf("hipMalloc %3zu GiB: %s  alloc %.3f s  touch %.3f s  free %.3f s\n", gib, hipGetErrorString(e), t1 - t, t2 - t1, t3 - t2);
    }
    // VMM
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity %zu (%s)\n", gran, hipGetErrorString(e));
    const size_t total = (size_t)100 << 30, chunk = (size_t)2 << 30;
    void *va = nullptr; double t = now();
    e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
    printf("reserve 100 GiB: %s %.3f s\n", hipGetErrorString(e), now() - t);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    double tm = 0; int n = 0;
    for (size_t off = 0; off < ((size_t)20 << 30); off += chunk) {
        double t0 = now();
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0); if (e != hipSuccess) { printf("create: %s\n", hipGetErrorString(e)); break; }
        e = hipMemMap((char *)va + off, chunk, 0, h, 0); if (e != hipSuccess) { printf("map: %s\n", hipGetErrorString(e)); break; }
        e = hipMemSetAccess((char *)va + off, chunk, &acc, 1); if (e != hipSuccess) { printf("access: %s\n", hipGetErrorString(e)); break; }
        hs.push_back(h); tm += now() - t0; n++;
    }
    printf("mapped %d chunks of 2 GiB: %.3f s each\n", n, n ? tm / n : 0.0);
    if (n) { hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (char *)va, (size_t)n * chunk, ((size_t)n * chunk) / (4096 * 256)); printf("touch: %s\n", hipGetErrorString(hipDeviceSynchronize())); }
    return 0;
}
