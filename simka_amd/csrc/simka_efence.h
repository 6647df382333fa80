// simka_efence.h -- test builds only (-DSIMKA_EFENCE, scripts/build_variant.sh efence -DSIMKA_EFENCE): every hipMalloc of the library ENDS at the
// end of a mapping of its own with nothing mapped behind it, so a kernel that reads or writes past the end of a buffer faults every time
// (with hipMalloc the bytes behind a buffer usually belong to the same pool block: an overrun shows up once in a while, where the block ends).
// Ranges are never given back (see g_vmm_retired_bytes in simka_ctx.hip: a range that is mapped again is read through stale translations).
#pragma once
#ifdef SIMKA_EFENCE
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
namespace simka_efence {
struct Info { void *va; size_t mapped; hipMemGenericAllocationHandle_t h; };
inline std::mutex &lock() { static std::mutex m; return m; }
inline std::map<void *, Info> &table() { static std::map<void *, Info> t; return t; }
inline hipError_t alloc(void **p, size_t bytes) {
    std::lock_guard<std::mutex> g(lock());
    int dev = 0; (void)hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) gran = (size_t)2 << 20;
    const size_t want = ((bytes ? bytes : 1) + 15) & ~(size_t)15;
    const size_t mapped = (want + gran - 1) / gran * gran;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, mapped + gran, gran, nullptr, 0);       // + one granule that stays unmapped
    if (e != hipSuccess) return e;
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, mapped, &prop, 0);
    if (e != hipSuccess) return e;
    e = hipMemMap(va, mapped, 0, h, 0);
    if (e != hipSuccess) { (void)hipMemRelease(h); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, mapped, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(va, mapped); (void)hipMemRelease(h); return e; }
    void *user = (char *)va + (mapped - want);
    table()[user] = Info{ va, mapped, h };
    *p = user;
    return hipSuccess;
}
inline hipError_t release(void *p) {
    if (!p) return hipSuccess;
    Info inf; bool mine = false;
    { std::lock_guard<std::mutex> g(lock()); auto it = table().find(p); if (it != table().end()) { inf = it->second; table().erase(it); mine = true; } }
    if (!mine) return (hipFree)(p);
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(inf.va, inf.mapped);
    (void)hipMemRelease(inf.h);
    return hipSuccess;
}
template <class T> inline hipError_t alloc_t(T **p, size_t bytes) { return alloc((void **)p, bytes); }
}
#define hipMalloc(p, n) simka_efence::alloc_t((p), (n))
#define hipFree(p) simka_efence::release((void *)(p))
#endif
