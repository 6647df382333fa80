"""does device memory come back after hipMemUnmap + hipMemRelease when the virtual range is kept (as simka_destroy does)?  Run plain and under rocprofv3 --kernel-trace."""
import ctypes as C, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
def free(): return torch.cuda.mem_get_info()[0] / 1e9
class Loc(C.Structure): _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Prop(C.Structure): _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p), ("allocFlags", C.c_ubyte * 8)]
class Acc(C.Structure): _fields_ = [("location", Loc), ("flags", C.c_int)]
prop = Prop(); prop.type = 1; prop.location.type = 1; prop.location.id = 0      # hipMemAllocationTypePinned, hipMemLocationTypeDevice
gran = C.c_size_t(); print("granularity rc", hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), gran.value)
print("free at start %.1f GB" % free())
SZ = 64 << 30; CH = 8 << 30
va = C.c_void_p(); print("reserve rc", hip.hipMemAddressReserve(C.byref(va), C.c_size_t(SZ), C.c_size_t(0), None, C.c_ulonglong(0)), "free %.1f" % free())
hs = []
for i in range(4):
    h = C.c_void_p(); rc = hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0))
    rc2 = hip.hipMemMap(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0))
    acc = Acc(); acc.location.type = 1; acc.location.id = 0; acc.flags = 3
    rc3 = hip.hipMemSetAccess(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.byref(acc), C.c_size_t(1))
    hs.append(h)
print("4 chunks of 8 GB mapped (rc %d %d %d): free %.1f" % (rc, rc2, rc3, free()))
t = torch.empty(0)
torch.cuda.synchronize()
for i, h in enumerate(hs):
    r1 = hip.hipMemUnmap(C.c_void_p(va.value + i * CH), C.c_size_t(CH)); r2 = hip.hipMemRelease(h)
print("unmapped + released (rc %d %d), range kept: free %.1f" % (r1, r2, free()))
import time; time.sleep(2); print("2 s later: free %.1f" % free())
print("address free rc", hip.hipMemAddressFree(va, C.c_size_t(SZ)), "free %.1f" % free())
