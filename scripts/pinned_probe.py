"""hipHostMalloc cost and H2D rates, pinned against pageable (sizing the dispatcher's batch buffers).  usage: python scripts/pinned_probe.py"""
import ctypes as C
import time
import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 1 << 30) == 0
for mb in (64, 200, 400, 800):
    n = mb << 20
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipHostMalloc(C.byref(p), n, 1)      # hipHostMallocPortable
    t1 = time.perf_counter()
    assert rc == 0
    C.memset(p, 1, n)
    t2 = time.perf_counter()
    hip.hipMemcpy(d, p, min(n, 1 << 30), 1)
    t3 = time.perf_counter()
    hip.hipMemcpy(d, p, min(n, 1 << 30), 1)
    t4 = time.perf_counter()
    a = np.empty(n, dtype=np.uint8)
    t5 = time.perf_counter()
    a[:] = 1
    t6 = time.perf_counter()
    hip.hipMemcpy(d, a.ctypes.data, min(n, 1 << 30), 1)
    t7 = time.perf_counter()
    hip.hipMemcpy(d, a.ctypes.data, min(n, 1 << 30), 1)
    t8 = time.perf_counter()
    t9 = time.perf_counter()
    hip.hipHostFree(p)
    t10 = time.perf_counter()
    print("%4d MB: hipHostMalloc %.1f ms, first touch %.1f ms, H2D pinned %.1f / %.1f ms (%.1f GB/s); pageable first touch %.1f ms, H2D %.1f / %.1f ms (%.1f GB/s); hipHostFree %.1f ms"
          % (mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, n / (t4 - t3) / 1e9, (t6 - t5) * 1e3, (t7 - t6) * 1e3, (t8 - t7) * 1e3, n / (t8 - t7) / 1e9, (t10 - t9) * 1e3), flush=True)
