"""hipMalloc / hipFree cost by size on this device (sizes the decision how large the grow-only workspaces may be)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize()
for gb in (0.25, 1, 4, 16, 64):
    n = int(gb * (1 << 30))
    p = C.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); t1 = time.perf_counter()
    hip.hipMemset(p, 0, min(n, 1 << 20)); hip.hipDeviceSynchronize()
    t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
    print("%.2f GB: hipMalloc %.1f ms (rc %d), hipFree %.1f ms" % (gb, (t1 - t0) * 1e3, rc, (t3 - t2) * 1e3), flush=True)
