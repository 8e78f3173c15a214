import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for gb in (0.5, 1, 2, 6, 6, 12):
    t0 = time.perf_counter()
    hip.hipMalloc(C.byref(p), int(gb * (1 << 30)))
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(int(gb * (1 << 30)))); hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    t3 = time.perf_counter()
    print(f"{gb} GB: hipMalloc {1e3*(t1-t0):.1f} ms, first memset {1e3*(t2-t1):.1f} ms, hipFree {1e3*(t3-t2):.1f} ms")
