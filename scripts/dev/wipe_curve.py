"""Development probe: what hipMemGetInfo reports while the driver wipes memory a process has just released: hold N GB (touched by a memset), free it, poll every 10 ms."""
import ctypes as C, sys, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]; hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
f, t = C.c_size_t(), C.c_size_t()
for rep in range(2):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), int(gb * 1e9)) == 0
    assert hip.hipMemset(p, 1, int(gb * 1e9)) == 0 and hip.hipDeviceSynchronize() == 0
    hip.hipMemGetInfo(C.byref(f), C.byref(t)); held = f.value
    t0 = time.perf_counter(); hip.hipFree(p); t_free = time.perf_counter() - t0
    series = []
    while time.perf_counter() - t0 < 9.0:
        hip.hipMemGetInfo(C.byref(f), C.byref(t)); series.append((time.perf_counter() - t0, f.value))
        time.sleep(0.01)
    # compress: print when free changed by > 1 GB, and the longest flat stretch while below the final value
    final = series[-1][1]; last_t, last_v = series[0]; out = [(round(last_t, 2), round(last_v / 1e9, 1))]; longest_flat = 0.0; flat_start = last_t
    for ts, v in series[1:]:
        if v - last_v > 1e9:
            out.append((round(ts, 2), round(v / 1e9, 1))); longest_flat = max(longest_flat, ts - flat_start) if last_v < final - 1e9 else longest_flat; last_v = v; flat_start = ts
    print("rep %d: hipFree took %.3f s; free right before the release %.1f GB, at the end %.1f GB; longest stretch without +1 GB while still below the end value: %.2f s" % (rep, t_free, held / 1e9, final / 1e9, longest_flat))
    print("   (t, free GB):", out[:6], "...", out[-4:], "points", len(out), flush=True)
