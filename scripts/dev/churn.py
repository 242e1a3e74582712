"""Development probe: scenes created, uploaded, rendered (every pass mode) and destroyed in a loop - device memory, host memory and handles must not creep."""
import importlib, os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
import psutil
lib = kz.abi.load_library()
proc = psutil.Process()
f, t = C.c_uint64(), C.c_uint64()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ref = None
for i in range(n):
    d = kz.scenes.cornell_box(160, 120, 8, sampler="independent", seed=0)
    sc = kz.Scene(d, device=0)
    sc.render(shadow_beside=i % 3, pass_halves=i % 3)
    film = sc.film()
    if ref is None:
        ref = film
    assert np.array_equal(film, ref), i
    if i % 7 == 0:
        sc.render(pass_items=160 * 120 * 2, passes_in_flight=3); assert np.array_equal(sc.film(), ref), i
    sc.close()
    if i % 50 == 0 or i == n - 1:
        lib.kz_device_mem_info(0, C.byref(f), C.byref(t))
        print("iteration %4d: device free %.3f GB, host rss %.1f MB, threads %d, fds %d" % (i, f.value / 1e9, proc.memory_info().rss / 1e6, proc.num_threads(), proc.num_fds()), flush=True)
