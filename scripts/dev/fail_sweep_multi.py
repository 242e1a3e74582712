"""Development probe (development library, aliased devices): the nth allocation made by ONE device thread of kz_render_multi fails (kz_debug_fail_device), n = 1 .. N, static and
dynamic dealing, a fresh scene each time: the call returns that device's KZ_ERR_OOM or goes through with the right film, nobody hangs, and the next call renders the film."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
lib.kz_debug_alias_devices(4)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
f, t = C.c_uint64(), C.c_uint64()
lib.kz_device_mem_info(0, C.byref(f), C.byref(t))
cap = int(0.8 * min(f.value, t.value) / 4)
d = kz.scenes.glass_scene(256, 192, 8)
sc = kz.Scene(d, lib=lib); good, _ = sc.render_multi([0], max_state_bytes=cap); sc.close()
failed = through = 0
t0 = time.time()
for dealing in (0, 1):
    for n in range(1, N):
        for dv in range(4): lib.kz_device_trim(dv)
        sc = kz.Scene(d, lib=lib)
        lib.kz_debug_fail_device(1 + n % 3, n)
        try:
            film, ms = sc.render_multi([0, 1, 2, 3], tile_dealing=dealing, max_state_bytes=cap, shadow_beside=n % 3, pass_halves=(n // 3) % 3)
            assert np.array_equal(film, good), ("film of a call that did not fail", dealing, n)
            through += 1
        except kz.abi.KzError as e:
            assert e.code == 6, (dealing, n, str(e))
            failed += 1
        finally:
            for dv in range(4): lib.kz_debug_fail_device(dv, 0)
        film, ms = sc.render_multi([0, 1, 2, 3], tile_dealing=dealing, max_state_bytes=cap)
        assert np.array_equal(film, good) and (ms > 0).all(), ("film after", dealing, n)
        sc.close()
print("%d calls failed with KZ_ERR_OOM, %d went through; every film right; %.0f s" % (failed, through, time.time() - t0))
