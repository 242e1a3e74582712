"""Development probe (development library): the nth device allocation from now on fails (kz_debug_fail_alloc), n = 1 .. N, in front of a render that uses every pass mode on a
fresh scene: the call either succeeds or fails with an error code - never anything else - and the next render on the same scene gives the film."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
d = kz.scenes.glass_scene(160, 128, 8)
sc = kz.Scene(d, device=0, lib=lib); sc.render(shadow_beside=1, pass_halves=1); ref = sc.film(); sc.close()
failed = ok = 0
for n in range(1, int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    for kw in (dict(pass_halves=2, shadow_beside=2), dict(pass_halves=1, shadow_beside=2), dict(pass_items=160 * 128 * 2, passes_in_flight=2, pass_halves=2)):
        lib.kz_device_trim(0)
        lib.kz_debug_fail_alloc(n)
        sc = None
        try:
            sc = kz.Scene(d, device=0, lib=lib)
            sc.render(**kw); sc.sync()
            assert np.array_equal(sc.film(), ref), ("film after a call that did not fail", n, kw)
            ok += 1
        except kz.abi.KzError as e:
            failed += 1
            assert e.code in (6, 7) or "alloc" in str(e) or "memory" in str(e), (n, kw, str(e))
        lib.kz_debug_fail_alloc(0)
        if sc is None or sc.device is None:
            if sc is not None: sc.close()
            sc = kz.Scene(d, device=0, lib=lib)
        sc.render(**kw)
        assert np.array_equal(sc.film(), ref), ("film of the render after the failure", n, kw)
        sc.close()
print("injected failures: %d calls failed with a code, %d went through; every film afterwards equal" % (failed, ok))
