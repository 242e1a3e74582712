import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
d = kz.scenes.glass_scene(160, 128, 8)
sc = kz.Scene(d, device=0, lib=lib); sc.render(shadow_beside=1, pass_halves=1); ref = sc.film(); sc.close()
for kw in (dict(pass_items=160 * 128 * 2, passes_in_flight=2, pass_halves=2), dict(pass_items=160 * 128 * 2, passes_in_flight=2), dict(pass_items=160 * 128 * 2, passes_in_flight=1)):
    for n in range(30, 60):
        lib.kz_device_trim(0)
        lib.kz_debug_fail_alloc(n)
        sc = kz.Scene(d, device=0, lib=lib) if True else None
        msg = "went through"
        try:
            sc.render(**kw); sc.sync()
        except kz.abi.KzError as e:
            msg = str(e)[:150]
        lib.kz_debug_fail_alloc(0)
        try:
            sc.render(**kw)
            f = sc.film()
            eq = np.array_equal(f, ref)
            extra = "" if eq else " max diff %.3g, texels differing %d, mean ratio %.4f" % (np.abs(f - ref).max(), int((f != ref).any(axis=-1).sum()), float(f[..., 3].sum() / ref[..., 3].sum()))
        except kz.abi.KzError as e:
            eq, extra = False, " second render failed: " + str(e)[:120]
        if not eq or n in (46, 47, 48):
            print(kw, n, "|", msg, "| next film equal:", eq, extra, flush=True)
        sc.close()
