"""Development probe (development library): injected allocation failures (kz_debug_fail_alloc, the nth allocation from now on) in front of the OTHER entry points - upload,
kz_render_tiles with a packed buffer, the dealer, kz_film_download_tiles, the 8-bit resolve, kz_render_multi - on a textured scene: an error code or the right answer, the right
answer on the retry, and the device's free memory where it started."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = kz.scenes.textured_scene(160, 128, 8)
tiles = kz.shard.deal_tiles(160, 128, 1, 0, 64)
sc = kz.Scene(d, device=0, lib=lib)
sc.render(); ref = sc.film(); ref_packed = sc.film_tiles(tiles); ref8 = sc.srgb8()
sc.close(); lib.kz_device_trim(0)
f, t = C.c_uint64(), C.c_uint64()
lib.kz_device_mem_info(0, C.byref(f), C.byref(t)); free0 = f.value
def attempt(name, n, fn, check):
    """fn(scene) under an armed failure; then again without; `check` on whatever comes back"""
    lib.kz_device_trim(0)
    out = {"failed": 0}
    sc = None
    lib.kz_debug_fail_alloc(n)
    try:
        sc = kz.Scene(d, lib=lib)
        sc.upload(0)
        r = fn(sc)
        assert check(r), (name, n, "wrong answer from a call that did not fail")
    except kz.abi.KzError as e:
        assert e.code == 6, (name, n, str(e))
        out["failed"] = 1
    finally:
        lib.kz_debug_fail_alloc(0)
    if sc is None:
        sc = kz.Scene(d, lib=lib)
    sc.upload(0)                                    # (a second upload of a resident scene is a no-op; after a failed one it is the retry)
    r = fn(sc)
    assert check(r), (name, n, "wrong answer on the retry")
    sc.close()
    return out["failed"]
def packed(sc): return sc.render_tiles(tiles, packed=True)
def dealt(sc):
    counter = np.zeros(2, np.uint32)
    took = sc.render_dealt(tiles, counter, takers=1, batch_tiles=2)
    return took, sc.film()
def rects(sc): sc.render(); return sc.film_tiles(tiles)
def eight(sc): sc.render(); return sc.srgb8()
def multi(sc): return sc.render_multi([0], tile_size=64)[0]
cases = [("upload + render", lambda sc: (sc.render(), sc.film())[1], lambda r: np.array_equal(r, ref)),
         ("render_tiles packed", packed, lambda r: np.array_equal(r, ref_packed)),
         ("dealer", dealt, lambda r: r[0] == tiles and np.array_equal(r[1], ref)),
         ("film_download_tiles", rects, lambda r: np.array_equal(r, ref_packed)),
         ("srgb8", eight, lambda r: np.array_equal(r, ref8)),
         ("render_multi", multi, lambda r: np.array_equal(r, ref))]
for name, fn, check in cases:
    nf = sum(attempt(name, n, fn, check) for n in range(1, N))
    lib.kz_device_trim(0)
    lib.kz_device_mem_info(0, C.byref(f), C.byref(t))
    print("%-22s %d of %d armed calls failed with KZ_ERR_OOM, the rest went through; every answer right; device free %+.1f MB against the start" % (name, nf, N - 1, (f.value - free0) / 1e6), flush=True)
