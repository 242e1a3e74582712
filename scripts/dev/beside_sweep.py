"""Development probe: the shadow rays of a bounce BESIDE its closest-hit rays (KzRenderOpts::shadowBeside = 2: the context's side stream, wfPass) against the one-stream
order (1) and the default (0): wall time of whole jobs (best of 5 calls incl. sync, after one) and film equality, over job sizes from 2^20 to 2^30 items."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
def q1_at(w, h, spp): return S.load_npz(q1, overrides={"camera": {"width": w, "height": h}, "sampler": {"type": "independent", "sampleCount": spp, "seed": 0}})
jobs = [("q1 asset 256x256x16 (C1)", lambda: q1_at(256, 256, 16)), ("q1 asset 256x256x64", lambda: q1_at(256, 256, 64)), ("q1 asset 512x512x64", lambda: q1_at(512, 512, 64)),
        ("q1 asset 1024x1024x64", lambda: q1_at(1024, 1024, 64)), ("q1 asset 1920x1080x64", lambda: q1_at(1920, 1080, 64)),
        ("cornell 256x256x16", lambda: S.cornell_box(256, 256, 16)), ("cornell 512x512x32", lambda: S.cornell_box(512, 512, 32)), ("sphere_env 512x512x64 (C2)", lambda: S.sphere_env(512, 512, 64)),
        ("glass 256x256x64", lambda: S.glass_scene(256, 256, 64)),
        ("random triangles 100k 512x512x16", lambda: S.random_triangles(100000, 512, 512, 16, sampler="independent")),
        ("random triangles 1M 960x540x16", lambda: S.random_triangles(1000000, 960, 540, 16, sampler="independent")),
        ("random triangles 1M 1920x1080x16", lambda: S.random_triangles(1000000, 1920, 1080, 16, sampler="independent")),
        ("random triangles 1M 1920x1080x64", lambda: S.random_triangles(1000000, 1920, 1080, 64, sampler="independent")),
        ("q1 asset 1920x1080x512", lambda: q1_at(1920, 1080, 512)),
        ("hero 960x540x16", lambda: S.hero_scene(960, 540, 16)), ("hero 1920x1080x64", lambda: S.hero_scene(1920, 1080, 64)), ("hero 1920x1080x256 (C3)", lambda: S.hero_scene(1920, 1080, 256))]
if len(sys.argv) > 1: jobs = [j for j in jobs if any(k in j[0] for k in sys.argv[1:])]
MODES = {"one stream": dict(shadow_beside=1, pass_halves=1), "beside": dict(shadow_beside=2, pass_halves=1), "halves": dict(shadow_beside=1, pass_halves=2),
         "halves + beside": dict(shadow_beside=2, pass_halves=2), "default": {}}
for name, make in jobs:
    d = make()
    sc = kz.Scene(d, device=0)
    items = sc.width * sc.height * d.sampler["sampleCount"]
    res = {}
    for mode, sb in MODES.items():
        ts = []
        for i in range(8 if mode == "default" else 6):
            t0 = time.perf_counter(); sc.render(**sb); sc.sync(); ts.append(time.perf_counter() - t0)
        res[mode] = (min(ts[5:] if mode == "default" else ts[1:]), sc.film().copy())
    a = res["one stream"]
    print("%-36s items 2^%.1f  one stream %8.3f ms  " % (name, np.log2(items), 1e3 * a[0]) + "  ".join("%s %8.3f ms (%+5.1f %%)" % (k, 1e3 * v[0], 100 * (v[0] / a[0] - 1)) for k, v in res.items() if k != "one stream")
          + "  films equal: %s" % all(np.array_equal(a[1], v[1]) for v in res.values()), flush=True)
    del sc
