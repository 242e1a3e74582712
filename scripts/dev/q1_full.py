"""The reference's own parameter-study scene (tests/golden/q1_default_m0_r0.5.npz = scene/2022_q1/parameters/default_m0_r0.5.xml + OBJ files, 36 378 triangles) at the
settings of its scene file - 1920 x 1080, 4096 spp, independent sampler, path_mis depth 5 - through the HIP path: time, Msamples/s, and the 8-bit sRGB picture (as
checked in, and with the light factors of tests/test_xmlscene.py) for the comparison with the published doc/2022_q1/img/param/default_m0_r0.5.png, which
scripts/dev/q1_compare.py makes in the build container (the reference does not travel to the GPU box):   python scripts/dev/q1_full.py [spp]
--all: all 22 scene files of scene/2022_q1/parameters/ (the npz + tests/golden/q1_params.json), each at 4096 spp with the light factors -> gpurun_out/q1_full/all/<name>.png"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
OUT = os.path.join(ROOT, "gpurun_out", "q1_full"); os.makedirs(OUT, exist_ok=True)
FACTORS = (0.976, 1.135, 1.034)
ALL = "--all" in sys.argv
BESIDE = "--beside" in sys.argv           # --all --beside: every picture a second time with KzRenderOpts::shadowBeside = 2 (time, and that the picture is the same)
argv = [a for a in sys.argv if a not in ("--all", "--beside")]
over = {"sampler": {"sampleCount": int(argv[1])}} if len(argv) > 1 else None
if ALL:
    params = json.load(open(os.path.join(ROOT, "tests", "golden", "q1_params.json")))["params"]
    os.makedirs(os.path.join(OUT, "all"), exist_ok=True)
    res = {}
    sc = None
    for name, bsdf in params.items():
        d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), over)
        d.meshes[4]["bsdf"] = {k: v for k, v in bsdf.items() if not k.startswith("_")}
        for l, f in zip([m["light"] for m in d.meshes if m["light"]], FACTORS):
            l["intensity"] *= f
        sc = kz.Scene(d, device=0)
        sc.render(0, min(64, sc.sample_count)); sc.sync()
        t0 = time.perf_counter(); sc.render(); sc.sync(); dt = time.perf_counter() - t0
        kz.output.save_png(os.path.join(OUT, "all", name), sc.srgb8())
        res[name] = {"render_s": round(dt, 3), "Msamples_per_s": round(sc.width * sc.height * sc.sample_count / dt / 1e6, 1), "spp": sc.sample_count, "mean_linear_rgb": float(sc.rgb().mean())}
        if BESIDE:
            film = sc.film()
            t0 = time.perf_counter(); sc.render(shadow_beside=2); sc.sync(); db = time.perf_counter() - t0
            res[name]["shadow_beside"] = {"render_s": round(db, 3), "Msamples_per_s": round(sc.width * sc.height * sc.sample_count / db / 1e6, 1), "film_equal": bool(np.array_equal(film, sc.film()))}
        print(name, res[name], flush=True)
        sc.close()
    json.dump(res, open(os.path.join(OUT, "q1_all.json"), "w"), indent=1)
    sys.exit(0)
res = {}
for tag, factors in (("as_checked_in", None), ("light_factors", FACTORS)):
    d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), over)
    if factors:
        for l, f in zip([m["light"] for m in d.meshes if m["light"]], factors):
            l["intensity"] *= f
    t0 = time.time(); sc = kz.Scene(d, device=0); tb = time.time() - t0
    n = sc.width * sc.height * sc.sample_count
    sc.render(0, min(64, sc.sample_count)); sc.sync()                      # warm-up: allocations, first launches
    t0 = time.perf_counter(); sc.render(); sc.sync(); dt = time.perf_counter() - t0
    px = sc.srgb8()
    kz.output.save_png(os.path.join(OUT, "hip_default_m0_r0.5_%s" % tag), px)
    res[tag] = {"width": sc.width, "height": sc.height, "spp": sc.sample_count, "sampler": d.sampler["type"], "maxDepth": d.integrator["maxDepth"], "tris": d.n_tris(),
                "scene_build_s": round(tb, 2), "render_s": round(dt, 3), "Msamples_per_s": round(n / dt / 1e6, 1), "mean_linear_rgb": float(sc.rgb().mean()), "passes": sc.last_pass_info()}
    print(tag, res[tag], flush=True)
    sc.close()
json.dump(res, open(os.path.join(OUT, "q1_full.json"), "w"), indent=1)
