"""Development probe (variant library with KZ_HALF_STAGGER read per pass): a pass as halves with the second half started one stage behind the first (its generate and camera
rays run beside the first half's first shade ... ) against halves started together and one stream."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
for name, make in (("q1 asset x256", lambda: S.load_npz(q1, overrides={"sampler": {"type": "independent", "sampleCount": 256, "seed": 0}})), ("C4 x256", lambda: S.random_triangles(1000000, 1920, 1080, 256, sampler="independent")),
                   ("materials_scene", lambda: S.materials_scene(1920, 1080, 256)), ("textured_scene", lambda: S.textured_scene(1920, 1080, 256)), ("hero (C3)", lambda: S.hero_scene(1920, 1080, 256, detail=2.0))):
    sc = kz.Scene(make(), device=0)
    n = sc.width * sc.height * sc.sample_count
    sc.render(shadow_beside=1, pass_halves=1); sc.sync(); sc.render(shadow_beside=1, pass_halves=1); sc.sync()
    ref = sc.film()
    out = []
    for what, env, kw in (("one stream", "0", dict(shadow_beside=1, pass_halves=1)), ("halves", "0", dict(shadow_beside=1, pass_halves=2)), ("halves, staggered", "1", dict(shadow_beside=1, pass_halves=2)),
                          ("halves", "0", dict(shadow_beside=1, pass_halves=2)), ("halves, staggered", "1", dict(shadow_beside=1, pass_halves=2)), ("one stream", "0", dict(shadow_beside=1, pass_halves=1))):
        os.environ["KZ_HALF_STAGGER"] = env
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); sc.render(**kw); sc.sync(); ts.append(time.perf_counter() - t0)
        out.append("%s %.1f%s" % (what, n / min(ts) / 1e6, "" if np.array_equal(sc.film(), ref) else " FILM DIFFERS"))
    print("%-16s %s" % (name, " | ".join(out)), flush=True)
    sc.close()
