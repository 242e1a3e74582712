"""Development probe: passes IN FLIGHT (two contexts) with the shadow rays of each pass in front of / beside its closest-hit rays, by pass size: where should the small-pass
rule end when passes already overlap each other?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
for name, make in (("C4-like 1920x1080x256", lambda: S.random_triangles(1000000, 1920, 1080, 256, sampler="independent")), ("q1 asset x256", lambda: S.load_npz(q1, overrides={"sampler": {"type": "independent", "sampleCount": 256, "seed": 0}})),
                   ("hero x256", lambda: S.hero_scene(1920, 1080, 256, detail=2.0))):
    sc = kz.Scene(make(), device=0)
    n = sc.width * sc.height * sc.sample_count
    sc.render(shadow_beside=1, pass_halves=1); sc.sync()
    ref = sc.film()
    for lg in (22, 24, 25, 26, 27):
        out = []
        for sb in (1, 2, 1, 2):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); sc.render(pass_items=1 << lg, passes_in_flight=2, shadow_beside=sb, pass_halves=1); sc.sync(); ts.append(time.perf_counter() - t0)
            out.append("%s %.1f" % ("front" if sb == 1 else "beside", n / min(ts) / 1e6))
        print("%-24s two passes of 2^%d in flight: %s  film equal %s" % (name, lg, " | ".join(out), np.array_equal(sc.film(), ref)), flush=True)
    sc.close()
