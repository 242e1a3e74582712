"""Development probe: the EXT scenes (and C3) at 1920x1080x256 as ONE pass (the default), with the shadow rays in front / beside, and as two passes in flight of half the size."""
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
    sc.render(shadow_beside=1); sc.sync(); sc.render(shadow_beside=1); sc.sync()
    ref = sc.film()
    for what, kw in (("one pass, shadow rays in front", dict(shadow_beside=1)), ("one pass, shadow rays beside", dict(shadow_beside=2)),
                     ("two passes of 2^28 in flight", dict(pass_items=1 << 28, passes_in_flight=2)), ("two passes of 2^28 in flight, beside", dict(pass_items=1 << 28, passes_in_flight=2, shadow_beside=2)),
                     ("four passes of 2^27, two in flight", dict(pass_items=1 << 27, passes_in_flight=2)),
                     ("one pass as halves (passHalves = 2)", dict(pass_halves=2, shadow_beside=1)), ("one pass as halves, shadow rays beside", dict(pass_halves=2, shadow_beside=2))):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); sc.render(**kw); sc.sync(); ts.append(time.perf_counter() - t0)
        print("%-16s %-40s %8.1f Msamples/s  (%.4f s)  film equal: %s  stages %s" % (name, what, n / min(ts) / 1e6, min(ts), np.array_equal(sc.film(), ref), sc.last_stage_ms() if "one pass" in what else ""), flush=True)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); sc.render(); sc.sync(); ts.append((round(time.perf_counter() - t0, 4), sc.last_pass_info()["shadowBeside"]))
    print("%-16s defaults, eight calls (seconds, how the pass ran: 0 one stream / 1 shadow rays beside / 2 halves): %s -> %.1f Msamples/s  film equal: %s" % (name, ts, n / ts[-1][0] / 1e6, np.array_equal(sc.film(), ref)), flush=True)
    sc.close()
