"""Development probe: one job with KzRenderOpts::shadowBeside = $KZ_SHADOW_BESIDE (default 0), three calls (for a kernel trace / counters of the last one)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
what, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if what == "q1":
    d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), overrides={"camera": {"width": w, "height": h}, "sampler": {"type": "independent", "sampleCount": spp, "seed": 0}})
elif what == "hero":
    d = kz.scenes.hero_scene(w, h, spp)
else:
    d = kz.scenes.random_triangles(1000000, w, h, spp, sampler="independent")
sc = kz.Scene(d, device=0)
sb = int(os.environ.get("KZ_SHADOW_BESIDE", "0"))
for i in range(3):
    t0 = time.perf_counter(); sc.render(shadow_beside=sb); sc.sync(); print("call %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
print("stages of the last pass:", sc.last_stage_ms(), "maxDepth", d.integrator["maxDepth"], flush=True)
