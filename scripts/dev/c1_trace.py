"""Development probe: the C1-sized job (BASELINE configs[0]: the q1 asset at 256x256x16) call by call - wall time per call, for a kernel trace of where a millisecond-sized job goes."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), overrides={"camera": {"width": 256, "height": 256}, "sampler": {"type": "independent", "sampleCount": 16, "seed": 0}})
sc = kz.Scene(d, device=0)
for i in range(6):
    t0 = time.perf_counter(); sc.render(); sc.sync(); print("call %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
print("stages of the last pass:", sc.last_stage_ms(), flush=True)
if os.environ.get("KZ_C1_MEGA"):
    for i in range(4):
        t0 = time.perf_counter(); sc.render(pipeline=1); sc.sync(); print("megakernel call %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
