"""Re-sweep of the persistent traversal's refill / postpone thresholds (trace kernel at 8 waves/SIMD)."""
import sys, os, importlib
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
def run(**env):
    for k, v in env.items(): os.environ[k] = str(v)
    sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
    st = sc.last_stage_ms()
    return "%.2f  closest %.2f shadow %.2f" % (sc.last_kernel_ms(), st["trace_closest"], st["trace_shadow"])
for refill in (32, 40, 48, 56, 60, 64):
    for post in (16, 28, 40):
        print("refill", refill, "postpone", post, run(KZ_TUNE_REFILL=refill, KZ_TUNE_POSTPONE=post), flush=True)
for batch in (64, 256, 512):
    print("batch", batch, run(KZ_TUNE_REFILL=40, KZ_TUNE_POSTPONE=28, KZ_TUNE_BATCH=batch), flush=True)
