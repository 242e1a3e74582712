"""Small frames: where does the time of a 1 M-sample render go? (C1: Cornell box 256x256x16)"""
import sys, os, importlib, time
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
for name, d in (("C1", kz.scenes.cornell_box(256, 256, 16)), ("C2", kz.scenes.sphere_env(512, 512, 64))):
    sc = kz.Scene(d, device=0)
    sc.render(); sc.sync()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); sc.render(); sc.sync(); best = min(best, time.perf_counter() - t)
    print(name, "wall %.3f ms" % (best * 1e3), "kernel_ms %.3f" % sc.last_kernel_ms(), sc.last_stage_ms(), flush=True)
