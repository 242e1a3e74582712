"""Shade kernel: sensitivity to resident blocks per CU (latency-bound or not?)."""
import sys, os, importlib
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
for sb in (1, 2, 3, 4, 6, 8):
    os.environ["KZ_TUNE_SHADE_BLOCKS"] = str(sb)
    sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
    print(sb, sc.last_kernel_ms(), sc.last_stage_ms(), flush=True)
