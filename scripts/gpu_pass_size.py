"""Pass size (KZ_PASS_ITEMS) vs throughput with two passes in flight: C4, one 128-spp call."""
import sys, os, importlib, time
import torch
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
for items in (1 << 27, 1 << 28):
    os.environ["KZ_PASS_ITEMS"] = str(items)
    sc.render(0, 256); sc.sync()
    t = time.perf_counter(); sc.render(256, 512); sc.sync(); dt = time.perf_counter() - t
    print("pass items 2^%d: %.1f Msamples/s" % (items.bit_length() - 1, 1920 * 1080 * 256 / dt / 1e6), flush=True)
