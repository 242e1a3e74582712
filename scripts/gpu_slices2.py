import sys, time, importlib, numpy as np
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024, sampler="independent", seed=0)
sc = kz.Scene(d, device=0)
for s0 in (0, 16, 32, 48, 64):
    sc.render(s0, s0+16, accumulate=True); sc.sync()
    print("independent", s0, "kernel ms %.1f" % sc.last_kernel_ms(), flush=True)
sc.close()
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
import os
for s0, n in ((0,4),(4,4),(8,4),(12,4),(32,4),(36,4),(0,1),(1,1),(2,1),(3,1),(32,1),(33,1),(0,64),(64,64),(128,64)):
    os.environ["KZ_PASS_ITEMS"] = str(1920*1080*n)
    sc.render(s0, s0+n, accumulate=True); sc.sync()
    print("pmj", s0, n, "kernel ms per 16spp-equivalent %.1f" % (sc.last_kernel_ms()*16/n), flush=True)
