import sys, time, importlib, numpy as np
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
for rep in range(2):
    for s0 in (0, 16, 32, 48, 64, 80, 96, 512, 1008, 32, 0):
        sc.render(s0, s0+16, accumulate=True); sc.sync()
        print(rep, s0, "kernel ms %.1f" % sc.last_kernel_ms(), flush=True)
sc.set_stats(True)
for s0 in (0, 32):
    sc.stats(reset=True); sc.render(s0, s0+16, accumulate=True); st = sc.stats()
    print(s0, {k: v/st['samples'] for k,v in st.items()})
