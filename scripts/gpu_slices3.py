import sys, os, importlib
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
for wide in (1,):
    os.environ["KZ_TUNE_WIDE"] = str(wide)
    for s0 in (0, 16, 32, 48, 64, 80, 96, 112, 512, 1008):
        sc.render(s0, s0+16); sc.sync()
        print("wide", wide, "slice", s0, "pass ms %.1f" % sc.last_kernel_ms(), flush=True)
