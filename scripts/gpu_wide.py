"""A/B of the traversal tree width (KZ_TUNE_WIDE = 4 / 8) on the C4 workload: stage times and film equality."""
import sys, os, importlib
import numpy as np
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
films = {}
for w in (1, 1):
    os.environ["KZ_TUNE_WIDE"] = str(w)
    sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
    print(w, sc.last_kernel_ms(), sc.last_stage_ms(), flush=True)
    films[w] = sc.film()
np.save("gpurun_out/film_new.npy", films[1][::8, ::8])
