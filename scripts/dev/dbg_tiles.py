import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
desc = kz.scenes.cornell_box(200, 136, 4, sampler="pmj02bn")
sc = kz.Scene(desc, device=0)
sc.render()
whole = sc.film()
b = sc.border
total = sc.empty_film()
for part in range(3):
    tiles = kz.shard.deal_tiles(200, 136, 3, part, 32)
    f = sc.render_tiles(tiles, device=0)
    packed = sc.film_tiles(tiles, device=0)
    # reference packing on the host from the downloaded film
    ref = np.concatenate([f[t[1]:t[1] + t[3] + 2 * b, t[0]:t[0] + t[2] + 2 * b].ravel() for t in tiles])
    m = sc.merge_tiles(sc.empty_film(), tiles, packed)
    print("part", part, "tiles", len(tiles), "merge of own rects == own film:", np.array_equal(m, f), float(np.abs(m - f).max()))
    sc.merge_tiles(total, tiles, packed)
print("sum of packed rects vs whole: allclose", np.allclose(total, whole, rtol=1e-5, atol=1e-6), float(np.abs(total - whole).max()))
