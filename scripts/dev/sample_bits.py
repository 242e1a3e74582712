"""Share of per-sample radiances (kz_render_samples vs the oracle) and of thin-lens camera rays that are equal to the last bit, on fuzz scenes:
python scripts/dev/sample_bits.py [first_seed] [n]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
kz = importlib.import_module("nano-kazen_amd")
import oracle as O
from test_gpu_parity import _fuzz_scene
first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
tot = eq = 0
worst = []
for seed in range(first, first + n):
    d = _fuzz_scene(kz.scenes, seed)
    sc = kz.Scene(d, device=0); o = O.OracleScene(d)
    w, h, s = d.camera["width"], d.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(s), indexing="ij")
    pxy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32); idx = ii.ravel().astype(np.uint32)
    g = sc.render_samples(pxy, idx); c = o.render_samples(pxy, idx)
    same = (g.view(np.uint32) == c.view(np.uint32)).all(axis=1)
    rel = np.abs(g[:, 2:] - c[:, 2:]).max(axis=1) / (1e-6 + np.abs(c[:, 2:]).max(axis=1))
    tot += same.size; eq += int(same.sum())
    print("seed %d %s depth %d: %d of %d samples differ in some bit, max relative difference %.2e" % (seed, d.camera["type"], d.integrator["maxDepth"], int((~same).sum()), same.size, rel.max()), flush=True)
    if (~same).any():
        k = int(np.argmax(rel)); worst.append((seed, pxy[k].tolist(), int(idx[k]), g[k, 2:].tolist(), c[k, 2:].tolist()))
    sc.close()
print("bit-identical samples: %d of %d (%.5f)" % (eq, tot, eq / tot))
for w in worst[:10]: print(w)
