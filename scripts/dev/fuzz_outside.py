"""The samples of one fuzz scene whose HIP radiance lies outside the oracle's tie bracket (tests/test_gpu_parity.py tie_bracket), with the BSDF of the
first hit: python scripts/dev/fuzz_outside.py <seed>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
kz = importlib.import_module("nano-kazen_amd")
import oracle as O
from test_gpu_parity import _fuzz_scene
seed = int(sys.argv[1])
d = _fuzz_scene(kz.scenes, seed)
sc = kz.Scene(d, device=0); ora = O.OracleScene(d)
w, h, n = d.camera["width"], d.camera["height"], sc.sample_count
yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(n), indexing="ij")
pxy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32); idx = ii.ravel().astype(np.uint32)
g = sc.render_samples(pxy, idx)[:, 2:5]; c = ora.render_samples(pxy, idx)[:, 2:5]
ora.set_tie_mode(+1); lo = ora.render_samples(pxy, idx)[:, 2:5]
ora.set_tie_mode(-1); hi = ora.render_samples(pxy, idx)[:, 2:5]
slack = 1e-3 * (1 + np.abs(hi))
out = ~((g >= lo - slack) & (g <= hi + slack)).all(axis=1)
print("integrator", d.integrator, "camera", {k: v for k, v in d.camera.items() if k != "toWorld"})
print("meshes:", [(i, (m["bsdf"] or {}).get("type")) for i, m in enumerate(d.meshes)])
print("outside:", int(out.sum()), "of", out.size, "; zero vs non-zero:", int(((g[out].max(axis=1) == 0) != (c[out].max(axis=1) == 0)).sum()))
for k in np.nonzero(out)[0][:int(sys.argv[2]) if len(sys.argv) > 2 else 1000]:
    print(pxy[k], idx[k], "hip", g[k], "oracle", c[k], "lo", lo[k], "hi", hi[k])
