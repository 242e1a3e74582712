"""One fuzz scene under single changes (camera, lights, integrator): share of samples whose HIP radiance differs from the literal oracle, and how many
camera rays are bit-identical: python scripts/dev/fuzz_variants.py <seed>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
kz = importlib.import_module("nano-kazen_amd")
import oracle as O
from test_gpu_parity import _fuzz_scene
seed = int(sys.argv[1])
def cmp(d, tag):
    sc = kz.Scene(d, device=0); o = O.OracleScene(d)
    w, h, n = d.camera["width"], d.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(n), indexing="ij")
    pxy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32); idx = ii.ravel().astype(np.uint32)
    g = sc.render_samples(pxy, idx)[:, 2:5]; c = o.render_samples(pxy, idx)[:, 2:5]
    off = (np.abs(g - c) > 1e-3 * (1 + np.abs(c))).any(axis=1)
    # camera rays of the first 4096 samples, bit for bit
    m = min(4096, idx.size)
    sxy = np.zeros((m, 2), np.float32); axy = np.zeros((m, 2), np.float32); ro = np.zeros((m, 6), np.float32)
    for k in range(m):
        st = o.sampler_stream(int(pxy[k, 0]), int(pxy[k, 1]), int(idx[k]), 4)
        sxy[k] = (pxy[k, 0] + st[0], pxy[k, 1] + st[1]); axy[k] = st[2:4]
        ro[k] = o.camera_ray(float(sxy[k, 0]), float(sxy[k, 1]), float(axy[k, 0]), float(axy[k, 1]))[0]
    rg = sc.camera_rays(sxy, axy)[:, :6]
    same = (rg.view(np.uint32) == ro.view(np.uint32)).all(axis=1)
    print("%-32s samples off %5d of %d (%.1e)   camera rays bit-equal %d of %d, max |d| %.1e" % (tag, int(off.sum()), off.size, off.mean(), int(same.sum()), m, np.abs(rg - ro).max()), flush=True)
    sc.close()
mods = [("as is", lambda x: None),
        ("aperture 0", lambda x: x.camera.update(apertureRadius=0.0)),
        ("aperture 1e-3", lambda x: x.camera.update(apertureRadius=1e-3)),
        ("maxDepth 1", lambda x: x.integrator.update(maxDepth=1)),
        ("maxDepth 2", lambda x: x.integrator.update(maxDepth=2)),
        ("all lights visible", lambda x: [m["light"].update(lightPrimaryVisibility=True) for m in x.meshes if m["light"]]),
        ("no background", lambda x: setattr(x, "background", None)),
        ("all bsdfs diffuse", lambda x: [m.update(bsdf=kz.scenes.diffuse((0.5, 0.5, 0.5))) for m in x.meshes if not m["light"]])]
for name, mod in mods:
    e = _fuzz_scene(kz.scenes, seed)
    mod(e)
    cmp(e, name)
