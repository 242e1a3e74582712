"""Per-sample radiance, HIP vs oracle, bit for bit on the named scenes of nano-kazen_amd/scenes.py and on the reference's own asset scene (small frames):
python scripts/dev/sample_bits_scenes.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
kz = importlib.import_module("nano-kazen_amd")
import oracle as O
S = kz.scenes
cases = {"cornell": lambda: S.cornell_box(48, 48, 8), "cornell pmj": lambda: S.cornell_box(48, 48, 8, sampler="pmj02bn", seed=1), "hero": lambda: S.hero_scene(64, 36, 8, detail=0.3),
         "sphere_env": lambda: S.sphere_env(48, 48, 8), "materials": lambda: S.materials_scene(64, 36, 8), "textured": lambda: S.textured_scene(64, 36, 8),
         "random_triangles 20k": lambda: S.random_triangles(20000, 64, 36, 8),
         "q1 asset": lambda: S.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), {"camera": {"width": 64, "height": 36}, "sampler": {"sampleCount": 8}})}
for name, mk in cases.items():
    d = mk()
    sc = kz.Scene(d, device=0); o = O.OracleScene(d)
    w, h, s = d.camera["width"], d.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(s), indexing="ij")
    pxy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32); idx = ii.ravel().astype(np.uint32)
    g = sc.render_samples(pxy, idx); c = o.render_samples(pxy, idx)
    same = ((g.view(np.uint32) == c.view(np.uint32)) | (np.isnan(g) & np.isnan(c))).all(axis=1)
    rel = np.abs(g[:, 2:] - c[:, 2:]).max(axis=1) / (1e-6 + np.abs(c[:, 2:]).max(axis=1))
    print("%-22s %d tris, depth %d: %d of %d samples differ in some bit (max relative difference %.2e)" % (name, d.n_tris(), d.integrator["maxDepth"], int((~same).sum()), same.size, rel.max()), flush=True)
    sc.close()
