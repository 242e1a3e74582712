"""Which BSDF routines are BIT-identical between the HIP path and the oracle? Share of random queries whose sampled direction / weight / eval / pdf agree
to the last bit, per plugin row. The parity tests bound these differences by tolerances; a deep path through small triangles amplifies every last-bit
difference, so this is the list to shorten: python scripts/dev/bsdf_bits.py [n]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
kz = importlib.import_module("nano-kazen_amd")
import oracle as O
S = kz.scenes
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rows = {"diffuse": S.diffuse((0.5, 0.6, 0.7)), "kazenstandard": S.kazenstandard((0.8, 0.5, 0.3), 0.4, 0.5, 0.3), "ggx": S.ggx((0.9, 0.6, 0.3), 0.3, 0.2),
        "roughconductor": S.roughconductor(0.3, "Au"), "roughplastic": S.roughplastic(0.3, kd=(0.2, 0.4, 0.7)), "roughdielectric": S.roughdielectric(0.4),
        "dielectric": S.dielectric(), "mirror": S.mirror()}
s = S.SceneDescription()
for r in rows.values():
    s.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=r)
s.camera.update(width=32, height=32)
sc = kz.Scene(s, device=0)
ora = O.OracleScene(s)
rng = np.random.default_rng(11)
wi = rng.normal(size=(m, 3)).astype(np.float32); wi[:, 2] = np.abs(wi[:, 2]) + 0.02; wi /= np.linalg.norm(wi, axis=1, keepdims=True)
wo = rng.normal(size=(m, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.02; wo /= np.linalg.norm(wo, axis=1, keepdims=True)
s3 = rng.random((m, 3)).astype(np.float32)
acc = np.zeros(m, np.float32)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
print("%-16s %10s %10s %10s %10s   (share of %d queries equal to the last bit)" % ("row", "sample wo", "weight", "eval", "pdf", m))
for r, (name, row) in enumerate(rows.items()):
    ev, pd, sm = sc.bsdf_query(np.full(m, r, np.int32), wi, wo, acc, s3)
    e = np.zeros((m, 3), np.float32); p = np.zeros(m, np.float32); so = np.zeros((m, 8), np.float32)
    for k in range(m):
        e[k] = ora.bsdf(r, "eval", wi[k], wo[k]); p[k] = ora.bsdf(r, "pdf", wi[k], wo[k])
        so[k] = ora.bsdf(r, "sample", wi[k], None, 0.0, float(s3[k, 0]), (float(s3[k, 1]), float(s3[k, 2])))
    ok = (so[:, 6] > 0) & (sm[:, 6] > 0)
    f = lambda a, b: float((bits(a) == bits(b)).reshape(len(a), -1).all(axis=1).mean()) if len(a) else float("nan")
    print("%-16s %10.4f %10.4f %10.4f %10.4f" % (name, f(sm[ok, 3:6], so[ok, 3:6]), f(sm[ok, :3], so[ok, :3]), f(ev, e), f(pd, p)), flush=True)
