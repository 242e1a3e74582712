import sys, os, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests/golden')
import oracle as O
kz=O.kz
import make_fp_goldens as mk
gold=np.load('/root/repo/tests/golden/fp_goldens.npz')
rows=mk.kiss_rows()
s = kz.scenes.SceneDescription()
tri = (np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32))
for r in rows: s.add_mesh(*tri, bsdf=r)
s.camera.update(width=32, height=32)
sc = kz.Scene(s, device=0)
wi, wo, s3 = gold["bsdf_wi"], gold["bsdf_wo"], gold["bsdf_s"]
m = wi.shape[0]
acc = np.where(np.arange(m) % 7 == 3, 0.25, 0.0).astype(np.float32)
for r in range(len(rows)):
    ev, pd, sm = sc.bsdf_query(np.full(m, r, np.int32), wi, wo, acc, s3)
    g = gold["bsdf_sample"][r]
    rel = np.abs(sm[:,:3]-g[:,:3])/np.maximum(np.abs(g[:,:3]),1e-6)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print(r, "max rel", rel.max(), "at", i, sm[i[0]], g[i[0]], "wi", wi[i[0]], "s", s3[i[0]])
    # recompute eval/pdf at the GPU's sampled direction, with the oracle
    k=i[0]
    e = O.bsdf(rows[r], "eval", wi[k], sm[k,3:6], float(acc[k])); p = O.bsdf(rows[r], "pdf", wi[k], sm[k,3:6], float(acc[k]))
    e2 = O.bsdf(rows[r], "eval", wi[k], g[k,3:6], float(acc[k])); p2 = O.bsdf(rows[r], "pdf", wi[k], g[k,3:6], float(acc[k]))
    print("   oracle eval/pdf at gpu wo", e/p, e, p, " at cpu wo", e2/p2, e2, p2)
