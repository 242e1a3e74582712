import sys, time, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import oracle as O
kz = O.kz
S = kz.scenes
cases = [("cornell", S.cornell_box(128,128,16)), ("cornell_pmj", S.cornell_box(96,96,16,sampler="pmj02bn",seed=1)),
         ("sphere", S.sphere_env(128,128,8)), ("hero", S.hero_scene(160,90,8,detail=0.3)), ("soup", S.random_triangles(100000,160,90,16))]
vis = S.cornell_box(64,64,8)
for m in vis.meshes:
    if m["light"]: m["light"]["lightPrimaryVisibility"] = True
cases.append(("cornell_visible_light", vis))
st2 = S.cornell_box(64,64,16)
st2.add_mesh(*S._vfnuv(S.quad((-0.4, 0.5, -0.4), (-0.4, 0.5, 0.4), (0.4, 0.5, 0.4), (0.4, 0.5, -0.4), flip=True)), bsdf=S.diffuse((0,0,0)), light=S.area((1,0.5,0.2), 6.0, False))
st2.add_mesh(*S._vfnuv(S.quad((-0.3, 0.75, -0.3), (-0.3, 0.75, 0.3), (0.3, 0.75, 0.3), (0.3, 0.75, -0.3), flip=True)), bsdf=S.diffuse((0,0,0)), light=S.area((0.2,0.5,1), 6.0, True))
cases.append(("stacked_lights", st2))
for name, desc in cases:
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render(pipeline=1); f1 = sc.film(); st1 = sc.stats(reset=True)
    sc.render(pipeline=2); f2 = sc.film(); st2 = sc.stats(reset=True)
    print(name, "films identical:", np.array_equal(f1, f2), "maxdiff", np.abs(f1-f2).max(), flush=True)
    print("   mega", st1); print("   wave", st2, flush=True)
d = S.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
import os
for wide, ls in ((1,12),(1,16)):
    os.environ["KZ_TUNE_WIDE"]=str(wide); os.environ["KZ_TUNE_LDS_STACK"]=str(ls)
    sc.render(32, 48, pipeline=2); sc.sync(); sc.render(32, 48, pipeline=2); sc.sync()
    print("C4 wavefront wide", wide, "ldsStack", ls, "pass ms %.1f" % sc.last_kernel_ms(), flush=True)
for mixed in (0, 1):
    os.environ["KZ_TUNE_MIXED"]=str(mixed)
    sc.render(32, 48, pipeline=2); sc.sync(); sc.render(32, 48, pipeline=2); sc.sync()
    print("C4 mixed", mixed, "pass ms %.2f" % sc.last_kernel_ms(), flush=True)
os.environ["KZ_TUNE_WIDE"]="1"; os.environ["KZ_TUNE_LDS_STACK"]="12"
sc.render(32,48,pipeline=1); a = sc.film(); sc.render(32,48,pipeline=2); b = sc.film()
print("C4 identical", np.array_equal(a,b), np.abs(a-b).max())
