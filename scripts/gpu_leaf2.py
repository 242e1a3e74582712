import sys, os, importlib
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
scenes = {"C4": kz.scenes.random_triangles(1000000, 1920, 1080, 1024), "hero": kz.scenes.hero_scene(1920, 1080, 16, detail=2.0), "cornell": kz.scenes.cornell_box(1920,1080,16)}
for cost in (0.2, 0.5, 0.7, 1.0, 2.0):
    os.environ["KZ_BVH_NODE_COST"] = str(cost)
    for name, d in scenes.items():
        sc = kz.Scene(d, device=0)
        sc.render(0,16); sc.sync(); sc.render(0,16); sc.sync()
        print("cost", cost, name, "nodes", sc.bvh_info()['nNodes'], "tris", sc.bvh_info()['nTris'], "pass ms %.2f" % sc.last_kernel_ms(), flush=True)
        sc.close()
