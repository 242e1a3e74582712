import sys, os, importlib
sys.path.insert(0,'/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
for leaf in (4,):
    os.environ["KZ_BVH_MAX_LEAF"] = str(leaf)
    sc = kz.Scene(d, device=0)
    sc.render(32,48); sc.sync(); sc.render(32,48); sc.sync()
    ms = sc.last_kernel_ms()
    sc.set_stats(True); sc.stats(reset=True); sc.render(32,48); st = sc.stats()
    print("leaf", leaf, sc.bvh_info()['nNodes'], "depth", sc.bvh_info()['maxDepth'], "pass ms %.1f" % ms, "nodes/sample %.1f tris/sample %.1f" % (st['nodeVisits']/st['samples'], st['triTests']/st['samples']), flush=True)
    sc.close()
