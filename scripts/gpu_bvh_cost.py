"""Builder cost-model sweep against the current traversal kernels (C4): SAH node cost and leaf size."""
import sys, os, importlib
os.environ["KZ_DUAL_STREAM"] = "0"
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
for leaf in (4, 2):
    for cost in (0.35, 0.5, 0.7, 1.0, 1.4, 2.0):
        os.environ["KZ_BVH_NODE_COST"] = str(cost); os.environ["KZ_BVH_MAX_LEAF"] = str(leaf)
        sc = kz.Scene(d, device=0)
        sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
        st = sc.last_stage_ms(); b = sc.bvh_info()
        print("leaf", leaf, "cost", cost, "nodes", b["nNodes"], "%.2f  closest %.2f shadow %.2f" % (sc.last_kernel_ms(), st["trace_closest"], st["trace_shadow"]), flush=True)
        sc.close()
