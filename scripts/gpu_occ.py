"""Is the traversal latency-bound or throughput-bound? Sweep the resident blocks per CU of the trace kernel and print stage times."""
import sys, os, importlib
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
sc = kz.Scene(d, device=0)
for tb in (6, 7, 8):
    os.environ["KZ_TUNE_TRAV_BLOCKS"] = str(tb)
    sc.render(32, 48); sc.sync(); sc.render(48, 64); sc.sync()
    print(tb, sc.last_kernel_ms(), sc.last_stage_ms(), flush=True)
