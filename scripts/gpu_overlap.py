"""Do a trace-heavy and a shade-heavy pass overlap when two half-passes run on two streams? Two uploads of the C4 scene, each
renders 8 of the 16 spp of a bench step; wall time of both vs one scene rendering all 16."""
import sys, os, importlib, time
import torch
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
A = kz.Scene(d, device=0)
B = kz.Scene(d, device=0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3

def seq():
    A.render(32, 48, stream=s1.cuda_stream)

def par():
    A.render(32, 40, stream=s1.cuda_stream)
    B.render(40, 48, stream=s2.cuda_stream)

def par_half_pixels():
    pass

for tb, sb in ((8, 6), (6, 3), (5, 2), (5, 1), (4, 2), (4, 1), (3, 1)):
    os.environ["KZ_TUNE_TRAV_BLOCKS"] = str(tb); os.environ["KZ_TUNE_SHADE_BLOCKS"] = str(sb)
    seq(); par()
    print("trav", tb, "shade", sb, "sequential %.2f ms   two streams %.2f ms" % (wall(seq), wall(par)), flush=True)
