"""How many full-size passes in flight pay off? k uploads of the C4 scene on k streams (KZ_DUAL_STREAM off inside each)."""
import sys, os, importlib, time
import torch
os.environ["KZ_DUAL_STREAM"] = "0"
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
scs = [kz.Scene(d, device=0) for _ in range(4)]
sts = [torch.cuda.Stream(priority=p) for p in (0, -1, 0, -1)]
def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
def par(k, n):
    for i in range(n):
        scs[i % k].render(16 * i, 16 * i + 16, stream=sts[i % k].cuda_stream, accumulate=True)
for k in (1, 2, 3, 4):
    par(k, 12)
    print("in flight", k, "%.2f ms/pass" % (wall(lambda: par(k, 12)) / 12), flush=True)
