"""Two FULL-size passes (16 spp each) of the C4 frame: one after the other on one stream vs side by side on two streams
(two uploads of the scene = two sets of path-state buffers). Film accumulation is per scene here, so no ordering is needed."""
import sys, os, importlib, time
import torch
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
A = kz.Scene(d, device=0)
B = kz.Scene(d, device=0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def wall(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3

def seq(n):
    for k in range(n):
        A.render(16 * k, 16 * k + 16, stream=s1.cuda_stream, accumulate=True)

def par(n):
    for k in range(n):
        (A if k % 2 == 0 else B).render(16 * k, 16 * k + 16, stream=(s1 if k % 2 == 0 else s2).cuda_stream, accumulate=True)

seq(2); par(2)
for n in (2,):
    print("passes", n, "one stream %.2f ms/pass   two streams %.2f ms/pass" % (wall(lambda: seq(n)) / n, wall(lambda: par(n)) / n), flush=True)
for tb in ():
    os.environ["KZ_TUNE_TRAV_BLOCKS"] = str(tb)
    print("trav blocks", tb, "one stream %.2f ms/pass   two streams %.2f ms/pass" % (wall(lambda: seq(4)) / 4, wall(lambda: par(4)) / 4), flush=True)
# the same through the library's own two-pass pipelining: one scene, one call of 32 / 64 spp
for dual in ("0", "1"):
    os.environ["KZ_DUAL_STREAM"] = dual
    A.render(0, 32, stream=s1.cuda_stream)
    print("KZ_DUAL_STREAM", dual, "32 spp call: %.2f ms/pass   64 spp call: %.2f ms/pass" % (wall(lambda: A.render(0, 32, stream=s1.cuda_stream)) / 2, wall(lambda: A.render(0, 64, stream=s1.cuda_stream)) / 4),
          "kernel_ms", A.last_kernel_ms(), flush=True)
