"""Library-internal two-pass pipelining (KZ_DUAL_STREAM) with the caller on the default stream, as bench.py is."""
import sys, os, importlib, time
import torch
sys.path.insert(0, '/root/repo')
kz = importlib.import_module("nano-kazen_amd")
d = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
A = kz.Scene(d, device=0)
st = torch.cuda.current_stream().cuda_stream
def wall(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
for dual in ("0", "1", "0", "1"):
    os.environ["KZ_DUAL_STREAM"] = dual
    A.render(0, 32, stream=st)
    print("KZ_DUAL_STREAM", dual, "32 spp call: %.2f ms/pass   64 spp call: %.2f ms/pass" % (wall(lambda: A.render(0, 32, stream=st)) / 2, wall(lambda: A.render(0, 64, stream=st)) / 4),
          "kernel_ms", A.last_kernel_ms(), flush=True)
