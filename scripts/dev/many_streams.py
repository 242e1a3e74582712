"""Development probe: does the overlap of the side streams survive in a process that owns many other streams (HIP maps streams onto a few hardware queues)? C1 and a 2^24-item
q1 job, shadow rays in front / beside, before and after 16 other streams have been created and used."""
import ctypes as C, importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = None
q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
def job(w, h, spp): return kz.scenes.load_npz(q1, overrides={"camera": {"width": w, "height": h}, "sampler": {"type": "independent", "sampleCount": spp, "seed": 0}})
def best(sc, **kw):
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); sc.render(**kw); sc.sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts[2:])
if "--others-first" in sys.argv:                      # (the library's side streams are then created AFTER the process's other sixteen)
    early = [torch.cuda.Stream(device=0) for _ in range(16)]
    y = torch.zeros(1 << 20, device="cuda:0")
    for s in early:
        with torch.cuda.stream(s):
            y.add_(1.0)
    torch.cuda.synchronize()
scs = [("C1 256x256x16", kz.Scene(job(256, 256, 16), device=0, lib=lib)), ("q1 512x512x64", kz.Scene(job(512, 512, 64), device=0, lib=lib)), ("materials 960x540x64 halves", kz.Scene(kz.scenes.materials_scene(960, 540, 64), device=0, lib=lib))]
def table(tag):
    for name, sc in scs:
        print("%-28s %-34s one stream %7.3f ms  beside %7.3f ms  halves %7.3f ms" % (tag, name, best(sc, shadow_beside=1, pass_halves=1), best(sc, shadow_beside=2, pass_halves=1), best(sc, shadow_beside=1, pass_halves=2)), flush=True)
table("fresh process")
others = [torch.cuda.Stream(device=0) for _ in range(16)]
x = torch.zeros(1 << 20, device="cuda:0")
for s in others:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
table("16 other streams in use")
for rep in range(3):
    for s in others:
        with torch.cuda.stream(s):
            x.add_(1.0)
table("... and busy a moment ago")
