"""Development probe: a render while something else holds most of the card (a torch tensor of total - KEEP GB in the same process): the pass contexts take what is left
minus the reserve for the HIP runtime, the job runs in smaller passes, the film is the same bits.   python scripts/dev/mem_pressure.py [keep_gb]"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
keep = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
d = kz.scenes.random_triangles(200000, 1920, 1080, 64, sampler="independent")
sc = kz.Scene(d, device=0)
sc.render(); sc.sync()
ref = sc.film(); info0 = sc.last_pass_info()
sc.close(); kz.abi.load_library().kz_device_trim(0)          # (the pooled contexts go back to the driver: what is held below is really gone)
free, total = torch.cuda.mem_get_info(0)
print("free %.1f GB of %.1f before the squeeze; unpressured job: %d passes of %d items" % (free / 1e9, total / 1e9, info0["passes"], info0["largestPassItems"]), flush=True)
hold = torch.empty(int(free - keep * 1e9), dtype=torch.uint8, device="cuda:0")
free2, _ = torch.cuda.mem_get_info(0)
print("holding %.1f GB, free now %.1f GB" % (hold.numel() / 1e9, free2 / 1e9), flush=True)
sc = kz.Scene(d, device=0)
for i in range(3):
    t0 = time.perf_counter(); sc.render(); sc.sync(); dt = time.perf_counter() - t0
    info = sc.last_pass_info()
    f3, _ = torch.cuda.mem_get_info(0)
    print("call %d: %.3f s, %d passes, largest %d items, state %.1f GB, free after %.2f GB, film equal: %s, note: %r" % (i, dt, info["passes"], info["largestPassItems"], info["stateBytes"] / 1e9, f3 / 1e9, np.array_equal(sc.film(), ref), sc.last_grow_note()), flush=True)
sc.close()
