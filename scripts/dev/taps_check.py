"""Development check: the two-lane tap-sum kernel (default) against the one-lane kernel of round 2 (KzTuning.filmGather = 3): the same film, bit for bit."""
import importlib, sys, numpy as np, time
sys.path.insert(0, "/root/repo")
kz = importlib.import_module("nano-kazen_amd")
for name, desc in (("cornell_gauss", kz.scenes.cornell_box(200, 150, 32)), ("hero", kz.scenes.hero_scene(640, 360, 64, detail=0.5))):
    sc = kz.Scene(desc, device=0)
    sc.render(); a = sc.film()
    sc.render(tune={"filmGather": 3}); b = sc.film()
    print(name, "taps2 == taps bitwise:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max()))
    sc.close()
for filt in ("tent", "box", "mitchell"):
    d = kz.scenes.cornell_box(96, 80, 16); d.camera["rfilter"] = {"type": filt}
    sc = kz.Scene(d, device=0)
    sc.render(); a = sc.film()
    sc.render(tune={"filmGather": 3}); b = sc.film()
    print(filt, np.array_equal(a.view(np.uint32), b.view(np.uint32)))
