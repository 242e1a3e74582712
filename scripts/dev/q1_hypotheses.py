"""Single-variable experiments on the reference's scene file default_m0_r0.5 at full size through the HIP path (data changes only), for the comparison with the published
picture in the build container (scripts/dev/q1_hyp_compare.py):   python scripts/dev/q1_hypotheses.py name=key:value[,key:value...] ...
keys: maxDepth, spp, f0 / f1 / f2 (factors on the three light intensities)"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
OUT = os.path.join(ROOT, "gpurun_out", "q1_hyp"); os.makedirs(OUT, exist_ok=True)
res = {}
for spec in sys.argv[1:]:
    name, _, kvs = spec.partition("=")
    kv = dict(p.split(":") for p in kvs.split(",") if p)
    d = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), {"sampler": {"sampleCount": int(kv.get("spp", 1024))}})
    if "maxDepth" in kv:
        d.integrator["maxDepth"] = int(kv["maxDepth"])
    for i, l in enumerate([m["light"] for m in d.meshes if m["light"]]):
        l["intensity"] *= float(kv.get("f%d" % i, 1.0))
    sc = kz.Scene(d, device=0)
    t0 = time.perf_counter(); sc.render(); sc.sync(); dt = time.perf_counter() - t0
    kz.output.save_png(os.path.join(OUT, name), sc.srgb8())
    res[name] = {"spec": kv, "render_s": round(dt, 2), "mean_linear_rgb": float(sc.rgb().mean())}
    print(name, res[name], flush=True)
    sc.close()
json.dump(res, open(os.path.join(OUT, "hyp.json"), "w"), indent=1)
