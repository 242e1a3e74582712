"""Development probe (round 6): kz_render_multi at BASELINE's C5 size (the 1 M-triangle scene at 3840x2160, sample indices [0, 64) of the 4096-spp pmj02bn table) with REAL host-thread
concurrency on one GPU - the development library's device aliasing (kz_debug_alias_devices) - against one replica: films compared bit for bit, wall time per device thread.
    python scripts/dev/alias_c5.py [spp]"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
lib = kz.abi.load_dev_library()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = dict(shadow_beside=1, pass_halves=1) if "--one-stream" in sys.argv else dict(shadow_beside=2, pass_halves=2) if "--halves-beside" in sys.argv else {}      # (--one-stream: no replica times its large passes; --halves-beside: every pass as halves with its shadow rays beside)
lib.kz_debug_alias_devices(8)
desc = kz.scenes.random_triangles(1000000, 3840, 2160, 4096, sampler="pmj02bn", seed=1)
sc = kz.Scene(desc, lib=lib)
out = {"workload": "C5: 1 000 028 triangles, 3840x2160, sample indices [0, %d) of the 4096-spp pmj02bn table; kz_render_multi over aliased replicas of ONE MI355X" % spp, "runs": []}
one = None
for n in (1, 2, 4):
    cap = int(160e9 / n)                                               # the aliases share one card: each replica's pass contexts capped
    for dealing in (0, 1):
        t0 = time.perf_counter()
        film, ms = sc.render_multi(list(range(n)), sample_begin=0, sample_end=spp, tile_dealing=dealing, max_state_bytes=cap, **modes)
        dt = time.perf_counter() - t0
        if one is None:
            one = film
        rec = {"aliases": n, "dealing": "dynamic" if dealing else "static", "wall_s": round(dt, 3), "device_ms": [round(float(m), 1) for m in ms],
               "equals_single_replica_film": bool(np.array_equal(film, one)), "Msamples_per_s": round(3840 * 2160 * spp / dt / 1e6, 1)}
        out["runs"].append(rec)
        print(json.dumps(rec), flush=True)
sc.render(0, spp, device=0, max_state_bytes=int(160e9))
out["kz_render_on_one_replica_equals"] = bool(np.array_equal(sc.film(), one))
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r06r_alias_c5"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06r_alias_c5", "alias_c5.json"), "w"), indent=1)
assert all(r["equals_single_replica_film"] for r in out["runs"]) and out["kz_render_on_one_replica_equals"]
