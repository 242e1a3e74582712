"""Development probe: host threads rendering DIFFERENT scenes on the same device at the same time (each its own replica; the per-device context pool, the thread-local error
state and the device's streams are what they share), creating and destroying scenes as they go: every film must be its scene's serial film."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
makers = [lambda: S.cornell_box(200, 160, 16, sampler="pmj02bn"), lambda: S.glass_scene(128, 128, 16), lambda: S.random_triangles(20000, 192, 128, 8, sampler="independent"),
          lambda: S.sphere_env(128, 128, 16)]
refs = []
for mk in makers:
    sc = kz.Scene(mk(), device=0); sc.render(shadow_beside=1, pass_halves=1); refs.append(sc.film()); sc.close()
bad = []
def work(k, rounds):
    try:
        for r in range(rounds):
            sc = kz.Scene(makers[k](), device=0)
            for j in range(4):
                sc.render(shadow_beside=(r + j) % 3, pass_halves=(r + 2 * j) % 3)
                if not np.array_equal(sc.film(), refs[k]):
                    bad.append((k, r, j))
            if r % 3 == 0:
                npx = sc.width * sc.height
                sc.render(pass_items=npx * 2, passes_in_flight=2)
                if not np.array_equal(sc.film(), refs[k]):
                    bad.append((k, r, "in flight"))
            sc.close()
    except Exception as e:                                    # noqa: BLE001
        bad.append((k, repr(e)))
t0 = time.time()
ths = [threading.Thread(target=work, args=(k, int(sys.argv[1]) if len(sys.argv) > 1 else 40)) for k in range(len(makers))]
for t in ths: t.start()
for t in ths: t.join()
print("%d threads, %.1f s, mismatches / errors: %s" % (len(ths), time.time() - t0, bad[:10]))
sys.exit(1 if bad else 0)
