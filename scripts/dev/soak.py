"""Development probe: one resident scene, a few thousand renders with randomly drawn options - pass mode, pass size, passes in flight, sample ranges accumulated in pieces, tile
subsets, caller streams - each compared with the film of the plain call for the same samples and tiles. A missing wait between streams shows up here, rarely, as a wrong film."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip = C.CDLL("libamdhip64.so")
streams = [None]
for _ in range(2):
    s = C.c_void_p(); assert hip.hipStreamCreate(C.byref(s)) == 0; streams.append(s)
q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
scenes = [kz.scenes.load_npz(q1, overrides={"camera": {"width": 320, "height": 256}, "sampler": {"type": "independent", "sampleCount": 32, "seed": 0}}),
          kz.scenes.glass_scene(192, 160, 32), kz.scenes.random_triangles(50000, 256, 192, 32, sampler="pmj02bn")]
bad = []
t0 = time.time()
for si, d in enumerate(scenes):
    sc = kz.Scene(d, device=0)
    npx = sc.width * sc.height
    tiles_all = kz.shard.deal_tiles(sc.width, sc.height, 1, 0, 64)
    refs = {}
    def ref(s0, s1, tkey):
        k = (s0, s1, tkey)
        if k not in refs:
            sc.render(s0, s1, tiles=None if tkey is None else [tiles_all[i] for i in tkey], shadow_beside=1, pass_halves=1); refs[k] = sc.film()
        return refs[k]
    for it in range(n // len(scenes)):
        s0 = int(rng.integers(0, 24)); s1 = int(rng.integers(s0 + 1, 33))
        tkey = None if rng.random() < 0.6 else tuple(sorted(rng.choice(len(tiles_all), size=int(rng.integers(1, len(tiles_all))), replace=False).tolist()))
        want = ref(s0, s1, tkey)
        kw = dict(shadow_beside=int(rng.integers(0, 3)), pass_halves=int(rng.integers(0, 3)))
        if rng.random() < 0.5:
            kw.update(pass_items=int(npx * rng.integers(1, 9)), passes_in_flight=int(rng.integers(1, 4)))
        tl = None if tkey is None else [tiles_all[i] for i in tkey]
        cuts = sorted(set([s0, s1] + [int(x) for x in rng.integers(s0, s1 + 1, size=int(rng.integers(0, 3)))]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            sc.render(a, b, tiles=tl, accumulate=(a != s0), stream=streams[int(rng.integers(0, 3))], **kw)
        if not np.array_equal(sc.film(), want):
            bad.append((si, it, s0, s1, tkey is not None, kw, cuts))
            print("MISMATCH", bad[-1], flush=True)
    sc.close()
    print("scene %d: %d rounds, %.0f s so far, mismatches %d" % (si, n // len(scenes), time.time() - t0, len(bad)), flush=True)
sys.exit(1 if bad else 0)
