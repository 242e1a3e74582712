#!/usr/bin/env python3
"""Mints nano-kazen_amd/data/bluenoise_vc_48x128.npz: the 48 void-and-cluster dither arrays scenes.make_pmj02bn_tables() hands to the PMJ02BN sampler
in place of the reference's missing BlueNoiseTextures blob (bluenoise.h:8-11). Deterministic (seeded per texture); ~1-2 s per texture per core."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if __name__ == "__main__":
    S = importlib.import_module("nano-kazen_amd.scenes")
    t0 = time.time()
    bn = S.mint_blue_noise_textures()
    path = os.path.join(ROOT, "nano-kazen_amd", "data", "bluenoise_vc_48x128.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, bn=bn)
    print("%s: %s %s in %.1f s, %d bytes" % (path, bn.shape, bn.dtype, time.time() - t0, os.path.getsize(path)))
