#!/usr/bin/env python3
"""What the dither textures do to a picture: C1's Cornell box through the pmj02bn sampler at a few spp, once with the white-noise stand-in of
rounds 1-4 and once with the void-and-cluster textures; the error against a 4096-spp render, raw and after a Gaussian blur (what the eye does): blue-noise
dithering leaves the RMS alone and moves the error to high frequencies."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes


def image(desc):
    sc = kz.Scene(desc); sc.upload(0); sc.render(); sc.sync()
    return sc.rgb(sc.film()).astype(np.float64)


def blur(img, sigma):
    n = int(3 * sigma) + 1
    k = np.exp(-np.arange(-n, n + 1) ** 2 / (2.0 * sigma * sigma)); k /= k.sum()
    for ax in (0, 1):
        img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, img)
    return img


if __name__ == "__main__":
    W = 256
    ref = image(S.cornell_box(W, W, 4096, sampler="pmj02bn", seed=7))
    out = {}
    for spp in (1, 4, 16):
        row = {}
        for name in ("white", "void_and_cluster"):
            d = S.cornell_box(W, W, spp, sampler="pmj02bn", seed=1)
            d.tables = S.make_pmj02bn_tables(dither=name)
            err = np.clip(image(d), 0, 2) - np.clip(ref, 0, 2)
            row[name] = {"rms": float(np.sqrt((err ** 2).mean())), "rms_blur_1.5px": float(np.sqrt((blur(err, 1.5) ** 2).mean())), "rms_blur_3px": float(np.sqrt((blur(err, 3.0) ** 2).mean()))}
        out["spp_%d" % spp] = row
    print(json.dumps(out, indent=1))
