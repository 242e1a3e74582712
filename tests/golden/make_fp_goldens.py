#!/usr/bin/env python3
"""Mints tests/golden/fp_goldens.npz: fp32 function tables of the CPU oracle (warp, GGX/kiss eval/pdf/sample,
light sampling, camera rays, filter table, sampler streams, per-sample radiance and two tiny golden images).

The reference ships no fp32 fixtures for this path and cannot be built here, so these vectors
pin the ORACLE (regressions, and the GPU against committed numbers on a box without the oracle source), not
the reference: their parity status is "unpinned (text-only restatement)". Inputs are seeded; rerun with
    python tests/golden/make_fp_goldens.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

kz = O.kz
S = kz.scenes


def kiss_rows():
    return [S.kazenstandard((0.75, 0.75, 0.75), 0.5, 0.0),
            S.kazenstandard((0.9, 0.6, 0.2), 0.3, 1.0),
            S.kazenstandard((0.2, 0.3, 0.8), 0.7, 0.0, sheen=1.0, sheenTint=0.5, specularTint=1.0),
            S.kazenstandard((0.8, 0.3, 0.3), 0.25, 0.0, clearcoat=1.0, clearcoatRoughness=0.1, anisotropy=0.5, specular=1.0),
            S.kazenstandard((0.0, 0.0, 0.0), 0.0, 0.5, clearcoat=0.5),
            S.diffuse((0.5, 0.25, 0.125))]


def main():
    rng = np.random.default_rng(20221)
    out = {}
    L = O.lib()
    # warp
    u = rng.random((256, 2)).astype(np.float32)
    u[0] = (0.5, 0.5)
    u[1] = (0.0, 0.0)
    u[2] = (0.999999, 0.25)
    w = np.zeros((256, 3), np.float32)
    dsk = np.zeros((256, 2), np.float32)
    for i in range(256):
        L.kzo_cosine_hemisphere(u[i, 0], u[i, 1], w[i].ctypes.data_as(O.abi.f32p))
        L.kzo_uniform_disk(u[i, 0], u[i, 1], dsk[i].ctypes.data_as(O.abi.f32p))
    out["warp_u"], out["warp_cos"], out["warp_disk"] = u, w, dsk
    # frames
    n = rng.normal(size=(64, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    fs, ft = np.zeros_like(n), np.zeros_like(n)
    for i in range(64):
        L.kzo_frame(n[i].ctypes.data_as(O.abi.f32p), fs[i].ctypes.data_as(O.abi.f32p), ft[i].ctypes.data_as(O.abi.f32p))
    out["frame_n"], out["frame_s"], out["frame_t"] = n, fs, ft
    # BSDF tables
    rows = kiss_rows()
    m = 96
    wi = rng.normal(size=(m, 3)).astype(np.float32)
    wi[:, 2] = np.abs(wi[:, 2]) + 0.05
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(m, 3)).astype(np.float32)
    wo[:, 2] = np.abs(wo[:, 2]) + 0.05
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi[0] = (0, 0, 1)
    wo[0] = (0, 0, 1)
    wi[1, 2] = -abs(wi[1, 2])          # back side
    s = rng.random((m, 3)).astype(np.float32)
    ev = np.zeros((len(rows), m, 3), np.float32)
    pd = np.zeros((len(rows), m), np.float32)
    sw = np.zeros((len(rows), m, 7), np.float32)
    for r, row in enumerate(rows):
        for i in range(m):
            acc = 0.25 if (i % 7 == 3) else 0.0
            ev[r, i] = O.bsdf(row, "eval", wi[i], wo[i], acc)
            pd[r, i] = O.bsdf(row, "pdf", wi[i], wo[i], acc)
            wt, d, ok = O.bsdf(row, "sample", wi[i], None, acc, float(s[i, 0]), (float(s[i, 1]), float(s[i, 2])))
            sw[r, i, :3], sw[r, i, 3:6], sw[r, i, 6] = wt, d, float(ok)
    out["bsdf_wi"], out["bsdf_wo"], out["bsdf_s"], out["bsdf_eval"], out["bsdf_pdf"], out["bsdf_sample"] = wi, wo, s, ev, pd, sw
    # scene-level vectors on the Cornell-style box
    for tag, desc in (("ind", S.cornell_box(32, 32, 4)), ("pmj", S.cornell_box(32, 32, 4, sampler="pmj02bn", seed=1))):
        o = O.OracleScene(desc)
        streams = np.stack([o.sampler_stream(px, py, idx, 12) for (px, py, idx) in ((0, 0, 0), (3, 5, 2), (31, 31, 3), (17, 2, 1))])
        out["stream_" + tag] = streams
        film = o.render(threads=1)
        out["film_" + tag] = film
        pxy = np.array([(x, y) for y in range(0, 32, 5) for x in range(0, 32, 5)], np.int32)
        idx = (np.arange(len(pxy)) % 4).astype(np.uint32)
        out["samples_pxy"], out["samples_idx"] = pxy, idx
        out["samples_" + tag] = o.render_samples(pxy, idx)
        if tag == "ind":
            rays = np.array([np.concatenate([o.camera_ray(x + 0.25, y + 0.75)[0], o.camera_ray(x + 0.25, y + 0.75)[1:]]) for x, y in pxy[:16]], np.float32)
            out["camera_rays"] = rays
            tab, r, b = o.filter_table()
            out["filter_table"] = tab
            ref = np.array([0.1, -0.2, 0.3], np.float32)
            lu = rng.random((16, 3)).astype(np.float32)
            out["light_u"] = lu
            out["light_ref"] = ref
            out["light_samples"] = np.stack([o.light_sample(0, ref, float(a), float(b2), float(c)) for a, b2, c in lu])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
