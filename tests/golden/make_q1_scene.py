#!/usr/bin/env python3
"""Flattens the reference's own asset scene scene/2022_q1/parameters/default_m0_r0.5.xml (+ its OBJ files: 36 378 triangles with smooth vertex
normals and uv tangents) into tests/golden/q1_default_m0_r0.5.npz: the vertex / index / normal / uv ARRAYS the reference's loader produces
(nano-kazen_amd/xmlscene.py restates parser.cpp + mesh.cpp:200-343: de-duplicated (p, n, uv) triples, quads split in its order, toWorld applied) and
the scene's parameters as JSON. Data only - no reference text. Only runs where /root/reference exists (this container); the .npz is the committed
fixture that travels to the GPU box, where tests/test_gpu_parity.py renders it through the HIP path against the oracle.

    python tests/golden/make_q1_scene.py
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
kz = importlib.import_module("nano-kazen_amd")
SRC = "/root/reference/scene/2022_q1/parameters/default_m0_r0.5.xml"


def main():
    d = kz.xmlscene.load_xml(SRC)
    arrays, meshes = {}, []
    for i, m in enumerate(d.meshes):
        for k in ("V", "F", "N", "UV"):
            if m[k] is not None:
                arrays["m%d_%s" % (i, k)] = m[k]
        meshes.append({"bsdf": m["bsdf"], "light": m["light"], "has": [k for k in ("V", "F", "N", "UV") if m[k] is not None]})
    cam = dict(d.camera)
    arrays["camera_toWorld"] = np.asarray(cam.pop("toWorld"), np.float32)
    meta = {"source": "scene/2022_q1/parameters/default_m0_r0.5.xml", "meshes": meshes, "camera": cam, "sampler": d.sampler, "integrator": d.integrator,
            "background": d.background, "n_tris": d.n_tris()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    out = os.path.join(HERE, "q1_default_m0_r0.5.npz")
    np.savez_compressed(out, **arrays)
    print("wrote %s: %d meshes, %d triangles, %.2f MB" % (out, len(meshes), d.n_tris(), os.path.getsize(out) / 1e6))


if __name__ == "__main__":
    main()
