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
    # The other 21 scene files of scene/2022_q1/parameters/ are this one with other kiss parameters on the object (mesh 4): geometry, lights, camera, sampler and
    # integrator are equal array for array (checked here). Their parameter sets are the second fixture: with the npz they ARE the 22 scene files.
    import glob
    params = {}
    for f in sorted(glob.glob(os.path.join(os.path.dirname(SRC), "*.xml"))):
        x = kz.xmlscene.load_xml(f)
        # (m0_r0_spec0.xml is checked in with sampleCount 8; the published pictures are 4096 spp, doc/2022_q1/2022_q1_report.md)
        assert len(x.meshes) == len(d.meshes) and dict(x.sampler, sampleCount=0) == dict(d.sampler, sampleCount=0) and x.integrator == d.integrator
        assert np.array_equal(np.asarray(x.camera["toWorld"]), np.asarray(d.camera["toWorld"]))
        for i, (a, b) in enumerate(zip(x.meshes, d.meshes)):
            assert all((a[k] is None) == (b[k] is None) and (a[k] is None or np.array_equal(a[k], b[k])) for k in ("V", "F", "N", "UV")) and a["light"] == b["light"]
            assert i == 4 or a["bsdf"] == b["bsdf"]
        params[os.path.basename(f)[:-4]] = dict(x.meshes[4]["bsdf"], _sampleCount=x.sampler["sampleCount"])
    with open(os.path.join(HERE, "q1_params.json"), "w") as fh:
        json.dump({"source": "scene/2022_q1/parameters/*.xml: the kiss parameters of the object (mesh 4 of q1_default_m0_r0.5.npz)", "params": params}, fh, indent=1)
    print("wrote q1_params.json: %d parameter sets" % len(params))


if __name__ == "__main__":
    main()
