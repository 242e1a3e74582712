"""Scene I/O adapter: the reference's XML + OBJ formats drive the core unchanged."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MINI = os.path.join(HERE, "golden", "xml", "mini.xml")
REF = "/root/reference/scene/2022_q1/parameters"


def test_mini_scene_parses_like_the_reference_loader(kz):
    d = kz.xmlscene.load_xml(MINI)
    assert d.integrator["maxDepth"] == 4 and abs(d.integrator["traceBias"] - 0.002) < 1e-9
    assert d.sampler == {"type": "correlated", "sampleCount": 9, "seed": 7, "resolution": 4}
    assert d.camera["rfilter"]["type"] == "mitchell" and d.camera["width"] == 48
    assert np.allclose(d.camera["toWorld"], kz.scenes.look_at((0, 1.5, 5), (0, 0.5, 0), (0, 1, 0)), atol=1e-6)
    assert d.background == {"color": (0.6, 0.7, 1.0), "intensity": 0.5}
    floor, cube, light = d.meshes
    # quad f 1 4 3 2 -> triangles (1,4,3), (2,1,3) in the reference's split order (verts[3], verts[0], verts[2])
    assert floor["F"].tolist() == [[0, 1, 2], [3, 0, 2]] and floor["UV"].shape == (4, 2) and floor["N"].shape == (4, 3)
    assert cube["F"].shape == (12, 3) and cube["V"].shape == (24, 3) and cube["UV"] is None        # (p, n) pairs deduplicated per face
    # transform order: scale, then rotate, then translate (each op left-multiplies, parser.cpp:243-267)
    c, s = np.cos(np.pi / 6), np.sin(np.pi / 6)
    p = np.array([1 * 0.5, 2 * 1.0, 1 * 0.5])
    expect = np.array([c * p[0] + s * p[2] + 0.2, p[1], -s * p[0] + c * p[2] - 0.3])
    assert np.min(np.linalg.norm(cube["V"] - expect.astype(np.float32), axis=1)) < 1e-5
    assert np.allclose(np.linalg.norm(cube["N"], axis=1), 1, atol=1e-6)
    assert cube["bsdf"]["type"] == "kazenstandard" and cube["bsdf"]["clearcoat"] == 1.0 and cube["bsdf"]["roughness"] == 0.35
    assert light["light"]["lightPrimaryVisibility"] is True and light["light"]["intensity"] == 20 and light["bsdf"] is None
    assert kz.Scene(d).bvh_info()["nTris"] == 2 + 12 + 2


def test_unsupported_content_raises(kz, tmp_path):
    txt = open(MINI).read()
    for old, new, msg in (('type="path_mis"', 'type="whitted"', "hot path"), ('type="correlated"', 'type="sobol"', "not supported"),
                          ('<bsdf type="diffuse">', '<bsdf type="principled">', "not supported"), ('<bsdf type="diffuse">', '<bsdf type="normalmap">', "needs a normal texture"),
                          ('type="constanttexture" id="baseColor"', 'type="checkerboard" id="baseColor"', "not supported"), ('<camera type="perspective">', '<camera type="fisheye">', "not supported")):
        p = tmp_path / "bad.xml"
        p.write_text(txt.replace(old, new).replace('value="floor.obj"', 'value="%s"' % os.path.join(HERE, "golden", "xml", "floor.obj"))
                     .replace('value="cube.obj"', 'value="%s"' % os.path.join(HERE, "golden", "xml", "cube.obj"))
                     .replace('value="light.obj"', 'value="%s"' % os.path.join(HERE, "golden", "xml", "light.obj")))
        with pytest.raises(ValueError) as e:
            kz.xmlscene.load_xml(str(p))
        assert msg in str(e.value)


TEXTURED = os.path.join(HERE, "golden", "xml", "textured.xml")


def test_textured_scene_parses_like_the_reference_loader(kz):
    """imagetexture / colorramp / blend children (matched by id, texture.cpp:241-262), lambertian and normalmap rows."""
    d = kz.xmlscene.load_xml(TEXTURED)
    floor, cube, light = d.meshes
    assert floor["bsdf"]["type"] == "diffuse" and floor["bsdf"]["albedo"]["type"] == "imagetexture"
    img = floor["bsdf"]["albedo"]
    assert img["image"].shape == (8, 8, 3) and img["image"].dtype == np.uint8 and img["scale"] == 4.0 and img["colorspace"] == "srgb"
    nm = cube["bsdf"]
    assert nm["type"] == "normalmap" and nm["normal"]["colorspace"] == "linear" and nm["normal"]["image"].shape == (32, 32, 3)
    kiss = nm["nested"]
    assert kiss["type"] == "kazenstandard" and kiss["metallic"] == 0.0
    bl = kiss["baseColor"]
    assert bl["type"] == "blend" and bl["blendmode"] == "mix" and bl["mask"]["image"].shape == (16, 16, 1) and bl["input1"]["color"] == (0.8, 0.3, 0.2)
    assert bl["input2"]["type"] == "imagetexture"
    r = kiss["roughness"]
    assert r["type"] == "colorramp" and (r["min"], r["max"]) == (0.2, 0.7) and r["nested"]["scale"] == 2.0
    cd = d.to_c()
    assert (cd.nBsdfs, cd.nTextures, cd.nImages) == (3, 8, 5) and cd.bsdfs[1].nested == 2 and cd.bsdfs[2].type == kz.abi.KZ_BSDF_KAZENSTANDARD
    assert d.integrator["regularization"] is True
    kz.Scene(d)                                                   # passes the product library's validation


def test_textured_scene_oracle_render(kz, O):
    o = O.OracleScene(kz.xmlscene.load_xml(TEXTURED))
    rgb = o.rgb(o.render(threads=2))
    assert np.isfinite(rgb).all() and rgb.mean() > 0.01


@pytest.mark.gpu
def test_textured_scene_gpu_matches_oracle(kz, O, gpu_lib):
    d = kz.xmlscene.load_xml(TEXTURED)
    sc = kz.Scene(d, device=0)
    sc.render()
    o = O.OracleScene(d)
    assert float(np.sqrt(np.mean((sc.rgb() - o.rgb(o.render(threads=0))) ** 2))) < 1e-3


def test_mini_scene_oracle_render(kz, O):
    d = kz.xmlscene.load_xml(MINI)
    o = O.OracleScene(d)
    film = o.render(threads=2)
    rgb = o.rgb(film)
    assert o.sample_count == 9 and np.isfinite(rgb).all() and rgb.mean() > 0.05


ENVMAP = os.path.join(HERE, "golden", "xml", "envmap.xml")


def test_background_with_an_environment_map_loads(kz, O):
    """<texture type="background"><texture type="imagetexture" .../></texture>: the nested texture is looked up by direction."""
    d = kz.xmlscene.load_xml(ENVMAP)
    assert d.background["texture"]["type"] == "imagetexture" and d.background["texture"]["image"].shape == (8, 16, 3) and d.background["intensity"] == 0.5
    o = O.OracleScene(d)
    up, down = o.background(np.array([0, 1, 0], np.float32)), o.background(np.array([0, -1, 0], np.float32))
    assert up[0] > 0.35 and down[0] < 0.05 and down[2] > up[2]                       # red fades towards the nadir, blue grows (x intensity 0.5)
    assert o.rgb(o.render(threads=0)).mean() > 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("path", [MINI, ENVMAP])
def test_mini_scene_gpu_matches_oracle(kz, O, gpu_lib, path):
    d = kz.xmlscene.load_xml(path)
    sc = kz.Scene(d, device=0)
    sc.render()
    o = O.OracleScene(d)
    assert float(np.sqrt(np.mean((sc.rgb() - o.rgb(o.render(threads=0))) ** 2))) < 1e-3


PIN_IMAGES = ["default_m0_r0.5", "m0.0_r0", "m0.0_r0.5", "m0.0_r1", "m0_r0_spec0", "m0_r0_spec0.5", "m0_r0_spec1", "m0_r0_spec1_st0.5", "m0_r0_spec1_st1",
              "m1_r0", "m1_r0.5", "m1_r1", "r0.5_c0", "r0.5_c0.5", "r0.5_c1", "r0.5_c1_cr0.5", "r0.5_c1_cr1", "r0_s0", "r0_s0.5", "r0_s1", "r0_s1_st0.5", "r0_s1_st1"]


# What the published pictures say about the scene DATA (scripts/pin_reference_pngs.py experiments, profiles/pin/): per-light basis images of
# default_m0_r0.5 fitted to its picture by least squares give these factors on the three area lights (back, right, left). They are a property of the
# lights the 22 scenes share, not of the object's BSDF: applied UNCHANGED to all 22 scene files they take the oracle from 0.953 of the picture (mean linear
# radiance, every image) to 0.9967 ... 1.0009.
PUBLISHED_LIGHT_FACTORS = (0.976, 1.135, 1.034)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("name", PIN_IMAGES)
def test_reference_scene_files_match_their_published_pngs(kz, O, name):
    """ALL 22 scene/2022_q1/parameters/*.xml (36 378 triangles, SURVEY 8d C1 geometry; a kiss parameter sweep over metallic,
    roughness, specular, specularTint, clearcoat, sheen) loaded through the adapter and rendered by the oracle at 240x135x64 against the
    reference's own 4096-spp PNG (doc/2022_q1/img/param/) - the only image-level pin the reference offers ('usable only as a
    statistical sanity check'). Blocks of the 16x9 grid in which the picture is clipped (value 255: the mirror-like objects' highlights,
    whose HDR energy the 8-bit picture has lost) are left out.

    Round 3 asserted the RESIDUAL (the oracle 1.0-2.2 % darker in sRGB, 4.7 % in linear radiance, the same in every image). This asserts the
    EXPLANATION: the three light intensities of the checked-in scene files, multiplied by ONE set of factors fitted on ONE image
    (PUBLISHED_LIGHT_FACTORS), reproduce the mean linear radiance of EVERY one of the 22 pictures to +-0.5 % and the mean of the 16x9 sRGB grid to
    0.004 - whatever the object's BSDF parameters are. A transport or BSDF error in the restatement would show as a dependence on those
    parameters, which is what the bound excludes; that the published renders used other light intensities than the checked-in XML stays a
    hypothesis about the reference's data (profiles/pin/README.md), not about its code."""
    from PIL import Image
    d = kz.xmlscene.load_xml(os.path.join(REF, name + ".xml"), {"camera": {"width": 240, "height": 135}, "sampler": {"sampleCount": 64}})
    assert d.n_tris() == 36378
    lights = [m["light"] for m in d.meshes if m["light"]]
    assert [l["intensity"] for l in lights] == [1.0, 1.5, 4.0]                                       # what the scene files say
    for l, f in zip(lights, PUBLISHED_LIGHT_FACTORS):
        l["intensity"] *= f
    o = O.OracleScene(d)
    rgb = o.rgb(o.render(threads=0)).astype(np.float64)
    png = np.asarray(Image.open("/root/reference/doc/2022_q1/img/param/%s.png" % name).convert("RGB"), np.float64) / 255
    lin_png = np.where(png <= 0.04045, png / 12.92, np.power((png + 0.055) / 1.055, 2.4))            # common.cpp:368-382
    clipped = (png >= 1.0).any(axis=2).reshape(27, 40, 48, 40).any(axis=(1, 3))
    lp = lin_png.mean(axis=2).reshape(27, 40, 48, 40).mean(axis=(1, 3))
    lo = rgb.mean(axis=2).reshape(27, 5, 48, 5).mean(axis=(1, 3))
    ok = ~clipped & (lp > 0.02)
    assert ok.sum() >= 1000
    ratio = lo[ok] / lp[ok]
    assert abs(ratio.mean() - 1.0) < 0.005, ratio.mean()                                             # explained to +-0.5 % (unweighted: 0.953)
    assert ratio.std() < 0.055                                                                       # the low-frequency shape no light factor removes (0.045-0.049)
    x = np.clip(rgb, 0, 1)
    srgb = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)                 # common.cpp:352-366
    ok9 = ~(png >= 1.0).any(axis=2).reshape(9, 120, 16, 120).any(axis=(1, 3))
    ref9, mine9 = png.reshape(9, 120, 16, 120, 3).mean(axis=(1, 3)), srgb.reshape(9, 15, 16, 15, 3).mean(axis=(1, 3))
    assert ok9.sum() >= 120
    # (the worst single block and channel: 0.028-0.032 for the rough objects, up to 0.057 next to the mirror-like objects' highlights at 64 spp)
    assert np.abs(mine9 - ref9)[ok9].max() < 0.06 and abs(mine9[ok9].mean() - ref9[ok9].mean()) < 0.004


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only present in the build container")
def test_every_reference_scene_file_loads(kz):
    """All checked-in scene XMLs of the reference (scene/2022_q1/**) stay inside the hot path's plugin set: they load through
    xmlscene.load_xml unchanged and pass kz_scene_create's validation on the host (no GPU needed)."""
    import glob
    files = sorted(glob.glob("/root/reference/scene/2022_q1/**/*.xml", recursive=True))
    assert len(files) >= 23
    for f in files:
        d = kz.xmlscene.load_xml(f, {"camera": {"width": 32, "height": 18}, "sampler": {"sampleCount": 1}})
        assert d.n_tris() > 0 and any(m["light"] for m in d.meshes), f
    sc = kz.Scene(d, device=None)                                                   # host-side build of the last one
    assert sc.bvh_info()["nTris"] == d.n_tris()


Q1_NPZ = os.path.join(HERE, "golden", "q1_default_m0_r0.5.npz")


def test_flattened_asset_scene_fixture(kz, O):
    """tests/golden/q1_default_m0_r0.5.npz (made by tests/golden/make_q1_scene.py from the reference's own XML + OBJ files) loads on a box without the
    reference and renders through the oracle; where the reference is present it is the loader's output array for array."""
    d = kz.scenes.load_npz(Q1_NPZ, {"camera": {"width": 48, "height": 48}, "sampler": {"sampleCount": 2}})
    assert d.n_tris() == 36378 and [m["F"].shape[0] for m in d.meshes] == [2, 12, 12, 1536, 34816]
    assert [bool(m["light"]) for m in d.meshes] == [True, True, True, False, False] and d.meshes[4]["bsdf"]["type"] == "kazenstandard"
    assert all(m["N"] is not None for m in d.meshes)
    o = O.OracleScene(d)
    assert 0.02 < o.rgb(o.render(threads=0)).mean() < 1.0
    src = os.path.join(REF, "default_m0_r0.5.xml")
    if os.path.exists(src):
        x = kz.xmlscene.load_xml(src, {"camera": {"width": 48, "height": 48}, "sampler": {"sampleCount": 2}})
        for ma, mb in zip(d.meshes, x.meshes):
            for k in ("V", "F", "N", "UV"):
                assert (ma[k] is None) == (mb[k] is None) and (ma[k] is None or np.array_equal(ma[k], mb[k]))
            assert ma["bsdf"] == mb["bsdf"] or (ma["bsdf"] and {k: (list(v) if isinstance(v, tuple) else v) for k, v in mb["bsdf"].items()} == ma["bsdf"])
        assert np.array_equal(np.asarray(d.camera["toWorld"]), np.asarray(x.camera["toWorld"]))


def test_parameter_sets_fixture(kz):
    """tests/golden/q1_params.json: the kiss parameters of the 22 scene files of scene/2022_q1/parameters/ (which differ in nothing else: make_q1_scene.py checks it when
    it writes the file). Where the reference is present every set is the loader's output for its file."""
    import json
    params = json.load(open(os.path.join(HERE, "golden", "q1_params.json")))["params"]
    assert len(params) == 22 and set(PIN_IMAGES) <= set(params)
    assert all(p["type"] == "kazenstandard" for p in params.values())
    if os.path.isdir(REF):
        for name, p in params.items():
            x = kz.xmlscene.load_xml(os.path.join(REF, name + ".xml"), {"camera": {"width": 16, "height": 9}, "sampler": {"sampleCount": 1}})
            want = {k: (list(v) if isinstance(v, tuple) else v) for k, v in x.meshes[4]["bsdf"].items()}
            assert {k: v for k, v in p.items() if not k.startswith("_")} == want, name
