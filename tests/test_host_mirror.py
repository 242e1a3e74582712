"""The C++ host mirror (nano-kazen_amd/host/kazen_host.hpp) keeps the reference's plugin surface: registry strings,
PropertyList names, defaults, addChild/activate order and exception behaviour. The test program is
driven like the reference's parser drives its object model; Python builds the same scene through its own path."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_cpp", "host_mirror_test.cpp")
LIBDIR = os.path.join(ROOT, "nano-kazen_amd", "csrc")
HOST = os.path.join(ROOT, "nano-kazen_amd", "host")
# the two files INTEGRATION.md adds to a kazen tree, compiled unchanged against the mirror (mirror_tree/kazen/*.h answer to the reference's header names)
ADAPTER = os.path.join(HOST, "adapter", "renderer_mi355x.cpp")
INCLUDES = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HOST, "mirror_tree"), "-I" + os.path.join(HOST, "adapter")]


@pytest.fixture(scope="module")
def exe(tmp_path_factory, kz):
    kz.abi.load_library()
    out = str(tmp_path_factory.mktemp("host") / "host_mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror"] + INCLUDES + ["-o", out, SRC, ADAPTER, "-L" + LIBDIR, "-lkazen_mi355x", "-Wl,-rpath," + LIBDIR])
    return out


def python_twin(kz):
    S = kz.scenes
    s = S.SceneDescription()

    def q(p, n):
        P = np.array(p, np.float32)
        return P, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), np.tile(np.array(n, np.float32), (4, 1)), np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    s.add_mesh(*q([(-2, -1, -2), (2, -1, -2), (2, -1, 2), (-2, -1, 2)], (0, 1, 0)))
    s.add_mesh(*q([(-2, -1, -2), (-2, 2, -2), (2, 2, -2), (2, -1, -2)], (0, 0, 1)), bsdf=S.diffuse((0.7, 0.3, 0.3)))
    s.add_mesh(*q([(-0.8, -0.6, 0), (0.8, -0.6, 0), (0.8, 0.6, -0.6), (-0.8, 0.6, -0.6)], (0, 0.70710678, 0.70710678)),
               bsdf=S.kazenstandard((0.8, 0.6, 0.2), roughness=0.4, metallic=0.0, clearcoat=1.0, sheen=0.5))
    s.add_mesh(*q([(-0.5, 1.5, -0.5), (-0.5, 1.5, 0.5), (0.5, 1.5, 0.5), (0.5, 1.5, -0.5)], (0, -1, 0)), light=S.area((1.0, 0.9, 0.8), 12.0, False))
    s.background = {"color": (0.5, 0.6, 1.0), "intensity": 0.25}
    s.camera.update(width=64, height=48, fov=40.0, nearClip=0.1, farClip=100.0, toWorld=S.look_at((0, 0.5, 3.0), (0, 0, 0), (0, 1, 0)))
    s.sampler = {"type": "independent", "sampleCount": 8, "seed": 0}
    s.integrator["maxDepth"] = 4
    return s


def test_flattened_description_and_errors(exe, kz):
    d = json.loads(subprocess.check_output([exe]).decode())
    assert (d["nMeshes"], d["nBsdfs"], d["nLights"], d["bvhTris"]) == (4, 4, 1, 8)
    assert d["meshBsdf"] == [0, 1, 2, 3] and d["meshLight"] == [-1, -1, -1, 0]            # no bsdf child -> Mesh::activate's default diffuse (mesh.cpp:25-28), described like any other
    assert np.allclose(d["kiss"], [1, 0.8, 0.6, 0.2, 0.4, 0, 0, 0.5, 0.5, 1, 0.5, 0.5, 0.5])     # reference defaults (bsdf.cpp:1160-1167)
    assert np.allclose(d["light"], [1, 0.9, 0.8, 12, 0])                                      # lightPrimaryVisibility defaults to false
    assert np.allclose(d["camera"], [64, 48, 40, 0.1, 100, 0, 2, 0.5]) and d["sampler"] == [0, 8, 0]
    assert np.allclose(d["integrator"], [0, 4, 0.001, 0, 0.5]) and np.allclose(d["background"], [1, 0.5, 0.6, 1.0, 0.25])
    assert np.allclose(d["toWorld"], kz.scenes.look_at((0, 0.5, 3.0), (0, 0, 0), (0, 1, 0)).reshape(-1), atol=1e-6)
    e = d["errors"]
    assert "not on the MI355X hot path" in e[0] and "could not be found" in e[1]
    assert e[2] == "No camera was specified!" and e[3] == "There can only be one sampler per scene!" and "is not supported" in e[4]


@pytest.mark.gpu
def test_cpp_render_equals_python_render(exe, kz, O, gpu_lib, tmp_path):
    out = str(tmp_path / "rgb.bin")
    info = json.loads(subprocess.check_output([exe, "--render", out]).decode())
    rgb = np.fromfile(out, np.float32).reshape(48, 64, 3)
    assert info["pixels"] == 64 * 48 and rgb.mean() > 0.01
    twin = python_twin(kz)
    sc = kz.Scene(twin, device=0)
    sc.render()
    # same description up to the last ulp of the look-at matrix (C++ forms it in float, numpy in double)
    assert float(np.sqrt(np.mean((sc.rgb() - rgb) ** 2))) < 1e-4
    ora = O.OracleScene(twin)
    cpu = ora.rgb(ora.render(threads=0))
    assert float(np.sqrt(np.mean((rgb - cpu) ** 2))) < 1e-3


@pytest.mark.gpu
def test_cpp_multi_device_render_equals_single_device(exe, kz, gpu_lib, tmp_path):
    """kazen::renderer::render(scene, devices) (kz_render_multi: one host thread per device, 64x64 tiles dealt by area, films summed on
    the host) on the devices of this box against renderer::render(scene, device)."""
    one, multi = str(tmp_path / "one.bin"), str(tmp_path / "multi.bin")
    subprocess.check_output([exe, "--render", one])
    devs = ",".join(str(d) for d in range(min(gpu_lib.kz_device_count(), 8)))
    info = json.loads(subprocess.check_output([exe, "--render-multi", multi, devs]).decode())
    assert info["pixels"] == 64 * 48 and info["devices"] == len(devs.split(","))
    a, b = np.fromfile(one, np.float32), np.fromfile(multi, np.float32)
    assert a.shape == b.shape and float(np.sqrt(np.mean((a - b) ** 2))) < 1e-5


@pytest.mark.gpu
def test_cpp_same_scene_on_each_device_in_turn(exe, kz, gpu_lib, tmp_path):
    """renderer::render(scene, device) once read the film of the PRIMARY replica whatever device it had rendered on. The adapter's DeviceScene gets the film
    from the device that rendered it (kz_render_tiles hands it back): one scene rendered on every device of the box in turn (twice on device 0 where there is only one:
    a second render must not see the first one's film either) gives the same picture every time."""
    n = min(gpu_lib.kz_device_count(), 8)
    devs = ",".join(str(d) for d in range(n)) if n > 1 else "0,0"
    stem = str(tmp_path / "turn")
    info = json.loads(subprocess.check_output([exe, "--render-in-turn", stem, devs]).decode())
    films = [np.fromfile("%s.%d" % (stem, k), np.float32) for k in range(info["renders"])]
    assert len(films) == len(devs.split(",")) and films[0].mean() > 0.01
    for f in films[1:]:
        assert np.array_equal(f.view(np.uint32), films[0].view(np.uint32))


# ---------------------------------------------------------------- SURVEY 8f rank 4 through the plugin surface
def _write_ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n# checker\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def textured_twin(kz):
    S = kz.scenes
    chk = S._test_images()[0]
    s = S.SceneDescription()

    def q(p, n):
        P = np.array(p, np.float32)
        return P, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), np.tile(np.array(n, np.float32), (4, 1)), np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    s.add_mesh(*q([(-2, -1, -2), (2, -1, -2), (2, -1, 2), (-2, -1, 2)], (0, 1, 0)), bsdf=S.lambertian(S.imagetexture(chk, 3.0, "srgb")))
    nrm = np.array([[[128, 128, 255], [160, 128, 240]], [[128, 160, 240], [100, 110, 235]]], np.uint8)
    kiss = S.kazenstandard(S.imagetexture(chk, 1.0, "srgb"), S.colorramp(S.blend(None, S.imagetexture(chk, 2.0, "linear"), S.constanttexture((0.9, 0.9, 0.9)), "multiply"), 0.2, 0.7),
                           0.0, clearcoat=1.0)
    s.add_mesh(*q([(-0.8, -0.6, 0), (0.8, -0.6, 0), (0.8, 0.6, -0.6), (-0.8, 0.6, -0.6)], (0, 0.70710678, 0.70710678)), bsdf=S.normalmap(S.imagetexture(nrm, 1.0, "linear"), kiss))
    s.add_mesh(*q([(-0.5, 1.5, -0.5), (-0.5, 1.5, 0.5), (0.5, 1.5, 0.5), (0.5, 1.5, -0.5)], (0, -1, 0)), light=S.area((1.0, 1.0, 1.0), 12.0, False))
    s.camera.update(width=64, height=48, fov=40.0, nearClip=0.1, farClip=100.0, toWorld=S.look_at((0, 0.5, 3.0), (0, 0, 0), (0, 1, 0)))
    s.sampler = {"type": "independent", "sampleCount": 8, "seed": 0}
    s.integrator.update(maxDepth=4, regularization=True)
    return s


def test_texture_plugins_flatten_like_the_python_path(exe, kz, tmp_path):
    a = kz.abi
    ppm = str(tmp_path / "checker.ppm")
    _write_ppm(ppm, kz.scenes._test_images()[0])
    d = json.loads(subprocess.check_output([exe, "--textured", ppm]).decode())
    assert (d["nBsdfs"], d["nImages"]) == (4, 4)
    lam, nm, _light_mesh_default, kiss = d["bsdfs"]
    assert lam[0] == a.KZ_BSDF_DIFFUSE and lam[1] > 0
    assert nm[0] == a.KZ_BSDF_NORMALMAP and nm[4] > 0 and nm[5] == 3                         # the wrapped row sits behind the per-mesh rows
    assert kiss[0] == a.KZ_BSDF_KAZENSTANDARD and kiss[1] > 0 and kiss[2] > 0 and kiss[3] == 0 and kiss[6] == 0.0     # metallic: a folded constanttexture
    T = d["textures"]
    ramp = T[kiss[2] - 1]
    assert ramp[0] == a.KZ_TEX_COLORRAMP and np.allclose(ramp[4:6], [0.2, 0.7])
    bl = T[ramp[7]]
    assert bl[0] == a.KZ_TEX_BLEND and bl[6] == a.KZ_BLEND_MULTIPLY and bl[7] == -1 and T[bl[8]][0] == a.KZ_TEX_IMAGE and T[bl[9]][0] == a.KZ_TEX_CONSTANT
    assert T[lam[1] - 1][:4] == [a.KZ_TEX_IMAGE, T[lam[1] - 1][1], 3.0, 1] and T[nm[4] - 1][3] == 0                   # srgb default / "linear"
    assert [8, 8, 3, a.KZ_PIXEL_U8] in d["images"] and [2, 2, 3, a.KZ_PIXEL_U8] in d["images"]
    assert "does not match any field" in d["errors"][0] and "cannot open" in d["errors"][1]


@pytest.mark.gpu
def test_cpp_textured_render_equals_python_render(exe, kz, O, gpu_lib, tmp_path):
    ppm, out = str(tmp_path / "checker.ppm"), str(tmp_path / "rgb.bin")
    _write_ppm(ppm, kz.scenes._test_images()[0])
    subprocess.check_output([exe, "--textured", ppm, out])
    rgb = np.fromfile(out, np.float32).reshape(48, 64, 3)
    twin = textured_twin(kz)
    sc = kz.Scene(twin, device=0)
    sc.render()
    assert float(np.sqrt(np.mean((sc.rgb() - rgb) ** 2))) < 1e-4
    ora = O.OracleScene(twin)
    assert float(np.sqrt(np.mean((rgb - ora.rgb(ora.render(threads=0))) ** 2))) < 1e-3


# ---------------------------------------------------------------- scene files through the C++ loader
MINI = os.path.join(ROOT, "tests", "golden", "xml", "mini.xml")


def _mesh_digest(m):
    V, F = m["V"], m["F"].reshape(-1).astype(np.uint64)
    w = (np.arange(F.size, dtype=np.uint64) % np.uint64(7)) + np.uint64(1)
    return [V.shape[0], m["F"].shape[0], int(m["N"] is not None), int(m["UV"] is not None), float(V.astype(np.float64).sum()),
            float(m["N"].astype(np.float64).sum()) if m["N"] is not None else 0.0, float(m["UV"].astype(np.float64).sum()) if m["UV"] is not None else 0.0,
            int((F * w).sum())]


def test_cpp_xml_loader_equals_the_python_one(exe, kz):
    d = json.loads(subprocess.check_output([exe, "--xml", MINI]).decode())
    assert "error" not in d, d
    py = kz.xmlscene.load_xml(MINI)
    assert d["nMeshes"] == len(py.meshes) and d["nLights"] == 1
    for cm, pm in zip(d["meshes"], py.meshes):
        g = _mesh_digest(pm)
        assert cm[:4] == g[:4] and cm[9] == g[7]                                   # counts, presence of N / UV, the index buffer
        assert np.allclose(cm[6:9], g[4:7], rtol=1e-5, atol=1e-4)                      # vertex / normal / uv sums (transform applied in float)
    a = kz.abi
    assert d["bsdfTypes"] == [a.KZ_BSDF_DIFFUSE, a.KZ_BSDF_KAZENSTANDARD, a.KZ_BSDF_DIFFUSE]      # (the emitter mesh has no <bsdf>: Mesh::activate gives it the default diffuse one, mesh.cpp:25-28)
    assert d["camera"][:3] == [a.KZ_CAMERA_PERSPECTIVE, 48, 32] and np.allclose(d["camera"][3:6], [45, 0.1, 50]) and d["camera"][6] == a.KZ_FILTER_MITCHELL
    assert np.allclose(d["toWorld"], np.asarray(py.camera["toWorld"]).reshape(-1), atol=1e-6)
    assert d["sampler"] == [a.KZ_SAMPLER_CORRELATED, 9, 7] and d["integrator"][0] == 4 and np.isclose(d["integrator"][1], 0.002)
    lp = [m["light"] for m in py.meshes if m["light"]][0]
    assert np.allclose(d["background"], [1, *py.background["color"], py.background["intensity"]])
    assert np.allclose(d["lights"][0], [*lp["color"], lp["intensity"], int(lp["lightPrimaryVisibility"])])


def test_cpp_xml_loader_errors(exe, tmp_path):
    txt = open(MINI).read()
    here = os.path.join(ROOT, "tests", "golden", "xml")
    for old, new, msg in (('type="path_mis"', 'type="whitted"', "not on the MI355X hot path"), ('<float name="fov" value="45"/>', '<float name="fov" value="wide"/>', "Could not parse floating point"),
                          ('<lookat origin="0, 1.5, 5"', '<lookat origin="0, 1.5"', "Expected 3 values"), ('<sampler type="correlated">', '<sampler type="correlated"><lookat origin="0 0 0" target="0 0 1" up="0 1 0"/>', "transform nodes can only contain"),
                          ('</scene>', '', "missing </scene>"), ('<integer name="maxDepth" value="4"/>', '<integer name="maxDepth"/>', "missing attribute"),
                          ('value="floor.obj"', 'value="nosuchfile.obj"', "Unable to open OBJ file")):
        p = tmp_path / "bad.xml"
        p.write_text(txt.replace(old, new, 1).replace('value="cube.obj"', 'value="%s"' % os.path.join(here, "cube.obj")).replace('value="light.obj"', 'value="%s"' % os.path.join(here, "light.obj"))
                     .replace('value="floor.obj"', 'value="%s"' % os.path.join(here, "floor.obj")))
        d = json.loads(subprocess.check_output([exe, "--xml", str(p)]).decode())
        assert msg in d.get("error", ""), (msg, d)


def test_cpp_xml_loader_environment_background(exe, kz):
    """the C++ loader hands the background's nested imagetexture over as a texture row (KzBackground.texture) + a raster"""
    d = json.loads(subprocess.check_output([exe, "--xml", os.path.join(ROOT, "tests", "golden", "xml", "envmap.xml")]).decode())
    assert "error" not in d, d
    assert d["background"][0] == 1 and np.isclose(d["background"][4], 0.5) and d["backgroundTexture"] >= 1 and d["nImages"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name,shape", [("mini.xml", (32, 48, 3)), ("envmap.xml", None)])
def test_cpp_xml_render_equals_python_render(exe, kz, gpu_lib, tmp_path, name, shape):
    """A scene FILE through kazen::loadFromXML, the mirror's private-member plugin classes, their describe() virtuals and the adapter of INTEGRATION.md
    (compiled unchanged) against the same file through the Python path: the same description, so the same picture - bit for bit."""
    path = os.path.join(ROOT, "tests", "golden", "xml", name)
    out = str(tmp_path / "rgb.bin")
    subprocess.check_output([exe, "--xml", path, out])
    sc = kz.Scene(kz.xmlscene.load_xml(path), device=0)
    sc.render()
    rgb = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 3)
    if shape:
        assert rgb.shape == shape
    assert rgb.mean() > 0.01 and np.array_equal(sc.rgb().view(np.uint32), rgb.view(np.uint32)), float(np.abs(sc.rgb() - rgb).max())


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene/2022_q1"), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("rel", ["parameters/default_m0_r0.5.xml", "parameters/m1_r0.xml", "WarmStudio/WarmStudio.xml"])
def test_cpp_xml_loader_on_reference_scene_files(exe, kz, rel):
    path = os.path.join("/root/reference/scene/2022_q1", rel)
    d = json.loads(subprocess.check_output([exe, "--xml", path]).decode())
    assert "error" not in d, d
    py = kz.xmlscene.load_xml(path)
    assert d["nMeshes"] == len(py.meshes) and d["nLights"] == sum(1 for m in py.meshes if m["light"])
    for cm, pm in zip(d["meshes"], py.meshes):
        g = _mesh_digest(pm)
        assert cm[:4] == g[:4] and cm[9] == g[7]
        assert np.allclose(cm[6:9], g[4:7], rtol=1e-5, atol=1e-2)
    assert d["camera"][1:3] == [py.camera["width"], py.camera["height"]] and np.allclose(d["toWorld"], np.asarray(py.camera["toWorld"]).reshape(-1), atol=1e-6)
    assert d["sampler"][1] == py.sampler["sampleCount"]


def test_example_main_builds_and_fails_loudly_without_a_gpu(kz, tmp_path):
    """nano-kazen_amd/host/example_main.cpp: main.cpp's shape on top of the library."""
    out = str(tmp_path / "kazen_mi355x")
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + INCLUDES + [os.path.join(HOST, "example_main.cpp"), ADAPTER, "-L" + LIBDIR, "-lkazen_mi355x",
                           "-Wl,-rpath," + LIBDIR, "-o", out])
    r = subprocess.run([out, "/nonexistent/scene.xml"], capture_output=True, text=True)
    assert r.returncode != 0 and "file not found" in r.stderr
    if kz.abi.load_library().kz_device_count() == 0:
        r = subprocess.run([out, MINI], capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr          # the product path has no CPU fallback


# ---------------------------------------------------------------- the adapter INTEGRATION.md hands to a kazen maintainer
def test_integration_md_quotes_the_adapter_verbatim():
    """The two NEW files of INTEGRATION.md are the files this suite compiles against the mirror, byte for byte."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for title, path in (("### NEW file `include/kazen/mi355x.h`", os.path.join(HOST, "adapter", "kazen", "mi355x.h")),
                        ("### NEW file `src/kazen/renderer_mi355x.cpp`", ADAPTER)):
        at = md.index(title)
        a = md.index("```cpp\n", at) + len("```cpp\n")
        b = md.index("\n```\n", a) + 1
        assert md[a:b] == open(path).read(), path


def test_adapter_reads_nothing_the_reference_keeps_private(tmp_path):
    """The mirror's plugin classes keep their parameters private, like the reference's (bsdf.cpp:1407-1418, light.cpp:61-65, integrator.cpp:350-354,
    camera.cpp:122-128): a translation unit that reaches for one does not compile, so the adapter - which does - reads none."""
    ok = tmp_path / "ok.cpp"
    ok.write_text('#include <kazen/scene.h>\nint main() { kazen::PropertyList p; kazen::Diffuse d(p); KzBSDF row{}; kazen::mi355x::Rows rows; return d.describe(row, rows) ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only"] + INCLUDES + [str(ok)])
    for cls, member in (("Diffuse", "m_albedo"), ("KazenStandardSurface", "m_specular"), ("AreaLight", "m_intensity"), ("PathMisIntegrator", "m_maxDepth"),
                        ("PerspectiveCamera", "m_fov"), ("PMJ02BN", "m_seed"), ("GaussianFilter", "m_stddev"), ("ConstantTexture", "m_color"), ("RoughConductor", "m_alpha")):
        bad = tmp_path / "bad.cpp"
        bad.write_text('#include <kazen/scene.h>\nint main() { kazen::PropertyList p; kazen::%s o(p); return (int)sizeof(o.%s); }\n' % (cls, member))
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + INCLUDES + [str(bad)], capture_output=True, text=True)
        assert r.returncode != 0 and ("private" in r.stderr or "protected" in r.stderr), (cls, member, r.stderr[-400:])
