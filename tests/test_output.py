"""Renderer output (bitmap.cpp:23-64, renderer.cpp:140-152): the device tone map and the PNG / EXR writers of the Python and C++
host sides."""
import os
import subprocess

import numpy as np
import pytest

from test_host_mirror import exe, python_twin       # noqa: F401  (the compiled C++ test program, the scene it builds)


def _gradient():
    w, h = 37, 5
    x, y, c = np.meshgrid(np.arange(w), np.arange(h), np.arange(3), indexing="xy")
    x, y, c = (a.transpose(0, 1, 2) for a in (x, y, c))
    scale = np.where(c == 0, np.float32(1.2), np.where(c == 1, np.float32(0.5), np.float32(0.01))).astype(np.float32)
    g = (x + 1).astype(np.float32) / np.float32(w) * scale + y.astype(np.float32) * np.float32(0.003) - np.where((c == 2) & (x == 0), np.float32(0.5), np.float32(0)).astype(np.float32)
    return g.astype(np.float32)


def _srgb8_reference(rgb):
    """bitmap.cpp:45-52 in numpy (double pow, so +-1 level at a rounding edge)."""
    t = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(np.maximum(rgb, 0).astype(np.float64), 1 / 2.4) - 0.055)
    return np.clip(255.0 * t, 0, 255).astype(np.uint8)


def test_python_writers_round_trip(kz, O, tmp_path):
    from PIL import Image
    g = _gradient()
    p = kz.output.save_exr(str(tmp_path / "g"), g)
    assert p.endswith(".exr") and np.array_equal(kz.output.load_exr(p), g)
    o = O.OracleScene(kz.scenes.cornell_box(37, 5, 1))
    film = np.zeros((g.shape[0] + 4, g.shape[1] + 4, 4), np.float32)
    film[2:-2, 2:-2, :3], film[2:-2, 2:-2, 3] = g * 2, 2.0                         # weight 2 everywhere
    px = o.srgb8(film)
    assert np.abs(px.astype(int) - _srgb8_reference(g).astype(int)).max() <= 1 and px[0, 0, 2] == 0 and px[0, -1, 0] == 255     # clamped both ways
    q = kz.output.save_png(str(tmp_path / "g"), px)
    assert q.endswith(".png") and np.array_equal(np.asarray(Image.open(q).convert("RGB")), px)


def test_cpp_writers_match_the_python_ones(exe, kz, O, tmp_path):       # noqa: F811
    from PIL import Image
    stem = str(tmp_path / "cpp")
    subprocess.check_call([exe, "--bitmap", stem])
    g = _gradient()
    assert open(stem + ".exr", "rb").read() == kz.output.exr_bytes(g)              # same file, byte for byte
    assert np.array_equal(kz.output.load_exr(stem + ".exr"), g)
    im = Image.open(stem + ".png")
    assert im.mode == "RGB" and im.size == (37, 5)
    o = O.OracleScene(kz.scenes.cornell_box(37, 5, 1))
    film = np.zeros((g.shape[0] + 4, g.shape[1] + 4, 4), np.float32)
    film[2:-2, 2:-2, :3], film[2:-2, 2:-2, 3] = g, 1.0
    assert np.array_equal(np.asarray(im), o.srgb8(film))                            # same float arithmetic as the oracle's restatement


@pytest.mark.gpu
def test_device_tone_map_matches_oracle(gpu_lib, kz, O):
    d = kz.scenes.cornell_box(96, 64, 16)
    sc = kz.Scene(d, device=0)
    sc.render()
    px = sc.srgb8()
    ref = O.OracleScene(d).srgb8(sc.film())                                          # the same film through the CPU restatement
    assert px.shape == (64, 96, 3) and np.array_equal(px, ref)                          # every byte: the sRGB curve's pow is the same arithmetic on both sides (kz_crmath.h)
    assert px.max() > 150 and px.min() < 30


@pytest.mark.gpu
def test_cpp_render_to_png(exe, kz, gpu_lib, tmp_path):       # noqa: F811
    from PIL import Image
    out = str(tmp_path / "frame.exr")                                                # any extension is replaced by .png (renderer.cpp:143-152)
    subprocess.check_call([exe, "--render-png", out])
    png = np.asarray(Image.open(str(tmp_path / "frame.png")).convert("RGB"))
    sc = kz.Scene(python_twin(kz), device=0)
    sc.render()
    diff = np.abs(png.astype(int) - sc.srgb8().astype(int))
    assert png.shape == (48, 64, 3) and diff.max() <= 2 and (diff != 0).mean() < 0.02      # look-at matrix: float vs double (1 ulp)
