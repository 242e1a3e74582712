"""No-GPU tests of the product library: it loads, exports every symbol of include/kazen_mi355x.h, validates
descriptions, builds the BVH on the host, and refuses to render without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(kz):
    lib = kz.abi.load_library()
    decl = lambda name: set(re.findall(r"^(?:int|void|const char \*)\s*\*?(kz_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", name)).read(), re.M))
    product, dev = decl("kazen_mi355x.h"), decl("kazen_mi355x_dev.h")
    # the product header is what a maintainer's adapter includes: at most twenty-two entry points (round 5: + kz_device_trim; round 6: + kz_film_merge_rects); everything else is the development surface
    assert product == set(kz.abi.PRODUCT_EXPORTS) and len(product) <= 22, product ^ set(kz.abi.PRODUCT_EXPORTS)
    declared = product | dev
    assert not (product & dev)
    hooks = set(kz.abi.DEV_ONLY_EXPORTS)
    assert declared == set(kz.abi.EXPORTS) | hooks, declared ^ (set(kz.abi.EXPORTS) | hooks)
    for sym in declared - hooks:
        assert getattr(lib, sym) is not None
    assert lib.kz_abi_version() == kz.abi.KZ_ABI_VERSION == 6
    # the hooks that are process-global state (failure injection, growth delay, trace, device aliasing) are NOT in the product library ...
    import subprocess
    exported = subprocess.check_output(["nm", "-D", "--defined-only", kz.abi.LIB_PATH], text=True)
    assert "kz_debug" not in exported
    assert lib.kz_build_flags() == 0
    # ... they live in the development variant, which exports everything the dev header declares
    if os.path.exists(kz.abi.DEV_LIB_PATH):
        dev_exported = set(re.findall(r" T (kz_[a-z0-9_]+)", subprocess.check_output(["nm", "-D", "--defined-only", kz.abi.DEV_LIB_PATH], text=True)))
        assert declared <= dev_exported, declared - dev_exported


def test_struct_sizes_match_the_header(kz, tmp_path):
    """sizeof of every structure as gcc sees the C header == sizeof of its ctypes mirror."""
    import subprocess
    a = kz.abi
    names = ["KzBSDF", "KzImage", "KzTexture", "KzLight", "KzMesh", "KzFilter", "KzCamera", "KzSampler", "KzIntegrator", "KzBackground",
             "KzSceneDesc", "KzTile", "KzTuning", "KzTileDealer", "KzRenderOpts", "KzPassInfo", "KzStats", "KzHit", "KzBvhInfo", "KzPlanQuery", "KzPlanAnswer", "KzPassModeInfo"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "kazen_mi355x_dev.h"\nint main(void){' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert C.sizeof(getattr(a, n)) == int(sizes[n]), n
    assert C.sizeof(a.KzBSDF) == 128 and C.sizeof(a.KzTexture) == 64
    # structures the library reads or writes THROUGH a caller's pointer carry no size field - their sizes are pinned to the ABI version (6)
    assert C.sizeof(a.KzTileDealer) == 48 and C.sizeof(a.KzPassInfo) == 64 and C.sizeof(a.KzRenderOpts) == 152


def test_render_opts_layout_v5(kz):
    """ABI v5: the v2 / v3 / v4 prefix of KzRenderOpts keeps its offsets (a zero-extended older struct means "defaults"); v4 appended tileDealing,
    v5 names the word behind it (packedOutput) and appends the dealer. KzTuning keeps its 16 words (the experiment fields are dev0..dev5 now)."""
    o = kz.abi.KzRenderOpts
    assert (o.sampleBegin.offset, o.sampleEnd.offset, o.tiles.offset, o.nTiles.offset, o.pipeline.offset, o.accumulate.offset, o.stream.offset) == (0, 4, 8, 16, 20, 24, 32)
    assert o.device.offset == 40 and o.passItems.offset == 48 and o.maxStateBytes.offset == 56 and o.tune.offset == 64
    assert C.sizeof(kz.abi.KzTuning) == 64 and o.tileDealing.offset == 128 and o.packedOutput.offset == 132 and o.dealer.offset == 136 and o.shadowBeside.offset == 144 and o.passHalves.offset == 148 and C.sizeof(o) == 152


@pytest.mark.parametrize("w,h,tile,parts", [(1920, 1080, 64, 8), (3840, 2160, 64, 8), (1920, 1080, 128, 3), (100, 70, 32, 5), (64, 64, 64, 4)])
def test_deal_tiles_partitions_the_image_by_area(kz, w, h, tile, parts):
    """kz_deal_tiles: every pixel in exactly one part, tiles on the 32-px block grid, areas balanced to one tile."""
    cover = np.zeros((h, w), np.int32)
    areas = []
    for p in range(parts):
        tl = kz.shard.deal_tiles(w, h, parts, p, tile)
        assert tl == sorted(tl, key=lambda t: (t[1], t[0]))              # row-major inside a part
        for (x0, y0, tw, th) in tl:
            assert x0 % 32 == 0 and y0 % 32 == 0 and 0 < tw <= tile and 0 < th <= tile
            cover[y0:y0 + th, x0:x0 + tw] += 1
        areas.append(sum(t[2] * t[3] for t in tl))
    assert (cover == 1).all()
    assert max(areas) - min(areas) <= tile * tile
    # 8 GPUs on the C4 / C5 frames: the imbalance the old 128-px round-robin had (ranks of 16 or 17 tiles) is gone
    if parts == 8:
        assert (max(areas) - min(areas)) / (w * h / parts) < 0.02


def test_deal_tiles_rejects_bad_arguments(kz):
    lib = kz.abi.load_library()
    n = C.c_uint32()
    assert lib.kz_deal_tiles(64, 64, 48, 2, 0, None, 0, C.byref(n)) == kz.abi.KZ_ERR_INVALID_ARG       # not a multiple of 32
    assert lib.kz_deal_tiles(64, 64, 32, 2, 2, None, 0, C.byref(n)) == kz.abi.KZ_ERR_INVALID_ARG       # part >= nParts
    assert lib.kz_deal_tiles(64, 64, 32, 2, 0, None, 0, C.byref(n)) == kz.abi.KZ_ERR_INVALID_ARG and n.value == 2   # no room: count still set


def test_film_merge_is_an_elementwise_sum(kz):
    lib = kz.abi.load_library()
    a = np.arange(40, dtype=np.float32)
    b = np.full(40, 0.5, np.float32)
    assert lib.kz_film_merge(a.ctypes.data_as(kz.abi.f32p), b.ctypes.data_as(kz.abi.f32p), 40) == 0
    assert np.array_equal(a, np.arange(40, dtype=np.float32) + 0.5)
    assert lib.kz_film_merge(None, b.ctypes.data_as(kz.abi.f32p), 40) == kz.abi.KZ_ERR_INVALID_ARG


def test_scene_create_and_bvh_on_host(kz):
    sc = kz.Scene(kz.scenes.random_triangles(20000, 64, 64, 4, sampler="independent"))
    info = sc.bvh_info()
    assert info["nTris"] == 20000 + 12 + 16 and info["maxLeafSize"] <= 4 and 0 < info["maxDepth"] <= 30
    assert info["nLeaves"] == info["nNodes"] + 1
    assert (sc.width, sc.height, sc.border) == (64, 64, 2)


def test_depth_cap_on_adversarial_input(kz):
    """A geometric progression of triangle sizes makes plain SAH trees degenerate; the builder must stay
    within the traversal stack."""
    s = kz.scenes.SceneDescription()
    n = 4000
    k = np.arange(n, dtype=np.float64)
    x = 1.0002 ** (k * 8) - 1.0
    V = np.zeros((n, 3, 3), np.float32)
    V[:, 0, 0], V[:, 1, 0], V[:, 2, 0] = x, x + 1e-3, x
    V[:, 2, 1] = 1e-3
    s.add_mesh(V.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3), np.tile([0, 0, 1], (3 * n, 1)).astype(np.float32))
    s.camera.update(width=32, height=32)
    assert kz.Scene(s).bvh_info()["maxDepth"] <= 30


def test_no_device_is_a_loud_error(kz):
    lib = kz.abi.load_library()
    if lib.kz_device_count() > 0:
        pytest.skip("a GPU is visible here")
    sc = kz.Scene(kz.scenes.cornell_box(16, 16, 1))
    with pytest.raises(kz.abi.KzError) as e:
        sc.upload(0)
    assert e.value.code == kz.abi.KZ_ERR_NO_DEVICE
    with pytest.raises(kz.abi.KzError) as e:
        sc.render()
    assert e.value.code == kz.abi.KZ_ERR_STATE
    assert sc.devices() == []
    with pytest.raises(kz.abi.KzError) as e:                      # the multi-device driver fails the same way, from its worker thread
        sc.render_multi([0])
    assert e.value.code == kz.abi.KZ_ERR_NO_DEVICE and "device 0" in str(e.value)


@pytest.mark.parametrize("mutate,code", [
    (lambda s: s.sampler.update(type="halton"), 2),
    (lambda s: s.integrator.update(type="whitted"), 2),
    (lambda s: s.camera.update(type="orthographic"), 2),
    (lambda s: s.sampler.update(type="stratified", resolution=0), 1),
    (lambda s: s.meshes[0].update(bsdf={"type": "principled"}), 2),
    (lambda s: s.sampler.update(sampleCount=0), 1),
    (lambda s: s.camera.update(width=0), 1),
    (lambda s: s.meshes[0]["F"].__setitem__((0, 0), 999), 1),
])
def test_invalid_descriptions(kz, mutate, code):
    s = kz.scenes.cornell_box(16, 16, 1)
    mutate(s)
    with pytest.raises(kz.abi.KzError) as e:
        kz.Scene(s)
    assert e.value.code == code
    assert len(str(e.value)) > 25        # carries a message


def test_null_arguments(kz):
    lib = kz.abi.load_library()
    h = C.c_void_p()
    assert lib.kz_scene_create(None, C.byref(h)) == kz.abi.KZ_ERR_INVALID_ARG
    assert b"null" in lib.kz_last_error()
    assert lib.kz_film_to_rgb(None, 4, 4, 2, None) == kz.abi.KZ_ERR_INVALID_ARG


def test_film_to_rgb_divides_by_weight(kz):
    lib = kz.abi.load_library()
    film = np.zeros((8, 8, 4), np.float32)
    film[2:6, 2:6] = (2.0, 4.0, 6.0, 2.0)
    film[3, 3] = (1.0, 1.0, 1.0, 0.0)             # zero weight -> black (color.h:94-99)
    rgb = np.zeros((4, 4, 3), np.float32)
    assert lib.kz_film_to_rgb(film.ctypes.data_as(kz.abi.f32p), 4, 4, 2, rgb.ctypes.data_as(kz.abi.f32p)) == 0
    assert np.allclose(rgb[0, 0], (1, 2, 3)) and np.allclose(rgb[1, 1], 0)


def test_scene_generators(kz):
    S = kz.scenes
    assert S.sphere_env(8, 8, 1).n_tris() == 9800
    c = S.cornell_box(8, 8, 1)
    assert c.n_tris() == 36
    for m in c.meshes:                            # vertex normals are unit length and every face index is valid
        assert np.allclose(np.linalg.norm(m["N"], axis=1), 1, atol=1e-6) and m["F"].max() < len(m["V"])
    r = S.random_triangles(800, 8, 8, 1)
    assert r.n_tris() == 800 + 12 + 16 and sum(1 for m in r.meshes if m["light"]) == 8
    P, N, UV, F = S.box((-1, -1, -1), (1, 1, 1), inward=True)
    ctr = P[F].mean(axis=1)
    assert (np.einsum("ij,ij->i", N[F[:, 0]], -ctr) > 0).all()        # inward normals point at the centre
    pmj, bn = S.make_pmj02bn_tables()
    assert pmj.shape == (5, 65536, 2) and bn.shape == (48, 128, 128)
    for k in (4, 16, 64, 256, 1024):                                   # every 4^k prefix... first set: (0,2) net per axis
        g = int(np.sqrt(k))
        x = (pmj[0, :k, 0].astype(np.float64) / 2 ** 32 * g).astype(int)
        y = (pmj[0, :k, 1].astype(np.float64) / 2 ** 32 * g).astype(int)
        assert len(set(zip(x, y))) == k
    assert float((pmj.astype(np.float64) * 2.0 ** -32).astype(np.float32).max()) < 1.0


def test_dither_textures_are_blue_noise(kz):
    """the stand-in for BlueNoiseTextures[48][128][128] (bluenoise.h:8-11, blob missing) is void-and-cluster blue noise, not the
    white noise of rounds 1-4: every texture holds each rank once, its low-frequency power is a thousandth of white noise's, the committed file is what
    the generator mints (texture 0 re-minted here), and thresholding at any level leaves no two points side by side that white noise would."""
    S = kz.scenes
    bn = S.blue_noise_textures()
    n = kz.abi.KZ_BLUENOISE_RES
    assert bn.shape == (48, n, n) and bn.dtype == np.uint16
    assert np.array_equal(S._mint_one(0), bn[0])
    white = S._white_noise_textures(2022)

    def band_power(t, lo, hi):
        f = np.abs(np.fft.fft2(t.astype(np.float64) / 65535 - 0.5)) ** 2
        ky, kx = np.meshgrid(np.fft.fftfreq(n) * n, np.fft.fftfreq(n) * n, indexing="ij")
        k = np.sqrt(kx * kx + ky * ky)
        return f[(k >= lo) & (k < hi)].mean()
    for t in (0, 17, 47):
        assert len(np.unique(bn[t])) == n * n                                      # ranks: a permutation, spread over the uint16 range
        assert band_power(bn[t], 1, 16) < 1e-3 * band_power(white[t], 1, 16)
        assert band_power(bn[t], 32, 64) > band_power(white[t], 32, 64)            # the energy sits at high frequencies instead
        for level in (0.03, 0.1):                                                 # no clusters: far fewer 4-neighbour pairs than a random set of that density
            def pairs(tex):
                m = tex < level * 65535
                return int((m & np.roll(m, 1, 0)).sum() + (m & np.roll(m, 1, 1)).sum())
            assert pairs(bn[t]) < 0.5 * pairs(white[t])
    assert len({bn[t].tobytes() for t in range(48)}) == 48                          # 48 different textures
    pmj, bn2 = S.make_pmj02bn_tables()
    assert bn2 is bn and np.array_equal(S.make_pmj02bn_tables(dither="white")[1], white)


def test_film_merge_tiles_adds_rects_in_list_order(kz):
    """kz_film_merge_tiles (host only): packed (h+2b) x (w+2b) rects are added into the film where the tiles sit, aprons overlapping;
    the result does not depend on the number of host threads."""
    lib = kz.abi.load_library()
    W, H, b = 100, 70, 2
    tiles = [(0, 0, 64, 64), (64, 0, 36, 64), (0, 64, 64, 6), (64, 64, 36, 6)]
    rng = np.random.default_rng(5)
    rects = [rng.random((t[3] + 2 * b, t[2] + 2 * b, 4), dtype=np.float32) for t in tiles]
    packed = np.concatenate([r.ravel() for r in rects])
    want = np.zeros((H + 2 * b, W + 2 * b, 4), np.float32)
    for t, r in zip(tiles, rects):
        want[t[1]:t[1] + t[3] + 2 * b, t[0]:t[0] + t[2] + 2 * b] += r
    arr = (kz.abi.KzTile * len(tiles))(*[kz.abi.KzTile(*t) for t in tiles])
    for threads in (1, 3, 0):
        film = np.zeros_like(want)
        assert lib.kz_film_merge_tiles(film.ctypes.data_as(kz.abi.f32p), W, H, b, arr, len(tiles), packed.ctypes.data_as(kz.abi.f32p), packed.size, threads) == 0
        assert np.array_equal(film, want), threads
    film = np.zeros_like(want)
    assert lib.kz_film_merge_tiles(film.ctypes.data_as(kz.abi.f32p), W, H, b, arr, len(tiles), packed.ctypes.data_as(kz.abi.f32p), packed.size - 4, 1) == kz.abi.KZ_ERR_INVALID_ARG
    bad = (kz.abi.KzTile * 1)(kz.abi.KzTile(90, 0, 64, 64))
    assert lib.kz_film_merge_tiles(film.ctypes.data_as(kz.abi.f32p), W, H, b, bad, 1, packed.ctypes.data_as(kz.abi.f32p), 68 * 68 * 4, 1) == kz.abi.KZ_ERR_INVALID_ARG


def test_merge_rects_equals_merge_tiles_and_checks_its_arguments(kz):
    """kz_film_merge_rects (round 6: the rects of a list may lie in different buffers - every rank's in that rank's shared-memory file) against kz_film_merge_tiles on the same
    rects, ragged tiles and overlapping aprons included; any number of band threads; null and out-of-frame arguments are errors, never a crash."""
    lib = kz.abi.load_library()
    rng = np.random.default_rng(3)
    w, h, b = 150, 100, 2
    tiles = kz.shard.deal_tiles(w, h, 1, 0, 64)                           # 3 x 2 tiles, ragged at the right and bottom edges
    rects = [rng.random((t[3] + 2 * b) * (t[2] + 2 * b) * 4, dtype=np.float32) for t in tiles]
    arr = (kz.abi.KzTile * len(tiles))(*[kz.abi.KzTile(*t) for t in tiles])
    packed = np.concatenate(rects)
    f1 = np.zeros(((h + 2 * b), (w + 2 * b), 4), np.float32)
    assert lib.kz_film_merge_tiles(f1.ctypes.data_as(kz.abi.f32p), w, h, b, arr, len(tiles), packed.ctypes.data_as(kz.abi.f32p), packed.size, 1) == 0
    for threads in (1, 3, 16):
        f2 = np.zeros_like(f1)
        ptrs = (kz.abi.f32p * len(tiles))(*[C.cast(r.ctypes.data, kz.abi.f32p) for r in rects])
        assert lib.kz_film_merge_rects(f2.ctypes.data_as(kz.abi.f32p), w, h, b, arr, ptrs, len(tiles), threads) == 0
        assert np.array_equal(f1, f2)
    # the order of the list is the order of the additions: reversed, the apron texels may differ in the last bit - and never by more
    rev = (kz.abi.KzTile * len(tiles))(*[kz.abi.KzTile(*t) for t in tiles[::-1]])
    ptrs = (kz.abi.f32p * len(tiles))(*[C.cast(r.ctypes.data, kz.abi.f32p) for r in rects[::-1]])
    f3 = np.zeros_like(f1)
    assert lib.kz_film_merge_rects(f3.ctypes.data_as(kz.abi.f32p), w, h, b, rev, ptrs, len(tiles), 0) == 0
    assert np.allclose(f3, f1, rtol=1e-6, atol=0)
    bad = (kz.abi.f32p * len(tiles))(*([C.cast(r.ctypes.data, kz.abi.f32p) for r in rects[:-1]] + [None]))
    assert lib.kz_film_merge_rects(f3.ctypes.data_as(kz.abi.f32p), w, h, b, arr, bad, len(tiles), 0) == kz.abi.KZ_ERR_INVALID_ARG
    out = (kz.abi.KzTile * 1)(kz.abi.KzTile(128, 64, 64, 64))
    assert lib.kz_film_merge_rects(f3.ctypes.data_as(kz.abi.f32p), w, h, b, out, ptrs, 1, 0) == kz.abi.KZ_ERR_INVALID_ARG
    assert lib.kz_film_merge_rects(None, w, h, b, arr, ptrs, len(tiles), 0) == kz.abi.KZ_ERR_INVALID_ARG
    assert lib.kz_film_merge_rects(f3.ctypes.data_as(kz.abi.f32p), w, h, b, arr, None, len(tiles), 0) == kz.abi.KZ_ERR_INVALID_ARG


def test_planner_entry_points_check_their_arguments(kz):
    lib = kz.abi.load_library()
    q, a = kz.abi.KzPlanQuery(), kz.abi.KzPlanAnswer()
    assert lib.kz_plan_passes(None, C.byref(a)) == kz.abi.KZ_ERR_INVALID_ARG
    assert lib.kz_plan_passes(C.byref(q), C.byref(a)) == kz.abi.KZ_ERR_INVALID_ARG            # an empty call
    q.nPix, q.sampleBegin, q.sampleEnd, q.limitBytes, q.dealer = 1000, 0, 8, 1 << 30, 1          # a dealer without a tile list
    assert lib.kz_plan_passes(C.byref(q), C.byref(a)) == kz.abi.KZ_ERR_INVALID_ARG
    q.dealer = 0
    assert lib.kz_plan_passes(C.byref(q), C.byref(a)) == 0 and a.need == 8000 and a.nPasses == 1
    n = C.c_uint32()
    av = (C.c_uint64 * 1)(0)                                                                    # a context that holds nothing: an error, not a loop
    assert lib.kz_plan_schedule(C.byref(q), av, 1, 0, 1000, None, 0, C.byref(n)) != 0
