"""CPU tests of the oracle (the checker itself): pinned against the reference's own headers where they compile
(integer KATs minted by oracle/kat_ref_main.cpp from hash.h / pcg32.h), against committed fp32 vectors, against a
brute-force Moeller-Trumbore search, and through domain properties."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(HERE, "golden", "int_kats.json")))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "fp_goldens.npz"))


# ---- integer functions: bit-exact against the reference's headers ---------------------------------
def test_hash_pixel_seed_matches_reference_header(O, kats):
    L = O.lib()
    for x, y, s, h in kats["hash_pixel_seed"]:
        assert L.kzo_hash_pixel_seed(x, y, int(s)) == int(h)
    # Hash({3,5}, u64 0) == 0xd539d46ed3159a89
    assert L.kzo_hash_pixel_seed(3, 5, 0) == 0xd539d46ed3159a89


def test_hash_pixel_dim_seed_matches_reference_header(O, kats):
    L = O.lib()
    for x, y, d, s, h in kats["hash_pixel_dim_seed"]:
        assert L.kzo_hash_pixel_dim_seed(x, y, d, int(s)) == int(h)


def test_murmur_all_tail_lengths(O, kats):
    L = O.lib()
    buf = bytes((i * 37 + 11) & 255 for i in range(40))
    for ln, s, h in kats["murmur64a"]:
        assert L.kzo_murmur64a(buf, ln, int(s)) == int(h), ln


def test_mixbits(O, kats):
    L = O.lib()
    for v, h in kats["mixbits"]:
        assert L.kzo_mixbits(int(v)) == int(h)
    assert L.kzo_mixbits(42) == 0x4942410a1f55400a


def test_pcg32_streams_bit_exact(O, kats):
    L = O.lib()
    for e in kats["pcg32_stream"]:
        u = np.zeros(8, np.uint32)
        f = np.zeros(8, np.float32)
        L.kzo_pcg32_stream(int(e["initseq"]), int(e["delta"]), 8, u.ctypes.data_as(O.abi.u32p), f.ctypes.data_as(O.abi.f32p), None)
        assert u.tolist() == e["u"]
        assert f.view(np.uint32).tolist() == e["fbits"]
    for e in kats["pcg32_seed2"]:
        u = np.zeros(4, np.uint32)
        L.kzo_pcg32_seed2(int(e["initstate"]), int(e["initseq"]), 4, u.ctypes.data_as(O.abi.u32p))
        assert u.tolist() == e["u"]


def test_survey_pcg_floats(O):
    """SURVEY 8c probed the reference header: after seed(Hash({3,5},0)), advance(2*65536), Point2f(nextFloat(),
    nextFloat()) printed (0.828320861, 0.502115607) under g++, i.e. x is the SECOND draw (H1)."""
    L = O.lib()
    u = np.zeros(2, np.uint32)
    f = np.zeros(2, np.float32)
    L.kzo_pcg32_stream(0xd539d46ed3159a89, 2 * 65536, 2, u.ctypes.data_as(O.abi.u32p), f.ctypes.data_as(O.abi.f32p), None)
    assert abs(float(f[0]) - 0.502115607) < 1e-7 and abs(float(f[1]) - 0.828320861) < 1e-7


def test_python_pcg32_equals_oracle(O, kz):
    f = kz.scenes.pcg32_floats(1, 1000)
    u = np.zeros(1000, np.uint32)
    g = np.zeros(1000, np.float32)
    O.lib().kzo_pcg32_stream(1, 0, 1000, u.ctypes.data_as(O.abi.u32p), g.ctypes.data_as(O.abi.f32p), None)
    assert np.array_equal(f, g)


def test_permute_is_a_bijection(O):
    L = O.lib()
    for l in (1, 2, 7, 16, 100, 1024):
        for p in (0, 1, 0xdeadbeef, 0x12345678):
            out = sorted(L.kzo_permute(i, l, p) for i in range(l))
            assert out == list(range(l))


def test_tea32_known_shape(O):
    L = O.lib()
    a, b = L.kzo_tea32(1, 2, 4), L.kzo_tea32(1, 2, 4)
    assert a == b and a != L.kzo_tea32(2, 1, 4) and L.kzo_tea32(5, 9, 0) == (9 << 32) + 5


# ---- fp32 functions: committed vectors + properties --------------------------------------------------
def test_fp_goldens_reproduce(O, kz, gold):
    """The committed vectors are regenerated bit-for-bit by the current oracle build."""
    L = O.lib()
    w = np.zeros(3, np.float32)
    for i in range(0, 256, 5):
        L.kzo_cosine_hemisphere(gold["warp_u"][i, 0], gold["warp_u"][i, 1], w.ctypes.data_as(O.abi.f32p))
        assert np.array_equal(w, gold["warp_cos"][i])
    S = kz.scenes
    for tag, desc in (("ind", S.cornell_box(32, 32, 4)), ("pmj", S.cornell_box(32, 32, 4, sampler="pmj02bn", seed=1))):
        o = O.OracleScene(desc)
        assert np.array_equal(o.sampler_stream(3, 5, 2, 12), gold["stream_" + tag][1])
        assert np.array_equal(o.render_samples(gold["samples_pxy"], gold["samples_idx"]), gold["samples_" + tag])
        assert np.array_equal(o.render(threads=1), gold["film_" + tag])


def test_cosine_hemisphere_properties(gold):
    v = gold["warp_cos"]
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=2e-6)
    assert (v[:, 2] > 0).all()
    assert np.allclose(v[0], (0, 0, 1))          # centre of the square maps to the pole


def test_frame_is_orthonormal(gold):
    n, s, t = gold["frame_n"], gold["frame_s"], gold["frame_t"]
    assert np.allclose(np.einsum("ij,ij->i", n, s), 0, atol=1e-6)
    assert np.allclose(np.einsum("ij,ij->i", n, t), 0, atol=1e-6)
    assert np.allclose(np.einsum("ij,ij->i", s, t), 0, atol=1e-6)
    assert np.allclose(np.cross(s, t), n, atol=1e-6)          # (s, t, n) right handed


def test_bsdf_sample_consistency(gold):
    """sample() returns eval/pdf at the sampled direction; back-side queries are zero (bsdf.cpp:1217-1218,1302-1303)."""
    sw = gold["bsdf_sample"]
    assert (sw[:, 1, :3] == 0).all() and (sw[:, 1, 6] == 0).all()        # wi[1] is below the surface
    assert (gold["bsdf_eval"][:, 1] == 0).all() and (gold["bsdf_pdf"][:, 1] == 0).all()
    ok = sw[:, :, 6] > 0
    assert np.isfinite(sw[ok]).all() and (sw[ok][:, :3] >= 0).all()
    # diffuse row: weight == albedo, pdf == cos/pi
    assert np.allclose(sw[5][ok[5]][:, :3], (0.5, 0.25, 0.125))
    assert np.allclose(gold["bsdf_pdf"][5][2:], gold["bsdf_wo"][2:, 2] / np.pi, rtol=1e-6)


def test_kiss_pdf_integrates_to_at_most_one(O, kz):
    """The mixture pdf is a density over the upper hemisphere: its Monte-Carlo integral is <= 1 (+noise)."""
    row = kz.scenes.kazenstandard((0.8, 0.8, 0.8), 0.6, 0.0, clearcoat=1.0)
    rng = np.random.default_rng(3)
    wi = np.array([0.3, 0.2, 0.93], np.float32)
    wi /= np.linalg.norm(wi)
    n = 4000
    z = rng.random(n)
    ph = 2 * np.pi * rng.random(n)
    r = np.sqrt(1 - z * z)
    wo = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)
    est = np.mean([O.bsdf(row, "pdf", wi, w) for w in wo]) * 2 * np.pi
    assert 0.85 < est < 1.1, est


def test_filter_table_matches_block_cpp(gold):
    tab = gold["filter_table"]
    x = 2.0 * np.arange(32, dtype=np.float32) / 32
    ref = np.maximum(0, np.exp(-2.0 * x * x) - np.exp(-8.0))       # alpha = -1/(2*0.25)
    assert np.allclose(tab[:32], ref, rtol=1e-5, atol=1e-8) and tab[32] == 0


def test_light_sample_geometry(gold):
    ls = gold["light_samples"]
    p, n, wi, pdf = ls[:, 0:3], ls[:, 3:6], ls[:, 6:9], ls[:, 9]
    assert np.allclose(p[:, 1], 0.99, atol=1e-6) and (np.abs(p[:, 0]) <= 0.25 + 1e-6).all()
    assert np.allclose(n, (0, -1, 0))
    d = p - gold["light_ref"]
    dist2 = np.sum(d * d, 1)
    cos = -np.einsum("ij,ij->i", n, wi)
    assert np.allclose(pdf, (1 / 0.25) * dist2 / cos, rtol=1e-5)          # light.cpp:47-48, area 0.5 x 0.5
    assert np.allclose(ls[:, 10:13], 15.0 / pdf[:, None], rtol=1e-5)


# ---- traversal: BVH vs brute force ---------------------------------------------------------------------
@pytest.mark.parametrize("scene_name", ["cornell", "sphere", "soup"])
def test_oracle_bvh_equals_brute_force(O, kz, scene_name):
    S = kz.scenes
    desc = {"cornell": lambda: S.cornell_box(32, 32, 1), "sphere": lambda: S.sphere_env(32, 32, 1),
            "soup": lambda: S.random_triangles(3000, 32, 32, 1, sampler="independent", s_edge=0.15)}[scene_name]()
    a, b = O.OracleScene(desc), O.OracleScene(desc, brute=True)
    rng = np.random.default_rng(7)
    n = 3000
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 50)]        # axis-aligned rays (zero components, inf reciprocals)
    ha, hb = a.trace_rays(o, d, 1e-3, np.inf), b.trace_rays(o, d, 1e-3, np.inf)
    for k in ("t", "u", "v", "mesh", "prim", "p", "sh_n", "sh_s", "uv"):
        assert np.array_equal(ha[k], hb[k], equal_nan=True), k
    assert (ha["mesh"] >= 0).mean() > 0.3


def test_oracle_tie_break_is_order_independent(O, kz):
    """Two coincident triangles: the lower (mesh, face) id wins, by BVH and by brute force."""
    s = kz.scenes.SceneDescription()
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    N = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    F = np.array([[0, 1, 2]], np.uint32)
    for _ in range(3):
        s.add_mesh(V, F, N)
    s.camera.update(width=32, height=32)
    for brute in (False, True):
        h = O.OracleScene(s, brute=brute).trace_rays([[0.2, 0.2, 1.0]], [[0, 0, -1]], 0, np.inf)
        assert h["mesh"][0] == 0 and h["t"][0] == 1.0


# ---- film -----------------------------------------------------------------------------------------------
def test_film_weight_equals_sample_count_times_filter_mass(O, kz):
    """Every valid sample adds the same total filter weight (sum of wx * sum of wy), whatever its radiance."""
    desc = kz.scenes.sphere_env(48, 40, 3)
    o = O.OracleScene(desc)
    film = o.render(threads=2)
    assert film.shape == (44, 52, 4) and o.stats()["droppedSamples"] == 0
    assert film[..., 3].sum() > 0 and np.isfinite(film).all()
    rgb = o.rgb(film)
    assert rgb.shape == (40, 48, 3) and rgb.min() >= 0


def test_film_tiles_sum_to_whole(O, kz):
    desc = kz.scenes.cornell_box(64, 64, 2)
    o = O.OracleScene(desc)
    whole = o.render(threads=2)
    tiles = kz.shard.make_tiles(64, 64, 32)
    parts = [o.render(tiles=kz.shard.tiles_for_rank(tiles, r, 2), threads=2) for r in range(2)]
    assert np.allclose(kz.shard.merge_films(parts), whole, rtol=1e-6, atol=1e-7)


def test_sample_ranges_add_up(O, kz):
    desc = kz.scenes.cornell_box(32, 32, 4)
    o = O.OracleScene(desc)
    whole = o.render(threads=1)
    acc = o.render(0, 2, threads=1)
    o.render(2, 4, threads=1, film=acc)
    assert np.allclose(acc, whole, rtol=1e-6, atol=1e-7)


def test_pmj02bn_pixel_samples_are_stratified(O, kz):
    """nextPixel2D of the 16 samples of a pixel falls in the 16 cells of a 4x4 grid (the (0,2) property the
    PMJ02BN constructor relies on, sampler.cpp:295-314)."""
    desc = kz.scenes.cornell_box(32, 32, 16, sampler="pmj02bn", seed=1)
    o = O.OracleScene(desc)
    for px, py in ((0, 0), (5, 9), (31, 31)):
        j = np.array([o.sampler_stream(px, py, i, 0)[:2] for i in range(16)])
        cells = set((int(x * 4), int(y * 4)) for x, y in j)
        assert len(cells) == 16 and (j >= 0).all() and (j < 1).all()


def test_unsupported_plugins_are_errors(O, kz):
    s = kz.scenes.cornell_box(16, 16, 1)
    s.sampler["type"] = "halton"
    with pytest.raises(kz.abi.KzError) as e:
        O.OracleScene(s)
    assert e.value.code == kz.abi.KZ_ERR_UNSUPPORTED


# ---- "next" rows of SURVEY 8f: stratified / correlated samplers, thin-lens camera -------------------------------------
@pytest.mark.parametrize("kind,spp,expect", [("stratified", 16, 16), ("stratified", 5, 16), ("stratified", 20, 25), ("correlated", 16, 16), ("correlated", 12, 12), ("correlated", 7, 8)])
def test_sample_count_rounding_follows_the_constructors(O, kz, kind, spp, expect):
    """Stratified: resolution starts at 4 and grows until res^2 >= sampleCount (sampler.cpp:86-92); Correlated: ry = floor(sqrt(n)),
    rx = ceil(n / ry) (sampler.cpp:181-187). The library and the oracle report the same rounded count."""
    d = kz.scenes.cornell_box(16, 16, spp, sampler=kind, seed=1)
    assert O.OracleScene(d).sample_count == expect
    assert kz.Scene(d).sample_count == expect


@pytest.mark.parametrize("kind", ["stratified", "correlated"])
def test_jittered_samplers_are_stratified(O, kz, kind):
    """The 16 pixel samples of one pixel: one per cell of the 4x4 grid; correlated multi-jitter also has one per 1/16 column and row."""
    o = O.OracleScene(kz.scenes.cornell_box(16, 16, 16, sampler=kind, seed=1))
    for px, py in ((0, 0), (7, 3), (15, 15)):
        j = np.array([o.sampler_stream(px, py, i, 2)[:2] for i in range(16)])
        assert (j >= 0).all() and (j < 1).all()
        assert len(set((int(x * 4), int(y * 4)) for x, y in j)) == 16
        if kind == "correlated":
            assert len(set(int(x * 16) for x, _ in j)) == 16 and len(set(int(y * 16) for _, y in j)) == 16
        # 1-D draws of the same dimension over the 16 samples: one per 1/16 stratum
        u = np.array([o.sampler_stream(px, py, i, 2)[4] for i in range(16)])
        assert sorted((u * 16).astype(int).tolist()) == list(range(16))


def test_thinlens_reduces_to_pinhole_at_zero_aperture(O, kz):
    a = kz.scenes.cornell_box(24, 24, 2)
    b = kz.scenes.cornell_box(24, 24, 2)
    b.camera.update(type="thinlens", apertureRadius=0.0, focusDistance=3.0)
    fa, fb = O.OracleScene(a).render(threads=1), O.OracleScene(b).render(threads=1)
    assert np.allclose(fa, fb, rtol=1e-4, atol=1e-5)
    c = kz.scenes.cornell_box(24, 24, 2)
    c.camera.update(type="thinlens", apertureRadius=0.3, focusDistance=3.6)
    fc = O.OracleScene(c).render(threads=1)
    assert np.isfinite(fc).all() and not np.allclose(fa, fc, rtol=1e-2, atol=1e-3)


def test_discrete_bsdfs(O, kz):
    """mirror / dielectric (bsdf.cpp:98-196): eval = pdf = 0, sample is a delta lobe; energy conserving weights of 1."""
    S = kz.scenes
    wi = np.array([0.3, -0.2, 0.93], np.float32)
    wi /= np.linalg.norm(wi)
    assert (O.bsdf(S.mirror(), "eval", wi, (0, 0, 1)) == 0).all() and O.bsdf(S.mirror(), "pdf", wi, (0, 0, 1)) == 0
    w, wo, ok = O.bsdf(S.mirror(), "sample", wi, None, 0, 0.3, (0.1, 0.2))
    assert ok and np.allclose(w, 1) and np.allclose(wo, (-wi[0], -wi[1], wi[2]))
    w, wo, ok = O.bsdf(S.mirror(), "sample", -wi, None, 0, 0.3, (0.1, 0.2))
    assert not ok and np.allclose(w, 0)
    g = S.dielectric()
    w, wo, ok = O.bsdf(g, "sample", wi, None, 0, 0.999, (0.1, 0.2))            # transmit
    assert ok and np.allclose(w, 1) and wo[2] < 0 and abs(np.linalg.norm(wo) - 1) < 1e-5
    # Snell: sin(t) = sin(i) * ext/int
    assert abs(np.hypot(wo[0], wo[1]) - np.hypot(wi[0], wi[1]) * 1.000277 / 1.5046) < 1e-5
    w, wo2, ok = O.bsdf(g, "sample", wi, None, 0, 0.0, (0.1, 0.2))             # reflect (fresnel > 0)
    assert np.allclose(wo2, (-wi[0], -wi[1], wi[2]))
    graz = np.array([0.9999, 0, -0.0141], np.float32)                          # from inside at grazing angle: total internal reflection
    w, wo3, ok = O.bsdf(g, "sample", graz, None, 0, 0.9999, (0.1, 0.2))
    assert np.allclose(wo3, (-graz[0], -graz[1], graz[2]))
    film = O.OracleScene(S.glass_scene(32, 32, 4)).render(threads=2)
    assert np.isfinite(film).all() and film[..., :3].sum() > 0


def test_rough_bsdfs_are_sane(O, kz):
    """ggx / roughconductor / roughplastic / roughdielectric (bsdf.cpp:629-1145): reciprocity-free sanity — finite, non-negative,
    sample() == eval/pdf for the reflection models, pdf integrates to <= 1, glass transmits to the other side."""
    S = kz.scenes
    rng = np.random.default_rng(5)
    wi = np.array([0.35, 0.1, 0.93], np.float32)
    wi /= np.linalg.norm(wi)
    for row in (S.ggx((0.8, 0.8, 0.8), 0.4), S.roughconductor(0.3, "Cu"), S.roughplastic(0.3)):
        for _ in range(40):
            s = rng.random(3)
            w, wo, ok = O.bsdf(row, "sample", wi, None, 0.0, float(s[0]), (float(s[1]), float(s[2])))
            assert ok and np.isfinite(w).all() and (w >= 0).all()
            if np.any(w > 0) and row["type"] != "ggx":
                e, p = O.bsdf(row, "eval", wi, wo), O.bsdf(row, "pdf", wi, wo)
                assert np.allclose(w, e / p, rtol=1e-5)
        n = 3000
        z = rng.random(n)
        ph = 2 * np.pi * rng.random(n)
        r = np.sqrt(1 - z * z)
        dirs = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1).astype(np.float32)
        est = np.mean([O.bsdf(row, "pdf", wi, d) for d in dirs]) * 2 * np.pi
        assert 0.6 < est < 1.15, (row["type"], est)
    glass = S.roughdielectric(0.3)
    below = 0
    for _ in range(200):
        s = rng.random(3)
        w, wo, ok = O.bsdf(glass, "sample", wi, None, 0.0, float(s[0]), (float(s[1]), float(s[2])))
        assert np.isfinite(w).all()
        below += int(np.any(w > 0) and wo[2] < 0)
    assert below > 100                                   # mostly transmission at near-normal incidence
    film = O.OracleScene(S.materials_scene(48, 32, 4)).render(threads=4)
    assert np.isfinite(film).all() and film[..., :3].sum() > 0


def test_permute_and_tea32_match_reference_text(O, kats):
    """random::permute / sampleTEA32 (common.cpp:304-344): vectors minted by compiling the reference's own text of that block
    (oracle/kat_ref_permute.cpp); keys are the callers' 64-bit expressions, truncated to uint32 as the call does."""
    L = O.lib()
    for i, l, key64, out in kats["permute"]:
        assert L.kzo_permute(i, l, int(key64) & 0xffffffff) == out, (i, l, key64)
    for e in kats["permute_full"]:
        got = [L.kzo_permute(i, e["l"], e["key"]) for i in range(e["l"])]
        assert got == e["out"]
        assert sorted(got) == list(range(e["l"]))                      # a permutation
    for v0, v1, rounds, out in kats["tea32"]:
        assert L.kzo_tea32(v0, v1, rounds) == int(out)


def test_fresnel_functions_match_reference_text_bit_for_bit(O, kats):
    """fresnel(cosThetaI, extIOR, intIOR) and fresnelDielectric(cosThetaI, eta, cosThetaT) (common.cpp:447-475, :492-518; the dielectric and rough BSDFs of
    SURVEY 8f rank 2): the one piece of the reference's FLOATING-POINT code that compiles without Eigen. Vectors = float bit patterns minted by compiling the
    reference's own text (oracle/kat_ref_fresnel.cpp); the restatement must give the same bits."""
    import ctypes as C
    L = O.lib()
    f = lambda u: np.array([u], np.uint32).view(np.float32)[0]
    b = lambda x: int(np.array([x], np.float32).view(np.uint32)[0])
    for c, e, i, out in kats["fresnel_ior"]:
        assert b(L.kzo_fresnel_ior(f(c), f(e), f(i))) == out, (f(c), f(e), f(i))
    for c, e, out, ct in kats["fresnel_dielectric"]:
        t = C.c_float(123.0)
        assert b(L.kzo_fresnel_dielectric(f(c), f(e), C.byref(t))) == out and b(t.value) == ct, (f(c), f(e))


def test_shadow_tie_modes_bracket_the_literal_loop(O, kz):
    """integrator.cpp:262-278: once a shadow ray has walked through an invisible light its far end (maxt - t, origin moved by t + eps) lies exactly
    on the sampled light - the reference's own tie. The oracle's two tie modes (tests only) decide every tie one way; the literal loop lies between
    them, they differ only where visible lights are sampled through invisible ones, and the scene without invisible lights has no tie at all."""
    from test_gpu_parity import _fuzz_scene
    desc = _fuzz_scene(kz.scenes, 5084)
    ora = O.OracleScene(desc)
    c = ora.rgb(ora.render(threads=0))
    ora.set_tie_mode(+1); lo = ora.rgb(ora.render(threads=0))
    ora.set_tie_mode(-1); hi = ora.rgb(ora.render(threads=0))
    assert (lo <= c + 1e-6).all() and (c <= hi + 1e-6).all()
    assert 0 < ((hi - lo).max(axis=2) > 1e-4).mean() < 0.5
    for m in desc.meshes:
        if m["light"]:
            m["light"]["lightPrimaryVisibility"] = True
    ora = O.OracleScene(desc)
    ora.set_tie_mode(+1); lo = ora.render(threads=0)
    ora.set_tie_mode(-1); hi = ora.render(threads=0)
    assert np.array_equal(lo, hi)


def test_transcendental_definitions_are_the_correctly_rounded_values(O):
    """oracle/kz_oracle_math.h defines sin / cos / tan / exp / log / atan / atan2 / acos / pow / hypot as fixed double sequences + one narrowing. Against the C
    library's DOUBLE functions narrowed to float (the correctly rounded value but for ~1 argument in 2^28) they must agree on every argument of a dense
    sample of the path's ranges; the census against the C library's FLOAT functions - what the reference's text executes on this machine - is printed: those
    are correctly rounded for 84 ... 99.9 % of the arguments (glibc 2.35: atan2f 84 %, acosf 92 %, tanf 96 %, sinf / cosf 98.7 %, expf / logf / powf 99.9 %),
    which is why "what libm returns" cannot be a bit-exact parity target and the correctly rounded value is."""
    rng = np.random.default_rng(5)
    n = 400000
    f64 = lambda a: a.astype(np.float64)
    x7 = rng.uniform(-7, 7, n).astype(np.float32)
    xe = (-np.exp(rng.uniform(-12, 4.4, n))).astype(np.float32)
    xl = np.exp(rng.uniform(-20, 20, n)).astype(np.float32)
    xa = (np.exp(rng.uniform(-15, 15, n)) * rng.choice([-1, 1], n)).astype(np.float32)
    xc = rng.uniform(-1, 1, n).astype(np.float32)
    y2, x2 = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    xp = rng.uniform(0.003, 50, n).astype(np.float32); pe = rng.choice(np.array([2.4, 1 / 2.4], np.float32), n)
    cases = [("sin", (x7,), np.sin(f64(x7))), ("cos", (x7,), np.cos(f64(x7))), ("tan", (x7,), np.tan(f64(x7))), ("exp", (xe,), np.exp(f64(xe))), ("log", (xl,), np.log(f64(xl))),
             ("atan", (xa,), np.arctan(f64(xa))), ("acos", (xc,), np.arccos(f64(xc))), ("atan2", (y2, x2), np.arctan2(f64(y2), f64(x2))),
             ("pow", (xp, pe), np.power(f64(xp), f64(pe))), ("hypot", (y2, x2), np.hypot(f64(y2), f64(x2))), ("cube", (xc,), f64(xc) ** 3)]
    for name, args, cr in cases:
        got = O.math_fn(name, *args)
        cr = cr.astype(np.float32)
        assert int((got.view(np.uint32) != cr.view(np.uint32)).sum()) <= 1, name
        libm = O.math_fn(name, *args, libm=True)
        print("%-6s libm float function differs from the correctly rounded value on %.3f %% of %d arguments" % (name, 100 * float((libm.view(np.uint32) != got.view(np.uint32)).mean()), n))


def test_both_statements_of_the_transcendentals_are_the_same_sequences():
    """csrc/kz_crmath.h (HIP) and oracle/kz_oracle_math.h (oracle) define the ten functions as the SAME sequences of double operations; the GPU test compares their results bit for bit,
    this one compares the statements themselves (code lines with the device attributes, the prefixes and the coefficient pinning taken out), so that a drift shows on a box without a GPU."""
    import re
    root = os.path.dirname(HERE)

    def norm(path, device):
        src = open(path).read()
        src = src[src.index("#pragma once"):]
        lines = []
        for l in src.splitlines():
            l = l.split("//")[0].rstrip()
            if not l.strip() or l.startswith("#"):
                continue
            if device:
                l = re.sub(r"KZ_K\(([^()]*)\)", r"\1", l)
                l = l.replace("kzcr", "kzom")
                l = re.sub(r"\bKZ_CR_(FN|CALL)\b", "KZO_FN", l)
                l = re.sub(r"\bkz(SinCos|Cos|Tan|Exp|Log|Pow|Atan2|Atan|Acos|Hypot|Cube)\b", r"kzo\1", l)
            lines.append(re.sub(r"\s+", " ", l).strip())
        return [l for l in lines if "kzomK(" not in l]                     # (the pinning helper exists on the device only)
    dev = norm(os.path.join(root, "nano-kazen_amd", "csrc", "kz_crmath.h"), True)
    ora = norm(os.path.join(root, "oracle", "kz_oracle_math.h"), False)
    assert len(ora) > 100 and dev == ora, [(a, b) for a, b in zip(dev, ora) if a != b][:3]


def test_resolved_alpha_rows_equal_raw_rows(kz, O):
    """KzBSDF.alphaResolved (ABI v5, round 5): what an adapter inside a kazen tree can read of the three rough BSDFs is the constructor's
    m_alpha = max(0.001, sqr(property)) (bsdf.cpp:696-700, :818-822, :956-959), not the property. A row that carries it (alphaResolved = 1) must be
    the row that carries the property, bit for bit - and the flag on any other model is refused."""
    S = kz.scenes
    rng = np.random.default_rng(4)
    wi = rng.normal(size=(24, 3)).astype(np.float32); wi[:, 2] = np.abs(wi[:, 2]) + 0.05; wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(24, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.05; wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    for raw in (S.roughconductor(0.3, "Cu"), S.roughconductor(0.01), S.roughplastic(0.25, kd=(0.3, 0.5, 0.2)), S.roughdielectric(0.35)):
        prop = np.float32(raw.get("alpha", raw.get("roughness")))
        res = dict(raw, alphaResolved=True)
        res["alpha" if "alpha" in raw else "roughness"] = float(np.maximum(np.float32(0.001), prop * prop))
        for k in range(24):
            assert (O.bsdf(raw, "eval", wi[k], wo[k]).view(np.uint32) == O.bsdf(res, "eval", wi[k], wo[k]).view(np.uint32)).all()
            assert O.bsdf(raw, "pdf", wi[k], wo[k]) == O.bsdf(res, "pdf", wi[k], wo[k])
    s = S.SceneDescription()
    s.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=dict(S.ggx((0.5, 0.5, 0.5), 0.3, 0.0), alphaResolved=True))
    s.camera.update(width=8, height=8)
    with pytest.raises(Exception):
        O.OracleScene(s)
    with pytest.raises(Exception, match="alphaResolved"):
        kz.Scene(s, device=None)


# ---------------------------------------------------------------- round 5: DiscretePDF and the power-of-4 helpers, pinned to the reference's own text
def _f32(bits):
    return np.array(bits, np.uint32).view(np.float32)


def test_discrete_pdf_matches_reference_text_bit_for_bit(O, kz, kats):
    """a19: the area CDF of a light mesh. struct DiscretePDF of the reference's dpdf.h (append: running sums; normalize: multiply by the reciprocal of the
    sum, last entry forced to 1, a zero sum left alone; sample: lower_bound - 1, clamped) was compiled where it lies (oracle/kat_ref_dpdf.cpp) on tables of
    1 .. 200 entries spanning 2^-20 .. 2^10 with zero entries; the oracle's restatement AND the host code of kz_scene_create reproduce every bit."""
    import ctypes as C
    L, lib = O.lib(), kz.abi.load_library()
    fp = lambda a: a.ctypes.data_as(kz.abi.f32p)
    assert len(kats["dpdf"]) >= 9
    n_samples = 0
    for t in kats["dpdf"]:
        vals, want = _f32(t["values"]), np.array(t["cdf"], np.uint32)
        for which in ("oracle", "library"):
            cdf, sn = np.zeros(len(vals) + 1, np.float32), np.zeros(2, np.float32)
            if which == "oracle":
                L.kzo_debug_dpdf(len(vals), fp(vals), fp(cdf), fp(sn))
            else:
                assert lib.kz_kat_dpdf(len(vals), fp(vals), fp(cdf), fp(sn)) == 0
            assert (cdf.view(np.uint32) == want).all(), (which, len(vals))
            assert sn.view(np.uint32).tolist() == [t["sum"], t["normalization"]], (which, len(vals))
        assert t["normalized"] == (1 if _f32([t["sum"]])[0] > 0 else 0)
        cdf = _f32(t["cdf"]).copy()
        for v_bits, idx in t["sample"]:
            assert L.kzo_debug_dpdf_sample(len(cdf), fp(cdf), float(_f32([v_bits])[0])) == idx, (len(vals), v_bits)
            n_samples += 1
    assert n_samples > 800


def test_power_of_4_helpers_match_reference_text(O, kz, kats):
    """a9: isPowerOf4 / log2i / log4i / roundUpPow4 (common.h:271-319) decide PMJ02BN's pixel tile (sampler.cpp:291); restated in the oracle AND in the
    library's host code, both against the reference's own text for EVERY sample count 1 .. 65536."""
    import ctypes as C
    L, lib = O.lib(), kz.abi.load_library()
    pw = set(kats["pow4"]["is_power_of_4"])
    changes = {c[0]: c[1:] for c in kats["pow4"]["changes"]}
    cur = None
    o4, l4 = (C.c_int * 4)(), (C.c_int32 * 4)()
    for spp in range(1, 65537):
        cur = changes.get(spp, cur)
        L.kzo_debug_pow4(spp, o4)
        assert lib.kz_kat_pow4(spp, l4) == 0
        want = [1 if spp in pw else 0] + cur
        assert list(o4) == want and list(l4) == want, (spp, list(o4), list(l4), want)


def test_no_reference_text_under_the_repository():
    """the KAT generators are compiled from the reference's own text WHERE IT LIES - the extracted blocks live in a temporary directory
    outside the repository for the duration of the compile (oracle/Makefile). Nothing under the repository root - tracked, git-ignored or shipped to the GPU box -
    is reference source text: no *.inc anywhere, and oracle/_ref/ holds compiled generators (ELF) only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in (".git", "gpurun_out", "__pycache__", ".pytest_cache")]
        for f in files:
            assert not f.endswith(".inc"), os.path.join(d, f)
    ref = os.path.join(root, "oracle", "_ref")
    if os.path.isdir(ref):
        for f in os.listdir(ref):
            assert open(os.path.join(ref, f), "rb").read(4) == b"\x7fELF", f
    # the oracle's own sources state the algorithm in their own words: none of them carries the reference's namespace macros (the host MIRROR must spell
    # NAMESPACE_BEGIN(kazen) - it is written to compile inside a kazen tree - the oracle and the kernels must not)
    for sub in ("oracle", os.path.join("nano-kazen_amd", "csrc")):
        for d, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith((".cpp", ".h", ".hip", ".py")) and not f.startswith("kat_ref_"):
                    assert "NAMESPACE_BEGIN(" not in open(os.path.join(d, f), errors="ignore").read(), os.path.join(d, f)


def test_canonical_film_order_of_the_oracle(kz, O):
    """kzo_render_canonical: the same samples and weights as kzo_render (the reference-shaped blocks merged in block order), the float additions in the order the HIP build
    fixes - per pixel and filter tap in sample order, texels resolved over the 64-px tile grid in tile order (SURVEY H10 leaves the order to the build). The two orders agree
    to rounding; the canonical one does not depend on the number of threads, a tile subset leaves the other pixels' contributions out, and a sample range split in two is NOT
    the same bits when added on the host (which is why the device keeps running sums)."""
    desc = kz.scenes.cornell_box(80, 56, 6, sampler="pmj02bn")
    desc.camera["rfilter"] = {"type": "gaussian", "radius": 2.0, "stddev": 0.5}
    ora = O.OracleScene(desc)
    blocks = ora.render(threads=0)
    canon = ora.render_canonical(threads=0)
    assert np.allclose(canon, blocks, rtol=2e-6, atol=1e-6) and abs(float(canon[..., 3].sum()) - float(blocks[..., 3].sum())) < 1e-2
    assert np.array_equal(canon, ora.render_canonical(threads=1)) and np.array_equal(canon, ora.render_canonical(threads=3))
    tiles = [(0, 0, 64, 56)]
    part = ora.render_canonical(tiles=tiles, threads=0)
    b = ora.border
    assert (part[:, 64 + 2 * b:, 3] == 0).all() and np.array_equal(part[:, :64 - 2 * b], canon[:, :64 - 2 * b])       # texels only the tile's pixels reach: the same sums
    halves = ora.render_canonical(0, 3, threads=0) + ora.render_canonical(3, 6, threads=0)
    assert np.allclose(halves, canon, rtol=2e-6, atol=1e-6)
    other = ora.render_canonical(threads=0, grid=32)                                 # another grid: another grouping where cells meet, the same film to rounding
    assert np.allclose(other, canon, rtol=2e-6, atol=1e-6)
