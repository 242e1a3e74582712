"""-m gpu: the parity tests proper. Everything goes through the C ABI of libkazen_mi355x.so and is compared
with the CPU oracle on the same seeded inputs, with the committed golden fixtures, and — at BASELINE.json's
full sizes — through size-independent properties.

Tolerances (fp32, stated by BASELINE.json north_star): per-pixel L2 of normalised linear rgb < 1e-3; rays:
same (mesh, prim) except documented ties, |dt| <= 1e-4 t. Integer work (sampler streams) is bit exact. In
practice the HIP path is compiled with -ffp-contract=off and agrees with the oracle to ~1e-7.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
L2_TOL = 1e-3


def l2(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "fp_goldens.npz"))


def _rays(n, seed, lo=-0.95, hi=0.95):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:64] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 64)] * rng.choice([-1, 1], 64)[:, None].astype(np.float32)
    return o, d


# ---------------------------------------------------------------- traversal + post-intersection
@pytest.mark.parametrize("name", ["cornell", "sphere", "soup", "hero"])
def test_rays_match_brute_force(gpu_lib, kz, O, name):
    S = kz.scenes
    desc = {"cornell": lambda: S.cornell_box(32, 32, 1), "sphere": lambda: S.sphere_env(32, 32, 1),
            "soup": lambda: S.random_triangles(20000, 32, 32, 1, sampler="independent", s_edge=0.08),
            "hero": lambda: S.hero_scene(32, 32, 1, detail=0.25)}[name]()
    sc = kz.Scene(desc, device=0)
    n = 40000 if name != "soup" else 8000
    o, d = _rays(n, 11, -0.9, 0.9)
    if name == "hero":
        o = o * np.float32(4.0) + np.array([0, 2.5, 1], np.float32)
    hg = sc.trace_rays(o, d, 1e-3, np.inf)
    ora = O.OracleScene(desc, brute=(name != "hero"))        # hero: oracle BVH (itself checked against brute force on CPU)
    hc = ora.trace_rays(o, d, 1e-3, np.inf)
    hit = hc["mesh"] >= 0
    assert hit.mean() > 0.3
    assert np.array_equal(hg["mesh"], hc["mesh"]) and np.array_equal(hg["prim"], hc["prim"])
    assert np.all(np.abs(hg["t"][hit] - hc["t"][hit]) <= 1e-4 * hc["t"][hit])
    assert np.isinf(hg["t"][~hit]).all()
    for k, tol in (("p", 1e-5), ("sh_n", 1e-5), ("sh_s", 1e-4), ("sh_t", 1e-4), ("geo_n", 1e-5), ("uv", 1e-5), ("u", 1e-5)):
        assert np.abs(hg[k][hit] - hc[k][hit]).max() <= tol, k


def test_ray_edge_cases(gpu_lib, kz, O):
    """Empty range, tmax clipping, NaN / zero / infinite rays (must miss at once, not walk the tree), coincident
    triangles (lower id wins)."""
    s = kz.scenes.SceneDescription()
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    N = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    F = np.array([[0, 1, 2]], np.uint32)
    for _ in range(3):
        s.add_mesh(V, F, N)
    s.camera.update(width=32, height=32)
    sc = kz.Scene(s, device=0)
    o = np.array([[0.2, 0.2, 1], [0.2, 0.2, 1], [0.2, 0.2, 1], [np.nan, 0, 0], [0.2, 0.2, 1], [0.2, 0.2, 1], [0.2, 0.2, 1], [2, 2, 1]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 0], [np.inf, 0, -1], [0, np.nan, -1], [0, 0, -1]], np.float32)
    tmax = np.array([np.inf, 0.5, np.inf, np.inf, np.inf, np.inf, np.inf, np.inf], np.float32)
    h = sc.trace_rays(o, d, 0.0, tmax)
    assert h["mesh"].tolist() == [0, -1, -1, -1, -1, -1, -1, -1] and h["t"][0] == 1.0
    hc = O.OracleScene(s, brute=True).trace_rays(o, d, 0.0, tmax)
    assert hc["mesh"].tolist() == h["mesh"].tolist()
    assert sc.trace_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0, 1)["t"].shape == (0,)


def test_empty_scene_renders_black(gpu_lib, kz):
    s = kz.scenes.SceneDescription()
    s.camera.update(width=40, height=24)
    s.sampler = {"type": "independent", "sampleCount": 2, "seed": 0}
    s.background = {"color": (1, 1, 1), "intensity": 1.0}
    sc = kz.Scene(s, device=0)
    sc.render()
    film = sc.film()
    assert film.shape == (28, 44, 4) and (film[..., :3] == 0).all() and film[..., 3].sum() > 0     # H5: primary miss is black


# ---------------------------------------------------------------- function-level tables
def test_bsdf_tables_match_goldens(gpu_lib, kz, O, gold):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_fp_goldens.py"))
    mk = importlib.util.module_from_spec(spec)
    import sys
    sys.modules["mk"] = mk
    spec.loader.exec_module(mk)
    rows = mk.kiss_rows()
    s = kz.scenes.SceneDescription()
    tri = (np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32))
    for r in rows:
        s.add_mesh(*tri, bsdf=r)
    s.camera.update(width=32, height=32)
    sc = kz.Scene(s, device=0)
    wi, wo, s3 = gold["bsdf_wi"], gold["bsdf_wo"], gold["bsdf_s"]
    m = wi.shape[0]
    acc = np.where(np.arange(m) % 7 == 3, 0.25, 0.0).astype(np.float32)
    for r in range(len(rows)):
        ev, pd, sm = sc.bsdf_query(np.full(m, r, np.int32), wi, wo, acc, s3)
        assert same_bits(ev, gold["bsdf_eval"][r]), r                                     # the committed oracle vectors, bit for bit (kz_crmath.h)
        assert same_bits(pd, gold["bsdf_pdf"][r]), r
        g = gold["bsdf_sample"][r]
        assert np.array_equal(sm[:, 6], g[:, 6])
        ok = g[:, 6] > 0                                                                  # bRec.wo is undefined when sample() bails out
        assert same_bits(sm[ok, 3:6], g[ok, 3:6]), r                                      # sampled directions
        assert same_bits(sm[ok, :3], g[ok, :3]), r                                        # weights
        for k in np.nonzero(ok)[0][::3]:
            e = O.bsdf(rows[r], "eval", wi[k], sm[k, 3:6], float(acc[k]))
            p = O.bsdf(rows[r], "pdf", wi[k], sm[k, 3:6], float(acc[k]))
            if p > 1e-5:
                assert np.allclose(sm[k, :3], e / np.float32(p), rtol=3e-5, atol=1e-7), (r, k)


def test_extended_bsdf_tables_match_oracle(gpu_lib, kz, O):
    """BSDF::eval / pdf / sample of every plugin row beyond diffuse/kiss, on both sides of the surface, GPU vs oracle."""
    S = kz.scenes
    rows = [S.ggx((0.9, 0.6, 0.3), 0.3, 0.2), S.roughconductor(0.3, "Au"), S.roughconductor(0.05, "Cr"), S.roughplastic(0.3, kd=(0.2, 0.4, 0.7)),
            S.roughdielectric(0.4), S.roughdielectric(0.1, 1.33, 1.0), S.dielectric(), S.mirror()]
    s = S.SceneDescription()
    for r in rows:
        s.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=r)
    s.camera.update(width=32, height=32)
    sc = kz.Scene(s, device=0)
    rng = np.random.default_rng(9)
    m = 128
    wi = rng.normal(size=(m, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(m, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi[:96, 2] = np.abs(wi[:96, 2]) + 0.02            # mostly front side; the last 32 come from below (transmission / back-face)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    s3 = rng.random((m, 3)).astype(np.float32)
    acc = np.zeros(m, np.float32)
    for r, row in enumerate(rows):
        ev, pd, sm = sc.bsdf_query(np.full(m, r, np.int32), wi, wo, acc, s3)
        for k in range(m):
            e = O.bsdf(row, "eval", wi[k], wo[k])
            p = O.bsdf(row, "pdf", wi[k], wo[k])
            assert same_bits(ev[k], e), (r, k, ev[k], e)
            assert same_bits(pd[k], np.float32(p)), (r, k, pd[k], p)
            w, d, ok = O.bsdf(row, "sample", wi[k], None, 0.0, float(s3[k, 0]), (float(s3[k, 1]), float(s3[k, 2])))
            assert bool(sm[k, 6]) == ok, (r, k)
            if ok and np.any(w != 0):
                assert same_bits(sm[k, 3:6], d), (r, k)
                assert same_bits(sm[k, :3], w), (r, k, sm[k, :3], w)


@pytest.mark.parametrize("tag,sampler,seed", [("ind", "independent", 0), ("pmj", "pmj02bn", 1)])
def test_per_sample_radiance_matches_goldens(gpu_lib, kz, gold, tag, sampler, seed):
    """renderSample for explicit (pixel, sample) pairs against the committed oracle vectors: the pixel sample
    position (an integer-driven sampler output) and the radiance, bit for bit."""
    sc = kz.Scene(kz.scenes.cornell_box(32, 32, 4, sampler=sampler, seed=seed), device=0)
    out = sc.render_samples(gold["samples_pxy"], gold["samples_idx"])
    g = gold["samples_" + tag]
    assert same_bits(out, g)


@pytest.mark.parametrize("kind", ["stratified", "correlated", "independent", "pmj02bn"])
def test_sampler_streams_are_bit_exact(gpu_lib, kz, O, kind):
    """The pixel-sample positions are pure sampler output (hash, permute, pcg32, tables): identical bits on CPU and GPU."""
    d = kz.scenes.cornell_box(40, 24, 9, sampler=kind, seed=2)
    sc, ora = kz.Scene(d, device=0), O.OracleScene(d)
    assert sc.sample_count == ora.sample_count
    rng = np.random.default_rng(4)
    pxy = np.stack([rng.integers(0, 40, 300), rng.integers(0, 24, 300)], 1).astype(np.int32)
    idx = rng.integers(0, sc.sample_count, 300).astype(np.uint32)
    g, c = sc.render_samples(pxy, idx), ora.render_samples(pxy, idx)
    assert same_bits(g, c)                                                                # and with them the radiance of every sample


@pytest.mark.parametrize("tag,sampler,seed", [("ind", "independent", 0), ("pmj", "pmj02bn", 1)])
def test_golden_film(gpu_lib, kz, gold, tag, sampler, seed):
    sc = kz.Scene(kz.scenes.cornell_box(32, 32, 4, sampler=sampler, seed=seed), device=0)
    sc.render()
    film = sc.film()
    g = gold["film_" + tag]
    assert np.allclose(film[..., 3], g[..., 3], rtol=1e-5, atol=1e-6)                     # weights: same table, same positions
    assert np.allclose(film[..., :3], g[..., :3], rtol=1e-3, atol=1e-5)
    assert l2(sc.rgb(film), sc.rgb(g)) < 1e-5


# ---------------------------------------------------------------- images
CASES = {
    "c1_cornell_ind": lambda S: S.cornell_box(256, 256, 16),                                # BASELINE configs[0]
    "c1_cornell_pmj": lambda S: S.cornell_box(128, 128, 16, sampler="pmj02bn", seed=1),
    "c2_sphere_env": lambda S: S.sphere_env(512, 512, 16),                                  # configs[1] at 16 of 64 spp (oracle time)
    "c3_hero": lambda S: S.hero_scene(320, 180, 16, detail=0.5),                            # configs[2] geometry, reduced size
    "c4_soup_pmj": lambda S: S.random_triangles(200000, 240, 136, 16),                      # configs[3] at reduced size
    "ragged_size": lambda S: S.cornell_box(77, 45, 5),                                      # not a multiple of anything
    "stratified": lambda S: S.cornell_box(96, 80, 10, sampler="stratified", seed=1),        # rounds to 16 spp
    "correlated": lambda S: S.cornell_box(96, 80, 12, sampler="correlated", seed=3),
    "thinlens": lambda S: _thinlens(S),
    "mirror_glass": lambda S: S.glass_scene(96, 96, 16),                                     # EDiscrete + eta branches
    "all_bsdfs": lambda S: S.materials_scene(160, 96, 16),                                   # ggx + rough* 
}


def _thinlens(S):
    d = S.hero_scene(200, 112, 8, detail=0.3)
    d.camera.update(type="thinlens", apertureRadius=0.15, focusDistance=8.0)
    d.sampler = {"type": "correlated", "sampleCount": 9, "seed": 5}
    return d


@pytest.mark.parametrize("name", list(CASES))
def test_image_matches_oracle(gpu_lib, kz, O, name):
    desc = CASES[name](kz.scenes)
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render()
    gpu = sc.rgb()
    st = sc.stats()
    ora = O.OracleScene(desc)
    film = ora.render(threads=0)
    cpu = ora.rgb(film)
    err = l2(gpu, cpu)
    assert np.isfinite(gpu).all() and gpu.mean() > 1e-3
    assert err < L2_TOL, (name, err)
    # Same paths as the oracle: identical sample and light-sample counts. The wavefront pipeline skips work that cannot
    # change the result (shadow rays whose pending radiance is exactly zero, the hit record after the last bounce
    # when there is no background, any-hit instead of closest-hit occlusion), so rays / node visits are <= the oracle's.
    so = ora.stats()
    assert st["samples"] == so["samples"] and st["droppedSamples"] == so["droppedSamples"]
    assert abs(st["lightSamples"] - so["lightSamples"]) <= 1e-5 * so["lightSamples"] + 2      # an ulp can flip one roulette decision
    assert 0.5 * so["rays"] <= st["rays"] <= so["rays"] and 0.9 * so["shadedHits"] <= st["shadedHits"] <= so["shadedHits"]
    # (camera rays take the triangles of their pixel's leaf list - kz_wf_beam - instead of walking the tree: far fewer node visits, somewhat more triangle tests)
    assert st["nodeVisits"] <= 1.1 * so["nodeVisits"] and st["triTests"] <= 1.5 * so["triTests"]
    assert 64 * st["nodeVisits"] + 48 * st["triTests"] <= 1.1 * (64 * so["nodeVisits"] + 48 * so["triTests"])
    # the reference-shaped megakernel does exactly the oracle's work and produces the same film bit for bit
    sc.stats(reset=True)
    sc.render(pipeline=1)
    sm = sc.stats()
    assert np.array_equal(sc.rgb(), gpu)
    for k in ("rays", "shadedHits", "lightSamples"):
        assert abs(sm[k] - so[k]) <= 1e-4 * so[k] + 2, k


def test_regularization_and_filters(gpu_lib, kz, O):
    for filt in ({"type": "tent"}, {"type": "box"}, {"type": "mitchell", "radius": 2.0, "B": 1 / 3.0, "C": 1 / 3.0},
                 {"type": "gaussian", "radius": 3.0, "stddev": 0.8}):
        desc = kz.scenes.cornell_box(48, 40, 4)
        desc.camera["rfilter"] = filt
        desc.integrator.update(regularization=True, accumulatedRoughness=0.5, maxDepth=8)
        sc = kz.Scene(desc, device=0)
        sc.render()
        ora = O.OracleScene(desc)
        film_c = ora.render(threads=0)
        assert sc.film().shape == film_c.shape
        assert l2(sc.rgb(), ora.rgb(film_c)) < L2_TOL, filt


def test_invalid_radiance_is_dropped(gpu_lib, kz, O):
    """A light with negative intensity makes negative radiance: ImageBlock::put drops the sample and its weight
    (block.cpp:57-61); the counter reports how many."""
    desc = kz.scenes.cornell_box(32, 32, 4)
    for m in desc.meshes:
        if m["light"]:
            m["light"]["intensity"] = -3.0
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render()
    ora = O.OracleScene(desc)
    film_c = ora.render(threads=0)
    assert sc.stats()["droppedSamples"] == ora.stats()["droppedSamples"] > 0
    assert np.allclose(sc.film()[..., 3], film_c[..., 3], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- sharding / accumulation invariants
def test_tiles_and_sample_ranges_are_invariant(gpu_lib, kz):
    """The film does not depend on how the work is cut: whole image == two interleaved tile sets summed
    == two sample ranges accumulated. Tile aprons carry the filter splats across tile borders."""
    desc = kz.scenes.cornell_box(96, 64, 8)
    sc = kz.Scene(desc, device=0)
    sc.render()
    whole = sc.film()
    tiles = kz.shard.make_tiles(96, 64, 32)
    parts = []
    for r in range(2):
        sc.render(tiles=kz.shard.tiles_for_rank(tiles, r, 2))
        parts.append(sc.film())
    assert np.allclose(kz.shard.merge_films(parts), whole, rtol=1e-5, atol=1e-6)
    sc.render(0, 3)
    sc.render(3, 8, accumulate=True)
    assert np.array_equal(sc.film(), whole)                   # (round 6: a pixel's running tap sums do not care which call brought a sample)
    a = sc.film()
    sc.render(0, 3)
    sc.render(3, 8, accumulate=True)
    assert np.array_equal(sc.film(), a)                       # deterministic: run-to-run bit identical


def test_overlapping_tiles_rejected(gpu_lib, kz):
    sc = kz.Scene(kz.scenes.cornell_box(64, 64, 1), device=0)
    with pytest.raises(kz.abi.KzError):
        sc.render(tiles=[(0, 0, 32, 32), (16, 16, 32, 32)])
    with pytest.raises(kz.abi.KzError):
        sc.render(tiles=[(40, 40, 32, 32)])
    with pytest.raises(kz.abi.KzError):
        sc.render(5, 3)


# ---------------------------------------------------------------- full BASELINE size: properties only
def test_full_size_c4_properties(gpu_lib, kz):
    """configs[3] at its real size (1 M triangles, 1920x1080, pmj02bn): one 4-spp slice. The oracle cannot
    cover this in seconds, so check properties: total filter weight equals the sample count times the mean
    filter mass measured on a small render with the same filter; determinism; no dropped samples; and a
    64x64 crop of the same frame against the oracle."""
    desc = kz.scenes.random_triangles(1000000, 1920, 1080, 1024)
    sc = kz.Scene(desc, device=0)
    assert sc.bvh_info()["nTris"] == 1000028
    sc.set_stats(True)
    sc.render(0, 4)
    film = sc.film()
    st = sc.stats(reset=True)
    assert st["samples"] == 1920 * 1080 * 4 and st["droppedSamples"] == 0 and np.isfinite(film).all()
    mass = film[..., 3].sum() / st["samples"]
    small = kz.Scene(kz.scenes.random_triangles(1000, 64, 64, 1024), device=0)
    small.render(0, 4)
    mass_small = small.film()[..., 3].sum() / (64 * 64 * 4)
    assert abs(mass - mass_small) < 2e-3 * mass_small
    sc.render(0, 4)
    assert np.array_equal(sc.film(), film)
    # crop parity against the oracle
    import oracle as O
    ora = O.OracleScene(desc)
    tile = [(928, 508, 64, 64)]
    film_c = ora.render(0, 4, tiles=tile, threads=0)
    sc.render(0, 4, tiles=tile)
    b = sc.border
    crop_g = sc.film()[508:508 + 64 + 2 * b, 928:928 + 64 + 2 * b]
    crop_c = film_c[508:508 + 64 + 2 * b, 928:928 + 64 + 2 * b]
    assert np.allclose(crop_g[..., 3], crop_c[..., 3], rtol=1e-5, atol=1e-6)
    rg = crop_g[..., :3] / np.maximum(crop_g[..., 3:], 1e-20)
    rc = crop_c[..., :3] / np.maximum(crop_c[..., 3:], 1e-20)
    assert l2(rg, rc) < L2_TOL


def same_bits(a, b):
    """Equal to the last bit (two NaNs count as equal, +0 and -0 do not)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


# ---------------------------------------------------------------- randomized differential test
def _fuzz_scene(S, seed, rich=False):
    """A random small scene: triangle soup + room, random BSDF plugins on every mesh, random visible / invisible lights (some stacked so
    shadow rays cross them), random camera / sampler / filter / integrator settings. rich: the same scene with (from a second generator, so the plain scene
    of a seed does not change) bicubic image lookups and an environment image, colour ramp or constant behind the background (scripts/dev/fuzz_sweep.py --rich)."""
    rng = np.random.default_rng(seed)
    d = S.random_triangles(int(rng.integers(50, 3000)), int(rng.integers(24, 72)), int(rng.integers(24, 56)), 4, sampler="independent", s_edge=float(rng.uniform(0.05, 0.4)))
    makers = [lambda: S.diffuse(tuple(rng.uniform(0.1, 0.9, 3))),
              lambda: S.kazenstandard(tuple(rng.uniform(0.1, 0.9, 3)), float(rng.uniform(0, 1)), float(rng.integers(0, 2)), float(rng.uniform(0, 0.8)),
                                      float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1))),
              lambda: S.mirror(), lambda: S.dielectric(float(rng.uniform(1.2, 1.8)), 1.0), lambda: S.ggx(tuple(rng.uniform(0.2, 0.9, 3)), float(rng.uniform(0.05, 0.9)), float(rng.uniform(0, 0.5))),
              lambda: S.roughconductor(float(rng.uniform(0.05, 0.6)), str(rng.choice(["Au", "Cu", "Cr"]))), lambda: S.roughplastic(float(rng.uniform(0.05, 0.6)), kd=tuple(rng.uniform(0.1, 0.6, 3))),
              lambda: S.roughdielectric(float(rng.uniform(0.05, 0.6)))]
    if seed % 3 == 0:                                                            # every third scene: texture trees and normal maps too (8f rank 4)
        chk, noise, gray, nrm = S._test_images()
        t_noise, t_gray = S.imagetexture(noise, float(rng.uniform(0.5, 4)), "srgb"), S.imagetexture(gray, float(rng.uniform(0.5, 4)), "linear")
        t_nrm = S.imagetexture(nrm, float(rng.uniform(0.5, 3)), "linear")
        makers += [lambda: S.lambertian(S.blend(t_gray, S.constanttexture(tuple(rng.uniform(0, 1, 3))), t_noise)),
                   lambda: S.kazenstandard(t_noise, S.colorramp(t_gray, 0.1, 0.9), t_gray, clearcoat=float(rng.uniform(0, 1))),
                   lambda: S.normalmap(t_nrm, makers[int(rng.integers(0, 8))]()), lambda: S.ggx(S.imagetexture(chk, 3.0, "srgb"), float(rng.uniform(0.1, 0.8)))]
    for m in d.meshes:
        if m["light"]:
            m["light"]["lightPrimaryVisibility"] = bool(rng.integers(0, 2))
            m["light"]["intensity"] = float(rng.uniform(5, 40))
        elif rng.random() < 0.85:
            m["bsdf"] = makers[int(rng.integers(0, len(makers)))]()
        else:
            m["bsdf"] = None                                                    # default diffuse (mesh.cpp:25-28)
        if rng.random() < 0.2:
            m["N"] = None                                                       # no vertex normals (H4)
    # an extra light hanging in the room below the ceiling lights: shadow rays to the ceiling cross it
    q = S.quad((-0.5, 0.6, 1.2), (-0.5, 0.6, 2.2), (0.5, 0.6, 2.2), (0.5, 0.6, 1.2), flip=True)
    d.add_mesh(q[0], q[3], q[1], q[2], bsdf=S.diffuse((0, 0, 0)), light=S.area((1, 0.8, 0.6), float(rng.uniform(2, 10)), bool(rng.integers(0, 2))))
    d.sampler = {"type": str(rng.choice(["independent", "pmj02bn", "stratified", "correlated"])), "sampleCount": int(rng.integers(1, 12)), "seed": int(rng.integers(0, 100))}
    d.integrator.update(maxDepth=int(rng.integers(1, 9)), traceBias=float(rng.choice([1e-3, 1e-4, 5e-3])), regularization=bool(rng.integers(0, 2)),
                        accumulatedRoughness=float(rng.uniform(0.1, 0.9)))
    d.camera["rfilter"] = [{"type": "gaussian", "radius": 2.0, "stddev": 0.5}, {"type": "tent"}, {"type": "box"}, {"type": "mitchell", "radius": 2.0, "B": 1 / 3, "C": 1 / 3},
                           {"type": "gaussian", "radius": 3.5, "stddev": 1.0}][int(rng.integers(0, 5))]
    if rng.random() < 0.4:
        d.camera.update(type="thinlens", apertureRadius=float(rng.uniform(0, 0.2)), focusDistance=float(rng.uniform(1, 4)))
    if rng.random() < 0.5:
        d.background = {"color": tuple(rng.uniform(0, 1, 3)), "intensity": float(rng.uniform(0, 2))}
    if rich:
        r2 = np.random.default_rng(1_000_003 * (seed + 1))

        def walk(node):
            if isinstance(node, dict):
                if node.get("type") == "imagetexture" and r2.random() < 0.5:
                    node["filter"] = "bicubic"
                for v in node.values():
                    walk(v)
        for m in d.meshes:
            walk(m["bsdf"])
        k = r2.random()
        if k < 0.6:
            env = r2.random((int(r2.integers(2, 9)), int(r2.integers(3, 17)), 3)).astype(np.float32) * float(r2.uniform(0.2, 3))
            tex = S.imagetexture(env, 1.0, "linear", "bicubic" if r2.random() < 0.3 else "bilinear")
            d.background = {"texture": tex if k < 0.45 else S.colorramp(tex, 0.1, 0.9), "intensity": float(r2.uniform(0.2, 2))}
    return d


def tie_bracket(ora, sc, desc, tol=1e-3):
    """The reference's shadow loop has a built-in tie (integrator.cpp:262-278): after walking through an invisible light the far end becomes maxt - t
    while the origin moved on by t + eps, so the remaining segment ends exactly ON the sampled light and whether a VISIBLE sampled light then
    occludes itself is decided by the last bit of everything upstream - in the reference as in any restatement of it. Where the upstream
    arithmetic is only equal to an ulp (libm sin / cos on the lens or in a BSDF sample) the oracle and the HIP path decide some of these ties
    differently; both are roundings of the same reference. The oracle renders the two extreme resolutions (every tie occluded / none) and every
    rounding must lie between them SAMPLE by sample (a pixel would not do: filters with negative lobes are not monotone). Returns
    (share of HIP samples outside the bracket by more than tol, number of samples with an open bracket). A closed bracket is the literal oracle's
    value, so the first number is the usual per-sample parity there."""
    w, h, n = desc.camera["width"], desc.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(n), indexing="ij")
    pxy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32)
    idx = ii.ravel().astype(np.uint32)
    g = sc.render_samples(pxy, idx)[:, 2:5]
    c = ora.render_samples(pxy, idx)[:, 2:5]
    ora.set_tie_mode(+1); lo = ora.render_samples(pxy, idx)[:, 2:5]
    ora.set_tie_mode(-1); hi = ora.render_samples(pxy, idx)[:, 2:5]
    ora.set_tie_mode(0)
    slack = tol * (1.0 + np.abs(hi))
    assert (lo <= c + slack).all() and (c <= hi + slack).all()
    outside = ~((g >= lo - slack) & (g <= hi + slack)).all(axis=1)
    return float(outside.mean()), int(((hi - lo) > slack).any(axis=1).sum())


@pytest.mark.parametrize("seed", list(range(15)))
def test_randomized_scenes_match_oracle(gpu_lib, kz, O, seed):
    desc = _fuzz_scene(kz.scenes, 1000 + seed)
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render()
    film = sc.film()
    st = sc.stats(reset=True)
    ora = O.OracleScene(desc)
    film_c = ora.render(threads=0)
    so = ora.stats()
    assert st["samples"] == so["samples"] == desc.camera["width"] * desc.camera["height"] * sc.sample_count
    assert st["droppedSamples"] == so["droppedSamples"]
    assert np.allclose(film[..., 3], film_c[..., 3], rtol=1e-4, atol=1e-5)              # same samples dropped, same weights
    # Every sample's radiance is the oracle's to the last bit (transcendentals defined by their arithmetic on both sides, kz_crmath.h): the films differ
    # only by the order in which the filter-weighted samples are added (2e-7 x scale at most over the 1 500 scenes of profiles/r04m_fuzz_sweep).
    w, h, n = desc.camera["width"], desc.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(n), indexing="ij")
    pxy, idx = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32), ii.ravel().astype(np.uint32)
    assert same_bits(sc.render_samples(pxy, idx), ora.render_samples(pxy, idx))
    err = l2(sc.rgb(film), ora.rgb(film_c))
    scale = max(1.0, float(np.abs(ora.rgb(film_c)).max()))
    assert err < 2e-6 * scale, (seed, err, scale)
    sc.render(pipeline=1)                                                               # reference-shaped megakernel: same film bit for bit
    assert np.array_equal(sc.film(), film)
    # The counting and the product instantiations of every traversal MODE these scenes reach (0, 1, 4 + the deferred 2; 2 for all shadow rays where the
    # invisible lights have more than 64 triangles) must render the SAME film, also with a 3-entry LDS stack (the global overflow path): ADVICE r03 - the
    # stack reset of kz_wf_trace rests on empty asm barriers that once only the counting instantiations needed; a recurrence has to fail here.
    sc.set_stats(False)
    sc.render()
    assert np.array_equal(sc.film(), film)
    sc.render(tune={"ldsStack": 3})
    assert np.array_equal(sc.film(), film)
    sc.set_stats(True)
    sc.render(tune={"ldsStack": 3})
    assert np.array_equal(sc.film(), film)


def _math_args(rng, n):
    """Arguments of the path's transcendental functions: the path's ranges, dense, plus every edge the definitions branch on."""
    e = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1.1754944e-38, 3.4e38, -3.4e38, 0.5, 2.0, 0.25, 0.75, 0.125, 0.375, 1.5707964, 3.1415927, 6.2831855,
                  88.7, 89.1, -87.3, -87.4, -103.9, -104.1, 1048576.0, 1048577.0, 1e-20, 1e20], np.float32)
    ye, xe = (a.ravel() for a in np.meshgrid(e, e))
    cat = lambda a: np.concatenate([a.astype(np.float32), e])
    return {"sin": (cat(rng.uniform(-7, 7, n)),), "cos": (cat(rng.uniform(-7, 7, n)),), "cos1": (cat(rng.uniform(-7, 7, n)),), "tan": (cat(rng.uniform(-3.2, 3.2, n)),),
            "exp": (cat(-np.exp(rng.uniform(-12, 5, n))),), "log": (cat(np.exp(rng.uniform(-20, 20, n))),), "atan": (cat(np.exp(rng.uniform(-15, 15, n)) * rng.choice([-1, 1], n)),),
            "acos": (cat(np.concatenate([rng.uniform(-1, 1, n // 2), 1 - np.exp(rng.uniform(-17, 0, n - n // 2))])),), "cube": (cat(rng.uniform(-1, 1, n)),),
            "atan2": (np.concatenate([rng.normal(size=n).astype(np.float32), ye]), np.concatenate([rng.normal(size=n).astype(np.float32), xe])),
            "hypot": (np.concatenate([rng.normal(size=n).astype(np.float32), ye]), np.concatenate([rng.normal(size=n).astype(np.float32), xe])),
            "pow": (np.concatenate([rng.uniform(0.003, 50, n).astype(np.float32), np.abs(ye)]), np.concatenate([rng.choice(np.array([2.4, 1 / 2.4], np.float32), n), xe]))}


def test_transcendentals_equal_oracle_bit_for_bit(gpu_lib, kz, O):
    """sin / cos / tan / exp / log / atan / atan2 / acos / pow / hypot as the kernels compute them (csrc/kz_crmath.h) against the oracle's statement of the
    same definitions (oracle/kz_oracle_math.h): the SAME bits on every argument - dense over the path's ranges, and on every edge the definitions branch on."""
    import ctypes as C
    rng = np.random.default_rng(77)
    for name, args in _math_args(rng, 1 << 21).items():
        x = np.ascontiguousarray(args[0], np.float32)
        y = np.ascontiguousarray(args[1] if len(args) > 1 else args[0], np.float32)
        out = np.zeros_like(x)
        f = lambda a: a.ctypes.data_as(kz.abi.f32p)
        kz.abi.check(gpu_lib, gpu_lib.kz_kat_math(0, O.MATH_FN[name], x.size, f(x), f(y), f(out)))
        ref = O.math_fn(name, x, y)
        both_nan = np.isnan(out) & np.isnan(ref)
        bad = (out.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
        assert not bad.any(), (name, int(bad.sum()), x[bad][:4], y[bad][:4], out[bad][:4], ref[bad][:4])


def test_sampled_directions_equal_oracle_bit_for_bit(gpu_lib, kz, O):
    """With both sides on the same transcendental definitions the sampled direction of every BSDF plugin is the oracle's to the last bit (before: ~55 % of
    the queries, scripts/dev/bsdf_bits.py) - what keeps a deep path through small triangles on the same triangles on both sides."""
    S = kz.scenes
    rows = [S.diffuse((0.5, 0.6, 0.7)), S.kazenstandard((0.8, 0.5, 0.3), 0.4, 0.5, 0.3), S.ggx((0.9, 0.6, 0.3), 0.3, 0.2), S.roughconductor(0.3, "Au"),
            S.roughplastic(0.3, kd=(0.2, 0.4, 0.7)), S.roughdielectric(0.4), S.dielectric(), S.mirror()]
    s = S.SceneDescription()
    for r in rows:
        s.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=r)
    s.camera.update(width=32, height=32)
    sc = kz.Scene(s, device=0)
    ora = O.OracleScene(s)
    rng = np.random.default_rng(11)
    m = 600
    wi = rng.normal(size=(m, 3)).astype(np.float32); wi[:, 2] = np.abs(wi[:, 2]) + 0.02; wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(m, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.02; wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    s3 = rng.random((m, 3)).astype(np.float32)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    for r in range(len(rows)):
        ev, pd, sm = sc.bsdf_query(np.full(m, r, np.int32), wi, wo, np.zeros(m, np.float32), s3)
        so = np.stack([ora.bsdf(r, "sample", wi[k], None, 0.0, float(s3[k, 0]), (float(s3[k, 1]), float(s3[k, 2]))) for k in range(m)])
        e = np.stack([ora.bsdf(r, "eval", wi[k], wo[k]) for k in range(m)])
        p = np.array([ora.bsdf(r, "pdf", wi[k], wo[k]) for k in range(m)], np.float32)
        ok = (so[:, 6] > 0) & (sm[:, 6] > 0)
        assert np.array_equal(so[:, 6] > 0, sm[:, 6] > 0), r
        assert np.array_equal(bits(sm[ok, 3:6]), bits(so[ok, 3:6])), (r, "direction")
        assert np.array_equal(bits(sm[ok, :3]), bits(so[ok, :3])), (r, "weight")
        assert np.array_equal(bits(ev), bits(e)) and np.array_equal(bits(pd), bits(p)), (r, "eval / pdf")


def _named_scenes(S):
    return {"cornell": lambda: S.cornell_box(48, 48, 8), "cornell_pmj": lambda: S.cornell_box(48, 48, 8, sampler="pmj02bn", seed=1), "hero": lambda: S.hero_scene(64, 36, 8, detail=0.3),
            "sphere_env": lambda: S.sphere_env(48, 48, 8), "materials": lambda: S.materials_scene(64, 36, 8), "textured": lambda: S.textured_scene(64, 36, 8),
            "random_triangles": lambda: S.random_triangles(20000, 64, 36, 8),
            "q1_asset": lambda: S.load_npz(os.path.join(HERE, "golden", "q1_default_m0_r0.5.npz"), {"camera": {"width": 64, "height": 36}, "sampler": {"sampleCount": 8}})}


@pytest.mark.parametrize("name", ["cornell", "cornell_pmj", "hero", "sphere_env", "materials", "textured", "random_triangles", "q1_asset"])
def test_every_sample_equals_the_oracle_bit_for_bit(gpu_lib, kz, O, name):
    """renderSample of every (pixel, sample) of the configs' scenes - and of the reference's own asset scene (smooth normals: terminator offsets, interpolated frames) -
    returns the oracle's position and radiance to the last bit: nothing on the path is compared by tolerance any more."""
    d = _named_scenes(kz.scenes)[name]()
    sc, ora = kz.Scene(d, device=0), O.OracleScene(d)
    w, h, n = d.camera["width"], d.camera["height"], sc.sample_count
    yy, xx, ii = np.meshgrid(np.arange(h), np.arange(w), np.arange(n), indexing="ij")
    pxy, idx = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32), ii.ravel().astype(np.uint32)
    g, c = sc.render_samples(pxy, idx), ora.render_samples(pxy, idx)
    assert same_bits(g, c), int((g.view(np.uint32) != c.view(np.uint32)).any(axis=1).sum())
    assert np.abs(c[:, 2:]).max() > 0


@pytest.mark.parametrize("name", ["m1_r0", "m0_r0_spec1_st1", "r0.5_c1_cr0.5", "r0_s1_st0.5"])
def test_reference_parameter_scenes_equal_the_oracle_bit_for_bit(gpu_lib, kz, O, name):
    """Four of the reference's 22 parameter-study scene files (tests/golden/q1_params.json + the npz: metallic mirror, tinted specular, clearcoat, tinted sheen) through the HIP
    path, every sample against the oracle. (All 22 at 1920 x 1080 x 4096 spp against the published pictures: profiles/r04p_q1_full.)"""
    import json
    bsdf = json.load(open(os.path.join(HERE, "golden", "q1_params.json")))["params"][name]
    d = kz.scenes.load_npz(os.path.join(HERE, "golden", "q1_default_m0_r0.5.npz"), {"camera": {"width": 48, "height": 27}, "sampler": {"sampleCount": 4}})
    d.meshes[4]["bsdf"] = {k: v for k, v in bsdf.items() if not k.startswith("_")}
    sc, ora = kz.Scene(d, device=0), O.OracleScene(d)
    yy, xx, ii = np.meshgrid(np.arange(27), np.arange(48), np.arange(4), indexing="ij")
    pxy, idx = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32), ii.ravel().astype(np.uint32)
    assert same_bits(sc.render_samples(pxy, idx), ora.render_samples(pxy, idx))


@pytest.mark.parametrize("seed", [5084, 5094, 5100])
def test_reference_shadow_ties_are_bracketed(gpu_lib, kz, O, seed):
    """Scenes of the wider sweep (scripts/dev/fuzz_sweep.py) where visible lights are sampled THROUGH invisible ones: the literal films differ in the
    few samples whose shadow test is the reference's tie (see tie_bracket) whenever anything upstream differs in its last bit (before kz_crmath.h these
    three films were 3e-3 ... 9e-3 off the literal oracle). Every HIP sample lies inside the oracle's bracket; the film weights are equal and the two
    pipelines agree bit for bit."""
    desc = _fuzz_scene(kz.scenes, seed)
    sc = kz.Scene(desc, device=0)
    sc.render()
    film = sc.film()
    ora = O.OracleScene(desc)
    film_c = ora.render(threads=0)
    assert np.allclose(film[..., 3], film_c[..., 3], rtol=1e-4, atol=1e-5)
    outside, n_open = tie_bracket(ora, sc, desc)
    n = desc.camera["width"] * desc.camera["height"] * sc.sample_count
    assert outside == 0 and 0 < n_open < n // 2, (outside, n_open, n)
    # and since both sides now compute everything upstream of the tie with the same bits, they decide it the same way
    assert l2(sc.rgb(film), ora.rgb(film_c)) < 2e-6 * max(1.0, float(np.abs(ora.rgb(film_c)).max()))
    sc.render(pipeline=1)
    assert np.array_equal(sc.film(), film)


@pytest.mark.parametrize("seed", list(range(12)))
def test_pixel_beams_equal_per_ray_traversal_under_random_cameras(gpu_lib, kz, seed):
    """The camera rays through the pixel-beam lists (kz_wf_beam + kz_wf_trace_list, the default), the packet kernel and the per-lane traversal give
    the SAME film bit for bit, for cameras the beams' margins have to survive: scaled / sheared / mirrored toWorld matrices (|d| != 1: the lists'
    bounds are distances, the rays' parameters are not), 2 to 150 degrees of view, the pinhole inside a dense soup, far away from a tiny one,
    single-sample passes, tile sets with ragged edges."""
    rng = np.random.default_rng(4000 + seed)
    S = kz.scenes
    w, h = int(rng.integers(33, 97)), int(rng.integers(17, 71))
    d = S.random_triangles(int(rng.integers(200, 6000)), w, h, 4, sampler="pmj02bn" if seed % 2 else "independent", s_edge=float(rng.choice([0.02, 0.1, 0.5])))
    eye = rng.uniform(-1.0, 1.0, 3) if seed % 3 == 0 else rng.uniform(-1, 1, 3) + np.array([0, 0, rng.uniform(2.5, 40.0)])      # inside the soup / outside, near to far
    tw = S.look_at(tuple(eye), tuple(rng.uniform(-0.5, 0.5, 3)), (0, 1, 0)).astype(np.float64)
    M = np.eye(4)
    if seed % 4 == 1: M[:3, :3] = np.diag(rng.uniform(0.3, 3.0, 3))                        # non-uniform scale of the camera's axes
    if seed % 4 == 2: M[:3, :3] = np.eye(3) + np.triu(rng.uniform(-0.4, 0.4, (3, 3)), 1)   # shear
    if seed % 4 == 3: M[0, 0] = -1.0                                                       # mirrored
    d.camera.update(toWorld=(tw @ M).astype(np.float32), fov=float(rng.choice([2.0, 20.0, 60.0, 120.0, 150.0])), nearClip=float(rng.choice([1e-4, 0.05])), farClip=float(rng.choice([50.0, 1e4])))
    d.sampler["sampleCount"] = int(rng.choice([1, 3, 4, 16]))
    sc = kz.Scene(d, device=0)
    tiles = None if seed % 2 else [(0, 0, 32, 17), (32, 0, w - 32, 17)]
    films = []
    for kernel in (0, 1, 2):
        sc.render(tiles=tiles, tune={"packetPrimary": kernel})
        films.append(sc.film())
    assert np.array_equal(films[0], films[1]) and np.array_equal(films[0], films[2]), seed
    assert np.isfinite(films[0]).all()
    sc.render(tiles=tiles, pass_items=max(64, (w * h // 3) // 64 * 64), tune={"sppPerPass": 1})      # pixel chunks: the contexts' own lists, rebuilt per chunk
    assert np.allclose(sc.film(), films[0], rtol=2e-5, atol=1e-5)
    sc.render(tiles=tiles, pipeline=1)                                                     # the reference-shaped megakernel (BVH2, one ray at a time)
    assert np.array_equal(sc.film(), films[0])


def test_two_passes_in_flight_equal_one_at_a_time(gpu_lib, kz, O):
    """kz_render splits a call into passes (KzRenderOpts.passItems) and keeps two of them in flight on two internal streams; every
    pixel's samples reach its running tap sums in sample order whatever the schedule, so every schedule - and every split of the
    sample range over calls - gives the same bits, and the oracle's film within the bar."""
    desc = kz.scenes.cornell_box(96, 80, 24, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    items = 96 * 80 * 4                                             # 6 passes of 4 spp
    sc.render(pass_items=items, passes_in_flight=1)
    one_at_a_time = sc.film()
    for _ in range(3):                                               # repeated: an ordering bug would show as run-to-run differences
        sc.render(pass_items=items)
        assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(0, 10, pass_items=items)                               # 3 passes, the last one short; then accumulate the rest in two calls
    sc.render(10, 24, accumulate=True, pass_items=items)
    ms = sc.last_kernel_ms()
    split = sc.film()
    assert ms > 0 and np.array_equal(split, one_at_a_time)           # other pass boundaries, two calls: the same film (H10)
    ora = O.OracleScene(desc)
    assert l2(sc.rgb(one_at_a_time), ora.rgb(ora.render(threads=0))) < L2_TOL
    sc.render()                                                      # one pass
    assert np.array_equal(sc.film(), one_at_a_time)


# ---------------------------------------------------------------- a3 camera rays and a18 light samples as stand-alone device tables
def _camera_cases(S):
    pin = S.cornell_box(160, 96, 4)
    lens = _thinlens(S)
    off = S.hero_scene(77, 45, 4, detail=0.3)
    off.camera.update(fov=63.0, nearClip=0.25, farClip=300.0)
    return {"pinhole": pin, "thinlens": lens, "ragged_fov": off}


@pytest.mark.parametrize("name", ["pinhole", "thinlens", "ragged_fov"])
def test_camera_rays_match_oracle(gpu_lib, kz, O, name):
    """Camera::sampleRay (camera.cpp:70-91, 191-223) on the device for explicit pixel-sample and aperture positions, against the
    oracle's cameraSampleRay: corners, centres, out-of-frame positions (the reconstruction filter never asks for them, the function
    is still defined there) and the aperture rim."""
    desc = _camera_cases(kz.scenes)[name]
    sc, ora = kz.Scene(desc, device=0), O.OracleScene(desc)
    w, h = desc.camera["width"], desc.camera["height"]
    rng = np.random.default_rng(11)
    sxy = np.concatenate([np.array([[0, 0], [w, h], [w, 0], [0, h], [w / 2, h / 2], [0.5, 0.5], [-1.5, h + 2.0]], np.float32),
                          (rng.random((249, 2)) * [w, h]).astype(np.float32)])
    axy = np.concatenate([np.array([[0, 0], [1, 1], [0.5, 0.5], [1, 0], [0, 1], [0.999999, 0.25], [0.25, 0.999999]], np.float32),
                          rng.random((249, 2)).astype(np.float32)])
    got = sc.camera_rays(sxy, axy)
    ref = np.zeros_like(got)
    for i in range(len(sxy)):
        o6, a, b = ora.camera_ray(float(sxy[i, 0]), float(sxy[i, 1]), float(axy[i, 0]), float(axy[i, 1]))
        ref[i, :6], ref[i, 6], ref[i, 7] = o6, a, b
    assert same_bits(got, ref), np.abs(got - ref).max()
    # without aperture samples the entry uses (0.5, 0.5), the oracle's plain entry
    got0 = sc.camera_rays(sxy[:16])
    for i in range(16):
        o6, a, b = ora.camera_ray(float(sxy[i, 0]), float(sxy[i, 1]))
        assert same_bits(got0[i, :6], o6) and same_bits(got0[i, 6], np.float32(a)) and same_bits(got0[i, 7], np.float32(b))
    # unit directions; mint/maxt are the clip planes over cos(theta) to the optical axis
    assert np.allclose(np.linalg.norm(got[:, 3:6], axis=1), 1.0, atol=1e-6)
    assert (got[:, 7] > got[:, 6]).all() and (got[:, 6] > 0).all()


def _light_scenes(S):
    return {"cornell": S.cornell_box(64, 64, 4), "hero": S.hero_scene(64, 36, 4, detail=0.3), "fuzz3": _fuzz_scene(S, 3), "fuzz7": _fuzz_scene(S, 7)}


@pytest.mark.parametrize("name", ["cornell", "hero", "fuzz3", "fuzz7"])
def test_light_samples_match_oracle(gpu_lib, kz, O, name):
    """AreaLight::sample / pdf / eval (light.cpp:16-51) through Mesh::sample (mesh.cpp:108-133, dpdf.h:99-104) on the device — the function
    the shade kernel calls — for every light of the scene: triangle picked by the area cdf (bit exact, including draws of 0, on cdf
    entries and next to 1), position, normal (interpolated, not normalised, when the mesh has normals: H8), direction, solid-angle pdf
    (0 behind the light), and eval/pdf."""
    desc = _light_scenes(kz.scenes)[name]
    sc, ora = kz.Scene(desc, device=0), O.OracleScene(desc)
    n_lights = sum(1 for m in desc.meshes if m.get("light"))
    assert n_lights > 0
    rng = np.random.default_rng(5)
    per = 192
    light = np.repeat(np.arange(n_lights, dtype=np.int32), per)
    n = len(light)
    ref = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    u3 = rng.random((n, 3)).astype(np.float32)
    edge = np.array([0.0, np.nextafter(np.float32(1), np.float32(0)), 0.5, 0.25, 0.75, np.float32(1e-8)], np.float32)
    for li in range(n_lights):
        u3[li * per:li * per + 6, 0] = edge
        u3[li * per + 6:li * per + 12, 1] = edge
        u3[li * per + 12:li * per + 18, 2] = edge
    got = sc.light_query(light, ref, u3)
    want = np.stack([ora.light_sample(int(light[i]), ref[i], float(u3[i, 0]), float(u3[i, 1]), float(u3[i, 2])) for i in range(n)])
    assert (got[:, 13] == want[:, 13]).all()                                  # the triangle: integer work
    assert np.array_equal(got[:, 9] == 0, want[:, 9] == 0)                    # back-facing samples have pdf 0 on both sides
    assert same_bits(got[:, :13], want[:, :13]), np.abs(got[:, :13] - want[:, :13]).max()
    assert (got[:, 9] > 0).any() and (got[:, 9] == 0).any()
    z = got[:, 9] == 0
    assert (got[z, 10:13] == 0).all()


def test_device_cdf_draws_match_the_reference_discrete_pdf(gpu_lib, kz):
    """a19 on the device against the REFERENCE'S OWN DiscretePDF (tests/golden/int_kats.json "dpdf": struct DiscretePDF of dpdf.h compiled where it lies,
    oracle/kat_ref_dpdf.cpp): every table of the fixture becomes a light mesh whose triangles have exactly the fixture's areas - legs (a, 2) with a = m 2^e,
    m < 2^11, so 0.5 * |e1 x e2| = a without rounding (zero areas: degenerate triangles) - and the triangle Mesh::sample picks for every minted draw (0,
    every CDF entry and its two neighbours, 1 - ulp, 1) through kz_light_query, the function the shade kernel calls, is DiscretePDF::sample's index. The
    tables have 1 .. 200 entries: both forms of the device search (counting for short tables, bisection for long ones)."""
    import json
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "int_kats.json")))["dpdf"]
    S = kz.scenes
    s = S.SceneDescription()
    tables = [t for t in kats if np.array(t["sum"], np.uint32).view(np.float32) > 0]          # (an all-zero table never normalises: nothing to draw from)
    assert len(tables) >= 8
    for t in tables:
        a = np.array(t["values"], np.uint32).view(np.float32)
        n = len(a)
        V = np.zeros((3 * n, 3), np.float32)
        V[1::3, 0] = a
        V[2::3, 1] = 2.0
        s.add_mesh(V, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), np.tile(np.array([0, 0, 1], np.float32), (3 * n, 1)), light=S.area((1, 1, 1), 1.0, False))
    s.camera.update(width=16, height=16)
    sc = kz.Scene(s, device=0)
    light, u0, want = [], [], []
    for li, t in enumerate(tables):
        for v_bits, idx in t["sample"]:
            v = np.array(v_bits, np.uint32).view(np.float32)
            if v < 1.0:                                             # (Sampler::next1D never returns 1)
                light.append(li); u0.append(v); want.append(idx)
    n = len(light)
    u3 = np.full((n, 3), 0.5, np.float32)
    u3[:, 0] = np.array(u0, np.float32)
    got = sc.light_query(np.array(light, np.int32), np.tile(np.array([[0.3, 0.4, 5.0]], np.float32), (n, 1)), u3)
    assert n > 800 and (got[:, 13].astype(np.int64) == np.array(want)).all(), np.nonzero(got[:, 13].astype(np.int64) != np.array(want))[0][:10]


def test_exact_reciprocal_and_sqrt_equal_ieee_for_every_float(gpu_lib, kz):
    """rcpExact / sqrtExact (kz_devfn.h: v_rcp_f32 / v_rsq_f32 + Newton steps with a range guard) replace the compiler's IEEE division by
    1 and sqrtf in the triangle test, the ray set-up and the BSDFs. They must not change a single bit: the library checks them against
    the compiler's sequences on ALL 2^32 float bit patterns (~1 s on the device)."""
    import ctypes as C
    r, s, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
    kz.abi.check(gpu_lib, gpu_lib.kz_kat_exact_math(0, C.byref(r), C.byref(s), C.byref(n)))
    assert n.value == 1 << 32
    assert r.value == 0 and s.value == 0, (r.value, s.value)


def test_permute_on_the_device_matches_the_reference_text(gpu_lib, kz):
    """random::permute as the sampler kernels compute it (kz_devfn.h permuteIdx, with its power-of-two shortcut) against the vectors minted by compiling
    the reference's own text of common.cpp:300-346 (oracle/kat_ref_permute.cpp -> tests/golden/int_kats.json)."""
    import json
    import os
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "int_kats.json")))
    rows = [(i, l, int(k) & 0xffffffff, o) for i, l, k, o in kats["permute"]]
    for e in kats["permute_full"]:
        rows += [(i, e["l"], e["key"], o) for i, o in enumerate(e["out"])]
    i, l, p, want = (np.array(c, np.uint32) for c in zip(*rows))
    got = np.zeros(len(rows), np.uint32)
    f = lambda a: a.ctypes.data_as(kz.abi.u32p)
    kz.abi.check(gpu_lib, gpu_lib.kz_kat_permute(0, len(rows), f(i), f(l), f(p), f(got)))
    assert np.array_equal(got, want)


Q1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "q1_default_m0_r0.5.npz")


def test_reference_asset_scene_through_the_hip_path(gpu_lib, kz, O):
    """The reference's OWN asset geometry (scene/2022_q1/parameters/default_m0_r0.5.xml: 36 378 triangles, smooth vertex normals on the kiss object and the
    backdrop - the Hanika terminator offset and the interpolated shading frames of accel.cpp:141-229 on real data - three invisible area lights) flattened to
    tests/golden/q1_default_m0_r0.5.npz by tests/golden/make_q1_scene.py, SURVEY 8d C1 at its stated size 256x256x16: HIP film vs the oracle, rays vs brute force."""
    d = kz.scenes.load_npz(Q1, {"camera": {"width": 256, "height": 256}, "sampler": {"sampleCount": 16}})
    assert d.n_tris() == 36378 and d.sampler["type"] == "independent" and d.integrator["maxDepth"] == 5
    sc = kz.Scene(d, device=0)
    sc.render()
    ora = O.OracleScene(d)
    cpu = ora.rgb(ora.render(threads=0))
    gpu = sc.rgb()
    assert gpu.mean() > 0.05
    assert float(np.sqrt(np.mean((gpu - cpu) ** 2))) < 1e-3
    assert np.array_equal(sc.film(), ora.render_canonical(threads=0))     # BASELINE configs[0] (C1) - the reference's own asset at 256 x 256 x 16 -: the whole film, bit for bit
    sc.render(pipeline=1)                                                 # the reference-shaped megakernel: the same film bit for bit
    assert np.array_equal(sc.rgb(), gpu)
    # closest hits of 20 000 rays around the object against the brute-force Moeller-Trumbore search over all 36 378 triangles
    rng = np.random.default_rng(5)
    lo, hi = np.min([m["V"].min(axis=0) for m in d.meshes[3:]], axis=0), np.max([m["V"].max(axis=0) for m in d.meshes[3:]], axis=0)
    c, ext = (lo + hi) / 2, (hi - lo) / 2
    o = (c + rng.uniform(-1.2, 1.2, (20000, 3)) * ext).astype(np.float32)
    tgt = (c + rng.uniform(-0.6, 0.6, (20000, 3)) * ext).astype(np.float32)
    dirs = tgt - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    hg = sc.trace_rays(o, dirs.astype(np.float32), 1e-3, np.inf)
    hc = O.OracleScene(d, brute=True).trace_rays(o, dirs.astype(np.float32), 1e-3, np.inf)
    hit = hc["mesh"] >= 0
    assert hit.mean() > 0.3
    assert np.array_equal(hg["mesh"], hc["mesh"]) and np.array_equal(hg["prim"], hc["prim"])
    assert np.all(np.abs(hg["t"][hit] - hc["t"][hit]) <= 1e-4 * hc["t"][hit])
    for k, tol in (("p", 2e-5), ("sh_n", 1e-5), ("sh_s", 1e-4), ("sh_t", 1e-4), ("geo_n", 1e-5), ("u", 1e-5)):
        assert np.abs(hg[k][hit] - hc[k][hit]).max() <= tol, k


def test_fresnel_on_the_device_matches_the_reference_text(gpu_lib, kz):
    """The device's fresnelIOR / fresnelDielectricT (kz_devfn.h) against the bit patterns minted from the reference's own text (oracle/kat_ref_fresnel.cpp):
    the divisions are the compiler's IEEE sequences and the square roots sqrtExact, so every bit must agree."""
    import json
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "int_kats.json")))
    f = lambda a: a.ctypes.data_as(kz.abi.f32p)
    rows = np.array(kats["fresnel_ior"], np.uint32)
    c, e, i = (np.ascontiguousarray(rows[:, k]).view(np.float32) for k in range(3))
    out = np.zeros((len(rows), 2), np.float32)
    kz.abi.check(gpu_lib, gpu_lib.kz_kat_fresnel(0, len(rows), 0, f(c), f(e), f(i), f(out)))
    assert np.array_equal(out[:, 0].view(np.uint32), rows[:, 3])
    rows = np.array(kats["fresnel_dielectric"], np.uint32)
    c, e = (np.ascontiguousarray(rows[:, k]).view(np.float32) for k in range(2))
    out = np.zeros((len(rows), 2), np.float32)
    kz.abi.check(gpu_lib, gpu_lib.kz_kat_fresnel(0, len(rows), 1, f(c), f(e), None, f(out)))
    assert np.array_equal(out[:, 0].view(np.uint32), rows[:, 2]) and np.array_equal(out[:, 1].view(np.uint32), rows[:, 3])


# ---------------------------------------------------------------- round 6: whole FILMS equal to the oracle's, bit for bit
def _film_cases(S):
    return {
        "cornell_pmj": lambda: S.cornell_box(96, 72, 16, sampler="pmj02bn"),
        "cornell_independent_ragged": lambda: S.cornell_box(77, 45, 9),
        "hero_kiss": lambda: S.hero_scene(160, 96, 8, detail=0.4),
        "materials": lambda: S.materials_scene(160, 96, 8),
        "textured": lambda: S.textured_scene(128, 80, 8),
        "sphere_env": lambda: S.sphere_env(96, 96, 8),
        "random_triangles_pmj": lambda: S.random_triangles(20000, 192, 108, 16, sampler="pmj02bn", seed=1),
    }


@pytest.mark.parametrize("name", ["cornell_pmj", "cornell_independent_ragged", "hero_kiss", "materials", "textured", "sphere_env", "random_triangles_pmj"])
def test_whole_films_equal_the_oracle_bit_for_bit(gpu_lib, kz, O, name):
    """Every sample's radiance has been the oracle's to the last bit since round 4, and the filter weights are ImageBlock::put's by construction; what kept the FILMS apart
    was the order of the float additions (H10: the reference's own depends on its thread timing). Round 6 fixes that order on the device - per pixel and tap in sample order,
    texels resolved over the 64-px tile grid in tile order - and the oracle states the same order (kzo_render_canonical): the whole film, weights channel and apron included,
    is now EQUAL - for the default gaussian, a tent and a 7-tap gaussian, under another pass schedule, and for a tile set."""
    desc = _film_cases(kz.scenes)[name]()
    ora = O.OracleScene(desc)
    want = ora.render_canonical(threads=0)
    sc = kz.Scene(desc, device=0)
    sc.render()
    assert np.array_equal(sc.film(), want)
    sc.render(pass_items=4096, passes_in_flight=3)
    assert np.array_equal(sc.film(), want)
    w, h = desc.camera["width"], desc.camera["height"]
    tiles = [(0, 0, min(64, w), min(40, h)), (min(64, w), 8, w - min(64, w), h - 8)] if w > 64 else [(0, 0, w, h // 2)]
    sc.render(tiles=tiles)
    assert np.array_equal(sc.film(), ora.render_canonical(tiles=tiles, threads=0))
    sc.close()
    for filt in ({"type": "tent"}, {"type": "gaussian", "radius": 3.0, "stddev": 0.8}):
        desc.camera["rfilter"] = filt
        s2 = kz.Scene(desc, device=0)
        s2.render()
        assert np.array_equal(s2.film(), O.OracleScene(desc).render_canonical(threads=0)), filt
        s2.close()
