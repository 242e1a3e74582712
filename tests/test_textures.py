"""Texture<Color3f> trees (texture.cpp) behind BSDF parameters and the NormalMap wrapper (bsdf.cpp:281-417).

CPU part: known answers of the oracle's restatement (the bilinear lookup is DECLARED, not pinned: OpenImageIO is absent from the
reference checkout) and the host-side validation of the product library. GPU part: the HIP path against the oracle."""
import numpy as np
import pytest


def _scene_with(kz, bsdfs):
    """One triangle per BSDF dict: a container to get the rows / textures into a KzScene."""
    S = kz.scenes
    d = S.SceneDescription()
    for b in bsdfs:
        d.add_mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=b)
    d.camera.update(width=8, height=8)
    return d


def _tex_scene(kz, textures):
    return _scene_with(kz, [kz.scenes.lambertian(t) for t in textures])


# ----------------------------------------------------------------------------- oracle known answers
def test_image_lookup_known_answers(kz, O):
    S = kz.scenes
    img = np.array([[[0.0, 0.1, 0.2], [1.0, 0.5, 0.2]],
                    [[0.5, 0.3, 0.2], [0.25, 0.7, 0.2]]], np.float32)                   # 2 x 2, row 0 = top
    o = O.OracleScene(_tex_scene(kz, [S.imagetexture(img, 1.0, "linear"), S.imagetexture(img, 2.0, "linear"), S.imagetexture(img, 1.0, "srgb"),
                                      S.imagetexture(img[:, :, :1], 1.0, "linear")]))
    # texel centres: s = (i + .5) / 2, t = 1 - v  ->  v = 0.75 is the TOP row (texture.cpp:55 flips v)
    c = o.texture(0, [(0.25, 0.75), (0.75, 0.75), (0.25, 0.25), (0.75, 0.25)])
    assert np.array_equal(c, img.reshape(4, 3))
    # halfway between the two top texels; halfway between all four
    assert np.allclose(o.texture(0, [(0.5, 0.75)])[0], (img[0, 0] + img[0, 1]) / 2, atol=1e-7)
    assert np.allclose(o.texture(0, [(0.5, 0.5)])[0], img.reshape(4, 3).mean(0), atol=1e-7)
    # periodic wrap (texture.cpp:48-49): one period away is the same texel; left of texel 0 blends with the last column
    assert np.array_equal(o.texture(0, [(1.25, -0.25), (-0.75, 1.75)]), np.stack([img[0, 0], img[0, 0]]))
    assert np.allclose(o.texture(0, [(0.0, 0.75)])[0], (img[0, 0] + img[0, 1]) / 2, atol=1e-7)
    # "scale" multiplies both coordinates AFTER the v flip: (u, 1 - v) * 2
    assert np.array_equal(o.texture(1, [(0.125, 0.875)]), o.texture(0, [(0.25, 0.75)]))
    # colorspace "srgb": Color3f::toLinearRGB of the filtered value (common.cpp:368-382)
    v = img[0, 1]
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    assert np.allclose(o.texture(2, [(0.75, 0.75)])[0], lin, rtol=2e-6)
    # one-channel file asked for three channels: missing channels are filled with 0
    assert np.array_equal(o.texture(3, [(0.75, 0.75)])[0], np.array([1.0, 0, 0], np.float32))


def test_bicubic_filter_known_answers(kz, O):
    """KzTexture.filter = KZ_TEXFILTER_BICUBIC: the 4 x 4 cubic B-spline (what OpenImageIO's default "smart bicubic" mode most likely evaluates for the
    reference's magnifying, zero-derivative lookups - the hazard recorded in DESIGN.md 2). A B-spline reproduces constants and linear ramps exactly
    (away from the periodic seam), blurs a single texel to the weights 1/6, 4/6, 1/6 per axis, and is not the bilinear lookup."""
    S = kz.scenes
    ramp = np.tile(np.linspace(0, 1, 16, dtype=np.float32)[None, :, None], (16, 1, 3))              # f(x) = x / 15 along s
    dot = np.zeros((8, 8, 1), np.float32); dot[3, 4, 0] = 1.0
    o = O.OracleScene(_tex_scene(kz, [S.imagetexture(np.full((4, 4, 3), 0.37, np.float32), 1.0, "linear", "bicubic"), S.imagetexture(ramp, 1.0, "linear", "bicubic"),
                                      S.imagetexture(ramp, 1.0, "linear", "bilinear"), S.imagetexture(dot, 1.0, "linear", "bicubic"), S.imagetexture(dot, 1.0, "linear")]))
    uv = np.random.default_rng(3).uniform(0, 1, (200, 2)).astype(np.float32)
    assert np.allclose(o.texture(0, uv), 0.37, atol=2e-7)
    inner = uv[(uv[:, 0] > 3 / 16) & (uv[:, 0] < 13 / 16)]
    want = (inner[:, 0] * 16 - 0.5) / 15                                                               # the ramp at the continuous texel coordinate
    assert np.allclose(o.texture(1, inner)[:, 0], want, atol=3e-6) and np.allclose(o.texture(2, inner)[:, 0], want, atol=3e-6)
    at = lambda j, i: ((i + 0.5) / 8, 1 - (j + 0.5) / 8)                                               # uv of the centre of texel (row j, column i)
    c = o.texture(3, [at(3, 4), at(3, 3), at(2, 4), at(2, 3), at(3, 6), at(5, 4)])[:, 0]
    assert np.allclose(c, [16 / 36, 4 / 36, 4 / 36, 1 / 36, 0, 0], atol=1e-7)
    assert np.allclose(o.texture(4, [at(3, 4), at(3, 3)])[:, 0], [1, 0])                               # bilinear at texel centres: the texels themselves


def test_u8_rasters_are_value_over_255(kz, O):
    img = np.array([[[255, 128, 0], [51, 102, 204]]], np.uint8)
    o = O.OracleScene(_tex_scene(kz, [kz.scenes.imagetexture(img, 1.0, "linear")]))
    assert np.array_equal(o.texture(0, [(0.25, 0.5), (0.75, 0.5)]), (img[0].astype(np.float32) * np.float32(1 / 255)))


def test_colorramp_and_blend_known_answers(kz, O):
    S = kz.scenes
    c = S.constanttexture((0.2, 1.5, -0.5))
    texs = [S.colorramp(c, 0.1, 0.9), S.colorramp(None, 0.1, 0.9),
            S.blend(S.constanttexture((0.25, 9, 9)), S.constanttexture((1, 2, 3)), S.constanttexture((5, 6, 7)), "mix"),
            S.blend(None, S.constanttexture((1, 2, 3)), S.constanttexture((5, 6, 7)), "multiply"),
            S.blend(None, None, None, "mix"), S.blend(None, None, None, "screen"),
            S.colorramp(S.blend(S.colorramp(c), S.blend(None, c, c, "multiply"), c), 0.0, 2.0)]
    d = _tex_scene(kz, texs)
    o = O.OracleScene(d)
    ids = {id(row): i for i, row in enumerate([])}
    # texture rows are numbered children-first; find each root through its lambertian row
    cd = d.to_c()
    roots = [cd.bsdfs[i].albedoTex - 1 for i in range(len(texs))]
    uv = [(0.3, 0.6)]
    assert np.allclose(o.texture(roots[0], uv)[0], [0.1 + 0.8 * 0.2, 0.9, 0.1])          # clamp to [0,1], then min + (max-min)*x
    assert np.array_equal(o.texture(roots[1], uv)[0], [0, 0, 0])                          # no nested texture: 0 (texture.cpp:170)
    assert np.allclose(o.texture(roots[2], uv)[0], [0.75 * 1 + 0.25 * 5, 0.75 * 2 + 0.25 * 6, 0.75 * 3 + 0.25 * 7])   # mask.x for all channels
    assert np.array_equal(o.texture(roots[3], uv)[0], [5, 12, 21])
    assert np.array_equal(o.texture(roots[4], uv)[0], [0.5, 0.5, 0.5])                    # defaults: mask .5, input1 0, input2 1
    assert np.array_equal(o.texture(roots[5], uv)[0], [0, 0, 0])                          # unknown blend mode (texture.cpp:236)
    assert np.all(np.isfinite(o.texture(roots[6], uv)))


def test_textured_parameters_equal_their_constant_rows(kz, O):
    """A constanttexture child and the folded constant are the same BSDF."""
    S = kz.scenes
    rows = [S.kazenstandard((0.7, 0.3, 0.2), 0.35, 0.6, clearcoat=0.4), S.kazenstandard(S.constanttexture((0.7, 0.3, 0.2)), S.constanttexture((0.35, 9, 9)), S.constanttexture((0.6, 9, 9)), clearcoat=0.4),
            S.diffuse((0.2, 0.5, 0.9)), S.lambertian(S.constanttexture((0.2, 0.5, 0.9))), S.ggx((0.9, 0.5, 0.1), 0.3), S.ggx(S.constanttexture((0.9, 0.5, 0.1)), 0.3)]
    o = O.OracleScene(_scene_with(kz, rows))
    rng = np.random.default_rng(3)
    for _ in range(50):
        wi = rng.normal(size=3); wi[2] = abs(wi[2]) + 0.05; wi /= np.linalg.norm(wi)
        wo = rng.normal(size=3); wo[2] = abs(wo[2]) + 0.05; wo /= np.linalg.norm(wo)
        s = rng.random(3)
        for a, b in ((0, 1), (2, 3), (4, 5)):
            assert np.array_equal(o.bsdf(a, "eval", wi, wo, 0.1), o.bsdf(b, "eval", wi, wo, 0.1))
            assert o.bsdf(a, "pdf", wi, wo, 0.1) == o.bsdf(b, "pdf", wi, wo, 0.1)
            assert np.array_equal(o.bsdf(a, "sample", wi, None, 0.1, s[0], (s[1], s[2])), o.bsdf(b, "sample", wi, None, 0.1, s[0], (s[1], s[2])))


def test_normalmap_known_answers(kz, O):
    S = kz.scenes
    flat = S.constanttexture((0.5, 0.5, 1.0))                         # n = (0, 0, 1): the perturbed frame is the shading frame
    tilt = S.constanttexture((0.5 + 0.5 * 0.6, 0.5, 0.5 + 0.5 * 0.8))      # n = (0.6, 0, 0.8)
    kiss = S.kazenstandard((0.7, 0.3, 0.2), 0.35, 0.2)
    rows = [kiss, S.normalmap(flat, kiss), S.normalmap(tilt, kiss), S.diffuse((0.4, 0.5, 0.6)), S.normalmap(flat, S.diffuse((0.4, 0.5, 0.6))),
            S.normalmap(flat, S.mirror()), S.mirror()]
    o = O.OracleScene(_scene_with(kz, rows))
    wi = np.array([0.3, -0.2, 0.9]); wi /= np.linalg.norm(wi)
    wo = np.array([-0.5, 0.1, 0.7]); wo /= np.linalg.norm(wo)
    # flat map, accumulatedRoughness 0: identical to the nested BSDF up to the rounding of the frame change
    assert np.allclose(o.bsdf(1, "eval", wi, wo), o.bsdf(0, "eval", wi, wo), rtol=1e-5)
    assert np.isclose(o.bsdf(1, "pdf", wi, wo), o.bsdf(0, "pdf", wi, wo), rtol=1e-5)
    # the perturbed record carries a fresh Intersection: the nested kiss does not see its.accumulatedRoughness (bsdf.cpp:301-311)
    assert np.allclose(o.bsdf(1, "eval", wi, wo, 0.3), o.bsdf(0, "eval", wi, wo, 0.0), rtol=1e-5)
    assert not np.allclose(o.bsdf(0, "eval", wi, wo, 0.3), o.bsdf(0, "eval", wi, wo, 0.0), rtol=1e-3)
    # tilted normal = the nested BSDF in the rotated frame: n' = (.6, 0, .8), s' = normalize(x - n'(n'.x)) = (.8, 0, -.6), t' = n' x s' = y
    R = np.array([[0.8, 0, -0.6], [0, 1, 0], [0.6, 0, 0.8]])
    assert np.allclose(o.bsdf(2, "eval", wi, wo), o.bsdf(0, "eval", R @ wi, R @ wo), rtol=2e-5, atol=1e-7)
    # wo below the perturbed horizon: cosTheta(wo) * cosTheta(wo') <= 0 -> 0 (bsdf.cpp:305-306)
    graze = np.array([-0.9, 0.0, 0.3]); graze /= np.linalg.norm(graze)
    assert (R @ graze)[2] < 0 and np.array_equal(o.bsdf(2, "eval", wi, graze), [0, 0, 0])
    # n.wi <= 0 with both directions above the horizon: the nested BSDF is used unperturbed (bsdf.cpp:295-296)
    back = np.array([-0.9, 0.0, 0.2]); back /= np.linalg.norm(back)
    assert np.dot([0.6, 0, 0.8], back) < 0 and np.array_equal(o.bsdf(2, "eval", back, wo, 0.2), o.bsdf(0, "eval", back, wo, 0.2))
    # H13: NormalMap::sample never copies the nested record's measure back (bsdf.cpp:348-362): the integrator's pdf(bRec) right
    # after sample() sees EUnknownMeasure, so a nested Diffuse reports pdf 0 and a nested mirror is not treated as EDiscrete
    s_plain = o.bsdf(3, "sample", wi, None, 0, 0.3, (0.4, 0.7))
    s_nm = o.bsdf(4, "sample", wi, None, 0, 0.3, (0.4, 0.7))
    assert np.allclose(s_nm[:6], s_plain[:6], rtol=1e-5, atol=1e-7) and s_plain[7] > 0 and s_nm[7] == 0
    m_nm, m_plain = o.bsdf(5, "sample", wi, None, 0, 0.3, (0.4, 0.7)), o.bsdf(6, "sample", wi, None, 0, 0.3, (0.4, 0.7))
    assert np.allclose(m_nm[:6], m_plain[:6], rtol=1e-5, atol=1e-7)


def test_textured_scene_renders_on_the_oracle(kz, O):
    d = kz.scenes.textured_scene(64, 40, 4)
    o = O.OracleScene(d)
    rgb = o.rgb(o.render(threads=0))
    assert np.isfinite(rgb).all() and rgb.mean() > 0.01 and o.stats()["droppedSamples"] == 0
    # brute-force traversal gives the same film: the texture / normalmap rows do not depend on the BVH
    o2 = O.OracleScene(d, brute=True)
    assert np.array_equal(o2.render(threads=0), o.render(threads=0))


# ----------------------------------------------------------------------------- product library, host side
def test_invalid_texture_descriptions_are_loud(kz):
    S, a = kz.scenes, kz.abi

    def create(mutate):
        d = _scene_with(kz, [S.lambertian(S.imagetexture(np.zeros((2, 2, 3), np.uint8))), S.normalmap(S.constanttexture(), S.diffuse())])
        cd = d.to_c()
        mutate(cd)
        h = a.C.c_void_p()
        return a.load_library().kz_scene_create(a.C.byref(cd), a.C.byref(h))

    assert create(lambda cd: None) == a.KZ_OK
    assert create(lambda cd: setattr(cd.bsdfs[0], "albedoTex", 7)) == a.KZ_ERR_INVALID_ARG                 # texture id out of range
    assert create(lambda cd: setattr(cd.textures[0], "image", 3)) == a.KZ_ERR_INVALID_ARG                  # image index out of range
    assert create(lambda cd: setattr(cd.textures[0], "type", 9)) == a.KZ_ERR_UNSUPPORTED                   # unknown texture plugin
    assert create(lambda cd: setattr(cd.bsdfs[1], "nested", 1)) == a.KZ_ERR_INVALID_ARG                    # normalmap nested in itself
    assert create(lambda cd: setattr(cd.bsdfs[1], "normalTex", 0)) == a.KZ_ERR_INVALID_ARG                 # normalmap without a texture
    assert create(lambda cd: setattr(cd.bsdfs[2], "roughnessTex", 1)) == a.KZ_ERR_INVALID_ARG              # diffuse has no roughness texture
    assert create(lambda cd: setattr(cd.images[0], "width", 0)) == a.KZ_ERR_INVALID_ARG

    def cyclic(cd):
        cd.textures[1].type = a.KZ_TEX_COLORRAMP
        cd.textures[1].child[0] = 1
    assert create(cyclic) == a.KZ_ERR_UNSUPPORTED


def test_deep_texture_trees_hit_the_declared_limit(kz):
    S = kz.scenes
    t = S.constanttexture()
    for _ in range(3):                               # each level keeps mask + input1 on the stack while input2 is evaluated
        t = S.blend(S.constanttexture(), S.constanttexture(), t)
    kz.Scene(_tex_scene(kz, [t]))                    # operand stack 7 <= KZ_TEX_MAX_DEPTH
    with pytest.raises(kz.abi.KzError) as e:
        kz.Scene(_tex_scene(kz, [S.blend(S.constanttexture(), S.constanttexture(), t)]))
    assert e.value.code == kz.abi.KZ_ERR_UNSUPPORTED


# ----------------------------------------------------------------------------- GPU parity
def _l2(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.mark.gpu
def test_texture_eval_matches_oracle(gpu_lib, kz, O):
    S = kz.scenes
    chk, noise, gray, nrm = S._test_images()
    lin = [S.imagetexture(chk, 6.0, "linear"), S.imagetexture(noise, 2.0, "linear"), S.imagetexture(gray, 3.0, "linear"), S.imagetexture(nrm, 0.7, "linear")]
    lin += [S.imagetexture(noise, 2.0, "linear", "bicubic"), S.imagetexture(gray, 0.9, "linear", "bicubic")]      # KzTexture.filter: cubic B-spline
    srgb = [S.imagetexture(chk, 6.0, "srgb"), S.imagetexture(noise, 1.3, "srgb")]
    tree = [S.colorramp(lin[2], 0.1, 0.8), S.blend(lin[2], S.constanttexture((0.8, 0.25, 0.2)), lin[1]), S.blend(None, lin[0], srgb[1], "multiply"),
            S.colorramp(S.blend(S.colorramp(lin[1]), S.blend(None, lin[0], lin[3], "multiply"), srgb[0]), -0.5, 2.0)]
    d = _tex_scene(kz, lin + srgb + tree)
    cd = d.to_c()
    roots = [cd.bsdfs[i].albedoTex - 1 for i in range(len(lin + srgb + tree))]
    sc = kz.Scene(d, device=0)
    o = O.OracleScene(d)
    rng = np.random.default_rng(11)
    uv = np.concatenate([rng.uniform(-3, 4, (3000, 2)), [[0, 0], [1, 1], [0.5, 0.5], [-1e6, 2e6], [np.inf, 0.2], [np.nan, 0.1]]]).astype(np.float32)
    for k, r in enumerate(roots):
        g = sc.texture_query(np.full(len(uv), r, np.int32), uv)
        c = o.texture(r, uv)
        both_nan = np.isnan(g) & np.isnan(c)
        assert ((g.view(np.uint32) == c.view(np.uint32)) | both_nan).all(), k                  # bit exact, the pow of the sRGB curve included (kz_crmath.h)


@pytest.mark.gpu
def test_textured_and_normalmapped_bsdfs_match_oracle(gpu_lib, kz, O):
    S = kz.scenes
    chk, noise, gray, nrm = S._test_images()
    t_noise, t_gray, t_nrm = S.imagetexture(noise, 2.0, "linear"), S.imagetexture(gray, 3.0, "linear"), S.imagetexture(nrm, 2.0, "linear")
    rows = [S.kazenstandard(t_noise, S.colorramp(t_gray, 0.1, 0.8), t_gray, clearcoat=0.5), S.lambertian(t_noise), S.ggx(t_noise, 0.35),
            S.normalmap(t_nrm, S.kazenstandard((0.8, 0.5, 0.2), S.colorramp(t_gray, 0.2, 0.6), 0.0, specular=0.8)),
            S.normalmap(t_nrm, S.diffuse((0.3, 0.7, 0.4))), S.normalmap(t_nrm, S.roughconductor(0.25, "Cu")), S.normalmap(t_nrm, S.dielectric()),
            S.normalmap(t_nrm, S.mirror()), S.normalmap(t_nrm, S.roughdielectric(0.3)), S.normalmap(t_nrm, S.lambertian(t_noise))]
    d = _scene_with(kz, rows)
    sc = kz.Scene(d, device=0)
    o = O.OracleScene(d)
    rng = np.random.default_rng(5)
    n = 400
    wi = rng.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi[: n // 2, 2] = np.abs(wi[: n // 2, 2])
    acc = rng.uniform(0, 0.5, n).astype(np.float32)
    s3 = rng.random((n, 3)).astype(np.float32)
    uv = rng.uniform(-1, 2, (n, 2)).astype(np.float32)
    wi, wo = wi.astype(np.float32), wo.astype(np.float32)
    for r in range(len(rows)):
        ev, pd, sm = sc.bsdf_query(np.full(n, r, np.int32), wi, wo, acc, s3, uv)
        for i in range(n):
            a = (wi[i], wo[i], float(acc[i]))
            assert np.array_equal(ev[i], o.bsdf(r, "eval", *a, uv=uv[i])), (r, i)               # every value below: the oracle's bits
            assert pd[i] == np.float32(o.bsdf(r, "pdf", *a, uv=uv[i])), (r, i)
            so = o.bsdf(r, "sample", wi[i], None, float(acc[i]), float(s3[i, 0]), (float(s3[i, 1]), float(s3[i, 2])), uv=uv[i])
            zero_g, zero_o = not sm[i, :3].any(), not so[:3].any()
            assert zero_g == zero_o, (r, i)
            if not zero_o:
                assert np.array_equal(sm[i, 3:6], so[3:6]), (r, i)
                assert np.array_equal(sm[i, :3], so[:3]), (r, i)
                assert sm[i, 7] == so[7], (r, i)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", ["independent", "pmj02bn"])
def test_textured_scene_matches_oracle(gpu_lib, kz, O, sampler):
    d = kz.scenes.textured_scene(160, 96, 16, sampler=sampler)
    sc = kz.Scene(d, device=0)
    sc.set_stats(True)
    sc.render()
    film = sc.film()
    st = sc.stats(reset=True)
    o = O.OracleScene(d)
    film_c = o.render(threads=0)
    so = o.stats()
    assert st["samples"] == so["samples"] and st["droppedSamples"] == so["droppedSamples"] == 0
    assert _l2(sc.rgb(film), o.rgb(film_c)) < 1e-3
    sc.render(pipeline=1)                                                               # megakernel: same film bit for bit
    assert np.array_equal(sc.film(), film)


# ---------------------------------------------------------------- environment lookup behind the background texture (texture.cpp:66-80, 104-145)
def _env_image(h=32, w=64):
    """a smooth HDR latitude-longitude map: sky gradient + a bright lobe, float32"""
    t = (np.arange(h, dtype=np.float32)[:, None] + 0.5) / h
    s = (np.arange(w, dtype=np.float32)[None, :] + 0.5) / w
    img = np.zeros((h, w, 3), np.float32)
    img[..., 0] = 0.3 + 0.7 * (1 - t) + 0 * s
    img[..., 1] = 0.2 + 0.5 * np.cos(2 * np.pi * s) ** 2 + 0 * t
    img[..., 2] = 1.5 * np.exp(-((s - 0.25) ** 2 + (t - 0.3) ** 2) * 40.0)
    return img


def _env_scene(kz, w=96, h=80, spp=8, nested="image"):
    d = kz.scenes.sphere_env(w, h, spp)
    tex = {"image": kz.scenes.imagetexture(_env_image(), 3.0, "srgb"),                      # scale / colorspace must be IGNORED by the direction lookup
           "ramp": kz.scenes.colorramp(kz.scenes.constanttexture((0.5, 0.5, 0.5)), 0.0, 1.0),
           "const": kz.scenes.constanttexture((0.25, 0.5, 1.0))}[nested]
    d.background = {"texture": tex, "intensity": 2.0}
    return d


def test_environment_lookup_known_answers(kz, O):
    """The declared mapping (include/kazen_mi355x.h): y-up latitude-longitude, s = atan2(-x, z) / 2pi + 0.5, t = 0.5 - atan2(y, hypot(z, -x)) / pi,
    bilinear, texel centres at (i + 0.5) / res. A 4 x 2 map whose texel (row r, column c) holds (r, c, 0): straight up reads the top row,
    +z reads the seam s = 0.5 (between columns 1 and 2), -x reads s = 0.75 (between columns 2 and 3: bilinear mid-point 2.5)."""
    img = np.zeros((2, 4, 3), np.float32)
    img[..., 0] = np.arange(2)[:, None]
    img[..., 1] = np.arange(4)[None, :]
    d = kz.scenes.sphere_env(16, 16, 1)
    d.meshes = d.meshes[:0]                                               # no geometry ... the scene needs one mesh: a far-away tiny triangle
    d.add_mesh(np.array([[100, 100, 100], [100.001, 100, 100], [100, 100.001, 100]], np.float32), np.array([[0, 1, 2]], np.uint32),
               np.tile([0, 0, 1], (3, 1)).astype(np.float32), None, bsdf=kz.scenes.diffuse((0.5, 0.5, 0.5)))
    d.background = {"texture": kz.scenes.imagetexture(img, 1.0, "linear"), "intensity": 1.0}
    o = O.OracleScene(d)
    bg = lambda v: o.background(np.asarray(v, np.float32))
    assert np.allclose(bg((0, 1, 0)), (0.0, bg((0, 1, 0))[1], 0.0)) and bg((0, 1, 0))[0] == 0.0          # zenith: row 0
    assert bg((0, -1, 0))[0] == 1.0                                                                    # nadir: row 1 (t clamped)
    assert np.allclose(bg((0, 0, 1)), (0.5, 1.5, 0.0), atol=1e-5)                                        # +z: s = 0.5, t = 0.5
    assert np.allclose(bg((-1, 0, 0)), (0.5, 2.5, 0.0), atol=1e-5)                                       # -x: s = 0.75
    assert np.allclose(bg((0, 0, -1)), (0.5, 1.5, 0.0), atol=1e-5)                                       # -z: s = 0 = 1: columns 3 and 0 wrap (periodic) -> 1.5
    assert np.allclose(bg((np.nan, 0, 1)), 0.0)                                                          # scene.cpp:71-76


def test_background_nested_texture_classes(kz, O):
    """BackgroundTexture::eval(Vector3f) -> nested->eval(Vector3f): constanttexture gives its colour, colorramp / blend inherit Texture::eval(Vector3f) = 0."""
    for nested, want in (("const", (0.5, 1.0, 2.0)), ("ramp", (0.0, 0.0, 0.0))):
        o = O.OracleScene(_env_scene(kz, 16, 16, 1, nested))
        assert np.allclose(o.background(np.array([0.3, 0.4, 0.5], np.float32)), want)


@pytest.mark.gpu
@pytest.mark.parametrize("nested", ["image", "const", "ramp"])
def test_environment_background_matches_oracle(gpu_lib, kz, O, nested):
    d = _env_scene(kz, nested=nested)
    sc = kz.Scene(d, device=0)
    sc.render()
    gpu = sc.rgb()
    ora = O.OracleScene(d)
    cpu = ora.rgb(ora.render(threads=0))
    assert float(np.sqrt(np.mean((gpu - cpu) ** 2))) < 1e-3
    if nested == "image":
        assert gpu.mean() > 0.05                                          # the sphere is lit by the map (primary misses stay black, H5)
    sc.render(pipeline=1)                                                  # the reference-shaped megakernel looks the map up the same way
    assert np.array_equal(sc.rgb(), gpu)
