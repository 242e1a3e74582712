"""Host-side scene description (numpy) -> KzSceneDesc, plus the synthetic scenes of BASELINE.json.

The description mirrors the reference's plugin surface by NAME: every dict key below is the
reference's registry string / PropertyList name, and every default is the
reference's default. `SceneDescription.to_c()` flattens it to the POD the C ABI takes.

Scenes are synthesized because (a) scene/2022_q2 is empty and no Cornell box exists in the
reference checkout, and (b) /root/reference is not present on the GPU box.
All randomness comes from a vectorised pcg32 (same generator as include/kazen/pcg32.h,
1-argument seed form), so scenes are bit-reproducible across numpy versions.
"""
import ctypes as C
import math

import numpy as np

from . import abi

# ----------------------------------------------------------------------------- pcg32 (vectorised)
_PCG_MULT = np.uint64(0x5851F42D4C957F2D)
_M64 = (1 << 64) - 1


def _mixbits(v):
    v &= _M64
    v ^= v >> 31
    v = (v * 0x7FB5D329728EA185) & _M64
    v ^= v >> 27
    v = (v * 0x81DADEF4BC2DD44D) & _M64
    v ^= v >> 33
    return v


def pcg32_floats(initseq, n):
    """n floats of pcg32.seed(initseq) [1-arg form: seed(MixBits(s), s)] via nextFloat(), in draw order."""
    mult = 0x5851F42D4C957F2D
    inc = ((initseq << 1) | 1) & _M64
    state = 0
    state = (state * mult + inc) & _M64
    state = (state + _mixbits(initseq)) & _M64
    state = (state * mult + inc) & _M64
    with np.errstate(over="ignore"):
        a = np.full(n, _PCG_MULT, dtype=np.uint64)
        a[0] = np.uint64(1)
        apow = np.multiply.accumulate(a)                    # A^k mod 2^64
        ssum = np.zeros(n, dtype=np.uint64)
        ssum[1:] = np.add.accumulate(apow[:-1])             # sum_{i<k} A^i
        old = apow * np.uint64(state) + np.uint64(inc) * ssum   # state before draw k
    xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
    rot = (old >> np.uint64(59)).astype(np.uint32)
    u = (xorshifted >> rot) | (xorshifted << ((~rot + np.uint32(1)) & np.uint32(31)))
    bits = (u >> np.uint32(9)) | np.uint32(0x3F800000)
    return bits.view(np.float32) - np.float32(1.0)


# ----------------------------------------------------------------------------- description
def _c3(v):
    return v if isinstance(v, dict) else tuple(v)


def diffuse(albedo=(0.5, 0.5, 0.5)):
    return {"type": "diffuse", "albedo": _c3(albedo)}


def kazenstandard(baseColor=(0.5, 0.5, 0.5), roughness=0.5, metallic=0.0, anisotropy=0.0, specular=0.5,
                  specularTint=0.5, clearcoat=0.0, clearcoatRoughness=0.5, sheen=0.0, sheenTint=0.5):
    return {"type": "kazenstandard", "baseColor": _c3(baseColor), "roughness": roughness, "metallic": metallic,
            "anisotropy": anisotropy, "specular": specular, "specularTint": specularTint, "clearcoat": clearcoat,
            "clearcoatRoughness": clearcoatRoughness, "sheen": sheen, "sheenTint": sheenTint}


def mirror():
    return {"type": "mirror"}


def dielectric(intIOR=1.5046, extIOR=1.000277):
    return {"type": "dielectric", "intIOR": intIOR, "extIOR": extIOR}


CONDUCTORS = {"Au": ((0.1431189557, 0.3749570432, 1.4424785571), (3.9831604247, 2.3857207478, 1.6032152899)),
              "Cu": ((0.2004376970, 0.9240334304, 1.1022119527), (3.9129485033, 2.4528477015, 2.1421879552)),
              "Cr": ((4.3696828663, 2.9167024892, 1.6547005413), (5.2064337956, 4.2313645277, 3.7549467933))}      # bsdf.cpp:795-806


def ggx(albedo=(0.5, 0.5, 0.5), roughness=0.5, anisotropy=0.0):
    return {"type": "ggx", "albedo": _c3(albedo), "roughness": roughness, "anisotropy": anisotropy}


def roughconductor(alpha=0.1, material="Au"):
    return {"type": "roughconductor", "alpha": alpha, "material": material}


def roughplastic(alpha=0.1, intIOR=1.5046, extIOR=1.000277, kd=(0.5, 0.5, 0.5)):
    return {"type": "roughplastic", "alpha": alpha, "intIOR": intIOR, "extIOR": extIOR, "kd": tuple(kd)}


def roughdielectric(roughness=0.1, intIOR=1.5046, extIOR=1.000277):
    return {"type": "roughdielectric", "roughness": roughness, "intIOR": intIOR, "extIOR": extIOR}


# Texture<Color3f> nodes (texture.cpp). Wherever the reference reads a parameter through a texture child (diffuse/lambertian/ggx
# "albedo", kiss "baseColor" / "roughness" / "metallic") the BSDF dicts above accept one of these instead of a constant.
def constanttexture(color=(0.5, 0.5, 0.5)):
    return {"type": "constanttexture", "color": tuple(color)}


def imagetexture(image, scale=1.0, colorspace="srgb", filter="bilinear"):
    """image: (H, W, C) uint8 or float32 raster, row 0 = top scan line (what OpenImageIO decodes from the file). filter: "bilinear" (the declared
    default) or "bicubic" (cubic B-spline: KzTexture.filter, the OpenImageIO hazard of DESIGN.md 2)."""
    a = np.asarray(image)
    if a.ndim == 2:
        a = a[:, :, None]
    a = np.ascontiguousarray(a, np.uint8 if a.dtype == np.uint8 else np.float32)
    return {"type": "imagetexture", "image": a, "scale": float(scale), "colorspace": colorspace, "filter": filter}


def colorramp(nested=None, min=0.0, max=1.0):
    return {"type": "colorramp", "min": float(min), "max": float(max), "nested": nested}


def blend(mask=None, input1=None, input2=None, blendmode="mix"):
    return {"type": "blend", "blendmode": blendmode, "mask": mask, "input1": input1, "input2": input2}


def lambertian(albedo):
    """"lambertian" (bsdf.cpp:202-276) = the diffuse model with its albedo read through a texture."""
    return {"type": "diffuse", "albedo": albedo}


def normalmap(normal, nested):
    """"normalmap" (bsdf.cpp:281-417): `normal` is a texture dict, `nested` a BSDF dict (not itself a normalmap)."""
    return {"type": "normalmap", "normal": normal, "nested": nested}


def area(color=(1.0, 1.0, 1.0), intensity=1.0, lightPrimaryVisibility=False):
    return {"type": "area", "color": tuple(color), "intensity": intensity,
            "lightPrimaryVisibility": bool(lightPrimaryVisibility)}


def look_at(origin, target, up):
    """Same convention as the reference's <lookat> (parser.cpp:268-287): columns = left, newUp, dir, origin."""
    o = np.asarray(origin, np.float64)
    d = np.asarray(target, np.float64) - o
    d /= np.linalg.norm(d)
    left = np.cross(np.asarray(up, np.float64), d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m.astype(np.float32)


class SceneDescription:
    """meshes: list of dicts {V (nV,3) f32, N (nV,3)|None, UV (nV,2)|None, F (nF,3) u32, bsdf: dict|None, light: dict|None}"""

    def __init__(self):
        self.meshes = []
        self.camera = {"type": "perspective", "width": 1280, "height": 720, "toWorld": np.eye(4, dtype=np.float32),
                       "fov": 30.0, "nearClip": 1e-4, "farClip": 1e4,
                       "rfilter": {"type": "gaussian", "radius": 2.0, "stddev": 0.5, "B": 1 / 3.0, "C": 1 / 3.0}}
        self.sampler = {"type": "independent", "sampleCount": 1, "seed": 0}
        self.integrator = {"type": "path_mis", "maxDepth": 5, "traceBias": 1e-3, "regularization": False,
                           "accumulatedRoughness": 0.5}
        self.background = None          # {"color": (r,g,b), "intensity": 1.0} or {"texture": <texture dict>, "intensity": 1.0}
        self.tables = None              # (pmj02bn u32 [5,65536,2], bluenoise u16 [48,128,128])
        self._keep = []

    def add_mesh(self, V, F, N=None, UV=None, bsdf=None, light=None):
        m = {"V": np.ascontiguousarray(V, np.float32), "F": np.ascontiguousarray(F, np.uint32),
             "N": None if N is None else np.ascontiguousarray(N, np.float32),
             "UV": None if UV is None else np.ascontiguousarray(UV, np.float32), "bsdf": bsdf, "light": light}
        self.meshes.append(m)
        return m

    def n_tris(self):
        return int(sum(m["F"].shape[0] for m in self.meshes))

    # -- flatten -----------------------------------------------------------------------------
    def to_c(self):
        keep = []
        bsdfs, lights = [], []
        cm = (abi.KzMesh * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            k = cm[i]
            k.V = m["V"].ctypes.data_as(abi.f32p)
            k.F = m["F"].ctypes.data_as(abi.u32p)
            k.N = m["N"].ctypes.data_as(abi.f32p) if m["N"] is not None else None
            k.UV = m["UV"].ctypes.data_as(abi.f32p) if m["UV"] is not None else None
            k.nV, k.nF = m["V"].shape[0], m["F"].shape[0]
            if m["bsdf"] is None:
                k.bsdf = -1
            else:
                k.bsdf = len(bsdfs)
                bsdfs.append(m["bsdf"])
            if m["light"] is None:
                k.light = -1
            else:
                k.light = len(lights)
                lights.append(m["light"])
        # normalmap rows reference their nested BSDF by row index: nested rows go behind the per-mesh rows
        i = 0
        while i < len(bsdfs):
            if bsdfs[i]["type"] == "normalmap":
                bsdfs.append(bsdfs[i]["nested"])
                bsdfs[i] = dict(bsdfs[i], _nested=len(bsdfs) - 1)
            i += 1
        textures, images, tex_ids = [], [], {}

        def tex_id(t):
            """1-based texture id of a texture dict (children first), 0 for None."""
            if t is None:
                return 0
            if id(t) in tex_ids:
                return tex_ids[id(t)]
            kids = {"colorramp": ("nested",), "blend": ("mask", "input1", "input2")}.get(t["type"], ())
            child = [tex_id(t.get(c)) - 1 for c in kids] + [-1] * (3 - len(kids))
            row = {"t": t, "child": child, "image": -1}
            if t["type"] == "imagetexture":
                row["image"] = len(images)
                images.append(t["image"])
            textures.append(row)
            tex_ids[id(t)] = len(textures)
            return len(textures)

        def param(k, b, name, field, texfield):
            v = b[name]
            if isinstance(v, dict):
                setattr(k, texfield, tex_id(v))
            elif isinstance(field, str) and field in ("roughness", "metallic"):
                setattr(k, field, v)
            else:
                getattr(k, field)[:] = v

        cb = (abi.KzBSDF * max(1, len(bsdfs)))()
        for i, b in enumerate(bsdfs):
            k = cb[i]
            if b["type"] == "diffuse":
                k.type = abi.KZ_BSDF_DIFFUSE
                param(k, b, "albedo", "albedo", "albedoTex")
                k.specular, k.specularTint, k.clearcoatRoughness, k.sheenTint = 0.5, 0.5, 0.5, 0.5
            elif b["type"] == "kazenstandard":
                k.type = abi.KZ_BSDF_KAZENSTANDARD
                param(k, b, "baseColor", "baseColor", "albedoTex")
                param(k, b, "roughness", "roughness", "roughnessTex")
                param(k, b, "metallic", "metallic", "metallicTex")
                for f in ("anisotropy", "specular", "specularTint", "clearcoat",
                          "clearcoatRoughness", "sheen", "sheenTint"):
                    setattr(k, f, b[f])
            elif b["type"] == "mirror":
                k.type = abi.KZ_BSDF_MIRROR
            elif b["type"] == "dielectric":
                k.type = abi.KZ_BSDF_DIELECTRIC
                k.intIOR, k.extIOR = b["intIOR"], b["extIOR"]
            elif b["type"] == "ggx":
                k.type = abi.KZ_BSDF_GGX
                param(k, b, "albedo", "albedo", "albedoTex")
                k.alpha, k.anisotropy = b["roughness"], b["anisotropy"]
            elif b["type"] == "roughconductor":
                k.type = abi.KZ_BSDF_ROUGHCONDUCTOR
                k.alpha = b["alpha"]
                k.condEta[:], k.condK[:] = CONDUCTORS[b["material"]]
            elif b["type"] == "roughplastic":
                k.type = abi.KZ_BSDF_ROUGHPLASTIC
                k.alpha, k.intIOR, k.extIOR = b["alpha"], b["intIOR"], b["extIOR"]
                k.albedo[:] = b["kd"]
            elif b["type"] == "roughdielectric":
                k.type = abi.KZ_BSDF_ROUGHDIELECTRIC
                k.alpha, k.intIOR, k.extIOR = b["roughness"], b["intIOR"], b["extIOR"]
            elif b["type"] == "normalmap":
                k.type = abi.KZ_BSDF_NORMALMAP
                k.normalTex = tex_id(b["normal"])
                k.nested = b["_nested"]
            else:
                k.type = 99      # unsupported plugin: the library must answer KZ_ERR_UNSUPPORTED
            k.alphaResolved = 1 if b.get("alphaResolved") else 0      # the value under "alpha" / "roughness" is then the constructor's m_alpha (KzBSDF.alphaResolved)
        bg_tex = tex_id(self.background.get("texture")) if self.background is not None else 0
        ct = (abi.KzTexture * max(1, len(textures)))()
        for i, row in enumerate(textures):
            t, k = row["t"], ct[i]
            k.type = {"constanttexture": abi.KZ_TEX_CONSTANT, "imagetexture": abi.KZ_TEX_IMAGE, "colorramp": abi.KZ_TEX_COLORRAMP,
                      "blend": abi.KZ_TEX_BLEND}.get(t["type"], 99)
            k.color[:] = t.get("color", (0.5, 0.5, 0.5))
            k.image, k.scale, k.srgb = row["image"], t.get("scale", 1.0), 1 if t.get("colorspace", "srgb") == "srgb" else 0
            k.filter = {"bilinear": 0, "bicubic": 1}[t.get("filter", "bilinear")]       # KZ_TEXFILTER_* (the OpenImageIO hazard: include/kazen_mi355x.h)
            k.rampMin, k.rampMax = t.get("min", 0.0), t.get("max", 1.0)
            k.blendMode = {"mix": abi.KZ_BLEND_MIX, "multiply": abi.KZ_BLEND_MULTIPLY}.get(t.get("blendmode", "mix"), abi.KZ_BLEND_NONE)
            k.child[:] = row["child"]
        ci = (abi.KzImage * max(1, len(images)))()
        for i, a in enumerate(images):
            ci[i].pixels = a.ctypes.data
            ci[i].height, ci[i].width, ci[i].channels = a.shape
            ci[i].format = abi.KZ_PIXEL_U8 if a.dtype == np.uint8 else abi.KZ_PIXEL_F32
        self.n_bsdf_rows, self.n_textures = len(bsdfs), len(textures)
        cl = (abi.KzLight * max(1, len(lights)))()
        for i, l in enumerate(lights):
            cl[i].color[:] = l["color"]
            cl[i].intensity = l["intensity"]
            cl[i].primaryVisibility = 1 if l["lightPrimaryVisibility"] else 0
        d = abi.KzSceneDesc()
        d.abiVersion = abi.KZ_ABI_VERSION
        d.meshes, d.nMeshes = cm, len(self.meshes)
        d.bsdfs, d.nBsdfs = cb, len(bsdfs)
        d.lights, d.nLights = cl, len(lights)
        d.textures, d.nTextures = ct, len(textures)
        d.images, d.nImages = ci, len(images)
        cam = self.camera
        d.camera.type = {"perspective": abi.KZ_CAMERA_PERSPECTIVE, "thinlens": abi.KZ_CAMERA_THINLENS}.get(cam["type"], 99)
        d.camera.apertureRadius = cam.get("apertureRadius", 1.0)
        d.camera.focusDistance = cam.get("focusDistance", 0.0)
        d.camera.width, d.camera.height = cam["width"], cam["height"]
        tw = np.ascontiguousarray(cam["toWorld"], np.float32).reshape(16)
        d.camera.toWorld[:] = tw.tolist()
        d.camera.fov, d.camera.nearClip, d.camera.farClip = cam["fov"], cam["nearClip"], cam["farClip"]
        d.camera.sampleToCamera = None
        rf = cam["rfilter"]
        d.camera.rfilter.type = {"gaussian": 0, "mitchell": 1, "tent": 2, "box": 3}[rf["type"]]
        d.camera.rfilter.radius = {"gaussian": rf.get("radius", 2.0), "mitchell": rf.get("radius", 2.0),
                                   "tent": 1.0, "box": 0.5}[rf["type"]]
        d.camera.rfilter.stddev = rf.get("stddev", 0.5)
        d.camera.rfilter.B, d.camera.rfilter.C = rf.get("B", 1 / 3.0), rf.get("C", 1 / 3.0)
        s = self.sampler
        d.sampler.type = {"independent": 0, "pmj02bn": 1, "stratified": 2, "correlated": 3}.get(s["type"], 99)
        d.sampler.resolution = s.get("resolution", 4)
        d.sampler.sampleCount = s["sampleCount"]
        d.sampler.seed = s.get("seed", 0 if s["type"] == "independent" else 1)
        if s["type"] == "pmj02bn":
            if self.tables is None:
                self.tables = make_pmj02bn_tables()
            pmj, bn = self.tables
            keep += [pmj, bn]
            d.sampler.pmj02bnSamples = pmj.ctypes.data_as(abi.u32p)
            d.sampler.blueNoise = bn.ctypes.data_as(abi.u16p)
        it = self.integrator
        d.integrator.type = 0 if it["type"] == "path_mis" else 99
        d.integrator.maxDepth = it["maxDepth"]
        d.integrator.traceBias = it["traceBias"]
        d.integrator.regularization = 1 if it["regularization"] else 0
        d.integrator.accumulatedRoughness = it["accumulatedRoughness"]
        if self.background is not None:
            d.background.present = 1
            d.background.color[:] = self.background.get("color", (0.0, 0.0, 0.0))
            d.background.intensity = self.background.get("intensity", 1.0)
            d.background.texture = bg_tex
        keep += [cm, cb, cl, ct, ci, images]
        self._keep = keep
        return d


# ----------------------------------------------------------------------------- sampler tables
def _reverse_bits32(x):
    x = x.astype(np.uint32)
    x = ((x >> np.uint32(1)) & np.uint32(0x55555555)) | ((x & np.uint32(0x55555555)) << np.uint32(1))
    x = ((x >> np.uint32(2)) & np.uint32(0x33333333)) | ((x & np.uint32(0x33333333)) << np.uint32(2))
    x = ((x >> np.uint32(4)) & np.uint32(0x0F0F0F0F)) | ((x & np.uint32(0x0F0F0F0F)) << np.uint32(4))
    x = ((x >> np.uint32(8)) & np.uint32(0x00FF00FF)) | ((x & np.uint32(0x00FF00FF)) << np.uint32(8))
    return (x >> np.uint32(16)) | (x << np.uint32(16))


def _owen_scramble(x, seed):
    """Laine-Karras hash applied to the bit-reversed value: a nested uniform (Owen) scramble."""
    with np.errstate(over="ignore"):
        x = _reverse_bits32(x)
        x = x + np.uint32(seed)
        x ^= x * np.uint32(0x6C50B47C)
        x ^= x * np.uint32(0xB82F1E52)
        x ^= x * np.uint32(0xC7AFE638)
        x ^= x * np.uint32(0x8D22F6E6)
        return _reverse_bits32(x)


def make_pmj02bn_tables(seed=2022, dither="void_and_cluster"):
    """Build-generated stand-ins for pmj02bnSamples[5][65536][2] / BlueNoiseTextures[48][128][128].

    The reference's table sources (src/kazen/pmj02table.cpp, bluenoise.cpp) are missing from the
    checkout (.MISSING_LARGE_BLOBS), so table VALUES are unpinned. The sample sets here are
    Owen-scrambled (0,2)-sequences (base-2 Sobol' pair): every prefix of 4^k points is a (0,2k,2)-net,
    which is the stratification property the PMJ02BN constructor relies on (sampler.cpp:295-314).
    The low 8 bits are cleared so that value * 2^-32 never rounds to 1.0f (the reference would index
    out of range there). The dither textures are genuine blue noise since round 5 (void-and-cluster,
    `blue_noise_textures`; rounds 1-4: white noise, still there as dither="white") — spectral quality is
    not on the parity path, but it is what the sampler's per-pixel shifts exist for (bluenoise.h:16-23,
    sampler.cpp:339-365). A user of the real tables passes them through KzSampler unchanged.
    """
    n = abi.KZ_PMJ02BN_SAMPLES
    i = np.arange(n, dtype=np.uint32)
    x0 = _reverse_bits32(i)                                       # van der Corput
    y0 = np.zeros(n, dtype=np.uint32)                              # 2nd Sobol' dimension
    dirv = np.uint32(0x80000000)
    for k in range(16):
        y0 ^= np.where((i >> np.uint32(k)) & np.uint32(1), dirv, np.uint32(0)).astype(np.uint32)
        dirv = np.uint32(dirv ^ (dirv >> np.uint32(1)))
    pmj = np.zeros((abi.KZ_PMJ02BN_SETS, n, 2), dtype=np.uint32)
    for s in range(abi.KZ_PMJ02BN_SETS):
        pmj[s, :, 0] = _owen_scramble(x0, (seed * 2654435761 + 2 * s + 1) & 0xFFFFFFFF)
        pmj[s, :, 1] = _owen_scramble(y0, (seed * 40503 + 7919 * (2 * s + 2)) & 0xFFFFFFFF)
    pmj &= np.uint32(0xFFFFFF00)
    bn = blue_noise_textures() if dither == "void_and_cluster" else _white_noise_textures(seed)
    return np.ascontiguousarray(pmj), np.ascontiguousarray(bn)


def _white_noise_textures(seed):
    """Rounds 1-4's stand-in: a 32-bit integer hash of the flat index (no spatial structure at all)."""
    m = abi.KZ_BLUENOISE_TEXTURES * abi.KZ_BLUENOISE_RES * abi.KZ_BLUENOISE_RES
    with np.errstate(over="ignore"):
        h = np.arange(m, dtype=np.uint32) + np.uint32(seed)
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    return (h >> np.uint32(16)).astype(np.uint16).reshape(abi.KZ_BLUENOISE_TEXTURES, abi.KZ_BLUENOISE_RES, abi.KZ_BLUENOISE_RES)


def void_and_cluster(n, seed, sigma=1.9):
    """One n x n blue-noise dither array by Ulichney's void-and-cluster method (Proc. SPIE 1913, 1993): returns the RANK of every cell (a permutation of
    0 .. n*n - 1) - thresholding the ranks at any level leaves a point set without clusters or voids on the torus. The "energy" of a cell is the minority
    pattern filtered by a toroidal Gaussian; the tightest cluster is the minority cell of largest energy, the largest void the majority cell of smallest."""
    rng = np.random.default_rng(seed)
    N = n * n
    ax = np.minimum(np.arange(n), n - np.arange(n)).astype(np.float64)
    k1 = np.exp(-ax * ax / (2.0 * sigma * sigma))
    K2 = np.tile(np.outer(k1, k1), (2, 2))                         # the kernel centred on (y, x) is the window K2[n - y : 2n - y, n - x : 2n - x]

    def kernel_at(y, x):
        return K2[n - y:2 * n - y, n - x:2 * n - x]

    ones0 = max(1, N // 10)
    pat = np.zeros(N, bool)
    pat[rng.permutation(N)[:ones0]] = True
    pat = pat.reshape(n, n)
    E = np.zeros((n, n))
    for y, x in zip(*np.nonzero(pat)):
        E += kernel_at(y, x)
    BIG = 1e30
    while True:                                                    # relax the random start: move the tightest cluster's point into the largest void
        cy, cx = np.unravel_index(np.argmax(np.where(pat, E, -BIG)), E.shape)
        pat[cy, cx] = False
        E -= kernel_at(cy, cx)
        vy, vx = np.unravel_index(np.argmin(np.where(pat, BIG, E)), E.shape)
        pat[vy, vx] = True
        E += kernel_at(vy, vx)
        if (vy, vx) == (cy, cx):
            break
    rank = np.zeros((n, n), np.int64)
    p, e = pat.copy(), E.copy()
    for r in range(ones0 - 1, -1, -1):                             # phase 1: take the initial points away, tightest cluster first
        y, x = np.unravel_index(np.argmax(np.where(p, e, -BIG)), e.shape)
        p[y, x] = False
        e -= kernel_at(y, x)
        rank[y, x] = r
    p, e = pat, E
    for r in range(ones0, N // 2):                                 # phase 2: fill the largest voids up to half coverage
        y, x = np.unravel_index(np.argmin(np.where(p, BIG, e)), e.shape)
        p[y, x] = True
        e += kernel_at(y, x)
        rank[y, x] = r
    q = ~p                                                         # phase 3: the zeros are the minority now: fill the tightest cluster of zeros
    e = np.zeros((n, n))
    for y, x in zip(*np.nonzero(q)):
        e += kernel_at(y, x)
    for r in range(N // 2, N):
        y, x = np.unravel_index(np.argmax(np.where(q, e, -BIG)), e.shape)
        q[y, x] = False
        e -= kernel_at(y, x)
        rank[y, x] = r
    return rank


_BLUE_NOISE = None


def blue_noise_textures():
    """BlueNoiseTextures[48][128][128] (bluenoise.h:8-11; the reference's blob is missing from the checkout): 48 independent void-and-cluster arrays, ranks
    spread over the uint16 range. Minted once by scripts/make_bluenoise.py (~1 s per texture) into nano-kazen_amd/data/; regenerated here if the file is gone."""
    global _BLUE_NOISE
    if _BLUE_NOISE is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bluenoise_vc_48x128.npz")
        if os.path.exists(path):
            bn = np.load(path)["bn"]
        else:
            bn = mint_blue_noise_textures()
            os.makedirs(os.path.dirname(path), exist_ok=True)
            np.savez_compressed(path, bn=bn)
        if bn.shape != (abi.KZ_BLUENOISE_TEXTURES, abi.KZ_BLUENOISE_RES, abi.KZ_BLUENOISE_RES) or bn.dtype != np.uint16:
            raise ValueError("%s: expected uint16 [%d][%d][%d]" % (path, abi.KZ_BLUENOISE_TEXTURES, abi.KZ_BLUENOISE_RES, abi.KZ_BLUENOISE_RES))
        _BLUE_NOISE = np.ascontiguousarray(bn)
    return _BLUE_NOISE


def _mint_one(t):
    n = abi.KZ_BLUENOISE_RES
    r = void_and_cluster(n, 0x6b7a0000 + t)
    return ((r * 65536 + 32768) // (n * n)).astype(np.uint16)      # rank -> the middle of its 1/(n*n) slice of [0, 65535]


def mint_blue_noise_textures(workers=None):
    import multiprocessing as mp
    import os
    workers = workers or min(8, os.cpu_count() or 1)
    with mp.get_context("spawn").Pool(workers) as pool:
        return np.stack(pool.map(_mint_one, range(abi.KZ_BLUENOISE_TEXTURES)))


# ----------------------------------------------------------------------------- geometry helpers
def quad(p0, p1, p2, p3, flip=False):
    """Two triangles (p0,p1,p2),(p0,p2,p3) with per-vertex face normals and unit UVs."""
    P = np.array([p0, p1, p2, p3], np.float32)
    n = np.cross(P[1] - P[0], P[2] - P[0])
    n = n / np.linalg.norm(n)
    if flip:
        n = -n
    N = np.tile(n.astype(np.float32), (4, 1))
    UV = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    F = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    return P, N, UV, F


def merge(parts):
    Vs, Ns, UVs, Fs, off = [], [], [], [], 0
    for P, N, UV, F in parts:
        Vs.append(P)
        Ns.append(N)
        UVs.append(UV)
        Fs.append(F + np.uint32(off))
        off += P.shape[0]
    return np.concatenate(Vs), np.concatenate(Ns), np.concatenate(UVs), np.concatenate(Fs)


def box(lo, hi, inward=False):
    """6 quads; normals point outward (inward=True: into the box, for rooms)."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),   # -x
             ((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)),   # +x
             ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),   # -y
             ((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)),   # +y
             ((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)),   # -z
             ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))]   # +z
    parts = []
    for f in faces:
        f = f[::-1] if inward else f
        parts.append(quad(*f))
    return merge(parts)


def uv_sphere(center=(0, 0, 0), radius=1.0, nu=70, nv=71):
    """Lat-long sphere with smooth normals and UVs: nu*(2*(nv-2)+2) triangles (70 x 71 -> 9800, the
    triangle count of the reference's scene/2022_q1/parameters/meshes/Sphere.obj)."""
    c = np.asarray(center, np.float64)
    V, N, UV = [], [], []
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu + 1):
            ph = 2 * math.pi * i / nu
            n = np.array([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
            V.append(c + radius * n)
            N.append(n)
            UV.append((i / nu, j / nv))
    F = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + 1
            c2 = a + (nu + 1)
            d = c2 + 1
            if j != 0:
                F.append((a, b, c2))
            if j != nv - 1:
                F.append((b, d, c2))
    return (np.array(V, np.float32), np.array(N, np.float32), np.array(UV, np.float32), np.array(F, np.uint32))


def torus(center, R, r, nu=64, nv=32, axis=1):
    c = np.asarray(center, np.float64)
    V, N, UV = [], [], []
    for j in range(nv + 1):
        ph = 2 * math.pi * j / nv
        for i in range(nu + 1):
            th = 2 * math.pi * i / nu
            n = np.array([math.cos(th) * math.cos(ph), math.sin(ph), math.sin(th) * math.cos(ph)])
            p = np.array([(R + r * math.cos(ph)) * math.cos(th), r * math.sin(ph), (R + r * math.cos(ph)) * math.sin(th)])
            if axis == 2:
                n = n[[0, 2, 1]]
                p = p[[0, 2, 1]]
            V.append(c + p)
            N.append(n)
            UV.append((i / nu, j / nv))
    F = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + 1
            c2 = a + (nu + 1)
            d = c2 + 1
            F.append((a, c2, b))
            F.append((b, c2, d))
    return (np.array(V, np.float32), np.array(N, np.float32), np.array(UV, np.float32), np.array(F, np.uint32))


def grid_sheet(fn, nu, nv):
    """Parametric sheet p(u,v) with finite-difference smooth normals; 2*nu*nv triangles."""
    us = np.linspace(0, 1, nu + 1)
    vs = np.linspace(0, 1, nv + 1)
    P = np.array([[fn(u, v) for u in us] for v in vs], np.float64)       # (nv+1, nu+1, 3)
    du = np.gradient(P, axis=1)
    dv = np.gradient(P, axis=0)
    n = np.cross(du, dv)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    UV = np.array([[(u, v) for u in us] for v in vs], np.float32)
    F = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            F.append((a, a + 1, a + nu + 1))
            F.append((a + 1, a + nu + 2, a + nu + 1))
    return (P.reshape(-1, 3).astype(np.float32), n.reshape(-1, 3).astype(np.float32), UV.reshape(-1, 2),
            np.array(F, np.uint32))


# ----------------------------------------------------------------------------- BASELINE configs
def cornell_box(width=256, height=256, spp=16, sampler="independent", seed=0, maxDepth=5):
    """C1 (BASELINE.json configs[0]): Cornell-style box, 34 triangles, kiss + diffuse, one ceiling light."""
    s = SceneDescription()
    white, red, green = diffuse((0.73, 0.73, 0.73)), diffuse((0.65, 0.05, 0.05)), diffuse((0.12, 0.45, 0.15))
    s.add_mesh(*_vfnuv(quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), flip=True)), bsdf=white)    # floor (n=+y)
    s.add_mesh(*_vfnuv(quad((-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1), flip=True)), bsdf=white)        # ceiling (n=-y)
    s.add_mesh(*_vfnuv(quad((-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1), flip=True)), bsdf=white)    # back (n=+z)
    s.add_mesh(*_vfnuv(quad((-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), flip=True)), bsdf=red)      # left (n=+x)
    s.add_mesh(*_vfnuv(quad((1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1), flip=True)), bsdf=green)        # right (n=-x)
    s.add_mesh(*_vfnuv(_rot_y(box((-0.3, -1.0, -0.3), (0.3, 0.2, 0.3)), 0.3, (-0.35, 0, -0.3))),
               bsdf=kazenstandard(baseColor=(0.75, 0.75, 0.75), roughness=0.5, metallic=0.0))
    s.add_mesh(*_vfnuv(_rot_y(box((-0.3, -1.0, -0.3), (0.3, -0.4, 0.3)), -0.3, (0.35, 0, 0.3))),
               bsdf=kazenstandard(baseColor=(0.9, 0.6, 0.2), roughness=0.3, metallic=1.0))
    s.add_mesh(*_vfnuv(quad((-0.25, 0.99, -0.25), (-0.25, 0.99, 0.25), (0.25, 0.99, 0.25), (0.25, 0.99, -0.25), flip=True)),
               bsdf=diffuse((0, 0, 0)), light=area((1, 1, 1), 15.0, False))
    s.camera.update(width=width, height=height, fov=39.0, nearClip=0.1, farClip=100.0,
                    toWorld=look_at((0, 0, 3.6), (0, 0, 0), (0, 1, 0)))
    s.sampler = {"type": sampler, "sampleCount": spp, "seed": seed}
    s.integrator["maxDepth"] = maxDepth
    return s


def glass_scene(width=128, height=128, spp=16, sampler="independent", seed=0, maxDepth=8):
    """Cornell-style box with a mirror block and a glass (dielectric) sphere: the EDiscrete / eta branches of Li."""
    s = cornell_box(width, height, spp, sampler, seed, maxDepth)
    s.meshes[5]["bsdf"] = mirror()
    del s.meshes[6]
    s.add_mesh(*_vfnuv(uv_sphere((0.4, -0.6, 0.35), 0.4, 24, 25)), bsdf=dielectric())
    s.meshes.append(s.meshes.pop(6))          # keep the light last is not required; order only fixes ids
    return s


def materials_scene(width=160, height=96, spp=16, sampler="independent", seed=0, maxDepth=6):
    """Studio backdrop + one sphere per BSDF plugin (ggx, roughconductor x3, roughplastic, roughdielectric, dielectric, mirror,
    kazenstandard, diffuse) + two area lights: every BSDF::eval/pdf/sample on the path in one frame."""
    s = SceneDescription()
    s.add_mesh(*_vfnuv(quad((-8, 0, -4), (8, 0, -4), (8, 0, 6), (-8, 0, 6), flip=True)), bsdf=diffuse((0.7, 0.7, 0.7)))
    s.add_mesh(*_vfnuv(quad((-8, 0, -4), (-8, 6, -4), (8, 6, -4), (8, 0, -4), flip=True)), bsdf=diffuse((0.6, 0.65, 0.7)))
    rows = [ggx((0.9, 0.6, 0.3), 0.3), roughconductor(0.3, "Au"), roughconductor(0.5, "Cu"), roughconductor(0.2, "Cr"), roughplastic(0.3, kd=(0.2, 0.4, 0.7)),
            roughdielectric(0.4), dielectric(), mirror(), kazenstandard((0.7, 0.2, 0.2), 0.4, clearcoat=1.0), diffuse((0.3, 0.7, 0.3))]
    for i, r in enumerate(rows):
        x = -4.5 + (i % 5) * 2.25
        z = 0.0 if i < 5 else 2.4
        s.add_mesh(*_vfnuv(uv_sphere((x, 0.8, z), 0.8, 20, 21)), bsdf=r)
    for (cx, cz, inten) in ((-3.0, 1.5, 14.0), (3.0, 2.5, 10.0)):
        q = quad((cx - 1, 5.0, cz - 1), (cx - 1, 5.0, cz + 1), (cx + 1, 5.0, cz + 1), (cx + 1, 5.0, cz - 1), flip=True)
        s.add_mesh(*_vfnuv(q), bsdf=diffuse((0, 0, 0)), light=area((1, 1, 1), inten, False))
    s.camera.update(width=width, height=height, fov=42.0, nearClip=0.1, farClip=100.0, toWorld=look_at((0, 3.2, 9.5), (0, 0.9, 1.0), (0, 1, 0)))
    s.sampler = {"type": sampler, "sampleCount": spp, "seed": seed}
    s.integrator["maxDepth"] = maxDepth
    return s


def _test_images(seed=7):
    """Small procedural rasters: an 8x8 sRGB checker, a 32x32 colour noise (u8), a 16x16 one-channel float map, and a 32x32
    tangent-space normal map of a bump field (linear, u8)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:8, 0:8]
    chk = np.where(((xx + yy) & 1)[:, :, None] == 0, np.array([230, 225, 210], np.uint8), np.array([40, 60, 120], np.uint8)).astype(np.uint8)
    noise = rng.integers(30, 255, (32, 32, 3), dtype=np.uint8)
    gray = rng.random((16, 16, 1), dtype=np.float32)
    y, x = np.mgrid[0:32, 0:32] / 32.0
    hgt = 0.06 * (np.sin(2 * np.pi * 3 * x) * np.cos(2 * np.pi * 2 * y))
    dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
    dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
    n = np.stack([-dx * 16, -dy * 16, np.ones_like(hgt)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    nrm = np.clip(np.rint((n * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8)
    return chk, noise, gray, nrm


def textured_scene(width=160, height=96, spp=16, sampler="independent", seed=0, maxDepth=6, regularization=True):
    """SURVEY 8f rank 4 in one frame: image / colorramp / blend textures behind diffuse ("lambertian"), ggx and kazenstandard
    parameters, and normalmap rows over kazenstandard, diffuse, roughconductor and dielectric."""
    chk, noise, gray, nrm = _test_images()
    t_chk = imagetexture(chk, scale=6.0, colorspace="srgb")
    t_noise = imagetexture(noise, scale=2.0, colorspace="srgb")
    t_gray = imagetexture(gray, scale=3.0, colorspace="linear")
    t_nrm = imagetexture(nrm, scale=2.0, colorspace="linear")
    s = SceneDescription()
    s.add_mesh(*_vfnuv(quad((-8, 0, -4), (8, 0, -4), (8, 0, 6), (-8, 0, 6), flip=True)), bsdf=lambertian(t_chk))
    wall = blend(mask=t_gray, input1=constanttexture((0.8, 0.25, 0.2)), input2=t_noise, blendmode="mix")
    s.add_mesh(*_vfnuv(quad((-8, 0, -4), (-8, 6, -4), (8, 6, -4), (8, 0, -4), flip=True)), bsdf=lambertian(wall))
    rows = [kazenstandard(t_noise, colorramp(t_gray, 0.1, 0.8), t_gray, clearcoat=0.5),
            normalmap(t_nrm, kazenstandard((0.8, 0.5, 0.2), colorramp(t_gray, 0.2, 0.6), 0.0, specular=0.8)),
            normalmap(t_nrm, diffuse((0.3, 0.7, 0.4))),
            ggx(blend(None, t_chk, t_noise, "multiply"), 0.35),
            normalmap(t_nrm, roughconductor(0.25, "Cu")),
            normalmap(t_nrm, dielectric()),
            normalmap(t_nrm, lambertian(t_chk)),
            kazenstandard((0.6, 0.6, 0.65), t_gray, 1.0)]
    for i, r in enumerate(rows):
        x = -3.0 + (i % 4) * 2.0
        z = 0.0 if i < 4 else 2.4
        s.add_mesh(*_vfnuv(uv_sphere((x, 0.8, z), 0.8, 20, 21)), bsdf=r)
    for (cx, cz, inten) in ((-3.0, 1.5, 14.0), (3.0, 2.5, 10.0)):
        q = quad((cx - 1, 5.0, cz - 1), (cx - 1, 5.0, cz + 1), (cx + 1, 5.0, cz + 1), (cx + 1, 5.0, cz - 1), flip=True)
        s.add_mesh(*_vfnuv(q), bsdf=diffuse((0, 0, 0)), light=area((1, 1, 1), inten, False))
    s.camera.update(width=width, height=height, fov=42.0, nearClip=0.1, farClip=100.0, toWorld=look_at((0, 3.2, 9.5), (0, 0.9, 1.0), (0, 1, 0)))
    s.sampler = {"type": sampler, "sampleCount": spp, "seed": seed}
    s.integrator["maxDepth"] = maxDepth
    s.integrator["regularization"] = regularization
    return s


def _vfnuv(m):
    P, N, UV, F = m
    return P, F, N, UV


def _rot_y(m, ang, trans):
    P, N, UV, F = m
    c, s_ = math.cos(ang), math.sin(ang)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float64)
    P2 = (P.astype(np.float64) @ R.T + np.asarray(trans, np.float64)).astype(np.float32)
    N2 = (N.astype(np.float64) @ R.T).astype(np.float32)
    return P2, N2, UV, F


def sphere_env(width=512, height=512, spp=64, seed=0):
    """C2 (configs[1]): one diffuse sphere (9800 tris, albedo .5), constant white background, no lights."""
    s = SceneDescription()
    s.add_mesh(*_vfnuv(uv_sphere((0, 0, 0), 1.0, 70, 71)), bsdf=diffuse((0.5, 0.5, 0.5)))
    s.background = {"color": (1.0, 1.0, 1.0), "intensity": 1.0}
    s.camera.update(width=width, height=height, fov=35.0, nearClip=0.1, farClip=100.0,
                    toWorld=look_at((0, 0.6, 4.0), (0, 0, 0), (0, 1, 0)))
    s.sampler = {"type": "independent", "sampleCount": spp, "seed": seed}
    return s


def hero_scene(width=1920, height=1080, spp=256, seed=0, detail=1.0):
    """C3 (configs[2], synthesized: scene/2022_q2 holds only TODO.md): studio backdrop + kiss objects that
    exercise every lobe (diffuse + spec + clearcoat + sheen, metallic 0/1) + 3 area lights."""
    s = SceneDescription()
    nb = max(8, int(96 * detail))

    def backdrop(u, v):   # floor that sweeps up into a back wall
        x = (u - 0.5) * 16.0
        t = v * 12.0
        if t < 6.0:
            return (x, 0.0, 4.0 - t)
        a = min((t - 6.0) / 2.0, math.pi / 2)
        if t < 6.0 + math.pi:
            return (x, 2.0 - 2.0 * math.cos(a), -2.0 - 2.0 * math.sin(a))
        return (x, 2.0 + (t - 6.0 - math.pi), -4.0)
    s.add_mesh(*_vfnuv(grid_sheet(backdrop, nb, nb)), bsdf=diffuse((0.8, 0.8, 0.8)))
    ns = max(12, int(96 * detail))
    mats = [kazenstandard((0.75, 0.2, 0.2), roughness=0.4, clearcoat=1.0, clearcoatRoughness=0.5, specular=0.5),
            kazenstandard((0.95, 0.8, 0.4), roughness=0.25, metallic=1.0),
            kazenstandard((0.2, 0.3, 0.8), roughness=0.7, sheen=1.0, sheenTint=0.5),
            kazenstandard((0.75, 0.75, 0.75), roughness=0.5, specular=1.0, specularTint=1.0),
            kazenstandard((0.1, 0.6, 0.3), roughness=0.15, clearcoat=0.5, clearcoatRoughness=0.1, anisotropy=0.5)]
    xs = [-3.2, -1.6, 0.0, 1.6, 3.2]
    for x, m in zip(xs, mats):
        s.add_mesh(*_vfnuv(uv_sphere((x, 0.7, 0.0), 0.7, ns, ns + 1)), bsdf=m)
    s.add_mesh(*_vfnuv(torus((0.0, 0.25, 1.8), 0.9, 0.25, max(16, int(128 * detail)), max(8, int(64 * detail)))),
               bsdf=kazenstandard((0.8, 0.5, 0.2), roughness=0.35, metallic=1.0, clearcoat=1.0))
    for (c, sz, inten) in [((-4.0, 5.0, 2.0), 1.5, 6.0), ((4.0, 4.0, 3.0), 1.0, 9.0), ((0.0, 6.0, -1.0), 2.0, 4.0)]:
        cx, cy, cz = c
        q = quad((cx - sz, cy, cz - sz), (cx - sz, cy, cz + sz), (cx + sz, cy, cz + sz), (cx + sz, cy, cz - sz), flip=True)
        s.add_mesh(*_vfnuv(q), bsdf=diffuse((0, 0, 0)), light=area((1, 1, 1), inten, False))
    s.camera.update(width=width, height=height, fov=35.489, nearClip=0.1, farClip=100.0,
                    toWorld=look_at((0, 2.2, 8.5), (0, 0.7, 0), (0, 1, 0)))
    s.sampler = {"type": "independent", "sampleCount": spp, "seed": seed}
    return s


def random_triangles(n_tris=1000000, width=1920, height=1080, spp=1024, sampler="pmj02bn", seed=1, s_edge=0.02):
    """C4/C5 (configs[3], [4]): n random triangles + 8 mesh lights in a closed diffuse room.

    Draw order (pcg32.seed(1), nextFloat): for triangle i, 9 floats f0..f8: centre = 2*f0..2-1,
    e1 = s*(2*f3..5-1), e2 = s*(2*f6..8-1); vertices (c, c+e1, c+e2); per-vertex normal = face normal
    (H4). Triangle i goes to mesh i%8 (kiss row i%8). Deviation from SURVEY 8d, stated in DESIGN.md:
    the room is [-1.2,1.2]^2 x [-1.2,3.6] instead of [-1.2,1.2]^3 so that the camera at (0,0,3.4) is INSIDE
    the closed box (with the cube the primary rays would all hit the outside of the front wall).
    """
    s = SceneDescription()
    f = pcg32_floats(1, 9 * n_tris).reshape(n_tris, 9).astype(np.float32)
    c = f[:, 0:3] * np.float32(2) - np.float32(1)
    e1 = (f[:, 3:6] * np.float32(2) - np.float32(1)) * np.float32(s_edge)
    e2 = (f[:, 6:9] * np.float32(2) - np.float32(1)) * np.float32(s_edge)
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    nrm = (nrm / ln).astype(np.float32)
    rows = [kazenstandard((0.8, 0.8, 0.8), roughness=0.5),
            kazenstandard((0.8, 0.3, 0.3), roughness=0.3, clearcoat=1.0),
            kazenstandard((0.3, 0.8, 0.3), roughness=0.7, sheen=1.0),
            kazenstandard((0.3, 0.3, 0.8), roughness=0.2, metallic=1.0),
            kazenstandard((0.9, 0.7, 0.3), roughness=0.4, metallic=1.0, clearcoat=0.5),
            kazenstandard((0.6, 0.6, 0.6), roughness=0.9, specular=1.0),
            kazenstandard((0.8, 0.5, 0.8), roughness=0.5, specularTint=1.0, sheen=0.5, sheenTint=1.0),
            kazenstandard((0.5, 0.8, 0.8), roughness=0.1, clearcoat=1.0, clearcoatRoughness=0.1)]
    for k in range(8):
        idx = np.arange(k, n_tris, 8)
        if idx.size == 0:
            continue
        m = idx.size
        V = np.empty((m, 3, 3), np.float32)
        V[:, 0], V[:, 1], V[:, 2] = c[idx], c[idx] + e1[idx], c[idx] + e2[idx]
        N = np.repeat(nrm[idx][:, None, :], 3, axis=1)
        F = np.arange(3 * m, dtype=np.uint32).reshape(m, 3)
        s.add_mesh(V.reshape(-1, 3), F, N.reshape(-1, 3), None, bsdf=rows[k])
    P, N, UV, F = box((-1.2, -1.2, -1.2), (1.2, 1.2, 3.6), inward=True)
    s.add_mesh(P, F, N, UV, bsdf=diffuse((0.7, 0.7, 0.7)))
    h = 0.2
    for cx in (-0.75, -0.25, 0.25, 0.75):
        for cz in (0.0, 2.3):
            q = quad((cx - h, 1.19, cz - h), (cx - h, 1.19, cz + h), (cx + h, 1.19, cz + h), (cx + h, 1.19, cz - h), flip=True)
            s.add_mesh(*_vfnuv(q), bsdf=diffuse((0, 0, 0)), light=area((1, 1, 1), 20.0, False))
    s.camera.update(width=width, height=height, fov=40.0, nearClip=0.01, farClip=100.0,
                    toWorld=look_at((0, 0, 3.4), (0, 0, 0), (0, 1, 0)))
    s.sampler = {"type": sampler, "sampleCount": spp, "seed": seed}
    return s


def load_npz(path, overrides=None):
    """A SceneDescription from a flattened scene file (tests/golden/make_q1_scene.py: the arrays a scene's loader produced + its parameters as JSON).
    `overrides`: dict of dicts merged into camera / sampler / integrator, as in xmlscene.load_xml."""
    import json
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    s = SceneDescription()
    for i, m in enumerate(meta["meshes"]):
        a = {k: (z["m%d_%s" % (i, k)] if k in m["has"] else None) for k in ("V", "F", "N", "UV")}
        s.add_mesh(a["V"], a["F"], a["N"], a["UV"], bsdf=m["bsdf"], light=m["light"])
    s.camera.update(meta["camera"])
    s.camera["toWorld"] = np.asarray(z["camera_toWorld"], np.float32)
    s.sampler.update(meta["sampler"]); s.integrator.update(meta["integrator"])
    s.background = meta["background"]
    for k, v in (overrides or {}).items():
        getattr(s, k).update(v)
    return s
