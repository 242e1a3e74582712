"""Scene I/O adapter: nano-kazen XML + Wavefront OBJ -> SceneDescription.

Follows the reference's loader so that the unchanged scene files drive the MI355X core:
  * tags and property types            src/kazen/parser.cpp:73-97, :203-292 (children first, then the object)
  * transform composition               parser.cpp:243-290 (each op LEFT-multiplies the running transform)
  * value parsing                       string::toVector3f / tokenize on ", " (common.cpp), toBool "true"/"false"
  * OBJ loading                         src/kazen/mesh.cpp:200-343: v transformed by toWorld, vn by the inverse transpose and
                                        normalised, (p,uv,n) triples deduplicated in encounter order, quads split (0,1,2),(3,0,2)
Plugins outside the supported set raise ValueError (never a silent fallback). Image files of <texture type="imagetexture"> are
decoded with PIL (8-bit PNG/JPEG -> u8 raster, 16-bit -> float), which is what OpenImageIO's reader hands the texture system.
Host-side I/O only: nothing here is on the per-sample path.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import scenes as S

OBJECT_TAGS = {"scene", "mesh", "bsdf", "light", "camera", "medium", "phase", "integrator", "sampler", "texture", "rfilter"}
TRANSFORM_OPS = {"translate", "matrix", "rotate", "scale", "lookat"}


def _floats(s):
    return [float(t) for t in s.replace(",", " ").split()]


def _vec3(s):
    v = _floats(s)
    if len(v) != 3:
        raise ValueError("Expected 3 values")                                  # string::toVector3f, common.cpp:271-279
    return v


def _bool(s):
    v = s.strip().lower()
    if v in ("true", "t"):
        return True
    if v in ("false", "f"):
        return False
    raise ValueError("Could not parse boolean value '%s'" % s)


def _transform(node):
    """parser.cpp:243-290, in float32 like Eigen::Affine3f."""
    m = np.eye(4, dtype=np.float32)
    for op in node:
        if op.tag is ET.Comment:
            continue
        if op.tag not in TRANSFORM_OPS:
            raise ValueError("transform nodes can only contain transform operations")
        t = np.eye(4, dtype=np.float32)
        if op.tag == "translate":
            t[:3, 3] = _vec3(op.get("value"))
        elif op.tag == "matrix":
            v = _floats(op.get("value"))
            if len(v) != 16:
                raise ValueError("Expected 16 values")
            t = np.array(v, np.float32).reshape(4, 4)
        elif op.tag == "scale":
            t[0, 0], t[1, 1], t[2, 2] = _vec3(op.get("value"))
        elif op.tag == "rotate":
            ang = np.float32(float(op.get("angle")) * (math.pi / 180.0))
            ax = np.array(_vec3(op.get("axis")), np.float64)
            ax /= np.linalg.norm(ax)
            c, s_ = math.cos(ang), math.sin(ang)
            x, y, z = ax
            t[:3, :3] = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                                  [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                                  [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]], np.float32)
        elif op.tag == "lookat":
            o = np.array(_vec3(op.get("origin")), np.float32)
            tg = np.array(_vec3(op.get("target")), np.float32)
            up = np.array(_vec3(op.get("up")), np.float32)
            d = tg - o
            d /= np.linalg.norm(d)
            left = np.cross(up / np.linalg.norm(up), d)
            left /= np.linalg.norm(left)
            nu = np.cross(d, left)
            nu /= np.linalg.norm(nu)
            t[:3, 0], t[:3, 1], t[:3, 2], t[:3, 3] = left, nu, d, o
        m = (t @ m).astype(np.float32)
    return m


def _props(node):
    """Property children of an object node -> dict (typed as the reference's PropertyList)."""
    p = {}
    for ch in node:
        if ch.tag is ET.Comment or ch.tag in OBJECT_TAGS:
            continue
        name = ch.get("name")
        if ch.tag == "string":
            p[name] = ch.get("value")
        elif ch.tag == "float":
            p[name] = float(ch.get("value"))
        elif ch.tag == "integer":
            p[name] = int(ch.get("value"))
        elif ch.tag == "boolean":
            p[name] = _bool(ch.get("value"))
        elif ch.tag in ("point", "vector", "color"):
            p[name] = tuple(_vec3(ch.get("value")))
        elif ch.tag == "transform":
            p[name] = _transform(ch)
        else:
            raise ValueError("unexpected tag \"%s\"" % ch.tag)
    return p


def _children(node, tag):
    return [c for c in node if c.tag == tag]


def _const_texture(node):
    if node.get("type") != "constanttexture":
        raise ValueError("texture type \"%s\" is not supported here (only constanttexture)" % node.get("type"))
    return tuple(_props(node).get("color", (0.5, 0.5, 0.5)))              # texture.cpp:13


def _load_image(path):
    from PIL import Image
    if not os.path.exists(path):
        raise ValueError("imagetexture: file \"%s\" does not exist" % path)
    try:
        im = Image.open(path)
        im.load()
    except Exception as e:                                                 # e.g. OpenEXR: no decoder in this environment
        raise ValueError("imagetexture: cannot decode \"%s\" (%s)" % (path, e))
    if im.mode in ("I;16", "I;16B", "I"):
        return (np.asarray(im, np.float32) / np.float32(65535.0))[:, :, None]
    if im.mode == "F":
        return np.asarray(im, np.float32)[:, :, None]
    if im.mode not in ("L", "RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.getbands() else "RGB")
    a = np.asarray(im, np.uint8)
    return a[:, :, None] if a.ndim == 2 else a


def _texture(node, base, folded=False):
    """<texture> -> texture dict (texture.cpp). folded=True returns a constanttexture as its colour tuple (the folded form)."""
    t = node.get("type")
    p = _props(node)
    kids = _children(node, "texture")
    if t == "constanttexture":
        c = tuple(p.get("color", (0.5, 0.5, 0.5)))
        return c if folded else S.constanttexture(c)
    if t == "imagetexture":
        if "filename" not in p:
            raise ValueError("imagetexture needs a \"filename\"")
        fn = p["filename"] if os.path.isabs(p["filename"]) else os.path.join(base, p["filename"])     # the resolver appends the scene's directory (main.cpp)
        return S.imagetexture(_load_image(fn), p.get("scale", 1.0), p.get("colorspace", "srgb"), p.get("filter", "bilinear"))      # ("filter": this library's own property, KzTexture.filter)
    if t == "colorramp":
        if len(kids) > 1:
            raise ValueError("colorramp takes one nested texture")
        return S.colorramp(_texture(kids[0], base) if kids else None, p.get("min", 0.0), p.get("max", 1.0))
    if t == "blend":
        slot = {}
        for k in kids:                                                      # texture.cpp:241-262: children are matched by id
            cid = k.get("id", "")
            if cid not in ("mask", "input1", "input2"):
                raise ValueError("The name of this texture does not match any field!")
            if cid in slot:
                raise ValueError("There is already an %s defined!" % cid)
            slot[cid] = _texture(k, base)
        return S.blend(slot.get("mask"), slot.get("input1"), slot.get("input2"), p.get("blendmode", "mix"))
    raise ValueError("texture type \"%s\" is not supported by the MI355X core" % t)


def _bsdf(node, base="."):
    t = node.get("type")
    p = _props(node)
    tkids = _children(node, "texture")
    tex = {c.get("id", ""): _texture(c, base, folded=True) for c in tkids}
    if t == "diffuse":
        return S.diffuse(p.get("albedo", (0.5, 0.5, 0.5)))
    if t == "lambertian":                                                   # bsdf.cpp:259-262: any texture child is the albedo
        if not tex:
            raise ValueError("lambertian needs an albedo texture child")
        return S.lambertian(_texture(tkids[-1], base))
    if t == "normalmap":                                                    # bsdf.cpp:391-404
        nested = _children(node, "bsdf")
        if not tkids or len(nested) != 1:
            raise ValueError("normalmap needs a normal texture child and one nested bsdf")
        if nested[0].get("type") == "normalmap":
            raise ValueError("a normalmap nested in a normalmap is not supported by the MI355X core")
        return S.normalmap(_texture(tkids[-1], base), _bsdf(nested[0], base))
    if t == "kazenstandard":
        for k in ("baseColor", "roughness", "metallic"):
            if k not in tex:
                raise ValueError("kazenstandard needs a texture child with id=\"%s\"" % k)
        rough, metal = (v if isinstance(v, dict) else v[0] for v in (tex["roughness"], tex["metallic"]))
        return S.kazenstandard(tex["baseColor"], rough, metal, p.get("anisotropy", 0.0), p.get("specular", 0.5),
                               p.get("specularTint", 0.5), p.get("clearcoat", 0.0), p.get("clearcoatRoughness", 0.5), p.get("sheen", 0.0), p.get("sheenTint", 0.5))
    if t == "mirror":
        return S.mirror()
    if t == "dielectric":
        return S.dielectric(p.get("intIOR", 1.5046), p.get("extIOR", 1.000277))
    if t == "ggx":
        if not tex:
            raise ValueError("ggx needs an albedo texture child")
        return S.ggx(list(tex.values())[-1], p.get("roughness", 0.5), p.get("anisotropy", 0.0))
    if t == "roughconductor":
        mat = p.get("material", "Au")
        if mat not in S.CONDUCTORS:
            raise ValueError("roughconductor: unknown material \"%s\"" % mat)
        return S.roughconductor(p.get("alpha", 0.1), mat)
    if t == "roughplastic":
        return S.roughplastic(p.get("alpha", 0.1), p.get("intIOR", 1.5046), p.get("extIOR", 1.000277), p.get("kd", (0.5, 0.5, 0.5)))
    if t == "roughdielectric":
        return S.roughdielectric(p.get("roughness", 0.1), p.get("intIOR", 1.5046), p.get("extIOR", 1.000277))
    raise ValueError("bsdf type \"%s\" is not supported by the MI355X core" % t)


def load_obj(path, to_world=None):
    """mesh.cpp:200-343. Returns V (n,3), F (m,3), N (n,3)|None, UV (n,2)|None as float32 / uint32."""
    pos, tex, nrm = [], [], []
    vmap, verts, idx = {}, [], []
    M = np.eye(4, dtype=np.float32) if to_world is None else np.asarray(to_world, np.float32)
    f32 = np.float32

    def xf_point(p):
        """Transform * Point3f (transform.h:59-62) in float, operation for operation what the C++ mirror's Transform::point does: the two loaders hand the
        library the same vertices bit for bit (tests/test_host_mirror.py renders the XML fixtures through both)"""
        q = [f32(f32(f32(f32(M[i, 0] * p[0]) + f32(M[i, 1] * p[1])) + f32(M[i, 2] * p[2])) + M[i, 3]) for i in range(4)]
        return np.array([q[0] / q[3], q[1] / q[3], q[2] / q[3]], np.float32)

    a = [float(v) for v in M.reshape(-1)]                    # Transform * Normal3f (transform.h:54-56): inverse transpose of the upper 3x3, the 4x4 inverse formed in
    inv = {}                                                 # double by cofactors - the C++ mirror's Transform::normal, expression for expression
    inv[0] = a[5]*a[10]*a[15]-a[5]*a[11]*a[14]-a[9]*a[6]*a[15]+a[9]*a[7]*a[14]+a[13]*a[6]*a[11]-a[13]*a[7]*a[10]
    inv[4] = -a[4]*a[10]*a[15]+a[4]*a[11]*a[14]+a[8]*a[6]*a[15]-a[8]*a[7]*a[14]-a[12]*a[6]*a[11]+a[12]*a[7]*a[10]
    inv[8] = a[4]*a[9]*a[15]-a[4]*a[11]*a[13]-a[8]*a[5]*a[15]+a[8]*a[7]*a[13]+a[12]*a[5]*a[11]-a[12]*a[7]*a[9]
    inv[12] = -a[4]*a[9]*a[14]+a[4]*a[10]*a[13]+a[8]*a[5]*a[14]-a[8]*a[6]*a[13]-a[12]*a[5]*a[10]+a[12]*a[6]*a[9]
    inv[1] = -a[1]*a[10]*a[15]+a[1]*a[11]*a[14]+a[9]*a[2]*a[15]-a[9]*a[3]*a[14]-a[13]*a[2]*a[11]+a[13]*a[3]*a[10]
    inv[5] = a[0]*a[10]*a[15]-a[0]*a[11]*a[14]-a[8]*a[2]*a[15]+a[8]*a[3]*a[14]+a[12]*a[2]*a[11]-a[12]*a[3]*a[10]
    inv[9] = -a[0]*a[9]*a[15]+a[0]*a[11]*a[13]+a[8]*a[1]*a[15]-a[8]*a[3]*a[13]-a[12]*a[1]*a[11]+a[12]*a[3]*a[9]
    inv[2] = a[1]*a[6]*a[15]-a[1]*a[7]*a[14]-a[5]*a[2]*a[15]+a[5]*a[3]*a[14]+a[13]*a[2]*a[7]-a[13]*a[3]*a[6]
    inv[6] = -a[0]*a[6]*a[15]+a[0]*a[7]*a[14]+a[4]*a[2]*a[15]-a[4]*a[3]*a[14]-a[12]*a[2]*a[7]+a[12]*a[3]*a[6]
    inv[10] = a[0]*a[5]*a[15]-a[0]*a[7]*a[13]-a[4]*a[1]*a[15]+a[4]*a[3]*a[13]+a[12]*a[1]*a[7]-a[12]*a[3]*a[5]
    det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12]

    def xf_normal(n):
        n = [float(v) for v in n]
        if det == 0.0:
            r = n
        else:
            r = [(inv[0] * n[0] + inv[4] * n[1] + inv[8] * n[2]) / det, (inv[1] * n[0] + inv[5] * n[1] + inv[9] * n[2]) / det, (inv[2] * n[0] + inv[6] * n[1] + inv[10] * n[2]) / det]
        r = [f32(v) for v in r]
        l2 = f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) + f32(r[2] * r[2]))
        if l2 > 0:
            ln = np.sqrt(l2, dtype=np.float32)
            r = [r[0] / ln, r[1] / ln, r[2] / ln]
        return np.array(r, np.float32)
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                pos.append(xf_point([f32(tok[1]), f32(tok[2]), f32(tok[3])]))
            elif tok[0] == "vt":
                tex.append((float(tok[1]), float(tok[2])))
            elif tok[0] == "vn":
                nrm.append(xf_normal([f32(tok[1]), f32(tok[2]), f32(tok[3])]))
            elif tok[0] == "f":
                vs = tok[1:5]
                keys = []
                for v in vs:
                    parts = v.split("/")
                    if len(parts) < 1 or len(parts) > 3:
                        raise ValueError("Invalid vertex data: \"%s\"" % v)
                    p_ = int(parts[0])
                    uv_ = int(parts[1]) if len(parts) >= 2 and parts[1] else -1
                    n_ = int(parts[2]) if len(parts) >= 3 and parts[2] else -1
                    keys.append((p_, uv_, n_))
                order = [0, 1, 2] if len(keys) == 3 else [0, 1, 2, 3, 0, 2]
                for o in order:
                    k = keys[o]
                    i = vmap.get(k)
                    if i is None:
                        i = len(verts)
                        vmap[k] = i
                        verts.append(k)
                    idx.append(i)
    if not verts:
        raise ValueError("OBJ file \"%s\" has no faces" % path)
    V = np.array([pos[k[0] - 1] for k in verts], np.float32)
    F = np.array(idx, np.uint32).reshape(-1, 3)
    N = np.array([nrm[k[2] - 1] for k in verts], np.float32) if nrm else None
    UV = np.array([tex[k[1] - 1] for k in verts], np.float32) if tex else None
    return V, F, N, UV


def load_xml(path, overrides=None):
    """Parse a nano-kazen scene file into a SceneDescription. `overrides`: dict of dicts merged into camera / sampler /
    integrator after parsing (e.g. {"camera": {"width": 256, "height": 256}, "sampler": {"sampleCount": 16}})."""
    root = ET.parse(path).getroot()
    if root.tag != "scene":
        raise ValueError("root element \"%s\" must be a kazen scene" % root.tag)
    base = os.path.dirname(os.path.abspath(path))
    s = S.SceneDescription()
    have_camera = have_integrator = False
    for node in root:
        if node.tag is ET.Comment:
            continue
        t = node.get("type")
        if node.tag == "integrator":
            if t != "path_mis":
                raise ValueError("integrator \"%s\" is not on the hot path (only path_mis)" % t)
            p = _props(node)
            s.integrator = {"type": "path_mis", "maxDepth": min(512, p.get("maxDepth", 5)), "traceBias": p.get("traceBias", 0.001),
                            "regularization": p.get("regularization", False), "accumulatedRoughness": p.get("accumulatedRoughness", 0.5)}
            have_integrator = True
        elif node.tag == "sampler":
            p = _props(node)
            defaults = {"independent": (1, 0), "pmj02bn": (16, 1), "stratified": (16, 1), "correlated": (16, 1)}
            if t not in defaults:
                raise ValueError("sampler \"%s\" is not supported" % t)
            s.sampler = {"type": t, "sampleCount": p.get("sampleCount", defaults[t][0]), "seed": p.get("seed", defaults[t][1]), "resolution": p.get("resolution", 4)}
        elif node.tag == "camera":
            if t not in ("perspective", "thinlens"):
                raise ValueError("camera \"%s\" is not supported" % t)
            p = _props(node)
            s.camera.update(type=t, width=p.get("width", 1280), height=p.get("height", 720), fov=p.get("fov", 30.0), nearClip=p.get("nearClip", 1e-4),
                            farClip=p.get("farClip", 1e4), toWorld=p.get("toWorld", np.eye(4, dtype=np.float32)))
            if t == "thinlens":
                s.camera.update(apertureRadius=p.get("apertureRadius", 1.0), focusDistance=p.get("focusDistance", 0.0))
            for rf in _children(node, "rfilter"):
                fp = _props(rf)
                ft = rf.get("type")
                if ft not in ("gaussian", "mitchell", "tent", "box"):
                    raise ValueError("rfilter \"%s\" is not supported" % ft)
                s.camera["rfilter"] = {"type": ft, "radius": fp.get("radius", 2.0), "stddev": fp.get("stddev", 0.5), "B": fp.get("B", 1 / 3.0), "C": fp.get("C", 1 / 3.0)}
            have_camera = True
        elif node.tag == "texture":
            if t != "background":
                raise ValueError("scene-level texture \"%s\" is not supported" % t)
            nested = _children(node, "texture")
            if nested:                                                       # texture.cpp:128-136: the last texture child is the nested one
                tx = _texture(nested[-1], base, folded=True)
                if isinstance(tx, tuple):
                    s.background = {"color": tx, "intensity": _props(node).get("intensity", 1.0)}
                else:                                                        # imagetexture: environment lookup; colorramp / blend: 0 (texture.h:13)
                    s.background = {"texture": tx, "intensity": _props(node).get("intensity", 1.0)}
        elif node.tag == "mesh":
            if t != "obj":
                raise ValueError("mesh type \"%s\" is not supported" % t)
            p = _props(node)
            V, F, N, UV = load_obj(os.path.join(base, p["filename"]), p.get("toWorld"))
            bs = _children(node, "bsdf")
            ls = _children(node, "light")
            if len(bs) > 1 or len(ls) > 1:
                raise ValueError("Mesh: tried to register multiple BSDF / light instances!")
            light = None
            if ls:
                if ls[0].get("type") != "area":
                    raise ValueError("light \"%s\" is not supported" % ls[0].get("type"))
                lp = _props(ls[0])
                light = S.area(lp.get("color", (1.0, 1.0, 1.0)), lp.get("intensity", 1.0), lp.get("lightPrimaryVisibility", False))
            s.add_mesh(V, F, N, UV, bsdf=_bsdf(bs[0], base) if bs else None, light=light)
        elif node.tag in OBJECT_TAGS:
            raise ValueError("Scene::addChild(<%s>) is not supported!" % node.tag)
        else:
            raise ValueError("unexpected tag \"%s\"" % node.tag)
    if not have_integrator:
        raise ValueError("No integrator was specified!")
    if not have_camera:
        raise ValueError("No camera was specified!")
    for k, v in (overrides or {}).items():
        getattr(s, k).update(v)
    return s
