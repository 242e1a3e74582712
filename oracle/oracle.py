"""ctypes wrapper of oracle/liboracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package (nano-kazen_amd/) never imports this module.
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
kz = importlib.import_module("nano-kazen_amd")
abi = kz.abi
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.kzo_last_error.restype = C.c_char_p
        L.kzo_scene_create.argtypes = [C.POINTER(abi.KzSceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.kzo_scene_destroy.argtypes = [C.c_void_p]
        L.kzo_scene_destroy.restype = None
        L.kzo_set_brute.argtypes = [C.c_void_p, C.c_int]
        L.kzo_set_brute.restype = None
        L.kzo_math.argtypes = [C.c_int, C.c_uint32, abi.f32p, abi.f32p, abi.f32p]
        L.kzo_math.restype = None
        L.kzo_set_tie_mode.argtypes = [C.c_void_p, C.c_int]
        L.kzo_set_tie_mode.restype = None
        L.kzo_film_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.kzo_sample_count.argtypes = [C.c_void_p]
        L.kzo_sample_count.restype = C.c_uint
        L.kzo_bvh_info.argtypes = [C.c_void_p, C.POINTER(abi.KzBvhInfo)]
        L.kzo_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(abi.KzTile), C.c_uint32, C.c_int, abi.f32p]
        L.kzo_render_canonical.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(abi.KzTile), C.c_uint32, C.c_int, C.c_int, abi.f32p]
        L.kzo_film_to_rgb.argtypes = [abi.f32p, C.c_int, C.c_int, C.c_int, abi.f32p]
        L.kzo_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.KzStats), C.c_int]
        L.kzo_trace_rays.argtypes = [C.c_void_p, C.c_uint32, abi.f32p, abi.f32p, abi.f32p, abi.f32p, C.POINTER(abi.KzHit)]
        L.kzo_hash_pixel_seed.argtypes = [C.c_int32, C.c_int32, C.c_uint64]
        L.kzo_hash_pixel_seed.restype = C.c_uint64
        L.kzo_hash_pixel_dim_seed.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint64]
        L.kzo_hash_pixel_dim_seed.restype = C.c_uint64
        L.kzo_murmur64a.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.kzo_murmur64a.restype = C.c_uint64
        L.kzo_mixbits.argtypes = [C.c_uint64]
        L.kzo_mixbits.restype = C.c_uint64
        L.kzo_permute.argtypes = [C.c_uint32] * 3
        L.kzo_permute.restype = C.c_uint32
        L.kzo_fresnel_ior.argtypes = [C.c_float] * 3
        L.kzo_fresnel_ior.restype = C.c_float
        L.kzo_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.kzo_fresnel_dielectric.restype = C.c_float
        L.kzo_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.kzo_debug_dpdf.argtypes = [C.c_uint32, abi.f32p, abi.f32p, abi.f32p]
        L.kzo_debug_dpdf.restype = None
        L.kzo_debug_dpdf_sample.argtypes = [C.c_uint32, abi.f32p, C.c_float]
        L.kzo_debug_dpdf_sample.restype = C.c_uint32
        L.kzo_debug_pow4.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.kzo_debug_pow4.restype = None
        L.kzo_tea32.restype = C.c_uint64
        L.kzo_pcg32_stream.argtypes = [C.c_uint64, C.c_int64, C.c_int, abi.u32p, abi.f32p, C.POINTER(C.c_uint64)]
        L.kzo_pcg32_stream.restype = None
        L.kzo_pcg32_seed2.argtypes = [C.c_uint64, C.c_uint64, C.c_int, abi.u32p]
        L.kzo_pcg32_seed2.restype = None
        L.kzo_sampler_stream.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_int, abi.f32p]
        L.kzo_sampler_stream.restype = None
        L.kzo_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, abi.f32p, abi.f32p, abi.f32p]
        L.kzo_camera_ray.restype = None
        L.kzo_camera_ray_lens.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, abi.f32p, abi.f32p, abi.f32p]
        L.kzo_camera_ray_lens.restype = None
        L.kzo_background.argtypes = [C.c_void_p, abi.f32p, abi.f32p]
        L.kzo_background.restype = None
        L.kzo_filter_table.argtypes = [C.c_void_p, abi.f32p, abi.f32p, C.POINTER(C.c_int)]
        L.kzo_filter_table.restype = None
        L.kzo_cosine_hemisphere.argtypes = [C.c_float, C.c_float, abi.f32p]
        L.kzo_cosine_hemisphere.restype = None
        L.kzo_uniform_disk.argtypes = [C.c_float, C.c_float, abi.f32p]
        L.kzo_uniform_disk.restype = None
        L.kzo_frame.argtypes = [abi.f32p, abi.f32p, abi.f32p]
        L.kzo_frame.restype = None
        L.kzo_bsdf.argtypes = [C.POINTER(abi.KzBSDF), C.c_int, abi.f32p, abi.f32p, C.c_float, C.c_float, C.c_float, C.c_float, abi.f32p]
        L.kzo_bsdf.restype = None
        L.kzo_scene_bsdf.argtypes = [C.c_void_p, C.c_int, C.c_int, abi.f32p, abi.f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, abi.f32p]
        L.kzo_scene_bsdf.restype = None
        L.kzo_rgb_to_srgb8.argtypes = [abi.f32p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.kzo_texture.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, abi.f32p]
        L.kzo_texture.restype = None
        L.kzo_ggx_sample_vndf.argtypes = [abi.f32p, C.c_float, C.c_float, C.c_float, C.c_float, abi.f32p]
        L.kzo_ggx_sample_vndf.restype = None
        L.kzo_light_sample.argtypes = [C.c_void_p, C.c_int, abi.f32p, C.c_float, C.c_float, C.c_float, abi.f32p]
        L.kzo_light_sample.restype = None
        L.kzo_render_samples.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), abi.u32p, abi.f32p]
        L.kzo_render_samples.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(abi.f32p)


class OracleScene:
    def __init__(self, desc, brute=False):
        self.L = lib()
        self.desc = desc
        cdesc = desc.to_c()
        h = C.c_void_p()
        rc = self.L.kzo_scene_create(C.byref(cdesc), 1 if brute else 0, C.byref(h))
        if rc != 0:
            raise abi.KzError(rc, self.L.kzo_last_error().decode())
        self.h = h
        w, hh, b = C.c_int(), C.c_int(), C.c_int()
        self.L.kzo_film_dims(self.h, C.byref(w), C.byref(hh), C.byref(b))
        self.width, self.height, self.border = w.value, hh.value, b.value
        self.sample_count = int(self.L.kzo_sample_count(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.kzo_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_brute(self, brute):
        self.L.kzo_set_brute(self.h, 1 if brute else 0)

    def set_tie_mode(self, mode):
        """How the shadow loop decides the reference's built-in tie (the segment left after walking through an invisible light
        ends exactly on the sampled light): 0 literal, +1 every tie occluded, -1 every tie unoccluded (kz_oracle.cpp Li)."""
        self.L.kzo_set_tie_mode(self.h, int(mode))

    def render(self, sample_begin=0, sample_end=0, tiles=None, threads=0, film=None):
        if film is None:
            film = np.zeros((self.height + 2 * self.border, self.width + 2 * self.border, 4), np.float32)
        tp, nt = None, 0
        if tiles is not None:
            tp = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
            nt = len(tiles)
        rc = self.L.kzo_render(self.h, sample_begin, sample_end, tp, nt, threads, _fp(film))
        if rc != 0:
            raise abi.KzError(rc, self.L.kzo_last_error().decode())
        return film

    def render_canonical(self, sample_begin=0, sample_end=0, tiles=None, threads=0, grid=64):
        """The same samples with the film's float additions in the order the HIP build fixes (per pixel and tap in sample order, texels resolved over the 64-px tile grid in
        tile order): the film the library must reproduce BIT FOR BIT."""
        film = np.zeros((self.height + 2 * self.border, self.width + 2 * self.border, 4), np.float32)
        tp, nt = None, 0
        if tiles is not None:
            tp = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
            nt = len(tiles)
        rc = self.L.kzo_render_canonical(self.h, sample_begin, sample_end, tp, nt, threads, int(grid), _fp(film))
        if rc != 0:
            raise abi.KzError(rc, self.L.kzo_last_error().decode())
        return film

    def rgb(self, film):
        out = np.empty((self.height, self.width, 3), np.float32)
        self.L.kzo_film_to_rgb(_fp(np.ascontiguousarray(film, np.float32)), self.width, self.height, self.border, _fp(out))
        return out

    def srgb8(self, film):
        rgb = self.rgb(film)
        out = np.zeros(rgb.shape, np.uint8)
        self.L.kzo_rgb_to_srgb8(_fp(rgb), rgb.shape[1], rgb.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def stats(self, reset=False):
        s = abi.KzStats()
        self.L.kzo_get_stats(self.h, C.byref(s), 1 if reset else 0)
        return s.as_dict()

    def bvh_info(self):
        info = abi.KzBvhInfo()
        self.L.kzo_bvh_info(self.h, C.byref(info))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def trace_rays(self, o, d, tmin, tmax):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        hits = (abi.KzHit * n)()
        self.L.kzo_trace_rays(self.h, n, _fp(o), _fp(d), _fp(tmin), _fp(tmax), hits)
        return kz.render.hits_to_arrays(hits, n)

    def sampler_stream(self, px, py, idx, n1):
        out = np.zeros(4 + n1, np.float32)
        self.L.kzo_sampler_stream(self.h, px, py, idx, n1, _fp(out))
        return out

    def background(self, direction):
        d = np.ascontiguousarray(direction, np.float32)
        out = np.zeros(3, np.float32)
        self.L.kzo_background(self.h, _fp(d), _fp(out))
        return out

    def camera_ray(self, sx, sy, ax=None, ay=None):
        o6 = np.zeros(6, np.float32)
        a, b = C.c_float(), C.c_float()
        if ax is None:
            self.L.kzo_camera_ray(self.h, sx, sy, _fp(o6), C.byref(a), C.byref(b))
        else:
            self.L.kzo_camera_ray_lens(self.h, sx, sy, ax, ay, _fp(o6), C.byref(a), C.byref(b))
        return o6, a.value, b.value

    def filter_table(self):
        tab = np.zeros(33, np.float32)
        r, b = C.c_float(), C.c_int()
        self.L.kzo_filter_table(self.h, _fp(tab), C.byref(r), C.byref(b))
        return tab, r.value, b.value

    def light_sample(self, light_idx, ref, u0, u1, u2):
        out = np.zeros(14, np.float32)
        ref = np.ascontiguousarray(ref, np.float32)
        self.L.kzo_light_sample(self.h, light_idx, _fp(ref), u0, u1, u2, _fp(out))
        return out

    def bsdf(self, row, which, wi, wo=None, acc_rough=0.0, s1=0.0, s2=(0.0, 0.0), uv=(0.0, 0.0)):
        """BSDF row `row` of this scene (texture children and normalmap rows included) on the identity frame at uv.
        'eval' -> rgb, 'pdf' -> float, 'sample' -> 8 floats (weight rgb, wo xyz, ok, pdf(bRec) after sample())."""
        wi = np.ascontiguousarray(wi, np.float32)
        wo_ = np.ascontiguousarray(wo if wo is not None else (0, 0, 1), np.float32)
        out = np.zeros(8, np.float32)
        self.L.kzo_scene_bsdf(self.h, row, {"eval": 0, "pdf": 1, "sample": 2}[which], _fp(wi), _fp(wo_), acc_rough, s1, s2[0], s2[1], uv[0], uv[1], _fp(out))
        return out[:3].copy() if which == "eval" else (float(out[0]) if which == "pdf" else out.copy())

    def texture(self, tex, uv):
        """Texture<Color3f>::eval of texture row `tex` at each uv (n, 2) -> (n, 3)."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 3), np.float32)
        tmp = np.zeros(3, np.float32)
        for i in range(uv.shape[0]):
            self.L.kzo_texture(self.h, tex, float(uv[i, 0]), float(uv[i, 1]), _fp(tmp))
            out[i] = tmp
        return out

    def render_samples(self, pxy, idx):
        pxy = np.ascontiguousarray(pxy, np.int32)
        idx = np.ascontiguousarray(idx, np.uint32)
        n = idx.shape[0]
        out = np.zeros((n, 5), np.float32)
        self.L.kzo_render_samples(self.h, n, pxy.ctypes.data_as(C.POINTER(C.c_int32)), idx.ctypes.data_as(abi.u32p), _fp(out))
        return out


def bsdf(params, which, wi, wo=None, acc_rough=0.0, s1=0.0, s2=(0.0, 0.0)):
    """which: 'eval' -> rgb, 'pdf' -> float, 'sample' -> (weight rgb, wo, ok)"""
    L = lib()
    row = kz.scenes.SceneDescription()
    row.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), bsdf=params)
    d = row.to_c()
    wi = np.ascontiguousarray(wi, np.float32)
    wo_ = np.ascontiguousarray(wo if wo is not None else (0, 0, 1), np.float32)
    out = np.zeros(7, np.float32)
    L.kzo_bsdf(d.bsdfs, {"eval": 0, "pdf": 1, "sample": 2}[which], _fp(wi), _fp(wo_), acc_rough, s1, s2[0], s2[1], _fp(out))
    if which == "eval":
        return out[:3].copy()
    if which == "pdf":
        return float(out[0])
    return out[:3].copy(), out[3:6].copy(), bool(out[6])


MATH_FN = {"sin": 0, "cos": 1, "exp": 2, "log": 3, "atan": 4, "atan2": 5, "acos": 6, "tan": 7, "pow": 8, "hypot": 9, "cube": 10, "cos1": 11}


def math_fn(name, x, y=None, libm=False):
    """The oracle's transcendental functions (kz_oracle_math.h) on float arrays; libm=True: the C library's float function instead (what the
    reference's text calls on this machine)."""
    x = np.ascontiguousarray(x, np.float32)
    y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.zeros_like(x)
    lib().kzo_math(MATH_FN[name] + (100 if libm else 0), x.size, _fp(x), _fp(y), _fp(out))
    return out
