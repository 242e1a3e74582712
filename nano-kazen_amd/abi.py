"""ctypes mirror of include/kazen_mi355x.h and the loader of libkazen_mi355x.so.

Plumbing only: every structure here is a field-for-field copy of the C header (which cites
the reference interface each one replaces). The loader FAILS LOUDLY when the HIP extension
is missing: there is no CPU fallback in the product path.
"""
import ctypes as C
import os

KZ_ABI_VERSION = 6

KZ_OK, KZ_ERR_INVALID_ARG, KZ_ERR_UNSUPPORTED, KZ_ERR_NO_DEVICE, KZ_ERR_HIP, KZ_ERR_STATE, KZ_ERR_OOM = range(7)
KZ_BSDF_DIFFUSE, KZ_BSDF_KAZENSTANDARD, KZ_BSDF_MIRROR, KZ_BSDF_DIELECTRIC = 0, 1, 2, 3
KZ_BSDF_GGX, KZ_BSDF_ROUGHCONDUCTOR, KZ_BSDF_ROUGHPLASTIC, KZ_BSDF_ROUGHDIELECTRIC, KZ_BSDF_NORMALMAP = 4, 5, 6, 7, 8
KZ_TEX_CONSTANT, KZ_TEX_IMAGE, KZ_TEX_COLORRAMP, KZ_TEX_BLEND = 0, 1, 2, 3
KZ_BLEND_MIX, KZ_BLEND_MULTIPLY, KZ_BLEND_NONE = 0, 1, 2
KZ_PIXEL_U8, KZ_PIXEL_F32 = 0, 1
KZ_TEX_MAX_DEPTH = 8
KZ_SAMPLER_INDEPENDENT, KZ_SAMPLER_PMJ02BN, KZ_SAMPLER_STRATIFIED, KZ_SAMPLER_CORRELATED = 0, 1, 2, 3
KZ_CAMERA_PERSPECTIVE, KZ_CAMERA_THINLENS = 0, 1
KZ_INTEGRATOR_PATH_MIS = 0
KZ_FILTER_GAUSSIAN, KZ_FILTER_MITCHELL, KZ_FILTER_TENT, KZ_FILTER_BOX = 0, 1, 2, 3
KZ_FILTER_RESOLUTION = 32
KZ_PMJ02BN_SETS, KZ_PMJ02BN_SAMPLES = 5, 65536
KZ_BLUENOISE_TEXTURES, KZ_BLUENOISE_RES = 48, 128

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)


class KzBSDF(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", C.c_float * 3), ("baseColor", C.c_float * 3),
                ("roughness", C.c_float), ("metallic", C.c_float), ("anisotropy", C.c_float),
                ("specular", C.c_float), ("specularTint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoatRoughness", C.c_float), ("sheen", C.c_float), ("sheenTint", C.c_float),
                ("intIOR", C.c_float), ("extIOR", C.c_float), ("alpha", C.c_float), ("condEta", C.c_float * 3),
                ("condK", C.c_float * 3), ("albedoTex", C.c_int32), ("roughnessTex", C.c_int32), ("metallicTex", C.c_int32),
                ("normalTex", C.c_int32), ("nested", C.c_int32), ("alphaResolved", C.c_int32), ("pad_", C.c_int32)]


class KzImage(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("format", C.c_int32)]


class KzTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("image", C.c_int32), ("scale", C.c_float), ("srgb", C.c_int32),
                ("rampMin", C.c_float), ("rampMax", C.c_float), ("blendMode", C.c_int32), ("child", C.c_int32 * 3), ("filter", C.c_int32), ("pad_", C.c_int32 * 2)]


class KzLight(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("intensity", C.c_float), ("primaryVisibility", C.c_int32)]


class KzMesh(C.Structure):
    _fields_ = [("V", f32p), ("N", f32p), ("UV", f32p), ("F", u32p), ("nV", C.c_uint32), ("nF", C.c_uint32),
                ("bsdf", C.c_int32), ("light", C.c_int32)]


class KzFilter(C.Structure):
    _fields_ = [("type", C.c_int32), ("radius", C.c_float), ("stddev", C.c_float), ("B", C.c_float), ("C", C.c_float)]


class KzCamera(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("toWorld", C.c_float * 16),
                ("fov", C.c_float), ("nearClip", C.c_float), ("farClip", C.c_float),
                ("apertureRadius", C.c_float), ("focusDistance", C.c_float),
                ("sampleToCamera", f32p), ("rfilter", KzFilter)]


class KzSampler(C.Structure):
    _fields_ = [("type", C.c_int32), ("sampleCount", C.c_uint32), ("resolution", C.c_int32), ("pad_", C.c_int32), ("seed", C.c_uint64),
                ("pmj02bnSamples", u32p), ("blueNoise", u16p)]


class KzIntegrator(C.Structure):
    _fields_ = [("type", C.c_int32), ("maxDepth", C.c_int32), ("traceBias", C.c_float),
                ("regularization", C.c_int32), ("accumulatedRoughness", C.c_float)]


class KzBackground(C.Structure):
    _fields_ = [("present", C.c_int32), ("color", C.c_float * 3), ("intensity", C.c_float), ("texture", C.c_int32)]


class KzSceneDesc(C.Structure):
    _fields_ = [("abiVersion", C.c_uint32),
                ("meshes", C.POINTER(KzMesh)), ("nMeshes", C.c_uint32),
                ("bsdfs", C.POINTER(KzBSDF)), ("nBsdfs", C.c_uint32),
                ("lights", C.POINTER(KzLight)), ("nLights", C.c_uint32),
                ("camera", KzCamera), ("sampler", KzSampler), ("integrator", KzIntegrator),
                ("background", KzBackground),
                ("textures", C.POINTER(KzTexture)), ("nTextures", C.c_uint32),
                ("images", C.POINTER(KzImage)), ("nImages", C.c_uint32)]


class KzTile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class KzTuning(C.Structure):
    _fields_ = [("refill", C.c_int32), ("postpone", C.c_int32), ("batch", C.c_int32), ("traceBlocksPerCU", C.c_int32),
                ("shadeBlocksPerCU", C.c_int32), ("ldsStack", C.c_int32), ("bvh2", C.c_int32), ("packetPrimary", C.c_int32),
                ("keyStack", C.c_int32), ("ldsTop", C.c_int32), ("filmGather", C.c_int32), ("leafQueue", C.c_int32), ("sppPerPass", C.c_int32), ("legacyTrace", C.c_int32),
                ("mixedLaunch", C.c_int32), ("streamPriority", C.c_int32)]


class KzTileDealer(C.Structure):
    _fields_ = [("counter", u32p), ("batchTiles", C.c_uint32), ("takers", C.c_uint32), ("taken", u32p), ("takenCap", C.c_uint32), ("nTaken", u32p), ("agreed", u32p)]


class KzRenderOpts(C.Structure):
    _fields_ = [("sampleBegin", C.c_uint32), ("sampleEnd", C.c_uint32), ("tiles", C.POINTER(KzTile)),
                ("nTiles", C.c_uint32), ("pipeline", C.c_int32), ("accumulate", C.c_int32), ("stream", C.c_void_p),
                ("device", C.c_int32), ("passesInFlight", C.c_int32), ("passItems", C.c_uint64), ("maxStateBytes", C.c_uint64),
                ("tune", KzTuning), ("tileDealing", C.c_int32), ("packedOutput", C.c_int32), ("dealer", C.POINTER(KzTileDealer)),
                ("shadowBeside", C.c_int32), ("passHalves", C.c_int32)]


class KzPassInfo(C.Structure):
    _fields_ = [("passes", C.c_uint32), ("passesInFlight", C.c_uint32), ("itemsPerPass", C.c_uint64), ("sppPerPass", C.c_uint32),
                ("pixels", C.c_uint32), ("stateBytes", C.c_uint64), ("pixelsPerPass", C.c_uint32), ("shadowBeside", C.c_uint32),
                ("firstPassItems", C.c_uint64), ("largestPassItems", C.c_uint64), ("contextItems", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class KzPassModeInfo(C.Structure):
    _fields_ = [("kept", C.c_int32), ("timedPasses", C.c_uint32), ("items", C.c_uint64), ("msOneStream", C.c_float * 2), ("msShadowBeside", C.c_float), ("msHalves", C.c_float)]


class KzPlanQuery(C.Structure):
    _fields_ = [("pipeline", C.c_int32), ("nPix", C.c_uint32), ("sampleBegin", C.c_uint32), ("sampleEnd", C.c_uint32), ("passItems", C.c_uint64),
                ("passesInFlight", C.c_int32), ("sppPerPass", C.c_int32), ("limitBytes", C.c_uint64), ("bytesPerItem", C.c_uint64), ("dealer", C.c_int32),
                ("takers", C.c_uint32), ("batchTiles", C.c_uint32), ("nTiles", C.c_uint32), ("tilePixOffset", u32p), ("heldItems", C.c_uint64)]


class KzPlanAnswer(C.Structure):
    _fields_ = [("autoShape", C.c_int32), ("nCtx", C.c_int32), ("multi", C.c_int32), ("grow", C.c_int32), ("S", C.c_uint32), ("pixPerPass", C.c_uint32),
                ("batchTiles", C.c_uint32), ("nPixSet", C.c_uint32), ("nPasses", C.c_uint32), ("need", C.c_uint64), ("wantItems", C.c_uint64),
                ("minStart", C.c_uint64), ("graceMs", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KzStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("nodeVisits", C.c_uint64), ("triTests", C.c_uint64),
                ("shadedHits", C.c_uint64), ("lightSamples", C.c_uint64), ("droppedSamples", C.c_uint64),
                ("beamPixels", C.c_uint64), ("beamListEntries", C.c_uint64), ("beamCompletePixels", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class KzHit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("mesh", C.c_int32), ("prim", C.c_int32),
                ("p", C.c_float * 3), ("uv", C.c_float * 2), ("sh_s", C.c_float * 3), ("sh_t", C.c_float * 3),
                ("sh_n", C.c_float * 3), ("geo_n", C.c_float * 3)]


class KzBvhInfo(C.Structure):
    _fields_ = [("nNodes", C.c_uint32), ("nLeaves", C.c_uint32), ("nTris", C.c_uint32), ("maxDepth", C.c_uint32),
                ("maxLeafSize", C.c_uint32), ("sahCost", C.c_float), ("buildSeconds", C.c_double)]


# every symbol include/kazen_mi355x.h (the product surface, PRODUCT_EXPORTS) and include/kazen_mi355x_dev.h declare (checked by tests/test_abi_cpu.py)
PRODUCT_EXPORTS = ["kz_scene_create", "kz_scene_destroy", "kz_scene_upload", "kz_scene_evict", "kz_render", "kz_render_tiles", "kz_render_multi", "kz_deal_tiles",
                   "kz_tiles_packed_floats", "kz_film_download_tiles", "kz_film_merge_tiles", "kz_film_merge_rects", "kz_film_download", "kz_film_clear", "kz_film_dims", "kz_film_to_rgb",
                   "kz_film_to_srgb8", "kz_sync", "kz_last_error", "kz_abi_version", "kz_device_count", "kz_device_trim"]
EXPORTS = ["kz_scene_create", "kz_scene_destroy", "kz_scene_bvh_info", "kz_scene_upload", "kz_render",
           "kz_film_download", "kz_film_clear", "kz_film_dims", "kz_film_to_rgb", "kz_trace_rays",
           "kz_set_stats", "kz_get_stats", "kz_sync", "kz_last_kernel_ms", "kz_last_error", "kz_abi_version",
           "kz_device_count", "kz_render_samples", "kz_bsdf_query", "kz_scene_sample_count", "kz_last_stage_ms", "kz_texture_query", "kz_film_to_srgb8",
           "kz_scene_evict", "kz_scene_devices", "kz_render_tiles", "kz_render_multi", "kz_deal_tiles", "kz_film_merge", "kz_film_download_on",
           "kz_film_clear_on", "kz_sync_on", "kz_last_pass_info", "kz_device_mem_info", "kz_camera_rays", "kz_light_query", "kz_kat_exact_math", "kz_kat_permute", "kz_kat_fresnel", "kz_kat_math", "kz_build_flags",
           "kz_tiles_packed_floats", "kz_film_download_tiles", "kz_film_merge_tiles", "kz_film_merge_rects", "kz_device_trim", "kz_kat_dpdf", "kz_kat_pow4", "kz_last_grow_note",
           "kz_plan_passes", "kz_plan_schedule", "kz_pass_mode_info"]
# exported by DEVELOPMENT builds of the library only (-DKZ_EXPERIMENTS): the hooks that are process-global state. The product library must NOT export them.
DEV_ONLY_EXPORTS = ["kz_debug_fail_alloc", "kz_debug_fail_device", "kz_debug_grow_delay", "kz_debug_trace", "kz_debug_alias_devices"]

_HERE = os.path.dirname(os.path.abspath(__file__))
# KZ_LIB_PATH: a development build of the library (scripts/build_variant.sh) instead of the in-tree one; probes only
LIB_PATH = os.environ.get("KZ_LIB_PATH") or os.path.join(_HERE, "csrc", "libkazen_mi355x.so")
DEV_LIB_PATH = os.path.join(_HERE, "csrc", "variants", "experiments", "libkazen_mi355x.so")
_libs = {}


class KzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kazen_mi355x error %d: %s" % (code, msg))
        self.code = code


def load_dev_library():
    """The development variant of the library (-DKZ_EXPERIMENTS: the same sources plus the kernels of rejected experiments and the hooks that are process-global
    state - kz_debug_fail_alloc / grow_delay / trace / alias_devices). Only tests load it; a second copy of the library in one process is a separate world
    (its own device pools, its own replicas)."""
    return load_library(DEV_LIB_PATH)


def load_library(path=None):
    """Load the HIP extension. No fallback: a missing build is an error."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the product path has no CPU fallback)" % path)
    lib = C.CDLL(path)
    lib.kz_last_error.restype = C.c_char_p
    lib.kz_scene_create.argtypes = [C.POINTER(KzSceneDesc), C.POINTER(C.c_void_p)]
    lib.kz_scene_destroy.argtypes = [C.c_void_p]
    lib.kz_scene_destroy.restype = None
    lib.kz_scene_bvh_info.argtypes = [C.c_void_p, C.POINTER(KzBvhInfo)]
    lib.kz_scene_sample_count.argtypes = [C.c_void_p, u32p]
    lib.kz_scene_upload.argtypes = [C.c_void_p, C.c_int]
    lib.kz_render.argtypes = [C.c_void_p, C.POINTER(KzRenderOpts)]
    lib.kz_film_download.argtypes = [C.c_void_p, f32p, C.c_size_t]
    lib.kz_film_clear.argtypes = [C.c_void_p, C.c_void_p]
    lib.kz_film_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.kz_film_to_rgb.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, f32p]
    lib.kz_trace_rays.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, f32p, C.POINTER(KzHit)]
    lib.kz_set_stats.argtypes = [C.c_void_p, C.c_int]
    lib.kz_get_stats.argtypes = [C.c_void_p, C.POINTER(KzStats), C.c_int]
    lib.kz_sync.argtypes = [C.c_void_p]
    lib.kz_last_kernel_ms.argtypes = [C.c_void_p, f32p]
    lib.kz_last_stage_ms.argtypes = [C.c_void_p, f32p]
    lib.kz_render_samples.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), u32p, f32p]
    lib.kz_bsdf_query.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p]
    lib.kz_film_to_srgb8.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t]
    lib.kz_texture_query.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), f32p, f32p]
    if hasattr(lib, "kz_camera_rays"):        # absent only in a KZ_LIB_PATH development build of older sources (same-call A/B runs)
        lib.kz_camera_rays.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p]
        lib.kz_light_query.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), f32p, f32p, f32p]
    lib.kz_scene_evict.argtypes = [C.c_void_p, C.c_int]
    lib.kz_scene_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, u32p]
    lib.kz_render_tiles.argtypes = [C.c_void_p, C.POINTER(KzRenderOpts), C.POINTER(KzTile), C.c_uint32, C.c_int, f32p, C.c_size_t]
    lib.kz_render_multi.argtypes = [C.c_void_p, C.POINTER(KzRenderOpts), C.POINTER(C.c_int32), C.c_uint32, C.c_int32, f32p, C.c_size_t, f32p]
    lib.kz_deal_tiles.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(KzTile), C.c_uint32, u32p]
    lib.kz_film_merge.argtypes = [f32p, f32p, C.c_size_t]
    lib.kz_tiles_packed_floats.argtypes = [C.c_void_p, C.POINTER(KzTile), C.c_uint32, C.POINTER(C.c_size_t)]
    lib.kz_film_download_tiles.argtypes = [C.c_void_p, C.c_int, C.POINTER(KzTile), C.c_uint32, f32p, C.c_size_t]
    lib.kz_film_merge_tiles.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(KzTile), C.c_uint32, f32p, C.c_size_t, C.c_int32]
    lib.kz_film_merge_rects.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(KzTile), C.POINTER(f32p), C.c_uint32, C.c_int32]
    lib.kz_film_download_on.argtypes = [C.c_void_p, C.c_int, f32p, C.c_size_t]
    lib.kz_film_clear_on.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.kz_sync_on.argtypes = [C.c_void_p, C.c_int]
    lib.kz_last_pass_info.argtypes = [C.c_void_p, C.POINTER(KzPassInfo)]
    lib.kz_pass_mode_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(KzPassModeInfo)]
    lib.kz_device_mem_info.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.kz_device_trim.argtypes = [C.c_int]
    lib.kz_last_grow_note.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    for hook in DEV_ONLY_EXPORTS:                 # development builds only
        if hasattr(lib, hook):
            getattr(lib, hook).argtypes = [C.c_int, C.c_int] if hook == "kz_debug_fail_device" else [C.c_int]
            getattr(lib, hook).restype = None
    if hasattr(lib, "kz_plan_passes"):
        lib.kz_plan_passes.argtypes = [C.POINTER(KzPlanQuery), C.POINTER(KzPlanAnswer)]
        lib.kz_plan_schedule.argtypes = [C.POINTER(KzPlanQuery), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p]
    lib.kz_kat_dpdf.argtypes = [C.c_uint32, f32p, f32p, f32p]
    lib.kz_kat_pow4.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
    if hasattr(lib, "kz_kat_math"):
        lib.kz_kat_math.argtypes = [C.c_int, C.c_int, C.c_uint32, f32p, f32p, f32p]
    if hasattr(lib, "kz_kat_fresnel"):
        lib.kz_kat_fresnel.argtypes = [C.c_int, C.c_uint32, C.c_int, f32p, f32p, f32p, f32p]
    if hasattr(lib, "kz_kat_permute"):
        lib.kz_kat_permute.argtypes = [C.c_int, C.c_uint32, u32p, u32p, u32p, u32p]
    if hasattr(lib, "kz_kat_exact_math"):
        lib.kz_kat_exact_math.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _libs[path] = lib
    return lib


def check(lib, rc):
    if rc != KZ_OK:
        msg = lib.kz_last_error()
        raise KzError(rc, msg.decode() if msg else "")
