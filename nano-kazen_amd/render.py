"""Thin handle around a KzScene* (the library's immutable scene + device tables + device film)."""
import ctypes as C

import numpy as np

from . import abi


class Scene:
    """kz_scene_create -> [kz_scene_upload] -> kz_render* -> kz_film_download."""

    def __init__(self, desc, device=None, lib=None):
        self.lib = lib or abi.load_library()      # (lib: a development build of the library, abi.load_dev_library() - tests only)
        self.desc = desc
        cdesc = desc.to_c()
        h = C.c_void_p()
        abi.check(self.lib, self.lib.kz_scene_create(C.byref(cdesc), C.byref(h)))
        self.h = h
        self.device = None
        w, hh, b = C.c_int32(), C.c_int32(), C.c_int32()
        abi.check(self.lib, self.lib.kz_film_dims(self.h, C.byref(w), C.byref(hh), C.byref(b)))
        self.width, self.height, self.border = w.value, hh.value, b.value
        n = C.c_uint32()
        abi.check(self.lib, self.lib.kz_scene_sample_count(self.h, C.byref(n)))
        self.sample_count = n.value
        if device is not None:
            self.upload(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kz_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_info(self):
        info = abi.KzBvhInfo()
        abi.check(self.lib, self.lib.kz_scene_bvh_info(self.h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def upload(self, device=0):
        """Adds a replica on `device` (the first one uploaded is the primary, which the calls without a device address)."""
        abi.check(self.lib, self.lib.kz_scene_upload(self.h, int(device)))
        if self.device is None:
            self.device = int(device)

    def evict(self, device=-1):
        abi.check(self.lib, self.lib.kz_scene_evict(self.h, int(device)))
        if device < 0 or device == self.device:
            left = self.devices()
            self.device = left[0] if left else None            # the next replica becomes the one the calls without a device address

    def devices(self):
        buf = (C.c_int32 * 64)()
        n = C.c_uint32()
        abi.check(self.lib, self.lib.kz_scene_devices(self.h, buf, 64, C.byref(n)))
        return [int(buf[i]) for i in range(n.value)]

    def _opts(self, sample_begin=0, sample_end=0, tiles=None, accumulate=False, pipeline=0, stream=None, device=None,
              passes_in_flight=0, pass_items=0, max_state_bytes=0, tune=None, tile_dealing=0, shadow_beside=0, pass_halves=0):
        o = abi.KzRenderOpts()
        o.sampleBegin, o.sampleEnd = sample_begin, sample_end
        keep = None
        if tiles is not None:
            keep = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
            o.tiles, o.nTiles = keep, len(tiles)
        o.pipeline = pipeline
        o.accumulate = 1 if accumulate else 0
        o.stream = stream
        o.device = (self.device or 0) if device is None else int(device)
        o.passesInFlight, o.passItems, o.maxStateBytes = int(passes_in_flight), int(pass_items), int(max_state_bytes)
        o.tileDealing = int(tile_dealing)
        o.shadowBeside = int(shadow_beside)
        o.passHalves = int(pass_halves)
        for k, v in (tune or {}).items():
            setattr(o.tune, k, int(v))
        return o, keep

    def render(self, sample_begin=0, sample_end=0, tiles=None, accumulate=False, pipeline=0, stream=None, **kw):
        """kz_render. Keywords: device, passes_in_flight, pass_items, max_state_bytes, shadow_beside, pass_halves, tune={KzTuning field: value}."""
        o, keep = self._opts(sample_begin, sample_end, tiles, accumulate, pipeline, stream, **kw)
        abi.check(self.lib, self.lib.kz_render(self.h, C.byref(o)))

    def render_tiles(self, tiles, device=None, sample_begin=0, sample_end=0, download=True, packed=False, **kw):
        """kz_render_tiles: blocking render of `tiles` on `device`. download=True returns that replica's whole film, packed=True the
        PACKED film rects of the tiles (1-D float32: every tile's (h+2b) x (w+2b) x 4 rect in list order), download=False nothing."""
        o, _ = self._opts(sample_begin, sample_end, **kw)
        arr = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
        dev = (self.device or 0) if device is None else int(device)
        n = self.packed_floats(tiles) if packed else (self.height + 2 * self.border) * (self.width + 2 * self.border) * 4
        o.packedOutput = 1 if packed else 0
        out = np.empty(n, np.float32) if (download or packed) else None
        abi.check(self.lib, self.lib.kz_render_tiles(self.h, C.byref(o), arr, len(tiles), dev,
                                                      out.ctypes.data_as(abi.f32p) if out is not None else None, n if out is not None else 0))
        if out is None or packed:
            return out
        return out.reshape(self.height + 2 * self.border, self.width + 2 * self.border, 4)

    def render_dealt(self, tiles, counter, takers=1, batch_tiles=0, device=None, sample_begin=0, sample_end=0, **kw):
        """kz_render_tiles with a KzTileDealer: `tiles` is the WHOLE list every taker passes, `counter` a numpy uint32 array of one element that all
        takers share (process-local, or a np.memmap of a file in /dev/shm for the ranks of a node; zeroed by the launcher). Renders the batches this
        call wins and returns the tiles it took, in the order it took them (hand them to film_tiles for the gather)."""
        o, _ = self._opts(sample_begin, sample_end, **kw)
        arr = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
        dev = (self.device or 0) if device is None else int(device)
        taken = np.zeros(2 * len(tiles) + 2, np.uint32)
        n_taken = np.zeros(1, np.uint32)
        assert counter.dtype == np.uint32 and counter.size >= 1
        # (a counter array of two or more words: word 1 is the takers' agreement word, KzTileDealer.agreed)
        agreed = C.cast(counter.ctypes.data + 4, abi.u32p) if counter.size >= 2 else None
        dl = abi.KzTileDealer(counter.ctypes.data_as(abi.u32p), int(batch_tiles), int(takers), taken.ctypes.data_as(abi.u32p), taken.size, n_taken.ctypes.data_as(abi.u32p), agreed)
        o.dealer = C.pointer(dl)
        abi.check(self.lib, self.lib.kz_render_tiles(self.h, C.byref(o), arr, len(tiles), dev, None, 0))
        out = []
        for k in range(0, int(n_taken[0]), 2):
            out += list(tiles[int(taken[k]):int(taken[k + 1])])
        return out

    def packed_floats(self, tiles):
        arr = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
        n = C.c_size_t()
        abi.check(self.lib, self.lib.kz_tiles_packed_floats(self.h, arr, len(tiles), C.byref(n)))
        return int(n.value)

    def film_tiles(self, tiles, device=None):
        """kz_film_download_tiles: the packed film rects of `tiles` from the replica's film."""
        arr = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
        dev = (self.device or 0) if device is None else int(device)
        out = np.empty(self.packed_floats(tiles), np.float32)
        abi.check(self.lib, self.lib.kz_film_download_tiles(self.h, dev, arr, len(tiles), out.ctypes.data_as(abi.f32p), out.size))
        return out

    def merge_tiles(self, film, tiles, packed, threads=0):
        """kz_film_merge_tiles: film += the packed rects, in list order (ImageBlock::put(ImageBlock&), block.cpp:87-96)."""
        arr = (abi.KzTile * len(tiles))(*[abi.KzTile(*t) for t in tiles])
        packed = np.ascontiguousarray(packed, np.float32)
        assert film.dtype == np.float32 and film.flags["C_CONTIGUOUS"]
        abi.check(self.lib, self.lib.kz_film_merge_tiles(film.ctypes.data_as(abi.f32p), self.width, self.height, self.border, arr, len(tiles),
                                                          packed.ctypes.data_as(abi.f32p), packed.size, int(threads)))
        return film

    def merge_rects(self, film, entries, threads=0):
        """kz_film_merge_rects: entries = [(tile, float32 array holding that tile's packed rect)], added to `film` in list order (row-major tile order = the
        film every other path gives). The arrays may be views into different buffers: nothing is copied."""
        n = len(entries)
        arr = (abi.KzTile * n)(*[abi.KzTile(*t) for t, _ in entries])
        ptrs = (abi.f32p * n)(*[C.cast(r.ctypes.data, abi.f32p) for _, r in entries])
        b = self.border
        for t, r in entries:
            if r.dtype != np.float32 or r.size != (t[2] + 2 * b) * (t[3] + 2 * b) * 4 or not r.flags["C_CONTIGUOUS"]:
                raise ValueError("merge_rects: the rect of tile %s must be %d contiguous float32" % (t, (t[2] + 2 * b) * (t[3] + 2 * b) * 4))
        abi.check(self.lib, self.lib.kz_film_merge_rects(film.ctypes.data_as(abi.f32p), self.width, self.height, self.border, arr, ptrs, n, int(threads)))
        return film

    def empty_film(self):
        return np.zeros((self.height + 2 * self.border, self.width + 2 * self.border, 4), np.float32)

    def render_multi(self, devices, tile_size=0, sample_begin=0, sample_end=0, **kw):
        """kz_render_multi: one host thread per device, tiles dealt by area, films summed on the host in device order.
        Returns (film, per-device ms)."""
        o, _ = self._opts(sample_begin, sample_end, **kw)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        n = (self.height + 2 * self.border) * (self.width + 2 * self.border) * 4
        out = np.empty(n, np.float32)
        ms = np.zeros(len(devices), np.float32)
        abi.check(self.lib, self.lib.kz_render_multi(self.h, C.byref(o), devs, len(devices), int(tile_size), out.ctypes.data_as(abi.f32p), n,
                                                      ms.ctypes.data_as(abi.f32p)))
        if self.device is None:
            self.device = int(devices[0])
        return out.reshape(self.height + 2 * self.border, self.width + 2 * self.border, 4), ms

    def last_pass_info(self):
        info = abi.KzPassInfo()
        abi.check(self.lib, self.lib.kz_last_pass_info(self.h, C.byref(info)))
        return info.as_dict()

    def pass_mode_info(self, device=-1):
        """kz_pass_mode_info: what the replica measured about its large passes and what it keeps (None while undecided)."""
        m = abi.KzPassModeInfo()
        abi.check(self.lib, self.lib.kz_pass_mode_info(self.h, int(device), C.byref(m)))
        d = {"kept": (None, "one stream", "shadow rays beside", "halves")[m.kept + 1], "timed_passes": int(m.timedPasses), "items": int(m.items)}
        if m.kept >= 0:
            d.update({"ms_one_stream": [round(float(m.msOneStream[0]), 2), round(float(m.msOneStream[1]), 2)], "ms_shadow_beside": round(float(m.msShadowBeside), 2), "ms_halves": round(float(m.msHalves), 2)})
        return d

    def last_grow_note(self):
        """Why the pass context of the last render stopped growing short of its target ('' if it did not)."""
        buf = C.create_string_buffer(512)
        abi.check(self.lib, self.lib.kz_last_grow_note(self.h, buf, 512))
        return buf.value.decode()

    def sync(self):
        abi.check(self.lib, self.lib.kz_sync(self.h))

    def film(self):
        n = (self.height + 2 * self.border) * (self.width + 2 * self.border) * 4
        out = np.empty(n, np.float32)
        abi.check(self.lib, self.lib.kz_film_download(self.h, out.ctypes.data_as(abi.f32p), n))
        return out.reshape(self.height + 2 * self.border, self.width + 2 * self.border, 4)

    def film_clear(self, stream=None):
        abi.check(self.lib, self.lib.kz_film_clear(self.h, stream))

    def rgb(self, film=None):
        film = self.film() if film is None else np.ascontiguousarray(film, np.float32)
        out = np.empty((self.height, self.width, 3), np.float32)
        abi.check(self.lib, self.lib.kz_film_to_rgb(film.ctypes.data_as(abi.f32p), self.width, self.height, self.border,
                                                     out.ctypes.data_as(abi.f32p)))
        return out

    def trace_rays(self, o, d, tmin, tmax):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        hits = (abi.KzHit * n)()
        abi.check(self.lib, self.lib.kz_trace_rays(self.h, n, o.ctypes.data_as(abi.f32p), d.ctypes.data_as(abi.f32p),
                                                   tmin.ctypes.data_as(abi.f32p), tmax.ctypes.data_as(abi.f32p), hits))
        return hits_to_arrays(hits, n)

    def render_samples(self, pxy, idx):
        pxy = np.ascontiguousarray(pxy, np.int32)
        idx = np.ascontiguousarray(idx, np.uint32)
        n = idx.shape[0]
        out = np.zeros((n, 5), np.float32)
        abi.check(self.lib, self.lib.kz_render_samples(self.h, n, pxy.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       idx.ctypes.data_as(abi.u32p), out.ctypes.data_as(abi.f32p)))
        return out

    def bsdf_query(self, bsdf, wi, wo, acc, s3, uv=None):
        """eval (n,3), pdf (n,), sample (n,8) = weight rgb, wo xyz, alive, pdf(bRec) after sample()."""
        bsdf = np.ascontiguousarray(bsdf, np.int32)
        n = bsdf.shape[0]
        wi, wo, s3 = (np.ascontiguousarray(a, np.float32) for a in (wi, wo, s3))
        acc = np.ascontiguousarray(acc, np.float32)
        uv = None if uv is None else np.ascontiguousarray(uv, np.float32)
        ev, pd, sm = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 8), np.float32)
        f = lambda a: a.ctypes.data_as(abi.f32p)
        abi.check(self.lib, self.lib.kz_bsdf_query(self.h, n, bsdf.ctypes.data_as(C.POINTER(C.c_int32)), f(wi), f(wo), f(acc), f(s3),
                                                   None if uv is None else f(uv), f(ev), f(pd), f(sm)))
        return ev, pd, sm

    def texture_query(self, tex, uv):
        tex = np.ascontiguousarray(tex, np.int32)
        uv = np.ascontiguousarray(uv, np.float32)
        out = np.zeros((tex.shape[0], 3), np.float32)
        abi.check(self.lib, self.lib.kz_texture_query(self.h, tex.shape[0], tex.ctypes.data_as(C.POINTER(C.c_int32)), uv.ctypes.data_as(abi.f32p),
                                                      out.ctypes.data_as(abi.f32p)))
        return out

    def camera_rays(self, sxy, axy=None):
        """Camera::sampleRay for pixel-sample positions sxy (n,2) and aperture samples axy (n,2) -> (n,8) = o, d, mint, maxt."""
        sxy = np.ascontiguousarray(sxy, np.float32)
        axy = None if axy is None else np.ascontiguousarray(axy, np.float32)
        out = np.zeros((sxy.shape[0], 8), np.float32)
        f = lambda a: a.ctypes.data_as(abi.f32p)
        abi.check(self.lib, self.lib.kz_camera_rays(self.h, sxy.shape[0], f(sxy), None if axy is None else f(axy), f(out)))
        return out

    def light_query(self, light, ref, u3):
        """AreaLight::sample of light rows from ref with Mesh::sample's three draws -> (n,14) = p, n, wi, pdf, eval/pdf, triangle."""
        light = np.ascontiguousarray(light, np.int32)
        ref, u3 = np.ascontiguousarray(ref, np.float32), np.ascontiguousarray(u3, np.float32)
        out = np.zeros((light.shape[0], 14), np.float32)
        f = lambda a: a.ctypes.data_as(abi.f32p)
        abi.check(self.lib, self.lib.kz_light_query(self.h, light.shape[0], light.ctypes.data_as(C.POINTER(C.c_int32)), f(ref), f(u3), f(out)))
        return out

    def srgb8(self):
        """(h, w, 3) uint8: the raster Bitmap::savePNG writes (bitmap.cpp:39-62), resolved on the device."""
        out = np.zeros((self.height, self.width, 3), np.uint8)
        abi.check(self.lib, self.lib.kz_film_to_srgb8(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
        return out

    def set_stats(self, enable=True):
        abi.check(self.lib, self.lib.kz_set_stats(self.h, 1 if enable else 0))

    def stats(self, reset=False):
        s = abi.KzStats()
        abi.check(self.lib, self.lib.kz_get_stats(self.h, C.byref(s), 1 if reset else 0))
        return s.as_dict()

    def last_stage_ms(self):
        out = np.zeros(6, np.float32)
        abi.check(self.lib, self.lib.kz_last_stage_ms(self.h, out.ctypes.data_as(abi.f32p)))
        d = dict(zip(("generate", "trace_bounce", "shade", "trace_shadow", "film", "trace_camera"), [round(float(x), 3) for x in out]))
        d["trace_closest"] = round(d["trace_bounce"] + d["trace_camera"], 3)
        return d

    def last_kernel_ms(self):
        ms = C.c_float()
        abi.check(self.lib, self.lib.kz_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


def hits_to_arrays(hits, n):
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(n, C.sizeof(abi.KzHit))
    w = C.sizeof(abi.KzHit) // 4
    f = raw.view(np.float32).reshape(n, w)
    i = raw.view(np.int32).reshape(n, w)
    return {"t": f[:, 0].copy(), "u": f[:, 1].copy(), "v": f[:, 2].copy(), "mesh": i[:, 3].copy(), "prim": i[:, 4].copy(),
            "p": f[:, 5:8].copy(), "uv": f[:, 8:10].copy(), "sh_s": f[:, 10:13].copy(), "sh_t": f[:, 13:16].copy(),
            "sh_n": f[:, 16:19].copy(), "geo_n": f[:, 19:22].copy()}
