"""Image output of the renderer (src/kazen/bitmap.cpp:23-64, renderer.cpp:140-152): the 8-bit sRGB PNG the reference
writes by default and the float RGB OpenEXR of Bitmap::saveEXR. Host-side file I/O; the tone map itself runs on the device
(Scene.srgb8 -> kz_film_to_srgb8). The reference writes through OpenImageIO; the EXR here is the plain uncompressed
scan-line form of the OpenEXR file layout (magic 20000630, version 2), readable by any EXR reader."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def save_png(path, rgb8):
    """rgb8: (h, w, 3) uint8, row 0 = top (what Scene.srgb8() returns). Appends ".png" like Bitmap::savePNG when missing."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("save_png expects an (h, w, 3) uint8 raster")
    h, w, _ = a.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()      # filter type 0 per scan line
    if not path.endswith(".png"):
        path += ".png"
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b""))
    return path


def _attr(name, typ, value):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value


def exr_bytes(rgb):
    """The bytes of an uncompressed scan-line OpenEXR holding (h, w, 3) float32 RGB (channels B, G, R as FLOAT)."""
    a = np.ascontiguousarray(rgb, np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("exr_bytes expects an (h, w, 3) float raster")
    h, w, _ = a.shape
    chl = b"".join(n + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = (struct.pack("<II", 20000630, 2) + _attr("channels", "chlist", chl) + _attr("compression", "compression", b"\0") + _attr("dataWindow", "box2i", box) +
           _attr("displayWindow", "box2i", box) + _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
           _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    line = 8 + 12 * w
    first = len(hdr) + 8 * h
    table = struct.pack("<%dQ" % h, *[first + y * line for y in range(h)])
    planar = a[:, :, ::-1].transpose(0, 2, 1)                                  # per scan line: all B, all G, all R
    body = b"".join(struct.pack("<ii", y, 12 * w) + planar[y].astype("<f4").tobytes() for y in range(h))
    return hdr + table + body


def save_exr(path, rgb):
    """rgb: (h, w, 3) float32 linear radiance (Scene.rgb()). Appends ".exr" like Bitmap::saveEXR when missing."""
    if not path.endswith(".exr"):
        path += ".exr"
    with open(path, "wb") as f:
        f.write(exr_bytes(rgb))
    return path


def load_exr(path):
    """Reader for the uncompressed scan-line FLOAT files written above (tests; Bitmap(filename), bitmap.cpp:7-21, 3 channels only)."""
    b = open(path, "rb").read()
    if struct.unpack_from("<I", b, 0)[0] != 20000630:
        raise ValueError("not an OpenEXR file")
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        n = struct.unpack_from("<i", b, pos)[0]; pos += 4
        attrs[name] = (typ, b[pos:pos + n]); pos += n
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("only uncompressed files are supported")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp); names.append(cl[cp:e].decode()); cp = e + 1
        if struct.unpack_from("<i", cl, cp)[0] != 2:
            raise ValueError("only FLOAT channels are supported")
        cp += 16
    if sorted(names) != ["B", "G", "R"]:
        raise ValueError("Bitmap: Only support 3 channel file for now")          # bitmap.cpp:14-15
    offs = struct.unpack_from("<%dQ" % h, b, pos)
    out = np.zeros((h, w, 3), np.float32)
    for o in offs:
        y, n = struct.unpack_from("<ii", b, o)
        row = np.frombuffer(b, "<f4", 3 * w, o + 8).reshape(3, w)
        for k, nm in enumerate(names):
            out[y - y0, :, "RGB".index(nm)] = row[k]
    return out
