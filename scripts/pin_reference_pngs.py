"""The only reference-produced evidence for this path: the 22 published parameter-sweep pictures
doc/2022_q1/img/param/*.png (1920x1080, 4096 spp, tone-mapped by Bitmap::savePNG) next to the scene files that made them,
scene/2022_q1/parameters/*.xml. This script renders every scene UNCHANGED with the CPU oracle at a reduced size, compares it
with its picture in LINEAR radiance and writes

    profiles/pin/table.json        per image: mean ratio oracle/PNG, its spread, the ratio per region (floor, object, top rows),
                                   block-wise sRGB differences (the numbers tests/test_xmlscene.py holds a bound on)
    profiles/pin/ratio_<name>.png  ratio map thumbnails (blue = oracle darker, red = brighter, +-12 % full scale)
    profiles/pin/experiments.json  single-variable experiments on the residual (see EXPERIMENTS below)
    profiles/pin/ratio_vs_level.json  the ratio binned by the PICTURE's pixel level over all 22 images

Runs only where /root/reference exists (the build container); nothing here ships or runs on the GPU box.
    python scripts/pin_reference_pngs.py table [--spp 256]
    python scripts/pin_reference_pngs.py experiments [--spp 256]
"""
import argparse
import copy
import glob
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF_XML = "/root/reference/scene/2022_q1/parameters"
REF_PNG = "/root/reference/doc/2022_q1/img/param"
OUT = os.path.join(ROOT, "profiles", "pin")
W, H = 480, 270
GRID = (54, 96)          # rows, cols of the comparison grid (5x5-pixel blocks)


def srgb_to_linear(v):
    return np.where(v <= 0.04045, v / 12.92, np.power((v + 0.055) / 1.055, 2.4))


def linear_to_srgb(x):
    x = np.clip(x, 0, None)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


def load_png_linear(name):
    """The picture in linear radiance at the oracle's resolution; `sat` marks blocks that contain clipped (255) pixels."""
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(REF_PNG, name + ".png")).convert("RGB"), np.float64) / 255.0
    sat = (a >= 1.0).any(axis=2)
    lin = srgb_to_linear(a)
    f = a.shape[0] // H
    lin = lin.reshape(H, f, W, f, 3).mean(axis=(1, 3))
    sat = sat.reshape(H, f, W, f).any(axis=(1, 3))
    return lin, sat, a


def blocks(x, grid=GRID):
    r, c = grid
    return x.reshape(r, x.shape[0] // r, c, x.shape[1] // c, *x.shape[2:]).mean(axis=(1, 3))


def render(desc, threads):
    import oracle as O
    o = O.OracleScene(desc)
    return o.rgb(o.render(threads=threads)).astype(np.float64)


def load(kz, name, spp, **over):
    ov = {"camera": {"width": W, "height": H}, "sampler": {"sampleCount": spp}}
    for k, v in over.items():
        ov.setdefault(k, {}).update(v)
    return kz.xmlscene.load_xml(os.path.join(REF_XML, name + ".xml"), ov)


def compare(rgb, name):
    """rgb: oracle, linear (H, W, 3). Returns the statistics of one image and the block ratio map."""
    lin, sat, _ = load_png_linear(name)
    lo = blocks(rgb.mean(axis=2)); lp = blocks(lin.mean(axis=2)); bs = blocks(sat.astype(np.float64)) > 0
    ok = (~bs) & (lp > 0.02)
    ratio = np.where(ok, lo / np.maximum(lp, 1e-9), np.nan)
    rows = GRID[0]
    top = ratio[: rows // 6]; floor = ratio[int(rows * 0.72):]; centre = ratio[int(rows * 0.30):int(rows * 0.65), int(GRID[1] * 0.38):int(GRID[1] * 0.62)]
    s_o = blocks(linear_to_srgb(np.clip(rgb, 0, 1)), (9, 16)); s_p = blocks(linear_to_srgb(np.clip(lin, 0, 1)), (9, 16))
    fine_o = blocks(linear_to_srgb(np.clip(rgb, 0, 1))); fine_p = blocks(linear_to_srgb(np.clip(lin, 0, 1)))
    st = {"mean_ratio": float(np.nanmean(ratio)), "sigma_ratio": float(np.nanstd(ratio)),
          "ratio_top_rows": float(np.nanmean(top)), "ratio_floor": float(np.nanmean(floor)), "ratio_object": float(np.nanmean(centre)),
          "srgb_16x9_max_abs": float(np.abs(s_o - s_p).max()), "srgb_16x9_mean_diff": float(s_o.mean() - s_p.mean()),
          "srgb_96x54_max_abs": float(np.abs(fine_o - fine_p).max()), "srgb_96x54_mean_abs": float(np.abs(fine_o - fine_p).mean()),
          "blocks_used": int(ok.sum())}
    return st, ratio, (lo, lp, ok)


def save_ratio_png(ratio, path, scale=0.12):
    from PIL import Image
    r = np.nan_to_num(ratio - 1.0, nan=0.0) / scale
    img = np.ones(ratio.shape + (3,))
    img[..., 0] = np.where(r < 0, 1 + np.clip(r, -1, 0), 1.0); img[..., 1] = 1 - np.clip(np.abs(r), 0, 1); img[..., 2] = np.where(r > 0, 1 - np.clip(r, 0, 1), 1.0)
    img[np.isnan(ratio)] = 0.5
    Image.fromarray((img * 255).astype(np.uint8)).resize((ratio.shape[1] * 4, ratio.shape[0] * 4), Image.NEAREST).save(path)


def cmd_table(args, kz):
    names = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(REF_XML, "*.xml")))
    table, levels = {}, []
    for n in names:
        t0 = time.time()
        rgb = render(load(kz, n, args.spp), args.threads)
        st, ratio, (lo, lp, ok) = compare(rgb, n)
        st["oracle_seconds"] = round(time.time() - t0, 1)
        table[n] = st
        save_ratio_png(ratio, os.path.join(OUT, "ratio_%s.png" % n))
        levels.append((lp[ok], (lo / np.maximum(lp, 1e-9))[ok]))
        print("%-22s mean %.4f sigma %.4f top %.4f floor %.4f object %.4f | sRGB 16x9 max %.3f mean %+.4f | 96x54 max %.3f" % (
            n, st["mean_ratio"], st["sigma_ratio"], st["ratio_top_rows"], st["ratio_floor"], st["ratio_object"], st["srgb_16x9_max_abs"],
            st["srgb_16x9_mean_diff"], st["srgb_96x54_max_abs"]), flush=True)
    json.dump({"size": [W, H], "spp": args.spp, "grid": GRID, "images": table}, open(os.path.join(OUT, "table.json"), "w"), indent=1)
    # is the residual a function of the picture's LEVEL (a transfer-curve difference) rather than of position?
    lv = np.concatenate([a for a, _ in levels]); rt = np.concatenate([b for _, b in levels])
    edges = np.array([0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.01])
    bins = [{"png_linear_level": [float(edges[i]), float(edges[i + 1])], "blocks": int(((lv >= edges[i]) & (lv < edges[i + 1])).sum()),
             "mean_ratio": float(rt[(lv >= edges[i]) & (lv < edges[i + 1])].mean()) if ((lv >= edges[i]) & (lv < edges[i + 1])).any() else None,
             "sigma": float(rt[(lv >= edges[i]) & (lv < edges[i + 1])].std()) if ((lv >= edges[i]) & (lv < edges[i + 1])).any() else None}
            for i in range(len(edges) - 1)]
    json.dump(bins, open(os.path.join(OUT, "ratio_vs_level.json"), "w"), indent=1)
    for b in bins:
        print(b)


def scale_lights(desc, factors):
    """factors by light order (back, right, left)."""
    k = 0
    for m in desc.meshes:
        if m["light"]:
            m["light"]["intensity"] = m["light"]["intensity"] * factors[k]
            k += 1
    return desc


def cmd_experiments(args, kz):
    name = args.image
    out = {}

    def run(tag, desc, note):
        t0 = time.time()
        rgb = render(desc, args.threads)
        st, ratio, _ = compare(rgb, name)
        st["note"] = note
        out[tag] = st
        save_ratio_png(ratio, os.path.join(OUT, "exp_%s_%s.png" % (name, tag)))
        print("%-28s mean %.4f sigma %.4f top %.4f floor %.4f object %.4f (%.0fs)  %s" % (tag, st["mean_ratio"], st["sigma_ratio"], st["ratio_top_rows"],
              st["ratio_floor"], st["ratio_object"], time.time() - t0, note), flush=True)
        return rgb

    base = run("baseline", load(kz, name, args.spp), "scene file unchanged")
    # --- which light's contribution is off: per-light basis images and the least-squares weights that best explain the picture
    basis = []
    for i, ln in enumerate(("back", "right", "left")):
        f = [0.0, 0.0, 0.0]; f[i] = 1.0
        basis.append(run("only_light_" + ln, scale_lights(load(kz, name, args.spp), f), "the other two lights at intensity 0 (still picked with pdf 1/3)"))
    lin, sat, _ = load_png_linear(name)
    ok = (~(blocks(sat.astype(np.float64)) > 0)) & (blocks(lin.mean(axis=2)) > 0.02)
    A = np.stack([blocks(b.mean(axis=2))[ok] for b in basis], axis=1); y = blocks(lin.mean(axis=2))[ok]
    wts, *_ = np.linalg.lstsq(A, y, rcond=None)
    fit = A @ wts
    out["light_reweighting"] = {"weights_back_right_left": [float(x) for x in wts], "sigma_ratio_after": float(np.std(fit / y)), "mean_ratio_after": float(np.mean(fit / y)),
                                "note": "least-squares weights on the three per-light images; sigma is what no re-weighting of the lights removes"}
    print("light re-weighting:", out["light_reweighting"], flush=True)
    # --- light model variants
    d = load(kz, name, args.spp)
    for m in d.meshes:
        if m["light"] and m["N"] is not None:
            m["N"] = -m["N"]
    run("light_normals_flipped", d, "vertex normals of the three light meshes negated (Mesh::sample interpolates them, H8)")
    d = load(kz, name, args.spp)
    for m in d.meshes:
        if m["light"]:
            m["N"] = None
    run("light_face_normals", d, "light meshes without vertex normals: Mesh::sample falls back to the face normal (mesh.cpp:125-131)")
    d = load(kz, name, args.spp)
    for m in d.meshes:
        if m["light"]:
            m["light"]["primaryVisibility"] = True
    run("lights_visible", d, "lightPrimaryVisibility = true")
    # --- transport knobs
    for md in (3, 8, 16):
        run("maxDepth_%d" % md, load(kz, name, args.spp, integrator={"maxDepth": md}), "path_mis maxDepth %d (file: default 5)" % md)
    run("traceBias_1e-4", load(kz, name, args.spp, integrator={"traceBias": 1e-4}), "traceBias 1e-4 (default 1e-3)")
    d = load(kz, name, args.spp)
    for m in d.meshes:
        if not m["light"]:
            m["N"] = None
    run("no_vertex_normals", d, "object + backdrop without vertex normals: no terminator offset, geometric shading frames (H4)")
    d = load(kz, name, args.spp)
    d.camera["rfilter"] = {"type": "box"}
    run("box_filter", d, "box reconstruction filter instead of the default gaussian")
    d = load(kz, name, args.spp)
    for m in d.meshes:
        if m["bsdf"] and m["bsdf"].get("type") == "diffuse":
            m["bsdf"]["albedo"] = [0.8, 0.8, 0.8]
    run("backdrop_albedo_0.8", d, "backdrop albedo 0.8 instead of 1.0 (sensitivity of the pattern to inter-reflection)")
    # --- a picture-side hypothesis: the pictures went through another transfer curve than Color3f::toSRGB
    _, _, srgb_png = load_png_linear(name)
    f = srgb_png.shape[0] // H
    alt = np.power(srgb_png, 2.2).reshape(H, f, W, f, 3).mean(axis=(1, 3))
    lo = blocks(base.mean(axis=2)); lp = blocks(alt.mean(axis=2))
    r = (lo / np.maximum(lp, 1e-9))[ok]
    out["png_decoded_as_gamma_2.2"] = {"mean_ratio": float(r.mean()), "sigma_ratio": float(r.std()), "note": "picture linearised with a pure 2.2 power instead of the inverse of toSRGB"}
    print("png as gamma 2.2:", out["png_decoded_as_gamma_2.2"], flush=True)
    json.dump({"image": name, "size": [W, H], "spp": args.spp, "experiments": out}, open(os.path.join(OUT, "experiments_%s.json" % name), "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["table", "experiments"])
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--image", default="default_m0_r0.5")
    args = ap.parse_args()
    if not os.path.isdir(REF_XML):
        raise SystemExit("the reference checkout is not present: this script only runs in the build container")
    os.makedirs(OUT, exist_ok=True)
    kz = importlib.import_module("nano-kazen_amd")
    {"table": cmd_table, "experiments": cmd_experiments}[args.cmd](args, kz)


if __name__ == "__main__":
    main()
