"""Build container only (needs /root/reference): the HIP pictures of scripts/dev/q1_full.py against the reference's published 4096-spp picture of the same scene file,
pixel by pixel at full resolution:   python scripts/dev/q1_compare.py [dir with the hip_*.png, default gpurun_out/q1_full]"""
import json, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "q1_full")
ref = np.asarray(Image.open("/root/reference/doc/2022_q1/img/param/default_m0_r0.5.png").convert("RGB"), np.float64)
lin = lambda p: np.where(p / 255 <= 0.04045, p / 255 / 12.92, np.power((p / 255 + 0.055) / 1.055, 2.4))
out = {}
for tag in ("as_checked_in", "light_factors"):
    hip = np.asarray(Image.open(os.path.join(D, "hip_default_m0_r0.5_%s.png" % tag)).convert("RGB"), np.float64)
    ok = ~((ref >= 255).any(axis=2) | (hip >= 255).any(axis=2))                       # clipped highlights carry no information
    d = (hip - ref)[ok]
    lr, lh = lin(ref)[ok].mean(axis=1), lin(hip)[ok].mean(axis=1)
    lit = lr > 0.02
    out[tag] = {"pixels_compared": int(ok.sum()), "share_of_frame": round(float(ok.mean()), 4),
                "mean_signed_diff_8bit": round(float(d.mean()), 3), "mean_abs_diff_8bit": round(float(np.abs(d).mean()), 3), "rms_diff_8bit": round(float(np.sqrt((d ** 2).mean())), 3),
                "share_within_1": round(float((np.abs(d) <= 1).all(axis=1).mean()), 4), "share_within_2": round(float((np.abs(d) <= 2).all(axis=1).mean()), 4),
                "share_within_4": round(float((np.abs(d) <= 4).all(axis=1).mean()), 4), "share_within_8": round(float((np.abs(d) <= 8).all(axis=1).mean()), 4),
                "mean_linear_ratio_hip_over_published": round(float(lh[lit].sum() / lr[lit].sum()), 5)}
    g = np.where(ok[..., None], hip - ref, np.nan).mean(axis=2).reshape(9, 120, 16, 120)
    out[tag]["signed_diff_8bit_on_a_16x9_grid"] = np.round(np.nanmean(g, axis=(1, 3)), 1).tolist()
    obj = np.zeros(ref.shape[:2], bool); obj[240:840, 720:1200] = True                 # the object and its contact shadow (columns 6-9, rows 2-6 of the grid)
    out[tag]["object_region_mean_abs_diff_8bit"] = round(float(np.abs((hip - ref)[ok & obj]).mean()), 3)
    out[tag]["backdrop_mean_abs_diff_8bit"] = round(float(np.abs((hip - ref)[ok & ~obj]).mean()), 3)
    print(tag, {k: v for k, v in out[tag].items() if k != "signed_diff_8bit_on_a_16x9_grid"})
json.dump(out, open(os.path.join(D, "q1_vs_published.json"), "w"), indent=1)
