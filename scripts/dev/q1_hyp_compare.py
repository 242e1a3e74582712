"""Build container only: the pictures of scripts/dev/q1_hypotheses.py against the published default_m0_r0.5.png."""
import glob, json, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.path.join(ROOT, "gpurun_out", "q1_hyp")
ref = np.asarray(Image.open("/root/reference/doc/2022_q1/img/param/default_m0_r0.5.png").convert("RGB"), np.float64)
lin = lambda p: np.where(p / 255 <= 0.04045, p / 255 / 12.92, np.power((p / 255 + 0.055) / 1.055, 2.4))
obj = np.zeros((1080, 1920), bool); obj[240:840, 720:1200] = True
np.set_printoptions(precision=1, suppress=True, linewidth=220)
out = {}
for f in sorted(glob.glob(os.path.join(D, "*.png"))):
    hip = np.asarray(Image.open(f).convert("RGB"), np.float64)
    ok = ~((ref >= 255).any(axis=2) | (hip >= 255).any(axis=2))
    d = hip - ref
    lr, lh = lin(ref).mean(axis=2), lin(hip).mean(axis=2)
    lit = ok & (lr > 0.02)
    g = np.nanmean(np.where(ok, d.mean(axis=2), np.nan).reshape(9, 120, 16, 120), axis=(1, 3))
    name = os.path.basename(f)[:-4]
    out[name] = {"ratio": round(float(lh[lit].sum() / lr[lit].sum()), 4), "abs_all": round(float(np.abs(d[ok]).mean()), 2), "abs_obj": round(float(np.abs(d[ok & obj]).mean()), 2),
                 "abs_back": round(float(np.abs(d[ok & ~obj]).mean()), 2), "column0_of_grid": np.round(g[:, 0], 1).tolist(), "column15": np.round(g[:, 15], 1).tolist()}
    print("%-26s ratio %.4f  abs all %.2f obj %.2f back %.2f   left column of the 16x9 grid %s" % (name, out[name]["ratio"], out[name]["abs_all"], out[name]["abs_obj"], out[name]["abs_back"], np.round(g[:, 0], 1)))
json.dump(out, open(os.path.join(D, "hyp_vs_published.json"), "w"), indent=1)
