"""Build container only: the 22 HIP pictures of `scripts/dev/q1_full.py --all` (4096 spp, light factors) against the reference's 22 published pictures, pixel by pixel:
python scripts/dev/q1_compare_all.py [dir, default gpurun_out/q1_full]"""
import json, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "q1_full")
lin = lambda p: np.where(p / 255 <= 0.04045, p / 255 / 12.92, np.power((p / 255 + 0.055) / 1.055, 2.4))
timing = json.load(open(os.path.join(D, "q1_all.json")))
obj = np.zeros((1080, 1920), bool); obj[240:840, 720:1200] = True                 # the object and its contact shadow
rows = {}
print("%-20s %8s %8s %9s %9s %9s %8s %8s" % ("scene file", "render s", "ratio", "abs all", "abs obj", "abs back", "<=8/255", "clipped"))
for name in sorted(timing):
    ref = np.asarray(Image.open("/root/reference/doc/2022_q1/img/param/%s.png" % name).convert("RGB"), np.float64)
    hip = np.asarray(Image.open(os.path.join(D, "all", name + ".png")).convert("RGB"), np.float64)
    ok = ~((ref >= 255).any(axis=2) | (hip >= 255).any(axis=2))
    d = hip - ref
    lr, lh = lin(ref).mean(axis=2), lin(hip).mean(axis=2)
    lit = ok & (lr > 0.02)
    r = {"render_s": timing[name]["render_s"], "Msamples_per_s": timing[name]["Msamples_per_s"], "mean_linear_ratio_hip_over_published": round(float(lh[lit].sum() / lr[lit].sum()), 4),
         "object_mean_linear_ratio": round(float(lh[lit & obj].sum() / lr[lit & obj].sum()), 4),
         "mean_abs_diff_8bit": round(float(np.abs(d[ok]).mean()), 2), "object_region_mean_abs_diff_8bit": round(float(np.abs(d[ok & obj]).mean()), 2),
         "backdrop_mean_abs_diff_8bit": round(float(np.abs(d[ok & ~obj]).mean()), 2), "share_within_8": round(float((np.abs(d[ok]) <= 8).all(axis=1).mean()), 4),
         "clipped_share": round(float(1 - ok.mean()), 4)}
    rows[name] = r
    print("%-20s %8.2f %8.4f %9.2f %9.2f %9.2f %8.4f %8.4f" % (name, r["render_s"], r["mean_linear_ratio_hip_over_published"], r["mean_abs_diff_8bit"], r["object_region_mean_abs_diff_8bit"],
                                                                 r["backdrop_mean_abs_diff_8bit"], r["share_within_8"], r["clipped_share"]))
json.dump(rows, open(os.path.join(D, "q1_all_vs_published.json"), "w"), indent=1)
v = lambda k: [r[k] for r in rows.values()]
print("all 22: render %.2f ... %.2f s; mean linear ratio %.4f ... %.4f; object-region ratio %.4f ... %.4f; object-region mean abs diff %.2f ... %.2f / 255" % (
    min(v("render_s")), max(v("render_s")), min(v("mean_linear_ratio_hip_over_published")), max(v("mean_linear_ratio_hip_over_published")),
    min(v("object_mean_linear_ratio")), max(v("object_mean_linear_ratio")), min(v("object_region_mean_abs_diff_8bit")), max(v("object_region_mean_abs_diff_8bit"))))
