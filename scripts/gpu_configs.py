"""Full renders of the BASELINE.json configs on one MI355X (wall time of kz_render + sync, film download excluded)."""
import sys, time, json, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
out = {}
def run(name, desc, check=None):
    t0 = time.time(); sc = kz.Scene(desc); tb = time.time() - t0
    t0 = time.time(); sc.upload(0); tu = time.time() - t0
    sc.render(); sc.sync()                                      # warm-up: the path-state buffers are sized by the first full call
    t0 = time.time(); sc.render(); sc.sync(); dt = time.time() - t0
    n = sc.width * sc.height * sc.sample_count
    rgb = sc.rgb()
    out[name] = {"width": sc.width, "height": sc.height, "spp": sc.sample_count, "tris": desc.n_tris(), "render_s": round(dt, 4),
                 "Msamples_per_s": round(n / dt / 1e6, 1), "scene_build_s": round(tb, 2), "upload_s": round(tu, 3), "image_mean": round(float(rgb.mean()), 5)}
    print(name, out[name], flush=True)
    if check:
        import oracle as O
        o = O.OracleScene(desc); t0 = time.time(); c = o.rgb(o.render(threads=0)); tc = time.time() - t0
        out[name]["oracle_s"] = round(tc, 2); out[name]["oracle_Msamples_per_s"] = round(n / tc / 1e6, 3)
        out[name]["l2_vs_oracle"] = float(np.sqrt(np.mean((rgb - c) ** 2)))
        print("   oracle %.1fs L2 %.2e" % (tc, out[name]["l2_vs_oracle"]), flush=True)
    try:
        from PIL import Image
        x = np.clip(rgb, 0, 1); x = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
        im = Image.fromarray((x * 255 + .5).astype(np.uint8)); im.thumbnail((480, 480)); im.save("/root/repo/gpurun_out/img_%s.png" % name)
    except Exception as e:
        print("no image", e)
    sc.close()
run("C1_cornell_256x256x16_independent", S.cornell_box(256, 256, 16), check=True)
run("C2_sphere_env_512x512x64", S.sphere_env(512, 512, 64), check=True)
run("C3_hero_1920x1080x256_kiss", S.hero_scene(1920, 1080, 256, detail=2.0))
run("C4_1Mtri_1920x1080x1024_pmj02bn", S.random_triangles(1000000, 1920, 1080, 1024))
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    run("C5_1Mtri_3840x2160x4096_pmj02bn_1gpu", S.random_triangles(1000000, 3840, 2160, 4096))
json.dump(out, open("/root/repo/gpurun_out/configs.json", "w"), indent=1)
