"""Development probes for the GPU box (run through gpurun); not part of the product or of the test suite.

    python scripts/probe.py configs [c5]                 full renders of the BASELINE.json configs (time, Msamples/s, L2 vs oracle for C1/C2)
    python scripts/probe.py stages [--tune k=v,...] [--opts passes_in_flight=1,...] [--scene c4|c3|c1|c2] [--spp N]
                                                         per-stage device times of one pass + Msamples/s of a full call
    python scripts/probe.py sweep --knob refill --values 24,32,40,48,56 [--scene c4]
                                                         the same for a list of values of one KzTuning field (or pass_items / passes_in_flight)
    python scripts/probe.py counters --values 'keyStack=1;keyStack=2'
                                                         executed node visits / triangle tests / rays per sample for each tuning
    python scripts/probe.py ab --variants tree,NAME[,NAME2...] [--scenes c4,c3] [--reps 2] [--spp 256] [--tune ...] [--out DIR]
                                                         same-call A/B: `stages` in a child process per (repetition, build, scene); NAME = a build made by
                                                         scripts/build_variant.sh NAME <flags> (tree = the in-tree product). Lines go to gpurun_out/DIR/ab.txt
    python scripts/probe.py dealing [--scene c5] [--spp 256] [--reps 2]
                                                         kz_render_multi with static and with dynamic dealing (KzTileDealer) on the devices of this box, same process
    KZ_LIB_PATH=nano-kazen_amd/csrc/variants/lanestat/libkazen_mi355x.so python scripts/probe.py lanestat
                                                         where the lanes of the traversal loop are (needs scripts/build_variant.sh lanestat -DKZ_LANESTAT)
    KZ_LIB_PATH=nano-kazen_amd/csrc/variants/shadestat/libkazen_mi355x.so python scripts/probe.py shadestat
                                                         shares of the shade kernel's wave time per section (scripts/build_variant.sh shadestat -DKZ_SHADESTAT)
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
kz = importlib.import_module("nano-kazen_amd")
S = kz.scenes
OUT = os.path.join(ROOT, "gpurun_out")


def scene(name):
    return {"c1": lambda: S.cornell_box(256, 256, 16), "c2": lambda: S.sphere_env(512, 512, 64), "c3": lambda: S.hero_scene(1920, 1080, 256, detail=2.0),
            "c4": lambda: S.random_triangles(1000000, 1920, 1080, 1024), "c5": lambda: S.random_triangles(1000000, 3840, 2160, 4096),
            # outside BASELINE.json: the EXT kernel variants (rough / dielectric BSDFs; image textures + normal map) at C3's size, for one timing beside the lean C3
            "ext_materials": lambda: S.materials_scene(1920, 1080, 256), "ext_textured": lambda: S.textured_scene(1920, 1080, 256)}[name]()


def all_kiss(desc):
    """experiment: every non-emitting diffuse mesh gets a kiss row (does the shade kernel pay for BSDF-type divergence?)"""
    kiss = next(m["bsdf"] for m in desc.meshes if m["bsdf"] and m["bsdf"].get("type") == "kazenstandard")
    for m in desc.meshes:
        if m["bsdf"] and m["bsdf"].get("type") == "diffuse" and not m["light"]:
            m["bsdf"] = dict(kiss, baseColor=list(m["bsdf"].get("albedo", [0.7, 0.7, 0.7])))
    return desc


def kv(s, cast=int):
    return {k: cast(v) for k, v in (p.split("=") for p in s.split(",") if p)} if s else {}


def timed(sc, s0, s1, **kw):
    sc.render(s0, s1, **kw); sc.sync()
    t0 = time.perf_counter(); sc.render(s0, s1, **kw); sc.sync()
    return time.perf_counter() - t0


def cmd_configs(a):
    out = {}
    for name in ["c1", "c2", "c3", "c4"] + (["c5"] if a.c5 else []):
        desc = scene(name)
        t0 = time.time(); sc = kz.Scene(desc); tb = time.time() - t0
        sc.upload(0)
        dt = timed(sc, 0, 0)
        n = sc.width * sc.height * sc.sample_count
        rgb = sc.rgb()
        out[name] = {"width": sc.width, "height": sc.height, "spp": sc.sample_count, "tris": desc.n_tris(), "render_s": round(dt, 4), "Msamples_per_s": round(n / dt / 1e6, 1),
                     "scene_build_s": round(tb, 2), "image_mean": round(float(rgb.mean()), 5), "passes": sc.last_pass_info()}
        if name in ("c1", "c2"):
            import oracle as O
            o = O.OracleScene(desc); t0 = time.time(); c = o.rgb(o.render(threads=0)); tc = time.time() - t0
            out[name].update(oracle_s=round(tc, 2), oracle_Msamples_per_s=round(n / tc / 1e6, 3), l2_vs_oracle=float(np.sqrt(np.mean((rgb - c) ** 2))))
        print(name, out[name], flush=True)
        sc.close()
    json.dump(out, open(os.path.join(OUT, "configs.json"), "w"), indent=1)


def one(sc, spp, tune, opts):
    info = None
    dt = timed(sc, 0, spp, tune=tune, **opts)
    info = sc.last_pass_info()
    sc.render(0, info["sppPerPass"], tune=tune, **dict(opts, passes_in_flight=1)); sc.sync()
    st = sc.last_stage_ms()
    return {"Msamples_per_s": round(sc.width * sc.height * spp / dt / 1e6, 1), "call_ms": round(dt * 1e3, 2), "stages_one_pass_alone": st, "passes": info["passes"], "spp_per_pass": info["sppPerPass"]}


def cmd_stages(a):
    desc = scene(a.scene)
    if a.all_kiss:
        desc = all_kiss(desc)
    sc = kz.Scene(desc, device=0)
    spp = a.spp or min(sc.sample_count, 128)
    print(json.dumps({"scene": a.scene, "tune": kv(a.tune), "opts": kv(a.opts), **one(sc, spp, kv(a.tune), kv(a.opts))}), flush=True)


def cmd_sweep(a):
    sc = kz.Scene(scene(a.scene), device=0)
    spp = a.spp or min(sc.sample_count, 128)
    for v in a.values.split(","):
        tune, opts = kv(a.tune), kv(a.opts)
        (opts if a.knob in ("pass_items", "passes_in_flight", "max_state_bytes") else tune)[a.knob] = int(v)
        print(json.dumps({"knob": a.knob, "value": int(v), **one(sc, spp, tune, opts)}), flush=True)


def cmd_counters(a):
    """executed work per sample (node visits, triangle tests, rays) for a list of tunings: --values 'keyStack=1;keyStack=2'"""
    sc = kz.Scene(scene(a.scene), device=0)
    sc.set_stats(True)
    spp = a.spp or 16
    for t in (a.values.split(";") if a.values else [""]):
        sc.stats(reset=True)
        sc.render(0, spp, passes_in_flight=1, tune=kv(t)); sc.sync()
        st = sc.stats(reset=True)
        n = max(1, st["samples"])
        print(json.dumps({"tune": t, **{k: round(v / n, 3) for k, v in st.items() if k != "samples"}}), flush=True)


def cmd_lanestat(a):
    sc = kz.Scene(scene(a.scene), device=0)
    sc.set_stats(True)
    spp = a.spp or 16
    sc.render(0, spp, passes_in_flight=1, tune=kv(a.tune)); sc.sync()
    print(json.dumps(sc.stats(reset=True)), flush=True)          # the library prints the lane statistics on stderr


def cmd_shadestat(a):
    """KZ_LIB_PATH=.../variants/shadestat/... : shares of the shade kernel's wave time per section (scripts/build_variant.sh shadestat -DKZ_SHADESTAT)"""
    sc = kz.Scene(scene(a.scene), device=0)
    spp = a.spp or 64
    sc.stats(reset=True)
    sc.render(0, spp, passes_in_flight=1, tune=kv(a.tune)); sc.sync()
    print(json.dumps(sc.stats(reset=True)), flush=True)          # the library prints the section shares on stderr


def cmd_sched(a):
    """the pass schedule: passes in flight x pass shape (sppPerPass) x pass size on one scene; every case renders the same --spp samples
    of every pixel twice (warm + timed). --values 'n,sppPerPass,log2(passItems);...'"""
    sc = kz.Scene(scene(a.scene), device=0)
    spp = a.spp or min(sc.sample_count, 512)
    ref = None
    for case in a.values.split(";"):
        n, sp, lg = (int(x) for x in case.split(","))
        kw = dict(passes_in_flight=n, pass_items=1 << lg, tune=dict(kv(a.tune), sppPerPass=sp))
        dt = timed(sc, 0, spp, **kw)
        info = sc.last_pass_info()
        film = sc.film()
        if ref is None:
            ref = film
        print(json.dumps({"in_flight": n, "sppPerPass": sp, "log2_items": lg, "Msamples_per_s": round(sc.width * sc.height * spp / dt / 1e6, 1), "call_ms": round(dt * 1e3, 1),
                          "passes": info["passes"], "ctx": info["passesInFlight"], "spp_per_pass": info["sppPerPass"], "pix_per_pass": info["pixelsPerPass"],
                          "state_GB": round(info["stateBytes"] / 2**30, 1), "film_equal_first": bool(np.array_equal(film, ref)),
                          "film_maxrel": float(np.max(np.abs(film - ref) / np.maximum(np.abs(ref), 1e-3)))}), flush=True)


def cmd_dealing(a):
    """static vs dynamic dealing of one fixed job on the devices of this box: kz_render_multi on C5 (or --scene) x --spp samples,
    each form warmed once and timed --reps times, same process. Dynamic dealing = one kz_render_tiles call per device with a KzTileDealer."""
    desc = scene(a.scene)
    sc = kz.Scene(desc)
    ndev = kz.abi.load_library().kz_device_count()
    devs = list(range(min(ndev, 8)))
    spp = a.spp or 256
    n = sc.width * sc.height * spp
    out = {"scene": a.scene, "spp": spp, "devices": devs}
    ref = None
    for name, kw in (("static", {}), ("dynamic", {"tile_dealing": 1}), ("static_again", {})):
        sc.render_multi(devs, sample_begin=0, sample_end=spp, **kw)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); film, ms = sc.render_multi(devs, sample_begin=0, sample_end=spp, **kw); ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = film
        out[name] = {"s": [round(t, 4) for t in ts], "Msamples_per_s": round(n / min(ts) / 1e6, 1), "device_ms": [round(float(x), 1) for x in ms],
                     "max_abs_diff_vs_static": float(np.max(np.abs(film - ref)))}
        print(name, out[name], flush=True)
    out["dynamic_over_static"] = round(min(out["dynamic"]["s"]) / min(min(out["static"]["s"]), min(out["static_again"]["s"])), 4)
    print(json.dumps(out), flush=True)
    # one device, the dealer driven directly: batch sizes (tiles) and pass shapes
    tiles = kz.shard.deal_tiles(sc.width, sc.height, 1, 0, 64)
    counter = np.zeros(1, np.uint32)
    for case in (a.values.split(";") if a.values else []):
        bt, spass = (int(x) for x in case.split(","))
        kw = {"tune": {"sppPerPass": spass}} if spass else {}
        ts = []
        for _ in range(a.reps + 1):
            counter[0] = 0
            t0 = time.perf_counter()
            if bt >= 0:
                took = sc.render_dealt(tiles, counter, takers=1, batch_tiles=bt, device=0, sample_begin=0, sample_end=spp, **kw)
            else:
                sc.render_tiles(tiles, device=0, sample_begin=0, sample_end=spp, download=False, **kw)
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"batch_tiles": bt, "sppPerPass": spass, "s": [round(t, 4) for t in ts[1:]], "Msamples_per_s": round(n / min(ts[1:]) / 1e6, 1), "passes": sc.last_pass_info()}), flush=True)


def cmd_ab(a):
    """Box-to-box variance is 3-6 %, so only numbers from ONE gpurun call compare: alternate the builds, each in its own child process."""
    import subprocess
    out = os.path.join(OUT, a.out or "ab"); os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ab.txt"), "a") as f:
        for rep in range(a.reps):
            for lib in a.variants.split(","):
                for sc in a.scenes.split(","):
                    env = dict(os.environ)
                    env.pop("KZ_LIB_PATH", None)
                    if lib != "tree":
                        env["KZ_LIB_PATH"] = os.path.join(ROOT, "nano-kazen_amd", "csrc", "variants", lib, "libkazen_mi355x.so")
                    cmd = [sys.executable, os.path.abspath(__file__), "stages", "--scene", sc, "--spp", str(a.spp or 256), "--tune", a.tune, "--opts", a.opts]
                    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                    line = "%s %s %s" % (lib, sc, (r.stdout.strip().splitlines() or ["FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])])[-1])
                    print(line, flush=True); f.write(line + "\n"); f.flush()


ap = argparse.ArgumentParser()
ap.add_argument("cmd", choices=["configs", "stages", "sweep", "lanestat", "counters", "shadestat", "sched", "ab", "dealing"])
ap.add_argument("--variants", default="tree"); ap.add_argument("--scenes", default="c4,c3"); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--out", default="")
ap.add_argument("c5", nargs="?")
ap.add_argument("--scene", default="c4"); ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--tune", default=""); ap.add_argument("--opts", default="")
ap.add_argument("--all-kiss", action="store_true"); ap.add_argument("--knob", default="refill"); ap.add_argument("--values", default="")
a = ap.parse_args()
{"configs": cmd_configs, "stages": cmd_stages, "sweep": cmd_sweep, "lanestat": cmd_lanestat, "counters": cmd_counters, "shadestat": cmd_shadestat, "sched": cmd_sched, "ab": cmd_ab, "dealing": cmd_dealing}[a.cmd](a)
