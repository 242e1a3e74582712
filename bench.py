#!/usr/bin/env python3
"""bench.py — Msamples/s of the path_mis hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--strong]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], "C4"): 1 M random triangles + 8 mesh lights in a closed diffuse room,
1920x1080, pmj02bn sampler (1024 spp), path_mis maxDepth 5. A "step" is one kz_render call over one batch: a 128-spp
slice of every pixel a GPU owns = 265 M (pixel, sample) paths per GPU per step, which the library runs as two 133 M-path
passes (its default pass size, 2^27 items) kept in flight together on two internal streams.

N GPUs (north_star: "the image tile grid shards embarrassingly across the GPUs; no RCCL needed; host gathers
tiles"): one process per GPU, each with a full scene replica; the 64x64 tiles of the frame are dealt over the ranks by area
(kz_deal_tiles, the dealing kz_render_multi uses in-process) and each rank renders 128*N spp of ITS tiles per step, so
per-GPU work is fixed (weak scaling). There is NO device collective anywhere: torch.distributed runs on the CPU (gloo) for the
barriers around the timed region and for the one host gather after it: every rank downloads the packed film rects of ITS tiles
(kz_film_download_tiles: one film's worth of texels in all, however many ranks) and rank 0 adds them in tile order
(shard.gather_tiles = ImageBlock::put(ImageBlock&), block.cpp:87-96). `value` is the render rate of the timed steps; `end_to_end`
adds that gather (download + gloo + merge, max over ranks) to the same steps' time. Scene tables, BVH and sampler tables are resident
in HBM before the timed region starts.

--strong: BASELINE.json configs[4], "C5" - the SAME scene at 3840x2160 x 4096 spp as ONE fixed job: a step is the whole frame, every rank
renders all 4096 spp of its tiles (scaling "strong": the work per GPU shrinks with N). Not the default: at N = 1 the driver's run must be C4.

`cold_job` (N = 1): the reference renders ONE frame per process (renderer.cpp:72-153). Before this process touches the GPU, two CHILD
processes each do exactly that job and report its wall time - kz_scene_upload of a freshly built scene, the WHOLE frame (C4: 1920x1080 x 1024 spp; the
reference's own scene file default_m0_r0.5 at its 4096 spp), kz_film_download - every allocation, the beam lists and the growth of the pass context included.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); 6.29 TB/s measured copy
SPP_PER_RANK_STEP = 512        # one pass of 2^30 (pixel, sample) items per call at 1920x1080: the library's default pass size on a 288 GB card (round 4; 128 = two passes of 2^27 before)
W, H, NTRIS, SPP = 1920, 1080, 1000000, 1024
W5, H5, SPP5 = 3840, 2160, 4096
TILE = 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes_per_sample(st):
    """64 B node packets, 48 B leaf triangles, 96 B shading gathers, 64 B light samples, +16 B film."""
    s = max(1, st["samples"])
    return (64.0 * st["nodeVisits"] + 48.0 * st["triTests"] + 96.0 * st["shadedHits"] + 64.0 * st["lightSamples"]) / s + 16.0


def git_head():
    """The commit of the sources: git where there is a repository, else the VERSION file __graft_entry__.build() wrote beside the library (the GPU box has no .git)."""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "VERSION")).read().strip() or None
    except Exception:
        return None


def source_hash():
    """sha256 (first 16 hex digits) of the kernel and ABI sources and of the build recipe (compiler flags): the counter facts of a profile describe ONE build"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "nano-kazen_amd", "csrc", "build.sh"), os.path.join(ROOT, "include", "kazen_mi355x.h"), os.path.join(ROOT, "include", "kazen_mi355x_dev.h")]):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def load_profile_facts(live_ms_per_pass_alone):
    """Counter-derived facts come from a committed rocprofv3 --pmc run of THIS command (profiles/pmc_latest.json, written by
    scripts/summarize_profile.py with the commit, the profile directory and a hash of the kernel sources). They describe one
    build, not this run: they are printed only when the sources are the profiled ones AND the live per-pass time agrees with the
    profiled build's to 5 %, otherwise they are withheld (null + the reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_latest.json"
    try:
        facts = json.load(open(path))
    except Exception as e:
        return None, "unreadable: %s" % e
    if facts.get("source_sha16") != source_hash():
        return None, "stale: profile %s (commit %s) was taken from other kernel sources (%s, now %s)" % (
            facts.get("profile"), facts.get("commit"), facts.get("source_sha16"), source_hash())
    ref = facts.get("path_kernels_ms_one_pass_alone")
    if not ref or abs(live_ms_per_pass_alone - ref) > 0.05 * ref:
        return None, "stale: profile %s (commit %s) measured %.2f ms per pass alone, this run %.2f ms" % (
            facts.get("profile"), facts.get("commit"), ref or 0.0, live_ms_per_pass_alone)
    return facts, None


def cold_job(which, tris):
    """One frame in a fresh process, as kazen::renderer::render is used (renderer.cpp:72-153: one render per process): prints one JSON object. The scene is
    built on the host first (kz_scene_create: BVH build, not part of the metric); the clock covers kz_scene_upload -> kz_render of every sample
    of every pixel -> kz_film_download, i.e. the table upload, the beam lists, every device allocation and the pass context growing while the first passes run."""
    import numpy as np
    kz = importlib.import_module("nano-kazen_amd")
    t0 = time.perf_counter()
    if which == "c4":
        desc = kz.scenes.random_triangles(tris, W, H, SPP, sampler="pmj02bn", seed=1)
        name = "C4: %d random triangles + 8 mesh lights, %dx%d, pmj02bn, all %d spp" % (tris, W, H, SPP)
    else:
        desc = kz.scenes.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"))
        name = "scene/2022_q1/parameters/default_m0_r0.5.xml (36 378 triangles, via tests/golden/q1_default_m0_r0.5.npz), 1920x1080, independent sampler, all of the file's spp"
    scene = kz.Scene(desc)
    t_build = time.perf_counter() - t0
    if os.environ.get("KZ_BENCH_TRACE"):
        getattr(scene.lib, "kz_debug_trace", lambda on: None)(1)      # (development builds of the library, KZ_LIB_PATH: the allocation / growth / pass timeline on stderr)
    t0 = time.perf_counter()
    n_dev = scene.lib.kz_device_count()                     # the first HIP call of the process: runtime start-up, a property of the process and reported beside the job
    if n_dev <= 0:
        raise SystemExit("cold job: no GPU")
    t_init = time.perf_counter() - t0
    free0 = C_mem_info(scene.lib)
    t0 = time.perf_counter()
    scene.upload(0)
    t1 = time.perf_counter()
    scene.render()                                         # every sample of every pixel, library defaults
    scene.sync()
    t2 = time.perf_counter()
    film = scene.film()
    t3 = time.perf_counter()
    info = scene.last_pass_info()
    n = scene.width * scene.height * scene.sample_count
    rec = {"workload": name, "job": "fresh process: kz_scene_upload -> kz_render (whole frame, every sample) -> kz_film_download; wall clock, every allocation, the beam lists and "
                                    "the growth of the pass context included; the host BVH build and the HIP runtime start-up are reported beside it",
           "value": round(n / (t3 - t0) / 1e6, 1), "unit": "Msamples/s", "samples": n, "seconds": round(t3 - t0, 4),
           "upload_s": round(t1 - t0, 4), "render_s": round(t2 - t1, 4), "download_s": round(t3 - t2, 4),
           "scene_build_s": round(t_build, 2), "hip_runtime_init_s": round(t_init, 3), "value_with_runtime_init": round(n / (t3 - t0 + t_init) / 1e6, 1),
           "passes": info["passes"], "first_pass_items": info["firstPassItems"], "largest_pass_items": info["largestPassItems"], "target_pass_items": info["itemsPerPass"],
           "state_gb": round(info["stateBytes"] / 1e9, 1), "device_free_gb_at_start": round(free0 / 1e9, 1), "grow_note": scene.last_grow_note(),
           "image_mean": round(float(scene.rgb(film).mean()), 5)}
    print(json.dumps(rec), flush=True)
    os._exit(0)                                            # (the job is done: what the process still holds goes back to the driver at exit, as the reference's does)


def C_mem_info(lib):
    import ctypes
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    return f.value if lib.kz_device_mem_info(0, ctypes.byref(f), ctypes.byref(t)) == 0 else 0


def profiler_attached():
    """True under rocprofv3 / rocprof: the profiler's preloaded library initialises the GPU before this program starts (with --pmc it does), and a process that
    holds the GPU must not start child processes on this pool."""
    if any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_")) for k in os.environ):
        return True
    return any(w in os.environ.get("LD_PRELOAD", "") for w in ("rocprofiler", "rocprof", "roctracer"))


def film_crc(rgbw_film, border, x0=928, y0=508, n=64):
    """crc32 of a fixed n x n crop of a film (float32 bits, rgb * w and w): since round 6 the film does not depend on how the work was cut - pass size, number of
    ranks, static or dynamic dealing - so the SAME value must appear at every N."""
    import zlib
    import numpy as np
    c = np.ascontiguousarray(rgbw_film[y0 + border:y0 + border + n, x0 + border:x0 + border + n])
    return "%08x" % (zlib.crc32(c.tobytes()) & 0xFFFFFFFF)


# film_crc of the parity render below (C4, sample indices [0, 64), crop 64 x 64 at (928, 508)) on ONE GPU: the value every N must reproduce.
# (measured at N = 1 on the GPU box in round 6, profiles/r06a_film/bench_n1.json; tests/test_gpu_multi.py::test_bench_parity_crc_is_independent_of_the_cut
# reproduces it through 2, 4 and 8 aliased replicas and through two ranks)
PARITY_CRC_N1 = "431bcc3e"
PARITY_SPP_TABLE = 1024        # the parity render's sampler table: pmj02bn's permutations depend on the table length, so every N renders THIS scene


def run_cold_jobs(tris):
    """The cold jobs as child processes, BEFORE this process initialises HIP (no fork from a process that holds the GPU). Each job is measured on a QUIET GPU:
    the driver wipes what a process releases at ~33 GB/s, and a process that starts inside that window waits for the whole wipe in its first larger
    allocation, its GPU work with it (profiles/r05a_alloc) - so bench.py waits `QUIET_S` seconds after a child before it starts the next. What a job started
    right BEHIND another one's exit costs is measured too, once (`c4_behind_a_release`), and reported beside the quiet numbers."""
    QUIET_S = 6.0
    out = {"quiet_seconds_between_jobs": QUIET_S}

    def child(which):
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-job", which, "--tris", str(tris)], capture_output=True, text=True, timeout=300)
            if os.environ.get("KZ_BENCH_TRACE"):
                log(r.stderr)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            rec = json.loads(line[-1]) if line else {"error": "no record (exit code %d): %s" % (r.returncode, r.stderr[-300:])}
        except Exception as e:                                 # noqa: BLE001 - a failed side job must not cost the headline record
            rec = {"error": repr(e)}
        rec["process_s"] = round(time.time() - t0, 1)
        return rec

    time.sleep(QUIET_S)                                        # (whatever ran on this GPU before bench.py)
    # q1 first: the FIRST process on a freshly leased box pays ~0.13 s once in its first host-to-device copies (profiles/r05d_cold_job: upload 0.18 s
    # instead of 0.05) - a property of the box, not of the job; the 4.4 s q1 job absorbs it (3 %), the 1.3 s C4 job would carry it as 10 %.
    out["order"] = ["q1", "c4", "c4_behind_a_release"]
    out["q1"] = child("q1")
    log("cold job q1: %s" % json.dumps(out["q1"]))
    time.sleep(QUIET_S)
    out["c4"] = child("c4")
    log("cold job c4: %s" % json.dumps(out["c4"]))
    behind = child("c4")                                       # at once: the C4 job's pass context (47 GB) is being wiped
    behind["job"] = "the C4 job again, started the moment the previous C4 job's process has exited (no quiet time): what the driver's wipe of the predecessor's memory costs a job"
    out["c4_behind_a_release"] = behind
    log("cold job c4 behind a release: %s" % json.dumps(behind))
    time.sleep(QUIET_S)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--tris", type=int, default=NTRIS)
    ap.add_argument("--strong", action="store_true", help="C5: 3840x2160 x 4096 spp as one fixed job split over the ranks (a step = the frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-c5", action="store_true", help="N > 1: skip the bounded C5 strong-scaling job that rides in the same JSON line")
    ap.add_argument("--strong-spp", type=int, default=256, help="samples per pixel of that job (sample indices [0, n) of the 4096-spp table)")
    ap.add_argument("--no-asset-scene", action="store_true", help="N = 1: skip the slice of the reference's own scene file that rides in the same JSON line")
    ap.add_argument("--asset-spp", type=int, default=512, help="samples per pixel of that slice (sample indices [0, n) of the scene file's 4096)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the oracle baseline sample")
    ap.add_argument("--cold-job", choices=["c4", "q1"], default=None, help="(child process mode) run ONE cold job and print its record")
    ap.add_argument("--no-cold-job", action="store_true", help="N = 1: skip the two one-frame-per-process jobs that ride in the same JSON line")
    ap.add_argument("--shadow-beside", type=int, default=0, help="KzRenderOpts::shadowBeside of the timed steps (0 = library default, 1 = one stream, 2 = shadow rays beside the closest-hit rays: scripts/r06_beside_bench.sh)")
    ap.add_argument("--pass-halves", type=int, default=0, help="KzRenderOpts::passHalves of the timed steps (0 = library default, 1 = never, 2 = every pass as two halves side by side)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity render (scripts/profile_bench.sh: the kernel statistics of a profiled run then hold the timed passes only)")
    ap.add_argument("--no-ext-scenes", action="store_true", help="N = 1: skip the EXT-kernel scenes and C1 / C2 / C3 at their BASELINE sizes (each beside the oracle's CPU time) that ride in the same JSON line")
    ap.add_argument("--profile-pass", action="store_true", help="(scripts/profile_bench.sh) every step asks for ONE pass of 2^30 items explicitly instead of earning it call by call: "
                                                                "the counter runs profile the first step of a process, and the pass they see must be the steady state's")
    args = ap.parse_args()
    if args.cold_job:
        return cold_job(args.cold_job, args.tris)
    if args.steps is None:
        args.steps = 1 if args.strong else 6
    if args.warmup is None:
        args.warmup = 0 if args.strong else 4          # (the default pass size is earned call by call: 2^27 items, then doubling - the fourth call runs passes of 2^30)
    commit = os.environ.get("GRAFT_HEAD") or git_head()          # before torch / HIP are loaded: no fork from a process that has initialised the GPU
    cold = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.strong and not args.no_cold_job:
        if profiler_attached():
            cold = {"skipped": "a profiler is attached (ROCPROFILER_* / ROCP_* in the environment or a rocprofiler preload): its library has initialised the GPU before "
                               "bench.py started, and a process that holds the GPU must not start child processes - pass --no-cold-job"}
            log("cold jobs skipped: %s" % cold["skipped"])
        else:
            cold = run_cold_jobs(args.tris)                      # (child processes; this one has not touched the GPU yet)

    # stdout carries ONE line, the JSON record: everything else that may write to file descriptor 1 (gloo's "[Gloo] Rank ... is
    # connected" banner comes from C++ and lands on stdout) is sent to stderr; the record goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal of N ranks on a 1-GPU box: KZ_BENCH_DEVICE=0 puts every rank on GPU 0 (each rank then caps its path state)
    device_index = int(os.environ.get("KZ_BENCH_DEVICE", local_rank))
    shared_device = "KZ_BENCH_DEVICE" in os.environ and world > 1
    torch.cuda.set_device(device_index)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)        # CPU process group: barriers + the host gather

    kz = importlib.import_module("nano-kazen_amd")
    # weak scaling: a rank owns 1/N of the pixels and renders N x 512 spp of them per step, so that every rank runs the SAME pass as N = 1 (one pass of 2^30
    # items). From N = 4 on that is more than C4's 1024-entry pmj02bn table holds; the table grows with it (N = 4: 2048, N = 8: 4096 entries - the table of
    # BASELINE's own 8-GPU config) instead of wrapping, which used to cut a rank's step into calls of 2^28 items (3 % slower per GPU: VERDICT r04)
    Wd, Hd, spp_table = (W5, H5, SPP5) if args.strong else (W, H, max(SPP, SPP_PER_RANK_STEP * world))
    t0 = time.time()
    desc = kz.scenes.random_triangles(args.tris, Wd, Hd, spp_table, sampler="pmj02bn", seed=1)
    t1 = time.time()
    scene = kz.Scene(desc)
    bvh = scene.bvh_info()
    scene.upload(device_index)
    t2 = time.time()
    if rank == 0:
        log("scene: %d tris, synth %.1fs, BVH %d nodes depth %d SAH %.1f built in %.2fs, upload+build %.1fs"
            % (desc.n_tris(), t1 - t0, bvh["nNodes"], bvh["maxDepth"], bvh["sahCost"], bvh["buildSeconds"], t2 - t1))

    tiles = kz.shard.deal_tiles(Wd, Hd, world, rank, TILE)
    render_tiles = tiles if world > 1 else None           # (one rank: the whole frame as one tile set - the same pixels, no list to walk)
    my_pixels = sum(t[2] * t[3] for t in tiles)
    if args.strong:
        spp_step = spp_table                              # the whole job every step
    else:
        spp_step = SPP_PER_RANK_STEP * world              # (<= the table: a step is ONE call)
    stream = torch.cuda.current_stream().cuda_stream
    kw = {}
    if args.shadow_beside:
        kw["shadow_beside"] = args.shadow_beside
    if args.pass_halves:
        kw["pass_halves"] = args.pass_halves
    if shared_device:
        kw["max_state_bytes"] = int(0.8 * torch.cuda.mem_get_info(device_index)[1] / world)
    if args.profile_pass:
        kw.update(pass_items=1 << 30, passes_in_flight=1)

    def step(k, accumulate=True):
        # sample indices [s0, s0 + spp_step) modulo the table: a slice that wraps is rendered piece by piece
        s0, left = (k * spp_step) % spp_table, spp_step
        while left > 0:
            n = min(left, spp_table - s0)
            scene.render(s0, s0 + n, tiles=render_tiles, accumulate=accumulate, stream=stream, **kw)
            accumulate, left, s0 = True, left - n, (s0 + n) % spp_table

    def barrier():
        # device first, then the ranks, then the device again: a rank must not start (or stop) its clock while its own earlier kernels still
        # run (with one rank per GPU the order is immaterial; with several ranks rehearsing on ONE GPU it is not: profiles/r03b_multiprocess)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scene.film_clear(stream)
    first_call_ms = None
    grow_wait_s = 0.0
    for k in range(args.warmup):
        t_w = time.perf_counter()
        step(k)
        torch.cuda.synchronize()
        if k == 0:                                        # what the timed region leaves out: the first call builds the beam lists of the rank's pixels and allocates the pass contexts
            first_call_ms = round(1e3 * (time.perf_counter() - t_w), 2)
        # Warm-up means "until the steady state": the pass context earns its size call by call (2^27 items, then doubling) and the library maps it on a side thread - at
        # ~33 GB/s on memory the driver still has to wipe when the GPU is idle, an eighth of that under a render (profiles/r05d_cold_job). bench.py gives it the idle
        # GPU between warm-up steps (bounded), so that the timed steps run on the context a process that keeps rendering ends up with.
        t_g = time.perf_counter()
        while time.perf_counter() - t_g < 8.0:
            i_ = scene.last_pass_info()
            if i_["contextItems"] >= i_["itemsPerPass"] or scene.last_grow_note():
                break
            time.sleep(0.05)
        grow_wait_s += time.perf_counter() - t_g
    # ... and "until the steady state" also means: until the replica has settled HOW its large passes run (KzRenderOpts::shadowBeside / passHalves at 0: it times four
    # passes of one size - one stream, shadow rays beside, halves, one stream - and keeps the fastest). With the pass size earned call by call those four land behind the
    # W warm-up steps; a process that keeps rendering leaves them behind once, so they are left behind here too (bounded; reported as `warmup_steps_until_settled`).
    settle = 0
    while args.warmup and not args.strong and settle < 6:
        m_ = scene.pass_mode_info()
        if m_["kept"] is not None or m_["timed_passes"] == 0:
            break
        step(args.warmup + settle)
        torch.cuda.synchronize()
        settle += 1
    if world > 1 and args.warmup:                         # the gather's one-time costs (pinned staging buffer, /dev/shm pages) belong to the warm-up too
        kz.shard.gather_tiles(scene, tiles, scene.film_tiles(tiles), rank, world)
    barrier()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms_last = scene.last_kernel_ms()      # hipEvents on the launch stream around the last call / its passes
    info = scene.last_pass_info()
    if scene.last_grow_note():
        log("rank %d: the pass context stopped growing: %s" % (rank, scene.last_grow_note()))
    # the gather, once per render: packed rects of the rank's tiles (D2H through pinned memory), gloo to rank 0, merge in tile order
    t_g = time.perf_counter()
    if world > 1:
        packed = scene.film_tiles(tiles)
        t_d = time.perf_counter()
        film_np = kz.shard.gather_tiles(scene, tiles, packed, rank, world)
    else:                                                 # one device: its film IS the frame
        film_np = packed = scene.film()
        t_d = time.perf_counter()
    gather_s = time.perf_counter() - t_g
    free_b, total_b = torch.cuda.mem_get_info(device_index)
    # audit line per rank: did N distinct devices render? was a rank capped by its state budget?
    log("rank %d/%d: device %d (%s), %d tiles = %d px, %.3f s for %d steps, passes %s, free %.1f of %.1f GB, download %.1f MB in %.1f ms, gather %.1f ms"
        % (rank, world, device_index, torch.cuda.get_device_name(device_index), len(tiles), my_pixels, elapsed, args.steps, info, free_b / 2**30, total_b / 2**30,
           packed.nbytes / 1e6, 1e3 * (t_d - t_g), 1e3 * gather_s))

    # ---- parity render (every rank; outside the timed region): sample indices [0, 64) of the frame, tiles dealt as above, gathered on rank 0. The film does not
    # depend on N, on the dealing or on the pass size (round 6: per-pixel running tap sums, rects merged in tile order), so the crc32 of a fixed crop is the same at
    # every N - the scaling run carries its own parity check.
    parity = None
    if not args.strong and not args.no_parity:
        # (weak scaling lengthens the sampler table from N = 4 on: those ranks build the 1024-entry scene for this render - outside every clock)
        psc = scene if spp_table == PARITY_SPP_TABLE else kz.Scene(kz.scenes.random_triangles(args.tris, Wd, Hd, PARITY_SPP_TABLE, sampler="pmj02bn", seed=1), device=device_index)
        psc.render(0, 64, tiles=render_tiles, accumulate=False, stream=stream, **kw)
        torch.cuda.synchronize()
        pf = kz.shard.gather_tiles(psc, tiles, psc.film_tiles(tiles), rank, world) if world > 1 else psc.film()
        if psc is not scene:
            psc.close()
        if rank == 0:
            parity = {"film_crc": film_crc(pf, scene.border), "film_crc_n1": PARITY_CRC_N1, "render": "C4 (%d triangles, %d-entry pmj02bn table) sample indices [0, 64) of every pixel, %d rank(s), "
                      "64 x 64 tiles; crc32 of the float32 film texels (rgb * w, w) of the 64 x 64 crop at (928, 508)" % (args.tris, PARITY_SPP_TABLE, world)}
            parity["equals_n1"] = (parity["film_crc"] == PARITY_CRC_N1) if args.tris == NTRIS else None
            log("parity render: %s" % json.dumps(parity))
        scene.film_clear(stream)

    el = torch.tensor([elapsed, gather_s, float(info["largestPassItems"]), -float(info["largestPassItems"])], dtype=torch.float64)
    px = torch.tensor([my_pixels], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(px, op=dist.ReduceOp.SUM)
    elapsed, gather_max = float(el[0].item()), float(el[1].item())
    items_per_pass_per_rank = [int(-el[3].item()), int(el[2].item())]          # [min, max] over the ranks: the pass every rank really ran in the last timed step
    total_samples = float(px.item()) * spp_step * args.steps
    value = total_samples / elapsed / 1e6

    if rank == 0:
        # ---- one pass run alone: per-stage device times (with two passes in flight the stage events of a pass overlap the other's)
        pass_spp = info["sppPerPass"]
        scene.render(0, pass_spp, tiles=render_tiles, accumulate=True, stream=stream, **dict(kw, shadow_beside=1))      # (one stream: the stage clock shows the shadow and the closest-hit kernels apart)
        scene.sync()
        stage_ms = scene.last_stage_ms()
        alone_ms = scene.last_kernel_ms()
        launch_samples = info["pixelsPerPass"] * pass_spp            # one pass: the unit of kernel_ms, traffic and the counter facts
        # ---- counters of the executed (wavefront) and of the reference-shaped (megakernel) pipelines on one slice
        scene.set_stats(True)
        s1 = min(spp_table, 128)
        scene.stats(reset=True)
        scene.render(0, s1, tiles=render_tiles, accumulate=True, stream=stream, **kw)
        scene.sync()
        st_exec = scene.stats(reset=True)
        scene.render(0, min(s1, 16 if args.strong else s1), tiles=render_tiles, accumulate=True, pipeline=1, stream=stream, **kw)      # reference-shaped
        scene.sync()
        st_ref = scene.stats(reset=True)
        scene.set_stats(False)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(desc, args.cpu_seconds, Wd, Hd, scene)
        bps_gpu_ref = algorithmic_bytes_per_sample(st_ref)
        bps = cpu["bytes_per_sample_oracle"] if cpu else bps_gpu_ref          # SURVEY 8d takes the counts from the CPU oracle
        bps_exec = algorithmic_bytes_per_sample(st_exec)
        hbm_algorithmic = bps * launch_samples / (kernel_ms_last * 1e-3) / 1e9
        # the datasheet peak beside a measured device-to-device stream copy (read + write bytes / time)
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda:%d" % device_index)      # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = round(8 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b

        # ---- roofline of the dominant kernel. The path is NOT HBM-bound (the 174 MB of tables sit in L2 / Infinity Cache and the
        # traversal is a VALU-issue problem): the roof is the VALU issue rate of the chip. `peak` is the guide's
        # (MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD-32) at the in-kernel clock scripts/micro/valu_clock.hip
        # measures under a saturated VALU stream; the rates that microbenchmark actually reaches (two-source VOP2 91 %, three-source VOP3
        # 49-55 % of it) are printed beside it. `achieved` = VALU wave-instructions the kz_wf_trace<0> launches of one pass execute
        # (a property of code + input, counted by rocprofv3 SQ_INSTS_VALU in the committed profile) / their LIVE device time
        # (hipEvents, one pass alone).
        facts, why = load_profile_facts(alone_ms)
        peak = json.load(open(os.path.join(ROOT, "profiles", "valu_peak.json")))
        peak_rate = peak["guide_peak_wave_instr_per_s"] / 1e9
        n_bounce = max(0, desc.integrator["maxDepth"] - 1)
        roofline = {"bound": "valu", "unit": "G wave-instr/s", "peak": round(peak_rate, 1),
                    "peak_source": "0.5 wave64 VALU instructions per cycle per SIMD (MI355X_MICROARCH.md) x 1024 SIMDs x the in-kernel clock measured under a "
                                   "saturated VALU stream (profiles/%s)" % peak["profile"],
                    "peak_measured": {"full_rate_class": round(peak["full_rate_class_wave_instr_per_s"] / 1e9, 1), "half_rate_class": round(peak["half_rate_class_wave_instr_per_s"] / 1e9, 1),
                                      "transcendental": round(peak["transcendental_wave_instr_per_s"] / 1e9, 1),
                                      "node_step_mix": peak.get("node_mix_wave_instr_per_s") and round(peak["node_mix_wave_instr_per_s"] / 1e9, 1),
                                      "classes": "mul/add/sub/fma f32, mov, and/or/xor, lshr/ashr, add/sub u32 issue in 2.2 cycles; min/max, cmp, cndmask, cvt, bfe/perm/lshl and 3-operand integer ops in 4; "
                                                 "rcp/sqrt/exp in 8 (profiles/r03u_valu_ops): the node step is mostly the 4-cycle class"},
                    "achieved": None, "frac": None, "traffic": None,
                    "kernel": "kz_wf_trace<0> (closest hit of the bounce rays), %d launches per pass" % n_bounce,
                    "kernel_ms_one_pass_alone": stage_ms["trace_bounce"], "pass_ms_in_flight": round(kernel_ms_last, 3),
                    "pass_ms_alone": round(alone_ms, 3), "stages_ms_one_pass_alone": stage_ms, "samples_per_launch": launch_samples,
                    "passes_per_step": info["passes"], "passes_in_flight": info["passesInFlight"]}
        if facts:
            k0 = facts["kernels"].get("kz_wf_trace<0>")
            if k0:
                ach = k0["valu_wave_instr_per_sample"] * launch_samples / (stage_ms["trace_bounce"] * 1e-3) / 1e9
                roofline.update({"achieved": round(ach, 1), "frac": round(ach / peak_rate, 4), "lanes_active_per_valu_instr": k0["lanes"],
                                 "useful_frac": round(ach / peak_rate * k0["lanes"] / 64.0, 4)})
            # the whole closest-hit side (camera rays: beam lists + list kernel + packet kernel; bounce rays) on its own line
            ks = [facts["kernels"][n] for n in ("kz_wf_trace_list", "kz_wf_trace_packet", "kz_wf_trace<0>") if n in facts["kernels"]]
            if ks and stage_ms["trace_closest"] > 0:
                instr = sum(k["valu_wave_instr_per_sample"] for k in ks)
                ach_all = instr * launch_samples / (stage_ms["trace_closest"] * 1e-3) / 1e9
                roofline["closest_hit_all_kernels"] = {"achieved": round(ach_all, 1), "frac": round(ach_all / peak_rate, 4), "ms": stage_ms["trace_closest"],
                                                       "valu_wave_instr_per_sample": round(instr, 2)}
            roofline.update({"traffic": facts.get("hbm_bytes_per_sample") and int(facts["hbm_bytes_per_sample"] * launch_samples),
                             "counter_facts": {"profile": facts["profile"], "commit": facts["commit"], "per_kernel": facts["kernels"]}})
        else:
            roofline["counter_facts_withheld"] = why
        # the SURVEY 8d HBM model, kept as the secondary figure (algorithmic bytes of the REFERENCE algorithm over the pass time)
        roofline["hbm_model"] = {"achieved_algorithmic": round(hbm_algorithmic, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac_algorithmic": round(hbm_algorithmic / HBM_PEAK_GBS, 4), "peak_copy_measured": copy_gbs,
                                 "traffic_frac_of_peak": facts and facts.get("hbm_bytes_per_sample") and
                                 round(facts["hbm_bytes_per_sample"] * launch_samples / (kernel_ms_last * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "bytes_per_sample": round(bps, 1), "bytes_per_sample_source": "cpu oracle counters" if cpu else "gpu megakernel counters",
                                 "bytes_per_sample_gpu_reference_shaped": round(bps_gpu_ref, 1), "bytes_per_sample_executed": round(bps_exec, 1),
                                 "counters_per_sample_reference_algorithm": {k: round(v / max(1, st_ref["samples"]), 3) for k, v in st_ref.items() if k != "samples"},
                                 "counters_per_sample_executed": {k: round(v / max(1, st_exec["samples"]), 3) for k, v in st_exec.items() if k != "samples"},
                                 "note": "algorithmic bytes are served by L1/L2/Infinity Cache, so this ratio is not a fraction of anything physical; "
                                         "the physical HBM figure is traffic_frac_of_peak"}
        rgb = scene.rgb(film_np)
        name = "C5" if args.strong else "C4"
        out = {"metric": "Msamples/s (w*h*spp/s) at %dx%d, 1 M-tri scene" % (Wd, Hd), "value": round(value, 3), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "end_to_end": {"value": round(total_samples / (elapsed + gather_max) / 1e6, 3), "unit": "Msamples/s", "gather_s": round(gather_max, 4),
                              "gather": "every rank: D2H of the packed rects of its tiles (%.1f MB on rank 0); rank 0: merge of all ranks' rects from shared memory, rank after rank" % (packed.nbytes / 1e6),
                              "render_s": round(elapsed, 4)},
               "config": {"workload": "%s: %d random triangles + 8 mesh lights in a closed diffuse room, %dx%d, pmj02bn %d spp, "
                                      "path_mis maxDepth 5; step = %d-spp slice of every rank-owned pixel = %d passes (%d in flight); "
                                      "%dx%d tiles dealt by area over ranks, host gather of tile rects"
                                      % (name, args.tris, Wd, Hd, spp_table, spp_step, info["passes"], info["passesInFlight"], TILE, TILE),
                          "samples_per_step": int(float(px.item()) * spp_step), "bvh_nodes": bvh["nNodes"], "bvh_depth": bvh["maxDepth"],
                          "image_mean": round(float(rgb.mean()), 5), "commit": commit,
                          "items_per_pass_per_rank": items_per_pass_per_rank, "sampler_table_spp": spp_table,
                          "how_the_last_timed_pass_ran": ("one stream", "its shadow rays beside its closest-hit rays", "as two halves side by side")[info["shadowBeside"]],
                          "large_passes_measured_by_the_replica": scene.pass_mode_info(), "warmup_steps_until_settled": settle, "warmup_wait_for_context_s": round(grow_wait_s, 2),
                          "first_call_ms": first_call_ms, "first_call": "the first (warm-up) step: beam lists of every pixel (kz_wf_beam, once per pixel and replica), the pass context "
                                                                        "growing (its memory is mapped on a side thread while the first passes run), then the step itself; a timed step is ms_per_step"},
               "roofline": roofline, "cpu_baseline": cpu, "cold_job": cold, "parity": parity}
    # ---- side jobs. The headline record above is complete: a failure below lands in the sub-record as {"error": ...} and never costs the measurement
    # (ADVICE r04). A collective side job is entered only when EVERY rank is fit for it, and every step of it checks the ranks' status first.
    if world > 1 and not args.strong and not args.no_strong_c5:
        # the same launch also carries the strong-scaling job (collective: every rank); this rank's C4 replica goes first - its pass contexts stay in the
        # device's pool and the C5 replica renders in them
        try:
            scene.close()
            sc5 = strong_c5(kz, rank, world, device_index, args.tris, kw, args.strong_spp)
        except Exception as e:                                 # noqa: BLE001
            sc5 = {"error": repr(e)}
            log("rank %d: strong_c5 failed: %r" % (rank, e))
        if rank == 0:
            out["strong_c5"] = sc5
    if world == 1 and not args.strong and not (args.no_asset_scene and args.no_ext_scenes):
        scene.close()                                          # (its pass contexts go to the device's pool: the scenes below render in them - until round 6 the reference's
                                                               #  scene ran BESIDE the benched one's 175 GB of path state, in passes a quarter of the size it would otherwise earn)
    if world == 1 and not args.strong and not args.no_asset_scene:
        try:
            out["reference_scene"] = reference_scene(kz, device_index, args.asset_spp)
        except Exception as e:                                 # noqa: BLE001
            out["reference_scene"] = {"error": repr(e)}
    if world == 1 and not args.strong and not args.no_ext_scenes:
        try:
            out["ext_scenes"] = ext_scenes(kz, device_index, cpu=not args.no_cpu_baseline)
        except Exception as e:                                 # noqa: BLE001
            out["ext_scenes"] = {"error": repr(e)}
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                                 # noqa: BLE001 - the record is out
            log("rank %d: shutdown: %r" % (rank, e))


def settled_calls(sc, call, most=14):
    """Wall times of repeated whole jobs on a resident scene until it runs the way it will go on running: the pass context earns its size call by call, and a replica
    whose passes are large then times four of them (one stream, shadow rays beside, halves, one stream) before it keeps the fastest way - then two more calls."""
    ts, after = [], 0
    while len(ts) < most and after < 2:
        t0 = time.perf_counter(); call(); sc.sync(); ts.append(time.perf_counter() - t0)
        m = sc.pass_mode_info()
        if len(ts) >= 3 and (m["kept"] is not None or m["timed_passes"] == 0):
            after += 1
    return ts


def ext_scenes(kz, device_index, cpu=True):
    """VERDICT r05 item 2a / SURVEY 8(d) "time C1-C3 in full": whole jobs, driver-timed, outside the timed region of `value` - (i) the two scenes that run the EXT
    shade kernels (rough BSDFs, mirror, dielectric; image textures, normal map, blend, ramp, environment map) at 1920 x 1080 x 256, and (ii)
    BASELINE configs C1, C2, C3 at the sizes BASELINE.json quotes, each beside the oracle's time on the host cores for the SAME job (C1, C2 in full; C3 a centre crop,
    scaled - the oracle needs minutes for 531 M samples). Every GPU figure is the best of the calls after the first on a resident scene (the first builds the beam lists)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    S = kz.scenes
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    jobs = [
        ("materials_scene", "mirror, dielectric, ggx, roughconductor, roughplastic, roughdielectric spheres (EXT kernels, no textures), 1920x1080, 256 spp, independent", lambda: S.materials_scene(1920, 1080, 256), None),
        ("textured_scene", "image textures, normal map, blend, colour ramp, environment map (EXT kernels + texture programs), 1920x1080, 256 spp, independent", lambda: S.textured_scene(1920, 1080, 256), None),
        ("C1", "BASELINE configs[0]: scene/2022_q1 default_m0_r0.5 (36 378 triangles), 256x256, 16 spp, independent",
         lambda: S.load_npz(os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz"), overrides={"camera": {"width": 256, "height": 256}, "sampler": {"type": "independent", "sampleCount": 16, "seed": 0}}), "full"),
        ("C2", "BASELINE configs[1]: diffuse sphere (9 800 triangles) + constant environment, 512x512, 64 spp, independent", lambda: S.sphere_env(512, 512, 64), "full"),
        ("C3", "BASELINE configs[2]: hero scene (508 k triangles), full kiss BSDF + 3 area lights, 1920x1080, 256 spp, independent", lambda: S.hero_scene(1920, 1080, 256, detail=2.0), "crop"),
    ]
    out = {"unit": "Msamples/s", "cpu_threads": threads, "note": "whole jobs on a resident scene (kz_render of every sample of every pixel + sync; best call after the first); C1 is 1 M samples - "
           "a job of a millisecond, bound by its ~40 kernel launches, not by the kernels"}
    t_all = time.perf_counter()
    for name, what, make, cpu_mode in jobs:
        try:
            t0 = time.perf_counter()
            desc = make()
            sc = kz.Scene(desc, device=device_index)
            build_s = time.perf_counter() - t0
            n = sc.width * sc.height * sc.sample_count
            ts = settled_calls(sc, lambda: sc.render())
            rec = {"workload": what, "samples": n, "value": round(n / min(ts[-2:] if n > (1 << 27) else ts[1:]) / 1e6, 1), "render_s": [round(t, 5) for t in ts], "scene_build_upload_s": round(build_s, 2),
                   "image_mean": round(float(sc.rgb().mean()), 5), "passes": sc.last_pass_info()["passes"], "tris": desc.n_tris(),
                   "how_the_last_pass_ran": ("one stream", "shadow rays beside", "halves")[sc.last_pass_info()["shadowBeside"]]}
            if n > (1 << 27):
                rec["large_passes_measured_by_the_replica"] = sc.pass_mode_info()
            if cpu and cpu_mode:
                import oracle as O
                ora = O.OracleScene(desc)
                if cpu_mode == "full":
                    t0 = time.perf_counter(); ora.render(threads=threads); dt = time.perf_counter() - t0
                    rec["cpu_oracle"] = {"value": round(n / dt / 1e6, 3), "seconds": round(dt, 2), "sample": "the whole job", "threads": threads, "kind": "port"}
                else:
                    tile = (sc.width // 2 - 192, sc.height // 2 - 108, 384, 216)
                    t0 = time.perf_counter(); ora.render(0, 16, tiles=[tile], threads=threads); dt = time.perf_counter() - t0
                    rate = tile[2] * tile[3] * 16 / dt
                    rec["cpu_oracle"] = {"value": round(rate / 1e6, 3), "seconds": round(dt, 2), "threads": threads, "kind": "port",
                                         "sample": "centre crop 384x216 at sample indices [0, 16) = %d samples; the whole job at this rate: %.0f s" % (tile[2] * tile[3] * 16, n / rate)}
                rec["gpu_over_cpu"] = round(rec["value"] / max(1e-9, rec["cpu_oracle"]["value"]), 1)
            sc.close()
        except Exception as e:                                 # noqa: BLE001 - one scene must not cost the others
            rec = {"workload": what, "error": repr(e)}
        out[name] = rec
        log("ext_scenes %s: %s" % (name, json.dumps(rec)))
        if time.perf_counter() - t_all > 140.0:                # (bounded: the default bench run stays within a few minutes)
            out["stopped"] = "time budget of 140 s used up after %s" % name
            break
    out["seconds"] = round(time.perf_counter() - t_all, 1)
    return out


def reference_scene(kz, device_index, spp):
    """The one job of the reference's own that this path can run as published: scene/2022_q1/parameters/default_m0_r0.5.xml + its OBJ files (36 378 triangles, kiss object on a
    smooth backdrop, three invisible area lights), flattened into tests/golden/q1_default_m0_r0.5.npz, at the settings of the scene file - 1920 x 1080, independent sampler,
    path_mis maxDepth 5 - for a slice of its 4096 samples per pixel. Outside the timed region of `value`; not a BASELINE.json config, so `vs_baseline` stays null: the only published
    timing of the reference is a caption for a job of this size on another scene of the same studio set (unstated CPU), quoted beside it."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
    if not os.path.exists(path):
        return None
    d = kz.scenes.load_npz(path)
    sc = kz.Scene(d, device=device_index)
    spp = min(spp, sc.sample_count)
    sc.render(0, min(64, spp)); sc.sync()
    ts = settled_calls(sc, lambda: sc.render(0, spp))
    kept = sc.last_pass_info()["shadowBeside"]
    n = sc.width * sc.height * spp
    # the same slice with KzRenderOpts::shadowBeside = 2 (the shadow rays of a bounce beside its closest-hit rays) and = 1 (in front): this scene's shadow rays are
    # short-lived and its kernels do not fill the VALUs by themselves (profiles/r06v_shadow_beside); `value` is the library default, i.e. what the replica measured and kept
    film0 = sc.film()
    tb = []
    t1 = []
    for mode, out in ((2, tb), (1, t1)):
        for _ in range(2):
            t0 = time.perf_counter(); sc.render(0, spp, shadow_beside=mode); sc.sync(); out.append(time.perf_counter() - t0)
    beside = {"value": round(n / min(tb) / 1e6, 1), "render_s": [round(t, 4) for t in tb], "film_equal": bool(np.array_equal(film0, sc.film())), "option": "KzRenderOpts::shadowBeside = 2",
              "one_stream": {"value": round(n / min(t1) / 1e6, 1), "render_s": [round(t, 4) for t in t1], "option": "KzRenderOpts::shadowBeside = 1"},
              "kept_by_the_replica": ("one stream", "shadow rays beside", "halves")[kept], "large_passes_measured_by_the_replica": sc.pass_mode_info()}
    rec = {"workload": "scene/2022_q1/parameters/default_m0_r0.5.xml (36 378 triangles, the reference's own scene file via tests/golden/q1_default_m0_r0.5.npz), %dx%d, independent sampler, "
                       "path_mis maxDepth %d, sample indices [0, %d) of the file's %d" % (sc.width, sc.height, d.integrator["maxDepth"], spp, sc.sample_count),
           "value": round(n / min(ts[-2:]) / 1e6, 1), "unit": "Msamples/s", "render_s": [round(t, 4) for t in ts], "image_mean": round(float(sc.rgb().mean()), 5),
           "shadow_beside": beside,
           "whole_job": "all 4096 spp: 4.4 s, 1 927 Msamples/s; against the published 4096-spp picture of this scene file: profiles/r04p_q1_full",
           "published_caption": {"job": "1920x1080, 4096 spp (another scene of the same studio set)", "seconds": 702, "Msamples_per_s": 12.1, "hardware": "unstated CPU",
                                 "source": "doc/2022_q1/2022_q1_report.md:226"}}
    sc.close()
    return rec


def strong_c5(kz, rank, world, device_index, tris, kw, spp=256):
    """The strong-scaling evidence inside the driver's N > 1 line: BASELINE.json configs[4] - the C4 scene at 3840x2160 with the
    4096-spp sampler table - as ONE fixed job split over the ranks, bounded to sample indices [0, spp): every rank renders all `spp` samples of the tiles it
    gets, then the tile rects are gathered on rank 0. Twice: tiles dealt beforehand by area (kz_deal_tiles), and taken in batches from a KzTileDealer whose
    counter the ranks share in /dev/shm (the reference's BlockGenerator, block.cpp:117-148). Collective: every rank calls it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    t0 = time.time()
    desc = kz.scenes.random_triangles(tris, W5, H5, SPP5, sampler="pmj02bn", seed=1)
    scene = kz.Scene(desc)
    scene.upload(device_index)
    build_s = time.time() - t0
    all_tiles = kz.shard.deal_tiles(W5, H5, 1, 0, TILE)
    mine = kz.shard.deal_tiles(W5, H5, world, rank, TILE)
    samples = float(W5 * H5 * spp)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def all_fit(ok):
        """every rank says whether it can go on; a rank that cannot makes every rank bail out instead of leaving the others in the next collective"""
        f = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
        if not int(f[0]):
            raise RuntimeError("strong_c5: a rank failed%s" % ("" if ok else " (this one)"))

    def run(dynamic):
        counter = cpath = None
        if dynamic:
            counter, cpath = kz.shard.shared_counter(rank, world)
        session = kz.shard.open_gather(rank, world)            # (collective, before the clock: the gather itself then needs none in front of the merge)
        sync()
        t = time.perf_counter()
        err = None
        took, packed = [], np.zeros(0, np.float32)
        try:
            if dynamic:
                took = scene.render_dealt(all_tiles, counter, takers=world, device=device_index, sample_begin=0, sample_end=spp, **kw)
            else:
                took = mine
                scene.render_tiles(mine, device=device_index, sample_begin=0, sample_end=spp, download=False, **kw)
        except Exception as e:                                 # noqa: BLE001 - every rank still enters the gather: it carries the failure to all
            err, took = e, []
        mine_s = time.perf_counter() - t                     # (kz_render_tiles is blocking: this rank's tiles are on its film)
        info = scene.last_pass_info() if err is None else {"largestPassItems": 0}
        # the gather starts here, per rank, with no barrier in front of it: rank 0 merges the ranks' rects as they land (shard.gather_tiles), so it overlaps
        # the tail of the slower ranks' renders; `end_to_end` is the one clock from the common start to the merged frame on rank 0
        t_g = time.perf_counter()
        film = None
        try:
            if err is None:
                if took:
                    packed = scene.film_tiles(took, device=device_index)
                film = kz.shard.gather_tiles(scene, took, packed, rank, world, session=session)
            else:                                              # this rank's render failed: a buffer of the wrong size fails the gather on EVERY rank, nobody waits for this one
                kz.shard.gather_tiles(scene, [(0, 0, 1, 1)], np.zeros(0, np.float32), rank, world, session=session)
        except Exception as e:                                 # noqa: BLE001
            err = err or e
        total_s = time.perf_counter() - t
        gather_own_s = time.perf_counter() - t_g
        if rank == 0 and cpath:
            os.unlink(cpath)
        all_fit(err is None)
        v = torch.tensor([mine_s, total_s, -mine_s, float(len(took)), -float(len(took)), gather_own_s, float(info["largestPassItems"]), -float(info["largestPassItems"])], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        rec = {"value": round(samples / v[0].item() / 1e6, 1), "end_to_end": round(samples / v[1].item() / 1e6, 1), "render_s": round(v[0].item(), 4),
               "end_to_end_s": round(v[1].item(), 4), "gather_after_last_render_s": round(v[1].item() - v[0].item(), 4), "gather_s_slowest_rank": round(v[5].item(), 4),
               "per_rank_ms": [round(-1e3 * v[2].item(), 1), round(1e3 * v[0].item(), 1)], "tiles_per_rank": [int(-v[4].item()), int(v[3].item())],
               "items_per_pass_per_rank": [int(-v[7].item()), int(v[6].item())]}
        return rec, film

    run(False)                                                # warm-up: allocations, beam lists of this rank's tiles, pinned staging, /dev/shm pages
    static, film_s = run(False)
    run(True)                                                 # (the dealer hands a rank other tiles than the static deal: their beam lists are built here)
    dynamic, film_d = run(True)
    out = {"workload": "C5 slice: %d random triangles, %dx%d, sample indices [0,%d) of the %d-spp pmj02bn table, ONE job over %d ranks, %dx%d tiles"
                       % (tris, W5, H5, spp, SPP5, world, TILE, TILE), "unit": "Msamples/s", "samples": int(samples), "scene_build_upload_s": round(build_s, 1),
           "static": static, "dynamic": dynamic}
    if rank == 0:
        out["films_agree"] = bool(np.array_equal(film_s, film_d))          # bit for bit since round 6: a tile's rect is what the tile's own pixels add, merged in tile order
        out["film_crc"] = film_crc(film_s, scene.border, x0=1888, y0=1048)  # (the same number at every N and for both dealings: compare across the SCALE run's lines)
        out["image_mean"] = round(float(scene.rgb(film_s).mean()), 5)
    scene.close()
    return out


def cpu_baseline(desc, target_seconds, Wd, Hd, scene=None):
    """The oracle (kind "port": the reference binary cannot be built here) timed on the host cores on a
    bounded sample of the SAME workload: a centre crop of the frame at the first sample indices."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    t0 = time.time()
    ora = O.OracleScene(desc)
    build_s = time.time() - t0
    # calibrate on a small crop, then size the timed sample (crop x spp) for ~target_seconds of CPU work
    cx, cy = Wd // 2, Hd // 2
    t0 = time.time()
    ora.render(0, 1, tiles=[(cx - 128, cy - 128, 256, 256)], threads=threads)
    rate = 256 * 256 / max(1e-3, time.time() - t0)
    want = rate * target_seconds
    tw, th = Wd, Hd
    if want < Wd * Hd * 2:
        tw = min(Wd, max(128, int((want / 2 * 16 / 9) ** 0.5) // 32 * 32))
        th = min(Hd, max(96, int(want / 2 / tw) // 32 * 32))
    spp = int(max(2, min(16, 0.5 * want // (tw * th))))
    tile = (cx - tw // 2, cy - th // 2, tw, th)
    ora.stats(reset=True)
    t0 = time.time()
    cpu_film = ora.render(0, spp, tiles=[tile], threads=threads)
    dt = time.time() - t0
    st = ora.stats()
    parity = None
    if scene is not None:
        # the SAME sample on the device, against the film the oracle has just been timed on (north_star: per-pixel L2 of normalised linear rgb < 1e-3): the oracle as the
        # checker, after every clock has stopped. (Its reference-shaped film adds in block order; in the build's own order the films are equal bit for bit: tests/)
        import numpy as np
        scene.render(0, spp, tiles=[tile])
        g, c = scene.rgb(scene.film()), ora.rgb(cpu_film)
        x0, y0 = tile[0], tile[1]
        g, c = g[y0:y0 + th, x0:x0 + tw], c[y0:y0 + th, x0:x0 + tw]
        parity = {"l2_per_pixel_vs_cpu_oracle": float(np.sqrt(np.mean((g.astype(np.float64) - c.astype(np.float64)) ** 2))), "bar": 1e-3, "pixels": int(tw * th), "spp": spp,
                  "max_abs": float(np.abs(g - c).max()), "image_mean": float(c.mean())}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": round(tw * th * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "centre crop %dx%d px at sample indices [0,%d) of the same frame = %d samples in %.1f s "
                      "(oracle BVH build %.1f s excluded)" % (tw, th, spp, tw * th * spp, dt, build_s),
            "cpu_model": model, "bytes_per_sample_oracle": round(algorithmic_bytes_per_sample(st), 1), "parity": parity}


if __name__ == "__main__":
    main()
