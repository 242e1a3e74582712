#!/usr/bin/env python3
"""bench.py — Msamples/s of the path_mis hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], "C4"): 1 M random triangles + 8 mesh lights in a closed diffuse room,
1920x1080, pmj02bn sampler (1024 spp), path_mis maxDepth 5. A "step" is one kz_render call over one batch: a 128-spp
slice of every pixel a GPU owns = 265 M (pixel, sample) paths per GPU per step, which the library runs as two 133 M-path
passes (its default pass size, 2^27 items) kept in flight together on two internal streams.

N GPUs (north_star: "the image tile grid shards embarrassingly across the GPUs; no RCCL needed; host gathers
tiles"): one process per GPU, each with a full scene replica; the 64x64 tiles of the frame are dealt over the ranks by area
(kz_deal_tiles, the dealing kz_render_multi uses in-process) and each rank renders 128*N spp of ITS tiles per step, so
per-GPU work is fixed (weak scaling). There is NO device collective anywhere: torch.distributed runs on the CPU (gloo) for the
barriers around the timed region and for the one host gather of the films after it (shard.gather_films = ImageBlock::put(ImageBlock&),
block.cpp:87-96, in rank order). Scene tables, BVH and sampler tables are resident in HBM before the timed region starts.

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
SPP_PER_RANK_STEP = 128        # two passes of 64 spp per call at 1920x1080 (2^27 items per pass)
W, H, NTRIS, SPP = 1920, 1080, 1000000, 1024
TILE = 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes_per_sample(st):
    """64 B node packets, 48 B leaf triangles, 96 B shading gathers, 64 B light samples, +16 B film."""
    s = max(1, st["samples"])
    return (64.0 * st["nodeVisits"] + 48.0 * st["triTests"] + 96.0 * st["shadedHits"] + 64.0 * st["lightSamples"]) / s + 16.0


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        return None


def source_hash():
    """sha256 (first 16 hex digits) of the kernel and ABI sources: the counter facts of a profile describe ONE build"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "nano-kazen_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "kazen_mi355x.h")]):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=NTRIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the oracle baseline sample")
    args = ap.parse_args()
    commit = os.environ.get("GRAFT_HEAD") or git_head()          # before torch / HIP are loaded: no fork from a process that has initialised the GPU

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
    t0 = time.time()
    desc = kz.scenes.random_triangles(args.tris, W, H, SPP, sampler="pmj02bn", seed=1)
    t1 = time.time()
    scene = kz.Scene(desc)
    bvh = scene.bvh_info()
    scene.upload(device_index)
    t2 = time.time()
    if rank == 0:
        log("scene: %d tris, synth %.1fs, BVH %d nodes depth %d SAH %.1f built in %.2fs, upload+build %.1fs"
            % (desc.n_tris(), t1 - t0, bvh["nNodes"], bvh["maxDepth"], bvh["sahCost"], bvh["buildSeconds"], t2 - t1))

    tiles = kz.shard.deal_tiles(W, H, world, rank, TILE) if world > 1 else None
    my_pixels = W * H if tiles is None else sum(t[2] * t[3] for t in tiles)
    spp_step = SPP_PER_RANK_STEP * world                  # weak scaling: a rank owns 1/N of the pixels and renders N x the spp
    if spp_step > SPP:
        raise SystemExit("bench.py: %d ranks x %d spp per step exceed the %d-spp sampler table (use <= %d ranks)"
                         % (world, SPP_PER_RANK_STEP, SPP, SPP // SPP_PER_RANK_STEP))
    stream = torch.cuda.current_stream().cuda_stream
    kw = {}
    if shared_device:
        kw["max_state_bytes"] = int(0.8 * torch.cuda.mem_get_info(device_index)[1] / world)

    def step(k, accumulate=True):
        # sample indices [s0, s0 + spp_step) modulo the table: a slice that wraps is rendered as its two halves
        s0 = (k * spp_step) % SPP
        s1 = s0 + spp_step
        if s1 <= SPP:
            scene.render(s0, s1, tiles=tiles, accumulate=accumulate, stream=stream, **kw)
        else:
            scene.render(s0, SPP, tiles=tiles, accumulate=accumulate, stream=stream, **kw)
            scene.render(0, s1 - SPP, tiles=tiles, accumulate=True, stream=stream, **kw)

    def barrier():
        # device first, then the ranks, then the device again: a rank must not start (or stop) its clock while its own earlier kernels still
        # run (with one rank per GPU the order is immaterial; with several ranks rehearsing on ONE GPU it is not: profiles/r03b_multiprocess)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scene.film_clear(stream)
    for k in range(args.warmup):
        step(k)
    barrier()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms_last = scene.last_kernel_ms()      # hipEvents on the launch stream around the last call / its passes
    info = scene.last_pass_info()
    # film merge: once per render, outside the per-step loop but reported (a HOST gather, not a data-path collective)
    t_m = time.perf_counter()
    film_np = kz.shard.gather_films(scene.film(), rank, world)
    merge_s = time.perf_counter() - t_m

    el = torch.tensor([elapsed], dtype=torch.float64)
    px = torch.tensor([my_pixels], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(px, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_samples = float(px.item()) * spp_step * args.steps
    value = total_samples / elapsed / 1e6

    if rank == 0:
        # ---- one pass run alone: per-stage device times (with two passes in flight the stage events of a pass overlap the other's)
        pass_spp = info["sppPerPass"]
        scene.render(0, pass_spp, tiles=tiles, accumulate=True, stream=stream, **kw)
        scene.sync()
        stage_ms = scene.last_stage_ms()
        alone_ms = scene.last_kernel_ms()
        launch_samples = my_pixels * pass_spp                       # one pass: the unit of kernel_ms, traffic and the counter facts
        # ---- counters of the executed (wavefront) and of the reference-shaped (megakernel) pipelines on the last slice
        scene.set_stats(True)
        s0 = ((args.warmup + args.steps - 1) * spp_step) % SPP
        s1 = min(SPP, s0 + spp_step)
        scene.stats(reset=True)
        scene.render(s0, s1, tiles=tiles, accumulate=True, stream=stream, **kw)
        scene.sync()
        st_exec = scene.stats(reset=True)
        scene.render(s0, s1, tiles=tiles, accumulate=True, pipeline=1, stream=stream, **kw)      # same slice, reference-shaped
        scene.sync()
        st_ref = scene.stats(reset=True)
        scene.set_stats(False)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(desc, args.cpu_seconds)
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
        # traversal is a VALU-issue problem): the roof is the VALU issue rate of the chip, calibrated with
        # scripts/micro/valu_peak.hip (independent v_fma_f32 at 8 waves/SIMD, the fastest VALU stream gfx950 sustains).
        # achieved = VALU wave-instructions the closest-hit traversal launches of one pass execute (a property of code + input,
        # counted by rocprofv3 SQ_INSTS_VALU in the committed profile) / their LIVE device time (hipEvents, one pass alone).
        facts, why = load_profile_facts(alone_ms)
        peak = json.load(open(os.path.join(ROOT, "profiles", "valu_peak.json")))
        peak_rate = peak["fma_wave_instr_per_s"] / 1e9
        roofline = {"bound": "valu", "unit": "G wave-instr/s", "peak": round(peak_rate, 1),
                    "peak_source": "scripts/micro/valu_peak.hip v_fma_f32, 8 waves/SIMD, measured (profiles/%s)" % peak["profile"],
                    "achieved": None, "frac": None, "traffic": None,
                    "kernel": "closest-hit traversal = kz_wf_trace_packet (camera rays) + %d launches of kz_wf_trace<0> (bounce rays) per pass" % (desc.integrator["maxDepth"] - 1),
                    "kernel_ms_one_pass_alone": stage_ms["trace_closest"], "pass_ms_in_flight": round(kernel_ms_last, 3),
                    "pass_ms_alone": round(alone_ms, 3), "stages_ms_one_pass_alone": stage_ms, "samples_per_launch": launch_samples,
                    "passes_per_step": info["passes"], "passes_in_flight": info["passesInFlight"]}
        if facts:
            ks = [facts["kernels"][n] for n in ("kz_wf_trace_packet", "kz_wf_trace<0>") if n in facts["kernels"]]     # camera rays + bounce rays
            instr = sum(k["valu_wave_instr_per_sample"] for k in ks)
            lanes = sum(k["valu_wave_instr_per_sample"] * k["lanes"] for k in ks) / instr
            ach = instr * launch_samples / (stage_ms["trace_closest"] * 1e-3) / 1e9
            roofline.update({"achieved": round(ach, 1), "frac": round(ach / peak_rate, 4),
                             "lanes_active_per_valu_instr": round(lanes, 1), "useful_frac": round(ach / peak_rate * lanes / 64.0, 4),
                             "mix_ceiling": peak.get("node_mix_wave_instr_per_s") and round(peak["node_mix_wave_instr_per_s"] / 1e9, 1),
                             "traffic": facts.get("hbm_bytes_per_sample") and int(facts["hbm_bytes_per_sample"] * launch_samples),
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
        out = {"metric": "Msamples/s (w*h*spp/s) at 1920x1080, 1 M-tri scene", "value": round(value, 3), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "C4: %d random triangles + 8 mesh lights in a closed diffuse room, %dx%d, pmj02bn %d spp, "
                                      "path_mis maxDepth 5; step = %d-spp slice of every rank-owned pixel = %d passes (%d in flight); "
                                      "%dx%d tiles dealt by area over ranks, host film gather"
                                      % (args.tris, W, H, SPP, spp_step, info["passes"], info["passesInFlight"], TILE, TILE),
                          "samples_per_step": int(float(px.item()) * spp_step), "bvh_nodes": bvh["nNodes"], "bvh_depth": bvh["maxDepth"],
                          "film_merge_s": round(merge_s, 4), "image_mean": round(float(rgb.mean()), 5), "commit": commit},
               "roofline": roofline, "cpu_baseline": cpu}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(desc, target_seconds):
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
    cx, cy = W // 2, H // 2
    t0 = time.time()
    ora.render(0, 1, tiles=[(cx - 128, cy - 128, 256, 256)], threads=threads)
    rate = 256 * 256 / max(1e-3, time.time() - t0)
    want = rate * target_seconds
    tw, th = W, H
    if want < W * H * 2:
        tw = min(W, max(128, int((want / 2 * 16 / 9) ** 0.5) // 32 * 32))
        th = min(H, max(96, int(want / 2 / tw) // 32 * 32))
    spp = int(max(2, min(16, 0.5 * want // (tw * th))))
    tile = (cx - tw // 2, cy - th // 2, tw, th)
    ora.stats(reset=True)
    t0 = time.time()
    ora.render(0, spp, tiles=[tile], threads=threads)
    dt = time.time() - t0
    st = ora.stats()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": round(tw * th * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "centre crop %dx%d px at sample indices [0,%d) of the same C4 frame = %d samples in %.1f s "
                      "(oracle BVH build %.1f s excluded)" % (tw, th, spp, tw * th * spp, dt, build_s),
            "cpu_model": model, "bytes_per_sample_oracle": round(algorithmic_bytes_per_sample(st), 1)}


if __name__ == "__main__":
    main()
