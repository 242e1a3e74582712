#!/usr/bin/env python3
"""bench.py — Msamples/s of the path_mis hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], "C4"): 1 M random triangles + 8 mesh lights in a closed diffuse room,
1920x1080, pmj02bn sampler (1024 spp), path_mis maxDepth 5. A "step" is one kz_render call over one batch: a 128-spp
slice (sample indices [128k, 128k+128)) of every pixel a GPU owns = 265 M (pixel, sample) paths per GPU per step, which
the library runs as two 133 M-path passes (its default pass size, 2^27 items) kept in flight together on two internal streams. With N GPUs the image tiles
(128x128) are dealt round-robin over the ranks and each rank
renders 128*N spp of ITS tiles per step, so per-GPU work is fixed (weak scaling); there is no data-path
collective — the per-rank films are summed once at the end (ImageBlock::put(ImageBlock&), block.cpp:87-96).
Scene tables, BVH and sampler tables are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); 6.29 TB/s measured copy
SPP_PER_STEP = 128            # two passes of PASS_SPP per call
PASS_SPP = 64                 # 2^27 (pixel, sample) items per pass (the library's default pass size) / 1920x1080 pixels
W, H, NTRIS, SPP = 1920, 1080, 1000000, 1024


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes_per_sample(st):
    """64 B node packets, 48 B leaf triangles, 96 B shading gathers, 64 B light samples, +16 B film."""
    s = max(1, st["samples"])
    return (64.0 * st["nodeVisits"] + 48.0 * st["triTests"] + 96.0 * st["shadedHits"] + 64.0 * st["lightSamples"]) / s + 16.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=NTRIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the oracle baseline sample")
    args = ap.parse_args()

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
    # rehearsal on a 1-GPU box: KZ_BENCH_BACKEND=gloo KZ_BENCH_DEVICE=0 runs every rank on GPU 0 with CPU collectives
    backend = os.environ.get("KZ_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("KZ_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    cdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

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

    tiles_all = kz.shard.make_tiles(W, H, 128)
    tiles = kz.shard.tiles_for_rank(tiles_all, rank, world) if world > 1 else None
    my_pixels = W * H if tiles is None else sum(t[2] * t[3] for t in tiles)
    spp_step = SPP_PER_STEP * world
    stream = torch.cuda.current_stream().cuda_stream

    def step(k, accumulate=True):
        s0 = (k * spp_step) % SPP
        scene.render(s0, s0 + spp_step, tiles=tiles, accumulate=accumulate, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scene.film_clear(stream)
    for k in range(args.warmup):
        step(k)
    barrier()
    t_start = time.perf_counter()
    kernel_ms = []
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms_last = scene.last_kernel_ms()      # HIP events on the launch stream, around the path kernels of the last step
    stage_ms_last = scene.last_stage_ms()
    # film merge: once per render, outside the per-step loop but reported (not a data-path collective)
    t_m = time.perf_counter()
    film = torch.from_numpy(scene.film()).to(cdev)
    if world > 1:
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    merge_s = time.perf_counter() - t_m

    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    px = torch.tensor([my_pixels], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(px, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_samples = float(px.item()) * spp_step * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        # ---- roofline: ALGORITHMIC bytes per pass / device time of the path kernels of one pass.
        # The per-sample figure is the reference algorithm's (closest-hit for every query, every shadow ray traced):
        # counted by the reference-shaped megakernel pipeline, whose counters the parity tests hold equal to the CPU
        # oracle's. The wavefront pipeline that is being timed does LESS than that for the same film (any-hit shadow
        # test, zero-contribution shadow rays skipped, BVH4 packets): its own counters are reported next to it.
        # per-stage device times of ONE pass run alone (with two passes in flight the stage events of a pass overlap the other's)
        scene.render(0, PASS_SPP * world, tiles=tiles, accumulate=True, stream=stream)
        scene.sync()
        stage_ms_last = scene.last_stage_ms()
        isolated_pass_ms = scene.last_kernel_ms()
        scene.set_stats(True)
        s0 = ((args.warmup + args.steps - 1) * spp_step) % SPP
        scene.stats(reset=True)
        scene.render(s0, s0 + spp_step, tiles=tiles, accumulate=True, stream=stream)
        scene.sync()
        st_exec = scene.stats(reset=True)
        scene.render(s0, s0 + spp_step, tiles=tiles, accumulate=True, pipeline=1, stream=stream)      # same slice, reference-shaped
        scene.sync()
        st_ref = scene.stats(reset=True)
        scene.set_stats(False)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(desc, args.cpu_seconds)
        # SURVEY 8d takes the counts from the CPU oracle; the GPU's reference-shaped counters stand in when the oracle leg is off
        bps_gpu_ref = algorithmic_bytes_per_sample(st_ref)
        bps = cpu["bytes_per_sample_oracle"] if cpu else bps_gpu_ref
        bps_exec = algorithmic_bytes_per_sample(st_exec)
        launch_samples = my_pixels * PASS_SPP * world          # one pass (the unit kernel_ms, traffic and the PMC figures refer to)
        achieved = bps * launch_samples / (kernel_ms_last * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the datasheet peak beside a measured device-to-device stream copy (read + write bytes / time)
        copy_gbs = None
        if True:
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
        pmc_note = None
        ppath = os.path.join(ROOT, "profiles", "bound_latest.json")
        if os.path.exists(ppath):
            try:
                pmc_note = json.load(open(ppath))
            except Exception:
                pmc_note = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "peak_copy_measured": copy_gbs,
                    "limiter_from_pmc": pmc_note,
                    "kernel": "wavefront pass = kz_wf_generate + maxDepth x (kz_wf_trace<0> closest-hit + kz_wf_shade + "
                              "kz_wf_trace<2> shadow) + kz_film_gather; two passes are in flight per call, kernel_ms = hipEvent span of "
                              "the call on the launch stream / passes of the call",
                    "kernel_ms": round(kernel_ms_last, 3), "stages_ms_one_pass_alone": stage_ms_last, "path_kernels_ms_one_pass_alone": round(isolated_pass_ms, 3),
                    "bytes_per_sample": round(bps, 1), "bytes_per_sample_source": "cpu oracle counters" if cpu else "gpu megakernel counters",
                    "bytes_per_sample_gpu_reference_shaped": round(bps_gpu_ref, 1), "bytes_per_sample_executed": round(bps_exec, 1),
                    "samples_per_launch": launch_samples,
                    "counters_per_sample_reference_algorithm": {k: round(v / max(1, st_ref["samples"]), 3) for k, v in st_ref.items() if k != "samples"},
                    "counters_per_sample_executed": {k: round(v / max(1, st_exec["samples"]), 3) for k, v in st_exec.items() if k != "samples"},
                    "note": "algorithmic bytes are mostly served by L1/L2/Infinity Cache (see profiles/: FETCH_SIZE per pass), "
                            "so achieved can exceed what HBM alone could deliver"}
        film_np = film.cpu().numpy()
        rgb = scene.rgb(film_np)
        out = {"metric": "Msamples/s (w*h*spp/s) at 1920x1080, 1 M-tri scene", "value": round(value, 3), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "C4: %d random triangles + 8 mesh lights in a closed diffuse room, %dx%d, pmj02bn %d spp, "
                                      "path_mis maxDepth 5; step = %d-spp slice of the frame (%d spp per rank-owned pixel) = 2 passes in flight, "
                                      "128x128 tiles round-robin over ranks" % (args.tris, W, H, SPP, spp_step, spp_step),
                          "samples_per_step": int(float(px.item()) * spp_step), "bvh_nodes": bvh["nNodes"], "bvh_depth": bvh["maxDepth"],
                          "film_merge_s": round(merge_s, 4), "image_mean": round(float(rgb.mean()), 5)},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(desc, target_seconds):
    """The oracle (kind "port": the reference binary cannot be built here) timed on the host cores on a
    bounded sample of the SAME workload: a centre crop of the frame at the first sample indices."""
    import numpy as np
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
