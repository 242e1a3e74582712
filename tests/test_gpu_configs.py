"""-m gpu: the BASELINE.json configs at their FULL sizes, and the multi-device entry points of ABI v3.

* C2 (sphere + environment, 512x512) at its full 64 spp against the oracle;
* C3 (hero scene, full detail = 508 k triangles, 1920x1080x256): properties + crop parity against the oracle;
* C5 (the 1 M-triangle scene at 3840x2160, pmj02bn 4096): a 4-spp slice rendered as EIGHT tile shares through
  kz_render_tiles (the unit of multi-GPU sharding), summed on the host, against the one-shot film and, on a crop, the oracle;
* kz_render_multi (one host thread per device, host gather) on the devices this box has;
* replicas, budget limits and error paths of ABI v3.
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L2_TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPERIMENTS_LIB = os.path.join(ROOT, "nano-kazen_amd", "csrc", "variants", "experiments", "libkazen_mi355x.so")


def l2(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _crop_rgb(film, x0, y0, w, h, b):
    c = film[y0:y0 + h + 2 * b, x0:x0 + w + 2 * b]
    return c[..., :3] / np.maximum(c[..., 3:], 1e-20), c[..., 3]


def test_c2_full_spp_matches_oracle(gpu_lib, kz, O):
    desc = kz.scenes.sphere_env(512, 512, 64)                       # BASELINE configs[1] as quoted: 512x512, 64 spp
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render()
    st = sc.stats()
    ora = O.OracleScene(desc)
    cpu = ora.rgb(ora.render(threads=0))
    assert st["samples"] == 512 * 512 * 64 == ora.stats()["samples"] and st["droppedSamples"] == 0
    assert l2(sc.rgb(), cpu) < L2_TOL
    assert np.array_equal(sc.film(), ora.render_canonical(threads=0))          # round 6: the WHOLE film of BASELINE configs[1] at its full size, bit for bit


def test_c3_full_size_properties_and_crop(gpu_lib, kz, O):
    """configs[2]: hero scene at full detail, full kiss BSDF + 3 lights, 1920x1080, 256 spp, all of it rendered."""
    desc = kz.scenes.hero_scene(1920, 1080, 256, detail=2.0)
    sc = kz.Scene(desc, device=0)
    assert sc.bvh_info()["nTris"] > 500000
    sc.set_stats(True)
    sc.render()
    film = sc.film()
    st = sc.stats(reset=True)
    assert st["samples"] == 1920 * 1080 * 256 and st["droppedSamples"] == 0 and np.isfinite(film).all()
    # total filter mass per sample equals that of a small render with the same filter (interior samples: the full kernel sum)
    small = kz.Scene(kz.scenes.cornell_box(64, 64, 16), device=0)
    small.render()
    mass, mass_small = film[..., 3].sum() / st["samples"], small.film()[..., 3].sum() / (64 * 64 * 16)
    assert abs(mass - mass_small) < 2e-3 * mass_small
    rgb = sc.rgb(film)
    assert 0.01 < rgb.mean() < 10 and (rgb >= 0).all()
    # the megakernel pipeline (reference-shaped) gives the same film bit for bit on a sample slice
    sc.set_stats(False)
    sc.render(0, 8)
    a = sc.film()
    sc.render(0, 8, pipeline=1)
    assert np.array_equal(sc.film(), a)
    # crop parity at the full 256 spp: two 48x48 windows (object + floor, backdrop) against the oracle
    ora = O.OracleScene(desc)
    b = sc.border
    for (x0, y0) in ((936, 520), (300, 760)):
        tile = [(x0, y0, 48, 48)]
        film_c = ora.render(tiles=tile, threads=0)
        sc.render(tiles=tile)
        rg, wg = _crop_rgb(sc.film(), x0, y0, 48, 48, b)
        rc, wc = _crop_rgb(film_c, x0, y0, 48, 48, b)
        assert np.allclose(wg, wc, rtol=1e-5, atol=1e-5)
        assert l2(rg, rc) < L2_TOL, (x0, y0)
        assert np.array_equal(sc.film(), ora.render_canonical(tiles=tile, threads=0)), (x0, y0)      # ... and in the build's fixed summation order: the same bits


def test_c5_workload_as_eight_tile_shares(gpu_lib, kz, O):
    """configs[4]: 3840x2160, 1 M triangles, pmj02bn 4096 spp; a 4-spp slice. Eight tile shares (what eight GPUs would get)
    rendered one after the other on this GPU through kz_render_tiles and summed on the host in share order."""
    Wd, Hd = 3840, 2160
    desc = kz.scenes.random_triangles(1000000, Wd, Hd, 4096)
    sc = kz.Scene(desc, device=0)
    assert sc.sample_count == 4096 and sc.bvh_info()["nTris"] == 1000028
    sc.render(100, 104)
    whole = sc.film()
    # the camera rays of this frame (33 M of them) through all three kernels: pixel beams + leaf lists (default), the per-lane traversal, the packet kernel
    for kernel in (1, 2):
        sc.render(100, 104, tune={"packetPrimary": kernel})
        assert np.array_equal(sc.film(), whole), kernel
    sc.set_stats(True); sc.stats(reset=True)
    sc.render(100, 101, tiles=[(0, 0, 1920, 1088)])                   # (another pixel set: the lists are rebuilt, and counted)
    st = sc.stats(reset=True); sc.set_stats(False)
    assert st["beamPixels"] == 1920 * 1088 and 0 < st["beamListEntries"] <= 32 * 1920 * 1088 and st["droppedSamples"] == 0
    total, areas, shares = None, [], []
    for part in range(8):
        tiles = kz.shard.deal_tiles(Wd, Hd, 8, part, 64)
        areas.append(sum(t[2] * t[3] for t in tiles))
        f = sc.render_tiles(tiles, device=0, sample_begin=100, sample_end=104)
        # the gather of this share as eight GPUs would do it: the packed rects of ITS tiles (each with its filter apron), not the film
        t0 = time.perf_counter()
        packed = sc.film_tiles(tiles, device=0)
        shares.append((tiles, packed, time.perf_counter() - t0))
        if total is None:
            total = f.copy()
        else:
            assert gpu_lib.kz_film_merge(total.ctypes.data_as(kz.abi.f32p), f.ctypes.data_as(kz.abi.f32p), f.size) == 0
    assert sum(areas) == Wd * Hd and (max(areas) - min(areas)) / (Wd * Hd / 8) < 0.01
    assert np.allclose(total, whole, rtol=1e-5, atol=1e-6)
    # "D2H volume for C5 = 133 MB total". The eight packed shares are 1.13 x the film in all (64 x 64 tiles with a 2-px apron) ...
    film_bytes = whole.nbytes
    assert sum(p.nbytes for _, p, _ in shares) <= 1.14 * film_bytes
    # ... and merging them in tile order gives the film of the whole-film sums; gather + merge of the C5 frame stay within 60 ms + 60 ms here
    # (eight GPUs download their shares at the same time: the slowest single download is what a frame waits for)
    rects = sorted(((tl, si, k) for si, (tiles, _, _) in enumerate(shares) for k, tl in enumerate(tiles)), key=lambda e: (e[0][1], e[0][0]))
    offs = []
    for tiles, _, _ in shares:
        o, acc = [], 0
        for tl in tiles:
            o.append(acc); acc += (tl[2] + 2 * sc.border) * (tl[3] + 2 * sc.border) * 4
        offs.append(o)
    allp = np.concatenate([shares[si][1][offs[si][k]:offs[si][k] + (tl[2] + 2 * sc.border) * (tl[3] + 2 * sc.border) * 4] for tl, si, k in rects])
    t0 = time.perf_counter()
    merged_tiles = sc.merge_tiles(sc.empty_film(), [e[0] for e in rects], allp)
    merge_s = time.perf_counter() - t0
    assert np.array_equal(merged_tiles, whole)                         # round 6: a rect = what the tile's own pixels add; merged in tile order = the device's own resolve
    assert max(dt for _, _, dt in shares[1:]) < 0.060, [round(dt, 4) for _, _, dt in shares]      # (the first call also allocates the pinned staging buffer)
    assert merge_s < 0.120, merge_s
    print("C5 gather: per-share download %s ms, merge of the frame %.1f ms" % ([round(1e3 * dt, 1) for _, _, dt in shares], 1e3 * merge_s))
    # the in-process driver (one host thread per device, tile gather) on the devices of this box, static and dynamic dealing
    devs = list(range(min(gpu_lib.kz_device_count(), 8)))
    merged, ms = sc.render_multi(devs, sample_begin=100, sample_end=104)
    assert np.array_equal(merged, whole) and (ms > 0).all()           # 64-px tiles = the resolve's canonical grid: the SAME bits as one device's whole-frame film
    merged_dyn, _ = sc.render_multi(devs, sample_begin=100, sample_end=104, tile_dealing=1)
    assert np.array_equal(merged_dyn, merged)                         # whoever renders a tile, in whatever batch: the same rect
    if len(devs) >= 2:                                                 # equal shares of a uniform scene: the devices finish together
        assert ms.max() <= 1.10 * ms.min() + 5.0, ms
    assert sc.devices()[0] == 0 and set(sc.devices()) == set(devs)
    # a 64x64 crop of the same slice against the oracle
    ora = O.OracleScene(desc)
    x0, y0, b = 1888, 1048, sc.border
    film_c = ora.render(100, 104, tiles=[(x0, y0, 64, 64)], threads=0)
    sc.render(100, 104, tiles=[(x0, y0, 64, 64)])
    rg, wg = _crop_rgb(sc.film(), x0, y0, 64, 64, b)
    rc, wc = _crop_rgb(film_c, x0, y0, 64, 64, b)
    assert np.allclose(wg, wc, rtol=1e-5, atol=1e-6)
    assert l2(rg, rc) < L2_TOL
    assert np.array_equal(sc.film(), ora.render_canonical(100, 104, tiles=[(x0, y0, 64, 64)], threads=0))


def test_render_multi_equals_single_device(gpu_lib, kz, O):
    desc = kz.scenes.cornell_box(200, 136, 8, sampler="pmj02bn")
    sc = kz.Scene(desc)                                               # not uploaded: kz_render_multi brings the replicas up itself
    devs = list(range(min(gpu_lib.kz_device_count(), 8)))
    merged, ms = sc.render_multi(devs, tile_size=32)
    assert sc.devices() == devs
    sc.render(device=devs[0])
    assert np.allclose(merged, sc.film(), rtol=1e-5, atol=1e-6)       # (32-px tiles: another grouping of the apron sums than the film's 64-px grid)
    m64, _ = sc.render_multi(devs)                                    # the default tile = the grid of the device's own resolve: the same bits
    assert np.array_equal(m64, sc.film())
    ora = O.OracleScene(desc)
    assert l2(sc.rgb(merged), ora.rgb(ora.render(threads=0))) < L2_TOL
    merged2, _ = sc.render_multi(devs, tile_size=32)
    assert np.array_equal(merged, merged2)                            # deterministic: fixed dealing, fixed merge order
    merged3, _ = sc.render_multi(devs, tile_size=32, tile_dealing=1)  # dynamic dealing: whoever renders a tile, the same paths and (round 6) the same rect
    assert np.array_equal(merged, merged3)
    with pytest.raises(kz.abi.KzError):
        sc.render_multi([0, 0])


def test_tile_sets_change_without_rebuilding_what_they_share(gpu_lib, kz, O):
    """A change of tile set costs a tile-descriptor upload and one expansion kernel (no stream synchronisation, no host-built index), and the beam
    lists live per pixel of the FRAME: rendering the frame as two tile sets, then as one, gives the one-shot film, and the whole-frame film is the same
    bit for bit before and after (the lists built for the halves are the ones the whole frame uses)."""
    desc = kz.scenes.cornell_box(200, 136, 8, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    sc.render()
    whole = sc.film()
    left = [(0, 0, 96, 136)]
    right = [(96, 0, 64, 72), (160, 0, 40, 136), (96, 72, 64, 64)]         # ragged tiles: partial 8x8 blocks at their right / bottom edges
    sc.set_stats(True); sc.stats(reset=True)
    sc.render(tiles=left)
    sc.render(tiles=right, accumulate=True)
    st = sc.stats(reset=True); sc.set_stats(False)
    assert st["samples"] == 200 * 136 * 8
    assert np.array_equal(sc.film(), whole)                            # two tile sets accumulated = the frame at once: the film is resolved from per-pixel sums
    sc.render()
    assert np.array_equal(sc.film(), whole)
    # the pixel list of a ragged tile is what the host used to build: every pixel of the tile once, 8x8 blocks row-major
    sc.render(tiles=[(160, 0, 40, 136)])
    f = sc.film()
    b = sc.border
    assert (f[b:-b, b + 160 + b:, 3] > 0).all() and (f[:, :b + 160 - b, 3] == 0).all()
    with pytest.raises(kz.abi.KzError) as e:
        sc.render(tiles=[(0, 0, 64, 64), (32, 32, 64, 64)])
    assert "overlap" in str(e.value)
    with pytest.raises(kz.abi.KzError) as e:
        sc.render(tiles=[(0, 0, 60, 60), (59, 10, 20, 20)])                # off the 8-px grid: pairwise test
    assert "overlap" in str(e.value)


def test_packed_output_is_said_not_inferred(gpu_lib, kz):
    """with a border-less (box) filter a full tiling packs to exactly the film's size. kz_render_tiles learns which layout the buffer
    has from KzRenderOpts.packedOutput, never from the size."""
    desc = kz.scenes.cornell_box(96, 64, 4)
    desc.camera["rfilter"] = {"type": "box"}
    sc = kz.Scene(desc, device=0)
    assert sc.border == 0
    tiles = kz.shard.deal_tiles(96, 64, 1, 0, 32)
    assert len(tiles) == 6 and sc.packed_floats(tiles) == 96 * 64 * 4
    f = sc.render_tiles(tiles, device=0)                                 # whole film, although the packed rects would fit the buffer exactly
    assert np.array_equal(f, sc.film())
    packed = sc.render_tiles(tiles, device=0, packed=True)
    assert not np.array_equal(packed.reshape(f.shape), f)                # tile-major rects, not scan lines
    assert np.array_equal(sc.merge_tiles(sc.empty_film(), tiles, packed), f)
    arr = (kz.abi.KzTile * 3)(*[kz.abi.KzTile(*t) for t in tiles[:3]])
    o = kz.abi.KzRenderOpts()
    o.packedOutput = 1                                                   # packed rects of three tiles do not fill a whole-film buffer: refused, not guessed
    assert sc.lib.kz_render_tiles(sc.h, C.byref(o), arr, 3, 0, f.ctypes.data_as(kz.abi.f32p), f.size) == kz.abi.KZ_ERR_INVALID_ARG


def test_tile_dealer_takes_every_tile_once(gpu_lib, kz, O):
    """KzTileDealer (ABI v5): ONE kz_render_tiles call renders the batches it wins from the counter, passes in flight across batch boundaries; the
    tiles it reports are the tiles on its film. One taker takes the whole list; a second call on a spent counter takes nothing."""
    desc = kz.scenes.cornell_box(200, 136, 8, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    sc.render()
    whole = sc.film()
    tiles = kz.shard.deal_tiles(200, 136, 1, 0, 32)
    counter = np.zeros(1, np.uint32)
    took = sc.render_dealt(tiles, counter, takers=1, batch_tiles=3, pass_items=32 * 32 * 3 * 4)       # two passes per batch of three tiles
    assert took == tiles and counter[0] >= len(tiles)
    assert np.array_equal(sc.film(), whole)
    assert np.allclose(sc.merge_tiles(sc.empty_film(), took, sc.film_tiles(took)), whole, rtol=1e-5, atol=1e-6)      # (32-px tiles against the film's 64-px grid)
    assert sc.render_dealt(tiles, counter, takers=1, batch_tiles=3) == []
    counter[0] = 0                                                        # one pass context: the dealer paces itself on that context's film event
    assert sc.render_dealt(tiles, counter, takers=1, batch_tiles=5, passes_in_flight=1, pass_items=32 * 32 * 5 * 2) == tiles
    assert np.array_equal(sc.film(), whole)
    # two takers, one after the other on this GPU (an 8-GPU node runs them side by side): the second starts where the first was stopped
    counter[0] = 0
    a = sc.render_dealt(tiles[:], counter, takers=2, batch_tiles=0)
    fa = sc.film_tiles(a)
    assert a == tiles                                                     # (alone at the counter, it wins every batch)
    assert np.allclose(sc.merge_tiles(sc.empty_film(), a, fa), whole, rtol=1e-5, atol=1e-6)


def test_replicas_and_device_addressing(gpu_lib, kz):
    sc = kz.Scene(kz.scenes.cornell_box(48, 48, 2))
    with pytest.raises(kz.abi.KzError) as e:
        sc.render(device=0)
    assert e.value.code == kz.abi.KZ_ERR_STATE
    sc.upload(0)
    sc.upload(0)                                                      # adding the same replica twice is a no-op
    assert sc.devices() == [0]
    with pytest.raises(kz.abi.KzError) as e:
        sc.render_tiles([(0, 0, 32, 32)], device=gpu_lib.kz_device_count())      # not resident there
    assert e.value.code == kz.abi.KZ_ERR_STATE and "not resident" in str(e.value)
    sc.render()
    a = sc.film()
    sc.evict(0)
    assert sc.devices() == []
    with pytest.raises(kz.abi.KzError):
        sc.film()
    sc.upload(0)
    sc.render()
    assert np.array_equal(sc.film(), a)


def test_frames_too_large_for_64_samples_per_pass_are_rendered_in_pixel_chunks(gpu_lib, kz, O):
    """The default pass shape: every pixel x as many samples as fit - unless fewer than 64 samples would fit while the call asks for at least 64
    (C5 on one GPU): then pixel chunks x up to 256 samples. The same film bit for bit (round 6: per-pixel running sums, resolved once per call)."""
    desc = kz.scenes.cornell_box(96, 80, 64, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    npx = 96 * 80
    sc.render(pass_items=npx * 64, passes_in_flight=1)
    whole = sc.film()
    assert (sc.last_pass_info()["sppPerPass"], sc.last_pass_info()["pixelsPerPass"]) == (64, npx)
    sc.render(pass_items=npx * 16)                                    # 16 samples of every pixel would fit: chunks of 1920 pixels x 64 samples instead
    info = sc.last_pass_info()
    assert (info["passes"], info["sppPerPass"], info["pixelsPerPass"]) == (4, 64, 1920)
    assert np.array_equal(sc.film(), whole)
    sc.render(0, 32, pass_items=npx * 16, passes_in_flight=1)         # a call of fewer than 64 samples keeps the plain shape
    assert (sc.last_pass_info()["sppPerPass"], sc.last_pass_info()["pixelsPerPass"]) == (16, npx)
    # more than 64 samples of every pixel fit: a multiple of 64 (every camera-ray wave inside one pixel) when that costs no extra pass
    sc2 = kz.Scene(kz.scenes.cornell_box(96, 80, 256, sampler="pmj02bn"), device=0)
    sc2.render(pass_items=npx * 129 + 77, passes_in_flight=1)
    assert (sc2.last_pass_info()["sppPerPass"], sc2.last_pass_info()["passes"]) == (128, 2)
    sc2.render(0, 129, pass_items=npx * 129 + 77, passes_in_flight=1)  # (129 samples asked for: one pass of 129, not two of 128 + 1)
    assert (sc2.last_pass_info()["sppPerPass"], sc2.last_pass_info()["passes"]) == (129, 1)


def test_state_budget_and_pass_options(gpu_lib, kz, O):
    """KzRenderOpts: pass size, pass shape, passes in flight and the state cap are per-call options; every schedule gives the film of
    pass-at-a-time BIT FOR BIT (round 6: a pixel's samples are added to its running tap sums in sample order whatever pass brings them), and the cap is respected."""
    desc = kz.scenes.cornell_box(96, 80, 24, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    npx = 96 * 80
    sc.render(pass_items=npx * 4, passes_in_flight=1)                 # 6 passes of 4 spp, one at a time
    one_at_a_time = sc.film()
    info = sc.last_pass_info()
    assert (info["passes"], info["passesInFlight"], info["sppPerPass"], info["pixels"], info["pixelsPerPass"]) == (6, 1, 4, npx, npx)
    for n in (2, 2, 2, 3, 4, 6, 8):                                   # several in flight, repeated: an ordering bug would show as run-to-run differences
        sc.render(pass_items=npx * 4, passes_in_flight=n)
        assert np.array_equal(sc.film(), one_at_a_time), n
        assert sc.last_pass_info()["passesInFlight"] == min(n, 6)     # never more contexts than passes
    with pytest.raises(kz.abi.KzError):
        sc.render(passes_in_flight=9)
    # pass shape: n samples of as many pixels as fit instead of a few samples of every pixel. All 24 samples of a pixel in ONE pass
    # add up in sample order; 4-sample slices of pixel chunks are the sums of the 4-spp passes above, bit for bit
    sc.render(pass_items=npx * 4, passes_in_flight=1, tune={"sppPerPass": 4})
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(pass_items=1920 * 4, passes_in_flight=3, tune={"sppPerPass": 4})            # 4 pixel chunks x 6 sample slices
    info = sc.last_pass_info()
    assert (info["passes"], info["passesInFlight"], info["sppPerPass"], info["pixelsPerPass"]) == (24, 3, 4, 1920)
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(pass_items=1920 * 4, passes_in_flight=1, tune={"sppPerPass": 4})
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(pass_items=1000 * 24, passes_in_flight=2, tune={"sppPerPass": 24})          # chunks of 960 pixels (a multiple of 64), all samples at once
    info = sc.last_pass_info()
    assert (info["passes"], info["sppPerPass"], info["pixelsPerPass"]) == (8, 24, 960)
    assert np.array_equal(sc.film(), one_at_a_time)
    per_item = 8 * 16 + 16 + 12 + 20                                  # path state + sample record per item (the film's tap sums and the beam lists, per FRAME pixel, are the replica's)
    cap = 2 * npx * 3 * per_item
    sc.render(max_state_bytes=cap, passes_in_flight=2)                # room for two contexts of 3 spp
    info = sc.last_pass_info()
    assert info["sppPerPass"] == 3 and info["passes"] == 8 and info["stateBytes"] <= cap + (64 << 20)      # + the traversal kernels' overflow stacks
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(max_state_bytes=cap)                                    # nothing said about the schedule: ONE pass at a time, as large as the cap allows (round 4)
    info = sc.last_pass_info()
    assert (info["passesInFlight"], info["sppPerPass"], info["passes"]) == (1, 6, 4) and info["stateBytes"] <= cap + (64 << 20)
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render()                                                       # and without a cap: the whole call in one pass
    assert (sc.last_pass_info()["passesInFlight"], sc.last_pass_info()["sppPerPass"], sc.last_pass_info()["passes"]) == (1, 24, 1)
    sc.render(max_state_bytes=npx * per_item, passes_in_flight=1)                         # one context of one sample
    assert sc.last_pass_info()["sppPerPass"] == 1 and sc.last_pass_info()["passesInFlight"] == 1 and sc.last_pass_info()["pixelsPerPass"] == npx
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(max_state_bytes=npx * per_item * 2 // 7, passes_in_flight=1)                # not even one sample of every pixel: pixel chunks
    info = sc.last_pass_info()
    chunk = npx * 2 // 7 // 64 * 64
    assert info["sppPerPass"] == 1 and info["pixelsPerPass"] == chunk and info["passes"] == 24 * ((npx + chunk - 1) // chunk)
    assert np.array_equal(sc.film(), one_at_a_time)
    with pytest.raises(kz.abi.KzError) as e:
        sc.render(max_state_bytes=1000)
    assert e.value.code == kz.abi.KZ_ERR_OOM
    sc.render(tune={"refill": 56, "postpone": 16, "batch": 64, "traceBlocksPerCU": 4, "shadeBlocksPerCU": 3, "ldsStack": 4})
    assert np.array_equal(sc.film(), one_at_a_time)                   # knobs change the schedule, never the paths - nor, since round 6, the film's bits
    # camera rays: wave-level packet traversal (default) and the per-lane kernel find the same hits, bit for bit
    sc.render(pass_items=npx * 4, tune={"packetPrimary": 1})
    assert np.array_equal(sc.film(), one_at_a_time)
    sc.render(pass_items=npx * 4, tune={"packetPrimary": 2})
    assert np.array_equal(sc.film(), one_at_a_time)
    # film reconstruction: the tap sums with two lanes per pixel (default, round 5) are the one-lane kernel's of round 2 bit for bit, for every filter width
    sc.render(pass_items=npx * 4, tune={"filmGather": 3})
    assert np.array_equal(sc.film(), one_at_a_time)
    for filt in ("tent", "box", "mitchell"):
        d2 = kz.scenes.cornell_box(96, 80, 8)
        d2.camera["rfilter"] = {"type": filt}
        s2 = kz.Scene(d2, device=0)
        s2.render()
        f2 = s2.film()
        s2.render(tune={"filmGather": 3})
        assert np.array_equal(s2.film(), f2), filt
        s2.close()
    # the staged gather kernel of rounds 1-5 is gone (every filter width runs on the running tap sums): asking for it is an error, not a silent default
    with pytest.raises(kz.abi.KzError) as e:
        sc.render(pass_items=npx * 4, tune={"filmGather": 1})
    assert e.value.code == kz.abi.KZ_ERR_UNSUPPORTED
    # a filter of 7 taps per axis (gaussian radius 3: four lane groups per pixel) is as independent of the schedule as the 5-tap default
    d7 = kz.scenes.cornell_box(96, 80, 24, sampler="pmj02bn")
    d7.camera["rfilter"] = {"type": "gaussian", "radius": 3.0, "stddev": 0.7}
    s7 = kz.Scene(d7, device=0)
    s7.render()
    f7 = s7.film()
    s7.render(pass_items=1920 * 4, passes_in_flight=3, tune={"sppPerPass": 4})
    assert s7.border == 3 and np.array_equal(s7.film(), f7)
    tl = kz.shard.deal_tiles(96, 80, 1, 0, 64)
    assert np.array_equal(s7.merge_tiles(s7.empty_film(), tl, s7.film_tiles(tl)), f7)
    s7.close()
    ora = O.OracleScene(desc)
    assert l2(sc.rgb(one_at_a_time), ora.rgb(ora.render(threads=0))) < L2_TOL
    # the kernels of rejected experiments are not in the default library: asking for one is an error, never a silent default
    if not (gpu_lib.kz_build_flags() & 1):
        for t in ({"bvh2": 1}, {"keyStack": 2}, {"ldsTop": 5}, {"leafQueue": 2}, {"legacyTrace": 1}, {"mixedLaunch": 1}):
            with pytest.raises(kz.abi.KzError) as e:
                sc.render(tune=t)
            assert e.value.code == kz.abi.KZ_ERR_UNSUPPORTED, t


@pytest.mark.skipif(not os.path.exists(EXPERIMENTS_LIB), reason="development variant not built (scripts/build_variant.sh experiments -DKZ_EXPERIMENTS)")
def test_experiment_kernels_stay_bit_identical():
    """The rejected experiments (kz_experiments.h: BVH2 per-lane traversal, per-lane key stack, LDS top-of-tree, decoupled leaf queue,
    mixed launches, the non-persistent round-1 launches) live in a -DKZ_EXPERIMENTS build only. It is run in a child process (one
    library per process) and every variant must reproduce the product kernels' film bit for bit."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("KZ_BVH_")}      # (the development build reads its builder sweeps from KZ_BVH_*: build the product's tree)
    env["KZ_LIB_PATH"] = EXPERIMENTS_LIB
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
kz = importlib.import_module("nano-kazen_amd")
assert kz.abi.load_library().kz_build_flags() & 1
desc = kz.scenes.cornell_box(96, 80, 24, sampler="pmj02bn")
sc = kz.Scene(desc, device=0)
npx = 96 * 80
sc.render(pass_items=npx * 4, passes_in_flight=1)
ref = sc.film()
for t in ({"bvh2": 1}, {"keyStack": 2, "ldsStack": 3}, {"ldsTop": 5}, {"leafQueue": 2}, {"leafQueue": 2, "ldsStack": 2, "refill": 64, "batch": 64},
          {"ldsTop": 1000, "ldsStack": 6, "keyStack": 2}, {"legacyTrace": 1}, {"mixedLaunch": 1}, {"bvh2": 1, "mixedLaunch": 1}):
    sc.render(pass_items=npx * 4, tune=t)
    f = sc.film()
    assert (np.array_equal(f, ref) if not t.get("bvh2") else np.allclose(f, ref, rtol=2e-5, atol=1e-5)), t
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_product_library_has_no_process_global_hooks(gpu_lib, dev_lib, kz):
    """failure injection, growth delay, trace and device aliasing are process-global state - they exist in development builds only."""
    for hook in kz.abi.DEV_ONLY_EXPORTS:
        assert not hasattr(gpu_lib, hook), hook
        assert hasattr(dev_lib, hook), hook
    assert gpu_lib.kz_build_flags() == 0


def test_failed_calls_release_their_device_memory(dev_lib, kz):
    """A failure in the middle of a call (kz_debug_fail_alloc - development builds of the library - makes the nth device allocation fail) returns KZ_ERR_OOM and
    leaves no device memory behind: hipMemGetInfo before == after."""
    gpu_lib = dev_lib
    desc = kz.scenes.cornell_box(64, 64, 4)
    sc = kz.Scene(desc, device=0, lib=dev_lib)
    sc.render()
    good = sc.film()
    o = np.zeros((1000, 3), np.float32)
    d = np.tile(np.array([[0, 0, -1]], np.float32), (1000, 1))
    sc.trace_rays(o, d, 1e-3, np.inf)
    uv = np.zeros((16, 2), np.float32)
    z3 = np.tile(np.array([[0, 0, 1]], np.float32), (16, 1))

    def free_now():
        f, t = C.c_uint64(), C.c_uint64()
        assert gpu_lib.kz_device_mem_info(0, C.byref(f), C.byref(t)) == 0
        return f.value

    calls = {
        "kz_trace_rays": (5, lambda: sc.trace_rays(o, d, 1e-3, np.inf)),
        "kz_render_samples": (3, lambda: sc.render_samples(np.zeros((8, 2), np.int32), np.zeros(8, np.uint32))),
        "kz_bsdf_query": (2, lambda: sc.bsdf_query(np.zeros(16, np.int32), z3, z3, np.zeros(16, np.float32), np.zeros((16, 3), np.float32), uv)),
    }
    for name, (n_allocs, call) in calls.items():
        call()                                                        # warm: the runtime's own pools are in their steady state
        for nth in range(1, n_allocs + 1):
            before = free_now()
            gpu_lib.kz_debug_fail_alloc(nth)
            with pytest.raises(kz.abi.KzError) as e:
                call()
            gpu_lib.kz_debug_fail_alloc(0)
            assert e.value.code == kz.abi.KZ_ERR_OOM, name
            assert free_now() == before, (name, nth)
        call()                                                        # and the call works again afterwards
    # a render whose state buffers fail half way: nothing half-allocated is used afterwards
    gpu_lib.kz_device_trim(0)                 # (no pass context of an earlier scene in the device's pool: this one allocates everything itself)
    big = kz.Scene(kz.scenes.cornell_box(64, 64, 4), device=0, lib=dev_lib)
    before = free_now()
    for nth in (1, 2, 3, 4, 5, 6):            # (the film's tap sums, beam lists, beam heads, the overflow stacks, queue counters, a path-state array: what is already there is kept)
        gpu_lib.kz_debug_fail_alloc(nth)
        try:
            with pytest.raises(kz.abi.KzError):
                big.render()
        finally:
            gpu_lib.kz_debug_fail_alloc(0)
    assert free_now() <= before
    big.render()
    assert np.array_equal(big.film(), good)
    # a failed upload leaves no replica and no memory behind
    sc2 = kz.Scene(desc, lib=dev_lib)
    before = free_now()
    gpu_lib.kz_debug_fail_alloc(7)
    with pytest.raises(kz.abi.KzError):
        sc2.upload(0)
    gpu_lib.kz_debug_fail_alloc(0)
    assert sc2.devices() == [] and free_now() == before


# ---------------------------------------------------------------- round 5: the growing pass context and the benched pass size
def _free_gb(gpu_lib):
    f, t = C.c_uint64(), C.c_uint64()
    assert gpu_lib.kz_device_mem_info(0, C.byref(f), C.byref(t)) == 0
    return f.value / 1e9


def test_c4_at_the_benched_pass_size(gpu_lib, kz, O):
    """the configuration bench.py times - C4, sample indices [0, 512), ONE default pass of 2^30 (pixel, sample) items, 175 GB of path
    state - against the same slice in passes of 2^27 items and, on a crop, against the oracle."""
    if _free_gb(gpu_lib) < 200:
        pytest.skip("needs 200 GB of free device memory (one default pass of 2^30 items)")
    gpu_lib.kz_device_trim(0)                                                # (the contexts of this test are its own)
    desc = kz.scenes.random_triangles(1000000, 1920, 1080, 1024, sampler="pmj02bn", seed=1)
    sc = kz.Scene(desc, device=0)
    sc.set_stats(True)
    sc.render(0, 512, pass_items=1 << 30, passes_in_flight=1)                # (said, not earned call by call: the pass bench.py's timed steps run)
    big = sc.film()
    st = sc.stats(reset=True)
    info = sc.last_pass_info()
    assert st["samples"] == 1920 * 1080 * 512 and st["droppedSamples"] == 0 and np.isfinite(big).all()
    assert info["itemsPerPass"] == 1920 * 1080 * 512 == info["largestPassItems"] and info["passes"] == 1
    # the default EARNS its pass size: a context the call's work amortises (items / 8, at least 2^27), doubling with every further call (here up to what
    # the memory `sc` leaves allows: 2^29)
    dflt = kz.Scene(desc, device=0)
    sizes = []
    for _ in range(3):
        dflt.render(0, 512)
        sizes.append(dflt.last_pass_info()["itemsPerPass"])
    assert sizes[:2] == [1 << 27, 1 << 28] and sizes[1] <= sizes[2] <= 1920 * 1080 * 256, sizes      # (the third: 2^29 unless the memory `sc` leaves caps it)
    assert np.array_equal(dflt.film(), big)                                  # passes of 2^27 .. 2^29 earned call by call: the film of the one 2^30 pass, bit for bit
    dflt.close()
    # the 2^30-item pass as two HALVES of its pixels side by side (views of the one context: the second half's arrays start 0.53 G items into the first's), its
    # shadow rays beside its closest-hit rays: the same film, bit for bit
    sc.set_stats(False)
    sc.render(0, 512, pass_items=1 << 30, passes_in_flight=1, pass_halves=2, shadow_beside=2)
    assert sc.last_pass_info()["shadowBeside"] == 2 and sc.last_pass_info()["passes"] == 1 and np.array_equal(sc.film(), big)
    sc.set_stats(True); sc.stats(reset=True)
    sc.render(0, 512, pass_items=1 << 27, passes_in_flight=2)
    small = sc.film()
    st2 = sc.stats(reset=True)
    assert st2["samples"] == st["samples"] and st2["droppedSamples"] == 0 and sc.last_pass_info()["passes"] >= 8
    # the same paths, the same film: a pixel's samples reach its running tap sums in sample order whatever the pass size (H10)
    assert np.array_equal(big, small)
    # a 64 x 64 crop of the same slice against the oracle
    x0, y0, b = 928, 508, sc.border
    ora = O.OracleScene(desc)
    cpu = ora.render(0, 512, tiles=[(x0, y0, 64, 64)], threads=0)
    tile = kz.Scene(desc, device=0)
    tile.render(0, 512, tiles=[(x0, y0, 64, 64)])
    assert np.array_equal(tile.film(), ora.render_canonical(0, 512, tiles=[(x0, y0, 64, 64)], threads=0))      # 2 M samples of C4 at the benched sample range: the crop's film, bit for bit
    g_rgb, g_w = _crop_rgb(tile.film(), x0, y0, 64, 64, b)
    c_rgb, c_w = _crop_rgb(cpu, x0, y0, 64, 64, b)
    inner = (slice(b + 2, -b - 2), slice(b + 2, -b - 2))
    assert l2(g_rgb[inner], c_rgb[inner]) < L2_TOL and np.allclose(g_w[inner], c_w[inner], rtol=1e-5)
    # ... and the crop of the whole-frame render agrees with the tile render where the tile's own samples decide the pixel
    f_rgb, _ = _crop_rgb(big, x0, y0, 64, 64, b)
    assert l2(f_rgb[inner], g_rgb[inner]) < L2_TOL


def test_a_pass_context_that_is_still_growing_renders_the_same_film(dev_lib, kz, O):
    """The pass context grows on a side thread while the first passes of a job already run (kz_arena.cpp): behind the driver's wipe of recently released
    memory the early passes are small and the later ones larger. kz_debug_grow_delay (development builds) makes that happen on demand: the film is the
    fixed-size render's BIT FOR BIT, every sample is rendered exactly once, and the passes did grow."""
    gpu_lib = dev_lib
    desc = kz.scenes.hero_scene(1280, 720, 256, detail=1.0)                   # 236 M items: four default levels and more
    one_pass = dict(pass_items=1280 * 720 * 256, passes_in_flight=1)         # (said: the call waits for the whole context and renders ONE pass)
    # The PRODUCT library is another copy of the library in this process, with a pool of its own: the test before this one left ~260 GB of pass contexts in it, and
    # this library's default state limit is what the device reports free (round 6: minus the 2 GB a context leaves to the HIP runtime) - 41.5 GB for this job fitted
    # by a margin that the reserve then ate, some runs. That pool goes back to the driver first.
    from conftest import wait_for_wipe
    kz.abi.load_library().kz_device_trim(0)
    wait_for_wipe(gpu_lib)
    ref = kz.Scene(desc, device=0, lib=dev_lib)
    ref.render(**one_pass)
    assert ref.last_pass_info()["passes"] == 1, (ref.last_pass_info(), ref.last_grow_note())
    want = ref.film()
    ref.close()
    assert gpu_lib.kz_device_trim(0) == 0                                    # the next scene starts from an empty context
    wait_for_wipe(gpu_lib)                                                   # (... on memory the driver has finished wiping: what is under test is the delay hook's schedule)
    try:
        gpu_lib.kz_debug_grow_delay(15)
        sc = kz.Scene(desc, device=0, lib=dev_lib)
        sc.set_stats(True)
        sc.render()
        got = sc.film()
        st = sc.stats(reset=True)
        info = sc.last_pass_info()
    finally:
        gpu_lib.kz_debug_grow_delay(0)
    assert st["samples"] == 1280 * 720 * 256 and st["droppedSamples"] == 0
    assert info["passes"] >= 3 and info["firstPassItems"] < info["largestPassItems"] <= info["itemsPerPass"], info
    assert np.array_equal(got, want)
    # the same samples in ONE pass on the context the job grew: bit-identical to the reference render
    sc.render(**one_pass)
    assert sc.last_pass_info()["passes"] == 1 and np.array_equal(sc.film(), want)


def test_pass_contexts_outlive_their_replica(gpu_lib, kz):
    """kz_scene_destroy hands the replica's pass contexts to the device's pool: the next scene renders in them (no allocation, no wait for the driver's wipe
    of what the first one would have released); kz_device_trim gives the memory back."""
    gpu_lib.kz_device_trim(0)
    t0 = time.perf_counter()                                                 # (what earlier tests left in the pool goes back to the driver, which wipes it first: ~30 ms per GB)
    while _free_gb(gpu_lib) < 250 and time.perf_counter() - t0 < 20.0:
        time.sleep(0.1)
    free0 = _free_gb(gpu_lib)
    desc = kz.scenes.cornell_box(512, 512, 256)                               # 67 M items
    a = kz.Scene(desc, device=0)
    a.render()
    film_a = a.film()
    held = a.last_pass_info()["stateBytes"] / 1e9
    assert held > 5
    a.close()
    assert _free_gb(gpu_lib) < free0 - 0.9 * held                            # still held: pooled
    b = kz.Scene(desc, device=0)
    t0 = time.perf_counter()
    b.render(); b.sync()
    assert np.array_equal(b.film(), film_a)
    b.close()
    assert gpu_lib.kz_device_trim(0) == 0
    t0 = time.perf_counter()                                                 # everything back (the driver reports released memory as free once it has wiped it: ~30 ms per GB)
    while _free_gb(gpu_lib) <= free0 - 0.5 and time.perf_counter() - t0 < 10.0:
        time.sleep(0.05)
    assert _free_gb(gpu_lib) > free0 - 0.5


@pytest.mark.parametrize("radius,taps", [(0.25, 1), (1.5, 4), (2.5, 6), (3.5, 8), (4.0, 9)])
def test_every_filter_width_runs_on_the_tap_sums(gpu_lib, kz, O, radius, taps):
    """Round 6: every reconstruction filter of 1 .. 9 taps per axis (radius up to 4) runs on the running per-pixel tap sums - `kz_film_taps<TAPS, GROUPS>` with two lane
    groups per pixel up to 5 taps, four beyond; the staged gather kernel of rounds 1-5 is gone. The widths the other tests do not reach (box 2, tent 3, default 5, gaussian
    r3 7), each against the oracle, across schedules, and through the tile rects merged in tile order."""
    desc = kz.scenes.cornell_box(100, 72, 12, sampler="pmj02bn")
    desc.camera["rfilter"] = {"type": "gaussian", "radius": radius, "stddev": 0.4 * max(radius, 1.0)}
    sc = kz.Scene(desc, device=0)
    assert 2 * sc.border + 1 >= taps - 1                                  # (the apron holds the footprint)
    sc.render()
    film = sc.film()
    ora = O.OracleScene(desc)
    cpu = ora.render(threads=0)
    assert np.allclose(film[..., 3], cpu[..., 3], rtol=1e-5, atol=1e-6) and l2(sc.rgb(film), ora.rgb(cpu)) < L2_TOL
    sc.render(pass_items=100 * 72 * 4, passes_in_flight=3)
    assert np.array_equal(sc.film(), film)
    sc.render(0, 5); sc.render(5, 12, accumulate=True)
    assert np.array_equal(sc.film(), film)
    tiles = kz.shard.deal_tiles(100, 72, 1, 0, 64)
    assert np.array_equal(sc.merge_tiles(sc.empty_film(), tiles, sc.film_tiles(tiles)), film)
    sc.close()


def test_calls_on_alternating_caller_streams_are_ordered(gpu_lib, kz):
    """kz_render is asynchronous on the caller's stream. A caller that alternates between two streams without synchronising still gets calls on ONE replica in call
    order (each call clears or extends the running tap sums the previous call's last kernels read): the frame in four accumulated slices, streams alternating, equals the
    one-shot film bit for bit - repeatedly."""
    hip = C.CDLL("libamdhip64.so")                                      # (the runtime the library itself is linked against: two plain HIP streams)
    s1, s2 = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(s1)) == 0 and hip.hipStreamCreate(C.byref(s2)) == 0
    desc = kz.scenes.cornell_box(320, 200, 32, sampler="pmj02bn")
    sc = kz.Scene(desc, device=0)
    sc.render()
    whole = sc.film()
    for _ in range(3):
        for k in range(4):
            sc.render(8 * k, 8 * k + 8, accumulate=k > 0, stream=(s1 if k % 2 == 0 else s2))
        sc.sync()
        assert np.array_equal(sc.film(), whole)
    sc.close()
    hip.hipStreamDestroy(s1); hip.hipStreamDestroy(s2)


def test_shadow_rays_beside_the_closest_hit_rays(gpu_lib, kz, O):
    """KzRenderOpts::shadowBeside: the shadow kernels of a bounce on the pass context's side stream, beside the bounce's closest-hit kernel (2), in front of it on one
    stream (1), or as the library decides (0: beside up to 2^27 items per pass). What the two kernels touch is disjoint and the next shade waits for both: the film is
    the SAME BITS - on scenes with visible and invisible lights (the walk-through launch rides on the side stream too), with the EXT shade kernels, with a background
    (the last bounce extends), with passes in flight, with the counting instantiations - and equals the oracle's."""
    q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
    cases = [("cornell (visible light)", kz.scenes.cornell_box(128, 96, 16, sampler="pmj02bn"), 1),
             ("glass (EXT models)", kz.scenes.glass_scene(96, 96, 16), 1),
             ("q1 asset (invisible lights)", kz.scenes.load_npz(q1, overrides={"camera": {"width": 160, "height": 120}, "sampler": {"type": "independent", "sampleCount": 8, "seed": 3}}), 1),
             ("random triangles", kz.scenes.random_triangles(20000, 128, 96, 8, sampler="independent"), 1),
             ("sphere + environment (no lights to sample)", kz.scenes.sphere_env(96, 96, 8), 0)]
    for name, desc, expect in cases:
        sc = kz.Scene(desc, device=0)
        sc.render(shadow_beside=1)
        one_stream = sc.film()
        assert sc.last_pass_info()["shadowBeside"] == 0, name
        for rep in range(3):                                          # (repeated: a missing wait would show as run-to-run differences)
            sc.render(shadow_beside=2)
            assert sc.last_pass_info()["shadowBeside"] == expect, name
            assert np.array_equal(sc.film(), one_stream), (name, rep)
        sc.render()
        assert sc.last_pass_info()["shadowBeside"] == expect, name    # a small pass: the default is "beside"
        assert np.array_equal(sc.film(), one_stream), name
        npx = sc.width * sc.height
        sc.render(shadow_beside=2, pass_items=npx * 2, passes_in_flight=3)              # every context has its own side stream
        assert np.array_equal(sc.film(), one_stream), name
        # KzRenderOpts::passHalves = 2: the pass as two halves of its pixels side by side (two views of the context's arrays, two streams) - with either placement of
        # the shadow rays, over several passes, on a tile set, accumulating
        for kw in (dict(pass_halves=2, shadow_beside=1), dict(pass_halves=2, shadow_beside=2), dict(pass_halves=2), dict(pass_halves=2, pass_items=npx * 3, passes_in_flight=1)):
            sc.render(**kw)
            assert sc.last_pass_info()["shadowBeside"] == 2, (name, kw)
            assert np.array_equal(sc.film(), one_stream), (name, kw)
        half = sc.sample_count // 2
        sc.render(0, half, pass_halves=2); sc.render(half, sc.sample_count, accumulate=True, pass_halves=2, shadow_beside=2)
        assert np.array_equal(sc.film(), one_stream), name
        tiles = [(16, 8, 64, 48), (0, 56, 96, 32)]
        sc.render(tiles=tiles, shadow_beside=1); on_tiles = sc.film()
        sc.render(tiles=tiles, pass_halves=2)
        assert sc.last_pass_info()["shadowBeside"] == 2 and np.array_equal(sc.film(), on_tiles), name
        sc.render(pass_halves=2, pass_items=npx * 2, passes_in_flight=2)                # (ignored with passes in flight: they overlap already)
        assert sc.last_pass_info()["shadowBeside"] != 2 and np.array_equal(sc.film(), one_stream), name
        sc.set_stats(True)
        sc.render(shadow_beside=2); s2 = sc.stats(reset=True)
        assert np.array_equal(sc.film(), one_stream), name
        sc.render(shadow_beside=1); s1 = sc.stats(reset=True)
        sc.set_stats(False)
        # (node visits and triangle tests are counted per wave step: with invisible lights they vary from run to run with the order the queues were filled in)
        assert {k: v for k, v in s1.items() if k not in ("triTests", "nodeVisits")} == {k: v for k, v in s2.items() if k not in ("triTests", "nodeVisits")}, name
        ora = O.OracleScene(desc)
        assert np.array_equal(one_stream, ora.render_canonical(threads=0)), name
        with pytest.raises(kz.abi.KzError):
            sc.render(shadow_beside=3)
        with pytest.raises(kz.abi.KzError):
            sc.render(pass_halves=3)
        sc.close()


def test_large_passes_measure_how_they_run(gpu_lib, kz):
    """KzRenderOpts::shadowBeside = passHalves = 0 on passes above 2^27 items: the replica runs its first large pass on one stream, the next one of that size with the shadow
    rays beside the closest-hit rays, a third as two halves, a fourth on one stream again, and keeps what was fastest for every later one (the decision itself depends on the scene and the clock -
    the reference's q1 asset gains 7 - 10 % beside at this size, profiles/r06v_shadow_beside - so only its shape is asserted: four probes, then one answer for good).
    Explicit values bypass it; the film is the same bits."""
    q1 = os.path.join(ROOT, "tests", "golden", "q1_default_m0_r0.5.npz")
    desc = kz.scenes.load_npz(q1, overrides={"camera": {"width": 1920, "height": 1080}, "sampler": {"type": "independent", "sampleCount": 128, "seed": 0}})
    sc = kz.Scene(desc, device=0)
    sc.render(shadow_beside=1)                                        # (earns the context its size: the passes below are all one pass of 1920 x 1080 x 128 > 2^27 items)
    sc.render(shadow_beside=1)
    sc.render(shadow_beside=1)
    info = sc.last_pass_info()
    assert info["passes"] == 1 and info["largestPassItems"] == 1920 * 1080 * 128 and info["shadowBeside"] == 0
    one_stream = sc.film()
    sc.render(); assert sc.last_pass_info()["shadowBeside"] == 0      # timed in front
    sc.render(); assert sc.last_pass_info()["shadowBeside"] == 1      # timed beside
    assert np.array_equal(sc.film(), one_stream)
    sc.render(); assert sc.last_pass_info()["shadowBeside"] == 2      # timed as halves
    assert np.array_equal(sc.film(), one_stream)
    sc.render(); assert sc.last_pass_info()["shadowBeside"] == 0      # timed in front once more (the yardstick is the better of the two)
    assert sc.pass_mode_info()["kept"] is None and sc.pass_mode_info()["timed_passes"] == 4 and sc.pass_mode_info()["items"] == 1920 * 1080 * 128
    sc.render(); kept = sc.last_pass_info()["shadowBeside"]           # waits for the fourth, decides
    m = sc.pass_mode_info()
    assert m["kept"] == ("one stream", "shadow rays beside", "halves")[kept] and min(m["ms_one_stream"] + [m["ms_shadow_beside"], m["ms_halves"]]) > 10.0, m
    assert np.array_equal(sc.film(), one_stream)
    for _ in range(2):
        sc.render(); assert sc.last_pass_info()["shadowBeside"] == kept
    print("q1 asset, passes of 2^28 items: the replica keeps", ("one stream", "its shadow rays beside the closest-hit rays", "halves")[kept], m)
    sc.render(shadow_beside=2); assert sc.last_pass_info()["shadowBeside"] == 1
    sc.render(shadow_beside=1); assert sc.last_pass_info()["shadowBeside"] == 0
    sc.render(pass_halves=2); assert sc.last_pass_info()["shadowBeside"] == 2 and np.array_equal(sc.film(), one_stream)
    sc.render(pass_halves=1); assert sc.last_pass_info()["shadowBeside"] == 0      # (one option said: the other follows its plain rule - small passes beside, large ones in front)
    sc.set_stats(True); sc.render(); sc.set_stats(False)              # the counting kernels are not what was timed: in front
    assert sc.last_pass_info()["shadowBeside"] == 0 and np.array_equal(sc.film(), one_stream)
    sc.render(0, 64); assert sc.last_pass_info()["shadowBeside"] == 1  # a small pass (just under 2^27 items): beside, whatever was decided for the large ones
    sc.close()


def test_rendering_on_a_card_that_is_nearly_full(gpu_lib, kz):
    """Something else holds the card but for ~5 GB (a plain hipMalloc here): a pass context stops growing where 2 GB would no longer stay free for the HIP runtime's own
    allocations (a queue's first scratch allocation that fails ABORTS the process, profiles/r06x_memory_pressure), the call runs on what there is - one level of 2^23
    items, so 33 M items take several passes - and the film is the same bits; kz_last_grow_note says why the context stopped."""
    from conftest import wait_for_wipe
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    desc = kz.scenes.cornell_box(1920, 1080, 16, sampler="independent")
    sc = kz.Scene(desc, device=0)
    sc.render()
    roomy = sc.film()
    assert sc.last_pass_info()["passes"] == 1
    sc.close()
    gpu_lib.kz_device_trim(0)                                           # the pooled path state goes back to the driver: what is held below is really gone
    free = wait_for_wipe(gpu_lib)
    block = C.c_void_p()
    if hip.hipMalloc(C.byref(block), free - (5 << 30)) != 0:
        pytest.skip("could not take the card's free memory in one allocation")
    try:
        sc = kz.Scene(desc, device=0)
        sc.render()
        info = sc.last_pass_info()
        assert info["passes"] >= 4 and info["largestPassItems"] <= 1 << 23, info
        assert "2 GB are left to the HIP runtime" in sc.last_grow_note(), sc.last_grow_note()
        assert np.array_equal(sc.film(), roomy)
        f, t = C.c_uint64(), C.c_uint64()
        assert gpu_lib.kz_device_mem_info(0, C.byref(f), C.byref(t)) == 0 and f.value >= (3 << 29), f.value      # >= 1.5 GB still free
        sc.render(pass_halves=2, shadow_beside=2)                       # (more streams, more queues: their first kernels find their scratch)
        assert np.array_equal(sc.film(), roomy)
        sc.close()
    finally:
        hip.hipFree(block)
        gpu_lib.kz_device_trim(0)
        wait_for_wipe(gpu_lib)                                          # (the driver wipes what was held: the tests behind this one start on a quiet card)
