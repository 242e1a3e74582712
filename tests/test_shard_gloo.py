"""world_size-2 tests of the N>1 path (one process per GPU, CPU process group): kz_deal_tiles + the host gather that
bench.py uses (shard.gather_tiles: gloo gather of the packed tile rects to rank 0, added in tile order = ImageBlock::put(ImageBlock&),
block.cpp:87-96) reproduce the single-process film.

* not gpu: each rank renders ITS tiles with the CPU oracle (test infrastructure) - what is under test is the host logic;
* gpu: the same two ranks render through the HIP path (kz_render_tiles on GPU 0) and the merged film is compared with the
  one-shot HIP film and with the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 160, 96, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, use_gpu):
    import importlib
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    kz = importlib.import_module("nano-kazen_amd")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    desc = kz.scenes.cornell_box(W, H, SPP)
    tiles = kz.shard.deal_tiles(W, H, world, rank, 32)
    if use_gpu:
        dev = rank % kz.abi.load_library().kz_device_count()          # distinct devices wherever the box has them
        sc = kz.Scene(desc, device=dev)
        packed = sc.render_tiles(tiles, device=dev, packed=True)       # the product path: every rank hands over the rects of ITS tiles
        dist.barrier()
        merged = kz.shard.gather_tiles(sc, tiles, packed, rank, world, 32)
    else:
        import oracle as O
        film = O.OracleScene(desc).render(tiles=tiles, threads=1)
        sc = kz.Scene(desc)                                            # host side only (no replica): sizes and kz_film_merge_tiles
        dist.barrier()
        merged = kz.shard.gather_tiles(sc, tiles, kz.shard.pack_rects_host(film, tiles, sc.border), rank, world, 32)
        whole = kz.shard.gather_films(film, rank, world)               # the same through whole films
        if rank == 0:
            assert np.array_equal(merged, whole)
    if rank == 0:
        np.save(out, merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_matches_single_process(kz, O, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, False), nprocs=2, join=True)
    merged = np.load(out)
    whole = O.OracleScene(kz.scenes.cornell_box(W, H, SPP)).render(threads=1)
    assert np.allclose(merged, whole, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_two_rank_hip_sharding_matches_one_shot(gpu_lib, kz, O, tmp_path):
    """The N > 1 HIP path itself: two processes, each with its own replica on GPU 0, tiles dealt by kz_deal_tiles, host gather."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    merged = np.load(out)
    desc = kz.scenes.cornell_box(W, H, SPP)
    sc = kz.Scene(desc, device=0)
    sc.render()
    assert np.allclose(merged, sc.film(), rtol=1e-5, atol=1e-6)
    ora = O.OracleScene(desc)
    assert float(np.sqrt(np.mean((sc.rgb(merged) - ora.rgb(ora.render(threads=0))) ** 2))) < 1e-3


@pytest.mark.parametrize("w,h,tile,world", [(1920, 1080, 128, 8), (96, 64, 32, 2), (77, 45, 32, 3)])
def test_tiles_partition_the_image(kz, w, h, tile, world):
    tiles = kz.shard.make_tiles(w, h, tile)
    cover = np.zeros((h, w), np.int32)
    per_rank = []
    for r in range(world):
        mine = kz.shard.tiles_for_rank(tiles, r, world)
        per_rank.append(sum(t[2] * t[3] for t in mine))
        for x0, y0, tw, th in mine:
            cover[y0:y0 + th, x0:x0 + tw] += 1
    assert (cover == 1).all()
    assert max(per_rank) - min(per_rank) <= 2 * tile * tile
