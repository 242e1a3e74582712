"""world_size-2 gloo test of the N>1 path on CPU: tile partition + film merge. The product kernels need a GPU,
so each rank renders ITS tiles with the CPU oracle (test infrastructure) — what is under test is the host
logic bench.py uses: make_tiles / tiles_for_rank cover the image exactly once, and summing the per-rank films
(ImageBlock::put(ImageBlock&), block.cpp:87-96) reproduces the single-process film."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import importlib
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    kz = importlib.import_module("nano-kazen_amd")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    desc = kz.scenes.cornell_box(96, 64, 4)
    tiles = kz.shard.tiles_for_rank(kz.shard.make_tiles(96, 64, 32), rank, world)
    ora = O.OracleScene(desc)
    film = torch.from_numpy(ora.render(tiles=tiles, threads=1))
    dist.barrier()
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out, film.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_matches_single_process(kz, O, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "film.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    merged = np.load(out)
    whole = O.OracleScene(kz.scenes.cornell_box(96, 64, 4)).render(threads=1)
    assert np.allclose(merged, whole, rtol=1e-6, atol=1e-7)


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
