"""Tile sharding of the image across GPUs: the analogue of the reference's
tbb::parallel_for over 32x32 ImageBlocks (src/kazen/renderer.cpp:94-127), one level up.

Every (pixel, sample) path is independent and samplers re-seed per (pixel, sample index)
(sampler.cpp:43-46, 333-337), so a tile renders to the same values on any GPU. Each rank
accumulates its tiles (with their filter aprons) into its own full-size film; the films are then
summed - ImageBlock::put(ImageBlock&) (block.cpp:87-96) - in rank order. No data-path collective.
"""


def make_tiles(width, height, tile=128):
    """Row-major list of (x0, y0, w, h); tile is a multiple of the reference's 32-px block."""
    assert tile % 32 == 0
    out = []
    for y in range(0, height, tile):
        for x in range(0, width, tile):
            out.append((x, y, min(tile, width - x), min(tile, height - y)))
    return out


def tiles_for_rank(tiles, rank, world):
    """Round-robin assignment (tile i -> rank i % world)."""
    return [t for i, t in enumerate(tiles) if i % world == rank]


def deal_tiles(width, height, parts, part, tile=64):
    """kz_deal_tiles (the dealing kz_render_multi uses): tiles go, largest first, to the part with the least area so far;
    row-major order inside a part. Every rank of a multi-process launch calls this with its own `part`."""
    import ctypes as C
    from . import abi
    lib = abi.load_library()
    n = C.c_uint32()
    cap = ((width + tile - 1) // tile) * ((height + tile - 1) // tile)
    buf = (abi.KzTile * cap)()
    abi.check(lib, lib.kz_deal_tiles(width, height, tile, parts, part, buf, cap, C.byref(n)))
    return [(buf[i].x0, buf[i].y0, buf[i].w, buf[i].h) for i in range(n.value)]


def gather_films(film, rank, world):
    """The host gather of a multi-process launch (one rank per GPU): every rank hands its host film to rank 0 over the CPU
    process group (gloo) and rank 0 adds them in rank order - ImageBlock::put(ImageBlock&), block.cpp:87-96. No RCCL, no
    device collective: the films are (h+2b) x (w+2b) x 4 floats, 33 MB at 1920x1080. Returns the merged film on rank 0, None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if world == 1:
        return film
    t = torch.from_numpy(np.ascontiguousarray(film))
    if rank == 0:
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, bufs, dst=0)
        return merge_films([b.numpy() for b in bufs])
    dist.gather(t, None, dst=0)
    return None


def merge_films(films):
    """Sum per-rank films in rank order (deterministic, H10)."""
    out = films[0].copy()
    for f in films[1:]:
        out += f
    return out
