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
    """The host gather of whole films (kept for the CPU-oracle test of the host logic): every rank hands its (h+2b) x (w+2b) x 4 film to
    rank 0 over the CPU process group (gloo) and rank 0 adds them in rank order - ImageBlock::put(ImageBlock&), block.cpp:87-96.
    The product path gathers TILES (gather_tiles): this one moves `world` whole films."""
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


def gather_tiles(scene, tiles, packed, rank, world, tile=None):
    """The host gather of a multi-process launch (one rank per GPU of ONE node), SURVEY 8e: every rank hands the PACKED film rects of the tiles IT
    rendered (kz_film_download_tiles: each tile with its filter apron, 1.13 x the tile's texels) to rank 0, and rank 0 adds them into one film, rank
    after rank, tiles in each rank's list order (kz_film_merge_tiles = ImageBlock::put(ImageBlock&), block.cpp:87-96, over row bands on host threads).
    `tiles` is THIS rank's list - whatever dealt it (kz_deal_tiles, or the batches a KzTileDealer handed out): the lists travel with the rects, nothing
    is recomputed on rank 0. The rects go through shared memory (/dev/shm: the ranks share a host), the CPU process group (gloo) carries the tile lists
    and the flags; without /dev/shm the rects travel by gloo too. No RCCL, no device collective; the volume is one film in all, however many ranks.
    A failure on any rank (or on rank 0 while it merges) is raised on EVERY rank instead of leaving the others in a barrier.
    Returns the film on rank 0, None elsewhere. (`tile` is accepted for callers of the round-3 signature and ignored.)"""
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    tiles = [tuple(int(v) for v in t) for t in tiles]
    packed = np.ascontiguousarray(packed, np.float32)
    if world == 1:
        return scene.merge_tiles(scene.empty_film(), tiles, packed)
    if packed.size != scene.packed_floats(tiles):
        raise ValueError("gather_tiles: %d floats for %d tiles, kz_tiles_packed_floats says %d" % (packed.size, len(tiles), scene.packed_floats(tiles)))
    token = [os.urandom(6).hex() if rank == 0 else None]
    dist.broadcast_object_list(token, src=0)
    lists = [None] * world
    dist.all_gather_object(lists, tiles)                       # every rank's tile list, as rendered
    path = lambda r: "/dev/shm/kz_gather_%s_%d.f32" % (token[0], r)
    ok = torch.ones(1, dtype=torch.int32)
    try:
        if packed.size:
            packed.tofile(path(rank))
    except OSError:
        ok[0] = 0
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                  # every rank wrote its file (this is also the barrier behind the writes)
    film, err = None, None
    try:
        if int(ok[0]):
            if rank == 0:
                try:
                    film = scene.empty_film()
                    for r in range(world):
                        if lists[r]:
                            scene.merge_tiles(film, lists[r], np.memmap(path(r), dtype=np.float32, mode="r"))      # (mapped, not copied; ranks without tiles wrote no file)
                except Exception as e:                         # noqa: BLE001 - reported to every rank below
                    err, film = e, None
        else:                                                  # no shared memory: the rects go through the process group
            sizes = [scene.packed_floats(lists[r]) if lists[r] else 0 for r in range(world)]
            buf = np.zeros(max(max(sizes), 1), np.float32)
            buf[:packed.size] = packed
            t = torch.from_numpy(buf)
            if rank == 0:
                bufs = [torch.empty_like(t) for _ in range(world)]
                dist.gather(t, bufs, dst=0)
                try:
                    film = scene.empty_film()
                    for r in range(world):
                        if lists[r]:
                            scene.merge_tiles(film, lists[r], bufs[r].numpy()[:sizes[r]])
                except Exception as e:                         # noqa: BLE001
                    err, film = e, None
            else:
                dist.gather(t, None, dst=0)
        done = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
        dist.broadcast(done, src=0)                            # rank 0 has read everything - or failed: every rank learns which
        if not int(done[0]):
            raise RuntimeError("gather_tiles: the merge on rank 0 failed%s" % (": %s" % err if err is not None else ""))
    finally:
        try:
            os.unlink(path(rank))
        except OSError:
            pass
    return film


def shared_counter(rank, world, name=None):
    """The counter of a KzTileDealer for the ranks of one node: a uint32 in a /dev/shm file that every rank maps (Scene.render_dealt takes the array).
    Rank 0 creates and zeroes it; collective (two gloo barriers). Returns (array, path); the caller unlinks the path on rank 0 when done."""
    import os
    import numpy as np
    import torch.distributed as dist
    token = [name or (os.urandom(6).hex() if rank == 0 else None)]
    if world > 1:
        dist.broadcast_object_list(token, src=0)
    path = "/dev/shm/kz_deal_%s.u32" % token[0]
    if rank == 0:
        np.zeros(16, np.uint32).tofile(path)                   # (one cache line: the counter is word 0)
    if world > 1:
        dist.barrier()
    arr = np.memmap(path, dtype=np.uint32, mode="r+", shape=(16,))
    if world > 1:
        dist.barrier()
    return arr, path


def pack_rects_host(film, tiles, border):
    """What kz_film_download_tiles hands back, formed on the host from a whole film (numpy; for hosts without a GPU: the CPU tests of the
    gather): every tile's (h + 2b) x (w + 2b) rect in list order, a texel that an earlier tile of the list already carried written as zero."""
    import numpy as np
    taken = np.zeros(film.shape[:2], bool)
    out = []
    for (x0, y0, w, h) in tiles:
        ys, xs = slice(y0, y0 + h + 2 * border), slice(x0, x0 + w + 2 * border)
        r = film[ys, xs].copy()
        r[taken[ys, xs]] = 0.0
        taken[ys, xs] = True
        out.append(r.ravel())
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def merge_films(films):
    """Sum per-rank films in rank order (deterministic, H10)."""
    out = films[0].copy()
    for f in films[1:]:
        out += f
    return out
