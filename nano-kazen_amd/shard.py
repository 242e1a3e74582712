"""Tile sharding of the image across GPUs: the analogue of the reference's
tbb::parallel_for over 32x32 ImageBlocks (src/kazen/renderer.cpp:94-127), one level up.

Every (pixel, sample) path is independent and samplers re-seed per (pixel, sample index)
(sampler.cpp:43-46, 333-337), so a tile renders to the same values on any GPU. Each rank
accumulates its tiles (with their filter aprons) into its own full-size film; the films are then
summed - ImageBlock::put(ImageBlock&) (block.cpp:87-96) - in tile order. No data-path collective.
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


def open_gather(rank, world):
    """Collective, BEFORE the clocks: the ranks agree on the names of the /dev/shm files of one gather_tiles (a token from rank 0). A gather that is
    handed the session needs no collective of its own before rank 0 starts merging, so a rank that finishes early does not wait for the slowest."""
    import os
    import torch.distributed as dist
    token = [os.urandom(6).hex() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(token, src=0)
    return token[0]


def _rects_of(tiles, packed, border):
    """[(tile, view of its packed rect)] for a rank's list."""
    out, off = [], 0
    for t in tiles:
        n = (t[2] + 2 * border) * (t[3] + 2 * border) * 4
        out.append((t, packed[off:off + n]))
        off += n
    return out


def gather_tiles(scene, tiles, packed, rank, world, tile=None, session=None):
    """The host gather of a multi-process launch (one rank per GPU of ONE node), SURVEY 8e: every rank hands the PACKED film rects of the tiles IT
    rendered (kz_film_download_tiles: each tile with its filter apron - what the tile's own pixels add, an ImageBlock of the tile - 1.13 x the tile's texels) to
    rank 0, and rank 0 adds ALL rects into one film in ROW-MAJOR TILE ORDER (kz_film_merge_rects = ImageBlock::put(ImageBlock&), block.cpp:87-96, over row bands
    on host threads). `tiles` is THIS rank's list - whatever dealt it (kz_deal_tiles, or the batches a KzTileDealer handed out): the list travels with the
    rects (one file per rank in /dev/shm: the ranks share a host), nothing is recomputed on rank 0.
    Round 6: the order of the additions is the TILE order whatever rank rendered a tile and whenever its file arrived - the merged film is, bit for bit, the
    film ONE device resolves for the same frame (64-px tiles) and the film kz_render_multi returns, for any number of ranks, static or dynamic dealing (H10).
    (Rounds 3-5 merged rank after rank as the files arrived - overlapping the tail of the render by a few milliseconds, at the price of a film whose last bits
    depended on the dealing.) A rank writes its file as soon as it has its rects and tells rank 0 with one point-to-point message; rank 0 maps the files as the
    messages arrive and merges once the last one is there. There is no collective in front of the merge (given a `session` from open_gather; without one the
    token broadcast is that collective). Without /dev/shm the list and the rects travel by gloo, point to point. No RCCL, no device collective; the volume is
    one film in all, however many ranks.
    A failure on ANY rank - a packed buffer of the wrong size, a file that cannot be written, the merge on rank 0 - is raised on EVERY rank: each rank sends
    exactly one status message and then waits for rank 0's verdict, rank 0 always receives world - 1 of them before it broadcasts it, nobody is left waiting.
    Returns the film on rank 0, None elsewhere. (`tile` is accepted for callers of the round-3 signature and ignored.)"""
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    tiles = [tuple(int(v) for v in t) for t in tiles]
    packed = np.ascontiguousarray(packed, np.float32)
    order = lambda e: (e[0][1], e[0][0])
    if world == 1:
        return scene.merge_rects(scene.empty_film(), sorted(_rects_of(tiles, packed, scene.border), key=order))
    token = session or open_gather(rank, world)
    path = lambda r: "/dev/shm/kz_gather_%s_%d.bin" % (token, r)
    SHM, GLOO, FAILED = 1, 2, -1
    err = None
    try:
        want = scene.packed_floats(tiles) if tiles else 0
        if packed.size != want:
            raise ValueError("gather_tiles: rank %d hands over %d floats for %d tiles, kz_tiles_packed_floats says %d" % (rank, packed.size, len(tiles), want))
    except Exception as e:                                     # noqa: BLE001 - reported to every rank below
        err = e
    if rank != 0:
        mode = FAILED if err is not None else SHM
        if mode == SHM and tiles:
            try:
                with open(path(rank), "wb") as f:
                    np.asarray(tiles, np.int32).tofile(f)
                    packed.tofile(f)
            except OSError:
                mode = GLOO
        dist.send(torch.tensor([mode, len(tiles), packed.size], dtype=torch.int64), dst=0)
        if mode == GLOO and tiles:
            dist.send(torch.from_numpy(np.asarray(tiles, np.int32).reshape(-1).copy()), dst=0)
            dist.send(torch.from_numpy(packed), dst=0)
        film = None
    else:
        film = None
        entries = [] if err is not None else _rects_of(tiles, packed, scene.border)
        keep = []                                              # the mapped files / received buffers stay alive until the merge is done
        for r in range(1, world):                              # a rank's message is there when its file is
            head = torch.zeros(3, dtype=torch.int64)
            dist.recv(head, src=r)
            mode, n_tiles, n_floats = (int(v) for v in head)
            try:
                if mode == FAILED:
                    raise RuntimeError("rank %d could not hand over its tile rects" % r)
                if n_tiles == 0:
                    continue
                if mode == GLOO:
                    tl, fl = torch.zeros(4 * n_tiles, dtype=torch.int32), torch.zeros(n_floats, dtype=torch.float32)
                    dist.recv(tl, src=r)
                    dist.recv(fl, src=r)
                    tl, fl = tl.numpy(), fl.numpy()
                else:
                    raw = np.memmap(path(r), dtype=np.uint8, mode="r")      # (mapped, not copied)
                    tl, fl = np.frombuffer(raw, np.int32, 4 * n_tiles), np.frombuffer(raw, np.float32, n_floats, 16 * n_tiles)
                    keep.append(raw)
                keep.append(fl)
                if scene.packed_floats([tuple(int(v) for v in t) for t in tl.reshape(-1, 4)]) != n_floats:
                    raise ValueError("rank %d sent %d floats for its %d tiles" % (r, n_floats, n_tiles))
                entries += _rects_of([tuple(int(v) for v in t) for t in tl.reshape(-1, 4)], fl, scene.border)
            except Exception as e:                             # noqa: BLE001
                err = err or e
        try:
            if err is None:
                film = scene.merge_rects(scene.empty_film(), sorted(entries, key=order))
        except Exception as e:                                 # noqa: BLE001
            err = e
        if err is not None:
            film = None
    done = torch.tensor([0 if (rank == 0 and err is not None) else 1], dtype=torch.int32)
    dist.broadcast(done, src=0)                                # rank 0 has read everything - or something failed somewhere: every rank learns which
    try:
        os.unlink(path(rank))
    except OSError:
        pass
    if not int(done[0]) or err is not None:
        raise RuntimeError("gather_tiles failed%s" % (": %s" % err if err is not None else " on another rank"))
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
