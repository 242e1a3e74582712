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


def merge_films(films):
    """Sum per-rank films in rank order (deterministic, H10)."""
    out = films[0].copy()
    for f in films[1:]:
        out += f
    return out
