"""No GPU: the pass planner of kz_render (nano-kazen_amd/csrc/kz_plan.cpp - pure host arithmetic) through kz_plan_passes / kz_plan_schedule.
VERDICT r05 item 5: this table IS the documentation of the pass policy: what a call of each BASELINE config is cut into, call after call,
with and without a dealer, under a tight limit - and the one invariant of the schedule: no pass ever exceeds what its context holds."""
import ctypes as C

import numpy as np
import pytest

GB = 1 << 30
PER_ITEM = 176                                             # 8 float4 + uint4 + 3 queue words (path state) + 5 floats (sample record)
CONFIGS = {"C1": (256 * 256, 16), "C2": (512 * 512, 64), "C3": (1920 * 1080, 256), "C4": (1920 * 1080, 1024), "C5": (3840 * 2160, 4096)}
DEFAULT_LIMIT = 216 * GB                                   # 3/4 of a 288 GB card


def plan(kz, **k):
    q = kz.abi.KzPlanQuery()
    keep = None
    if "tiles" in k:                                       # a dealer's tile list: (w, h) of every tile
        offs = np.concatenate([[0], np.cumsum([w * h for w, h in k.pop("tiles")])]).astype(np.uint32)
        keep = offs
        q.dealer, q.nTiles, q.tilePixOffset = 1, len(offs) - 1, offs.ctypes.data_as(kz.abi.u32p)
        k.setdefault("nPix", int(offs[-1]))
    for a, b in k.items():
        setattr(q, a, b)
    q.limitBytes = q.limitBytes or DEFAULT_LIMIT
    ans = kz.abi.KzPlanAnswer()
    rc = kz.abi.load_library().kz_plan_passes(C.byref(q), C.byref(ans))
    return rc, ans.as_dict(), q, keep


def schedule(kz, q, avail, b0, b1, cap=4096):
    lib = kz.abi.load_library()
    av = (C.c_uint64 * len(avail))(*avail)
    out = np.zeros(4 * cap, np.uint32)
    n = C.c_uint32()
    rc = lib.kz_plan_schedule(C.byref(q), av, len(avail), b0, b1, out.ctypes.data_as(kz.abi.u32p), cap, C.byref(n))
    return rc, out[:4 * min(n.value, cap)].reshape(-1, 4)


# config, items the first context held before the call -> (contexts, pixels per pass, samples per pass, passes, may start on a growing context)
TABLE = [
    # a one-frame job (fresh process): small frames are ONE pass; C3 starts at 2^27 items, C4 / C5 (>= 2^30 items of work) at 2^28
    ("C1", 0, (1, 65536, 16, 1, 0)), ("C2", 0, (1, 262144, 64, 1, 0)),
    ("C3", 0, (1, 524288, 256, 4, 1)), ("C4", 0, (1, 1048576, 256, 8, 1)), ("C5", 0, (1, 1048576, 256, 128, 1)),
    # a process that keeps rendering: the context doubles with every call, up to 2^30 items (C4: the whole frame x 512 samples = bench.py's timed step)
    ("C4", 1 << 28, (1, 2073600, 256, 4, 1)), ("C4", 1 << 29, (1, 2073600, 512, 2, 1)), ("C4", 1 << 30, (1, 2073600, 512, 2, 1)),
    ("C3", 1 << 27, (1, 1048576, 256, 2, 1)), ("C3", 1 << 28, (1, 2073600, 256, 1, 1)),
    ("C5", 1 << 29, (1, 4194304, 256, 32, 1)), ("C5", 1 << 30, (1, 4194304, 256, 32, 1)),
]


@pytest.mark.parametrize("name,held,want", TABLE)
def test_default_policy_table(kz, name, held, want):
    npx, spp = CONFIGS[name]
    rc, a, _, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp, heldItems=held)
    assert rc == 0 and a["autoShape"] == 1
    assert (a["nCtx"], a["pixPerPass"], a["S"], a["nPasses"], a["grow"]) == want, a
    assert a["need"] == a["pixPerPass"] * a["S"] <= max(a["wantItems"], 64) and a["need"] * PER_ITEM <= DEFAULT_LIMIT


def test_explicit_options_and_limits(kz):
    npx, spp = CONFIGS["C4"]
    # passItems said: two contexts of that size, the call waits for them (no growing passes) - what rounds 1-3 benched
    rc, a, _, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp, passItems=1 << 27)
    assert rc == 0 and (a["autoShape"], a["nCtx"], a["multi"], a["S"], a["grow"], a["minStart"]) == (0, 2, 1, 64, 0, a["need"])
    # one pass of 2^30 said explicitly (scripts/profile_bench.sh)
    rc, a, _, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=512, passItems=1 << 30, passesInFlight=1)
    assert rc == 0 and (a["nCtx"], a["S"], a["nPasses"], a["need"]) == (1, 512, 1, npx * 512)
    # never more contexts than passes
    rc, a, _, _ = plan(kz, nPix=96 * 80, sampleBegin=0, sampleEnd=24, passItems=96 * 80 * 4, passesInFlight=8)
    assert rc == 0 and (a["nCtx"], a["nPasses"]) == (6, 6)
    # a tight limit: fewer samples first ...
    small = 96 * 80
    rc, a, _, _ = plan(kz, nPix=small, sampleBegin=0, sampleEnd=24, limitBytes=small * PER_ITEM * 10)
    assert rc == 0 and (a["nCtx"], a["S"], a["pixPerPass"], a["nPasses"]) == (1, 10, small, 3)
    # ... then, from one sample, fewer pixels (a multiple of 64)
    rc, a, _, _ = plan(kz, nPix=small, sampleBegin=0, sampleEnd=24, limitBytes=small * PER_ITEM // 3)
    assert rc == 0 and a["S"] == 1 and a["pixPerPass"] == small // 3 // 64 * 64
    # (C4 under ten samples' worth of the frame: the 256-sample chunk shape keeps its pixels and gives up samples)
    rc, a, _, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp, limitBytes=npx * PER_ITEM * 10)
    assert rc == 0 and (a["nCtx"], a["S"], a["pixPerPass"]) == (1, 19, 1 << 20)
    rc, a, _, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp, limitBytes=1000)
    assert rc == kz.abi.KZ_ERR_OOM
    # eight ranks (or eight aliased replicas) rehearsing on ONE card, each capped at 0.8 x 288 GB / 8: a rank's eighth of C5 in whole-share passes of 128 samples
    rc, a, _, _ = plan(kz, nPix=3840 * 2160 // 8, sampleBegin=0, sampleEnd=4096, limitBytes=int(0.8 * 288e9 / 8), heldItems=1 << 30)
    assert rc == 0 and (a["S"], a["pixPerPass"]) == (128, 3840 * 2160 // 8) and a["need"] * PER_ITEM <= 0.8 * 288e9 / 8
    # sppPerPass = n: n samples of as many pixels as fit
    rc, a, _, _ = plan(kz, nPix=96 * 80, sampleBegin=0, sampleEnd=24, passItems=1920 * 4, passesInFlight=3, sppPerPass=4)
    assert rc == 0 and (a["S"], a["pixPerPass"], a["nPasses"], a["nCtx"]) == (4, 1920, 24, 3)


def test_dealer_batches_do_not_depend_on_a_takers_history(kz):
    """the takers of one counter must resolve the SAME batch size - what a device's context has earned (its history) may size its passes, never its batches."""
    tiles = [(64, 64)] * (60 * 33) + [(64, 48)] * 60                      # C5's 64-px tiles
    got = set()
    for held in (0, 1 << 27, 1 << 28, 1 << 29, 1 << 30):
        rc, a, _, _ = plan(kz, tiles=list(tiles), sampleBegin=0, sampleEnd=1024, takers=2, heldItems=held)
        assert rc == 0 and a["nCtx"] == 2 and a["need"] <= 1 << 29
        got.add(a["batchTiles"])
    assert len(got) == 1 and got.pop() == min(2 * (1 << 29) // (64 * 64 * 1024), len(tiles) // (4 * 2))    # two passes' worth of a dealer's full-size context, at most 1 / (4 x takers) of the list
    # a short list: at most 1 / (4 x takers) of it per batch
    rc, a, _, _ = plan(kz, tiles=[(32, 32)] * 35, sampleBegin=0, sampleEnd=8, takers=2)
    assert rc == 0 and a["batchTiles"] == 4
    rc, a, _, _ = plan(kz, tiles=[(32, 32)] * 35, sampleBegin=0, sampleEnd=8, takers=2, batchTiles=3)
    assert rc == 0 and a["batchTiles"] == 3 and a["nPixSet"] == 3 * 32 * 32


def _covered(passes, b0, b1, s0, s1):
    cover = np.zeros((b1 - b0, s1 - s0), np.int32)
    for p0, w, s, sp in passes:
        cover[p0 - b0:p0 - b0 + w, s - s0:s - s0 + sp] += 1
    return cover


def test_schedule_on_a_growing_context_never_exceeds_what_is_mapped(kz):
    """frames above 2^23 pixels at 8 .. 255 spp kept every pixel of the range in the column whatever the context held - a pass of more items
    than were mapped. The schedule now narrows the column to what is there; every (pixel, sample) is still rendered exactly once, samples of a pixel in order."""
    npx, spp = 4096 * 2160, 16                                             # 8.8 M pixels, 141 M items: the case named in ADVICE
    rc, a, q, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp)
    assert rc == 0 and a["grow"] == 1 and a["minStart"] == 1 << 20
    grow = [1 << 20, 1 << 20, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, a["need"]]
    rc, passes = schedule(kz, q, grow, 0, npx)
    assert rc == 0 and len(passes) > 2                                      # (kz_plan_schedule itself fails with KZ_ERR_STATE on a pass beyond `avail`)
    assert int((passes[:, 1].astype(np.int64) * passes[:, 3]).max()) <= a["need"]
    assert passes[0][1] * passes[0][3] <= 1 << 20                          # the first pass fits the first megabyte-items
    cover = _covered(passes[:, :], 0, npx, 0, spp) if npx * spp < 2e8 else None
    assert cover is not None and (cover == 1).all()
    # a complete context: the fixed schedule "pixel chunks x sample ranges of S"
    rc, full = schedule(kz, q, [a["need"]], 0, npx)
    assert rc == 0 and len(full) == a["nPasses"]


def test_schedule_with_two_contexts_of_different_size(kz):
    """A dealer's two contexts: the column is sized from the context of ITS first pass; when the next sample range lands in a context that holds less than one
    sample of that column, the rest of the column is finished in narrower columns instead of a pass beyond the mapped memory."""
    tiles = [(64, 64)] * 64
    rc, a, q, _ = plan(kz, tiles=tiles, sampleBegin=0, sampleEnd=512, takers=1, batchTiles=16, passItems=1 << 22, passesInFlight=2)
    assert rc == 0 and a["nCtx"] == 2
    b0, b1 = 0, 16 * 64 * 64
    avail = [1 << 22, 1 << 15, 1 << 15, 1 << 22, 1 << 14, 1 << 22]       # context A complete, context B far behind
    rc, passes = schedule(kz, q, avail, b0, b1)
    assert rc == 0
    assert (_covered(passes, b0, b1, 0, 512) == 1).all()
    for p in range(b0, b1, 4099):                                          # samples of a pixel arrive in ascending order (the running tap sums need exactly that)
        mine = [s for p0, w, s, sp in passes if p0 <= p < p0 + w]
        assert mine == sorted(mine)


def test_a_call_left_to_the_library_runs_on_what_there_is(kz):
    """A small job (need <= 2^26 items) with everything left to the library waits for its context (grow = 0: nothing to gain from starting early) but does NOT fail
    when the context stops short - the card is full, or the arena keeps its reserve for the HIP runtime: minStart is 2^20 items (or the whole need, if smaller), and
    the schedule on a context that stopped at one level covers every (pixel, sample) once. An explicit pass size still insists on what it asked for."""
    npx, spp = 1920 * 1080, 16                                             # 33 M items
    rc, a, q, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp)
    assert rc == 0 and a["autoShape"] == 1 and a["grow"] == 0 and a["minStart"] == 1 << 20 and a["graceMs"] < 0
    rc, passes = schedule(kz, q, [1 << 23], 0, npx)                         # the context stopped at one level of 2^23 items
    assert rc == 0 and len(passes) >= 4 and int((passes[:, 1].astype(np.int64) * passes[:, 3]).max()) <= 1 << 23
    assert (_covered(passes, 0, npx, 0, spp) == 1).all()
    rc, a, q, _ = plan(kz, nPix=64 * 64, sampleBegin=0, sampleEnd=4)       # 16 K items: the whole need
    assert rc == 0 and a["minStart"] == a["need"] == 64 * 64 * 4
    rc, a, q, _ = plan(kz, nPix=npx, sampleBegin=0, sampleEnd=spp, passItems=1 << 24)
    assert rc == 0 and a["autoShape"] == 0 and a["minStart"] == a["need"]
