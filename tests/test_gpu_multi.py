"""-m gpu: the in-process multi-device driver with REAL concurrency on a box with one GPU.

kz_render_multi (kz_multi.cpp: one host thread per device, a shared dealer counter, per-thread kz_render_tiles + kz_film_download_tiles, tile-ordered merge)
is what the drop-in takes BY DEFAULT on a multi-GPU node (host/adapter/renderer_mi355x.cpp: empty device list = every visible device) - and until round 6 it
had never executed with more than one thread: every box had one GPU. The development build of the library can ALIAS devices (kz_debug_alias_devices(n):
n logical devices on the one physical GPU; replicas, pass contexts, pools, growth threads and streams are per LOGICAL device), so the threaded path, the VMM
growth threads (kz_arena.cpp) and the per-device pools run N-fold here. Since round 6 a tile's rect is what the tile's own pixels add, merged in tile order:
every film below is required to be EQUAL, bit for bit, to the single-replica film - not merely close.
The host half of the same path (threads, dealer words, merge) runs under ThreadSanitizer in tests/test_multi_host_cpu.py.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def aliased(dev_lib, kz):
    """8 logical devices on GPU 0 for the duration of a test (the hook is process-global state of the DEVELOPMENT library: set and cleared here)."""
    dev_lib.kz_debug_alias_devices(8)
    assert dev_lib.kz_device_count() == 8
    yield dev_lib
    for d in range(8):
        dev_lib.kz_device_trim(d)
    dev_lib.kz_debug_alias_devices(0)


def _budget(lib, n):
    f, t = C.c_uint64(), C.c_uint64()
    assert lib.kz_device_mem_info(0, C.byref(f), C.byref(t)) == 0
    return int(0.8 * min(f.value, t.value) / n)                          # what VERDICT names: every replica capped at 0.8 x total / N (the aliases share one card)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_aliased_replicas_render_the_single_replica_film(aliased, kz, n):
    """Static and dynamic dealing over n host threads / n replicas of ONE GPU on a C5-shaped frame (16:9, pmj02bn, the 1 M-triangle scene's little sister):
    the merged film equals the single-replica film bit for bit, every device worked, the batches a dealer handed out partition the list."""
    desc = kz.scenes.random_triangles(20000, 960, 544, 64, sampler="pmj02bn", seed=1)
    sc = kz.Scene(desc, lib=aliased)
    cap = _budget(aliased, n)
    one, ms1 = sc.render_multi([0], max_state_bytes=cap)
    assert sc.devices() == [0] and ms1[0] > 0
    sc.render(device=0, max_state_bytes=cap)
    assert np.array_equal(one, sc.film())                                # the driver with ONE device = kz_render on it (64-px tiles = the resolve's grid)
    devs = list(range(n))
    for dealing in (0, 1):
        for _ in range(2):                                               # twice: an ordering bug would show as run-to-run differences
            film, ms = sc.render_multi(devs, tile_dealing=dealing, max_state_bytes=cap)
            assert np.array_equal(film, one), (n, dealing)
            assert len(ms) == n and (ms > 0).all()
    assert sorted(sc.devices()) == devs
    # a sample slice of the same frame, static and dynamic: the same film
    a, _ = sc.render_multi(devs, sample_begin=16, sample_end=48, max_state_bytes=cap)
    b, _ = sc.render_multi(devs, sample_begin=16, sample_end=48, tile_dealing=1, max_state_bytes=cap)
    c, _ = sc.render_multi([0], sample_begin=16, sample_end=48, max_state_bytes=cap)
    assert np.array_equal(a, c) and np.array_equal(b, c)
    sc.close()


def test_dealer_batches_partition_the_list_across_threads(aliased, kz):
    """The KzTileDealer protocol with four takers running SIDE BY SIDE on four replicas (Python threads around the blocking kz_render_tiles; ctypes releases
    the GIL): the batches they took are disjoint and cover the list, the merged rects are the film."""
    import threading
    desc = kz.scenes.random_triangles(20000, 960, 544, 32, sampler="pmj02bn", seed=1)
    sc = kz.Scene(desc, lib=aliased)
    n = 4
    cap = _budget(aliased, n)
    for d in range(n):
        sc.upload(d)
    sc.render(device=0, max_state_bytes=cap)
    whole = sc.film()
    tiles = kz.shard.deal_tiles(960, 544, 1, 0, 64)
    counter = np.zeros(16, np.uint32)                                    # word 0 = the counter, word 1 = `agreed`
    took, errs = [None] * n, []

    def taker(d):
        try:
            took[d] = sc.render_dealt(tiles, counter, takers=n, device=d, max_state_bytes=cap)
        except Exception as e:                                           # noqa: BLE001
            errs.append((d, e))

    th = [threading.Thread(target=taker, args=(d,)) for d in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    flat = [t for part in took for t in part]
    assert sorted(flat) == sorted(tiles) and len(set(flat)) == len(tiles)          # a partition of the list
    assert sum(1 for part in took if part) >= 2                          # more than one taker got work
    film = sc.empty_film()
    order = sorted(((t, d) for d in range(n) for t in took[d]), key=lambda e: (e[0][1], e[0][0]))
    rects = {d: (took[d], sc.film_tiles(took[d], device=d)) for d in range(n) if took[d]}
    offs = {}
    for d, (tl, _) in rects.items():
        acc = 0
        for t in tl:
            offs[(d, t)] = acc
            acc += (t[2] + 2 * sc.border) * (t[3] + 2 * sc.border) * 4
    packed = np.concatenate([rects[d][1][offs[(d, t)]:offs[(d, t)] + (t[2] + 2 * sc.border) * (t[3] + 2 * sc.border) * 4] for t, d in order])
    assert np.array_equal(sc.merge_tiles(film, [t for t, _ in order], packed), whole)
    sc.close()


def test_one_failing_device_thread_fails_the_call(aliased, kz):
    """A device allocation fails inside ONE of the four device threads of kz_render_multi (kz_debug_fail_device: the countdown is handed to the thread that addresses
    that replica) while the other three render: the call returns that device's KZ_ERR_OOM with its message, nobody hangs, and the next call renders the reference film."""
    desc = kz.scenes.cornell_box(256, 192, 16, sampler="pmj02bn")
    ref = kz.Scene(desc, lib=aliased)
    good, _ = ref.render_multi([0], max_state_bytes=_budget(aliased, 4))
    ref.close()
    for dealing in (0, 1):
        for nth in (1, 2, 4):                                            # the tile set's pixel list, ..., a later allocation of the first render on a fresh replica
            sc = kz.Scene(desc, lib=aliased)
            aliased.kz_debug_fail_device(2, nth)
            try:
                with pytest.raises(kz.abi.KzError) as e:
                    sc.render_multi([0, 1, 2, 3], tile_dealing=dealing, max_state_bytes=_budget(aliased, 4))
            finally:
                aliased.kz_debug_fail_device(2, 0)
            assert e.value.code == kz.abi.KZ_ERR_OOM and "device 2" in str(e.value), (dealing, nth, str(e.value))
            again, ms = sc.render_multi([0, 1, 2, 3], tile_dealing=dealing, max_state_bytes=_budget(aliased, 4))
            assert np.array_equal(again, good) and (ms > 0).all(), (dealing, nth)
            sc.close()


def test_a_failing_replica_fails_the_call_without_hanging_the_others(aliased, kz):
    """kz_debug_fail_alloc is per calling thread, so a failure is injected into ONE device's thread through its budget instead: a replica whose state cap cannot
    hold 64 items fails (KZ_ERR_OOM) while the other threads render; the call returns that device's error, the next call is whole again."""
    desc = kz.scenes.cornell_box(256, 192, 16, sampler="pmj02bn")
    sc = kz.Scene(desc, lib=aliased)
    good, _ = sc.render_multi([0, 1, 2, 3], max_state_bytes=_budget(aliased, 4))
    with pytest.raises(kz.abi.KzError) as e:
        sc.render_multi([0, 1, 2, 3], max_state_bytes=1000)             # every thread fails: each error is carried from its thread to the caller's
    assert e.value.code == kz.abi.KZ_ERR_OOM and "device 0" in str(e.value)
    with pytest.raises(kz.abi.KzError) as e:
        sc.render_multi([0, 1, 9], max_state_bytes=_budget(aliased, 4))  # device 9 does not exist (8 aliases): the upload fails before any thread starts
    assert "device 9" in str(e.value)
    again, ms = sc.render_multi([0, 1, 2, 3], tile_dealing=1, max_state_bytes=_budget(aliased, 4))
    assert np.array_equal(again, good) and (ms > 0).all()
    sc.close()


def test_injected_allocation_failure_inside_a_growing_context(dev_lib, kz):
    """kz_debug_fail_alloc used to be taken back from the arena when the call returned while the growth thread was still mapping - a failure
    injected beyond the first levels never fired. It stays armed on the arena now: the context stops growing where the failure hits, the call succeeds on what
    there is, and the note says why."""
    from conftest import wait_for_wipe
    dev_lib.kz_device_trim(0)
    wait_for_wipe(dev_lib)
    desc = kz.scenes.hero_scene(1280, 720, 256, detail=1.0)                # 236 M items: 28 levels of 2^23
    sc = kz.Scene(desc, device=0, lib=dev_lib)
    try:
        dev_lib.kz_debug_grow_delay(5)
        dev_lib.kz_debug_fail_alloc(17 * 6 + 3)                           # (17 arrays per level, a handful of other allocations of the call first: fails in the sixth or seventh level, long after the first pass has started)
        sc.render()
        info, note = sc.last_pass_info(), sc.last_grow_note()
    finally:
        dev_lib.kz_debug_fail_alloc(0)
        dev_lib.kz_debug_grow_delay(0)
    assert "kz_debug_fail_alloc" in note and info["contextItems"] in (5 << 23, 6 << 23) and info["firstPassItems"] < info["largestPassItems"] <= info["contextItems"], (info, note)
    got = sc.film()
    sc.close()
    dev_lib.kz_device_trim(0)
    ref = kz.Scene(desc, device=0, lib=dev_lib)
    ref.render(pass_items=1280 * 720 * 256, passes_in_flight=1)
    assert np.array_equal(got, ref.film())                               # small passes on a stunted context: the same film, bit for bit
    ref.close()


def test_cxx_adapter_renders_on_an_alias_list(aliased, kz, tmp_path):
    """The C++ side of the drop-in (mi355x::DeviceScene::render of host/adapter/renderer_mi355x.cpp, compiled unchanged) on a list of four
    aliased devices - kz_render_multi with four host threads - static and dynamic dealing, against the Python path on one replica: the same film, bit for bit."""
    import shutil
    host = os.path.join(ROOT, "nano-kazen_amd", "host")
    xml_dir = tmp_path / "xml"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "xml"), xml_dir)
    xml = xml_dir / "mini.xml"                                           # the hand-written fixture at 200 x 136: 4 x 3 tiles of 64 px, ragged at the right and bottom edges
    xml.write_text(xml.read_text().replace('name="width" value="48"', 'name="width" value="200"').replace('name="height" value="32"', 'name="height" value="136"'))
    src = tmp_path / "alias_main.cpp"
    src.write_text(r"""
#include <kazen/renderer.h>
#include <kazen/scene.h>
#include <kazen/mi355x.h>
#include "kazen_sceneio.hpp"
#include <cstdio>
#include <cstring>
extern "C" void kz_debug_alias_devices(int n);
using namespace kazen;
int main(int argc, char **argv) {
    kz_debug_alias_devices(4);
    try {
        std::unique_ptr<Object> root(loadFromXML(argv[1]));
        Scene *scene = static_cast<Scene *>(root.get());
        mi355x::DeviceScene ds(scene);
        ImageBlock result(scene->getCamera()->getOutputSize(), scene->getCamera()->getReconstructionFilter());
        KzRenderOpts o;
        std::memset(&o, 0, sizeof o);
        o.maxStateBytes = (uint64_t)30 << 30;
        o.tileDealing = argc > 3 ? 1 : 0;
        ds.render(result, {0, 1, 2, 3}, &o);
        FILE *f = std::fopen(argv[2], "wb");
        std::fwrite(result.data(), 4 * sizeof(float), result.size(), f);
        std::fclose(f);
    } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
""")
    exe = str(tmp_path / "alias_main")
    libdir = os.path.dirname(kz.abi.DEV_LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(host, "mirror_tree"), "-I" + os.path.join(host, "adapter"), "-I" + host,
                           "-o", exe, str(src), os.path.join(host, "adapter", "renderer_mi355x.cpp"), "-L" + libdir, "-lkazen_mi355x", "-Wl,-rpath," + libdir])
    outs = []
    for extra in ([], ["dyn"]):
        out = str(tmp_path / ("film%d.bin" % len(outs)))
        r = subprocess.run([exe, str(xml), out] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(out, np.float32))
    assert np.array_equal(outs[0], outs[1])
    sc = kz.Scene(kz.xmlscene.load_xml(str(xml)), device=0, lib=aliased)
    sc.render(max_state_bytes=30 << 30)
    assert (sc.width, sc.height) == (200, 136) and np.array_equal(outs[0].reshape(sc.film().shape), sc.film())
    sc.close()


def test_bench_parity_crc_is_independent_of_the_cut(aliased, gpu_lib, kz):
    """bench.py's `parity` sub-record: the crc32 of a fixed 64 x 64 crop of C4's film at sample indices [0, 64) is ONE number (bench.PARITY_CRC_N1, measured at N = 1 on
    the product library) however the frame was cut: here through the development library on one replica, through 2 / 4 / 8 aliased replicas with tiles dealt
    beforehand and taken from a counter, and from the packed rects of eight tile shares merged in tile order (what eight ranks hand to rank 0)."""
    sys.path.insert(0, ROOT)
    import bench
    desc = kz.scenes.random_triangles(bench.NTRIS, bench.W, bench.H, bench.PARITY_SPP_TABLE, sampler="pmj02bn", seed=1)
    prod = kz.Scene(desc, device=0)                                       # the product library, as bench.py at N = 1
    prod.render(0, 64)
    assert bench.film_crc(prod.film(), prod.border) == bench.PARITY_CRC_N1
    entries = []                                                         # eight ranks' worth of rects, each rank rendering only ITS tiles
    for part in range(8):
        tiles = kz.shard.deal_tiles(bench.W, bench.H, 8, part, bench.TILE)
        packed = prod.render_tiles(tiles, device=0, sample_begin=0, sample_end=64, packed=True)
        entries += kz.shard._rects_of(tiles, packed, prod.border)
    merged = prod.merge_rects(prod.empty_film(), sorted(entries, key=lambda e: (e[0][1], e[0][0])))
    assert bench.film_crc(merged, prod.border) == bench.PARITY_CRC_N1
    prod.close()
    gpu_lib.kz_device_trim(0)
    sc = kz.Scene(desc, lib=aliased)
    for n in (1, 2, 4, 8):
        cap = _budget(aliased, n)
        for dealing in (0, 1):
            film, _ = sc.render_multi(list(range(n)), sample_begin=0, sample_end=64, tile_dealing=dealing, max_state_bytes=cap)
            assert bench.film_crc(film, sc.border) == bench.PARITY_CRC_N1, (n, dealing)
    sc.close()


def test_a_frame_of_more_pixels_than_a_growing_context_holds(dev_lib, kz):
    """frames above 2^23 pixels at 8 .. 255 spp kept EVERY pixel of the frame in a pass's column whatever the growing context held - one sample
    of 8.8 M pixels planned on a context of 2^20 items: writes into reserved, unmapped address space. The planner now narrows the column to what is mapped
    (tests/test_plan_cpu.py has the arithmetic); this is the same case on the device, slowed-down growth and all - alone and under a dealer with two contexts."""
    from conftest import wait_for_wipe
    dev_lib.kz_device_trim(0)
    wait_for_wipe(dev_lib)
    desc = kz.scenes.cornell_box(4096, 2160, 16, sampler="pmj02bn")       # 8.8 M pixels x 16 spp = 141 M items
    ref = kz.Scene(desc, device=0, lib=dev_lib)
    ref.render(pass_items=1 << 26, passes_in_flight=1)                    # (said: every pass waits for its full context)
    want = ref.film()
    ref.close()
    dev_lib.kz_device_trim(0)
    wait_for_wipe(dev_lib)
    sc = kz.Scene(desc, device=0, lib=dev_lib)
    try:
        dev_lib.kz_debug_grow_delay(10)
        sc.render()                                                        # library defaults: passes start on 2^20 mapped items
        info = sc.last_pass_info()
        got = sc.film()
        assert info["firstPassItems"] <= 1 << 23 < 4096 * 2160 and info["passes"] > 2, info
        assert np.array_equal(got, want)
        tiles = kz.shard.deal_tiles(4096, 2160, 1, 0, 64)
        counter = np.zeros(16, np.uint32)
        dev_lib.kz_device_trim(0)
        took = sc.render_dealt(tiles, counter, takers=1)                   # a dealer: two contexts growing at their own pace
        assert sorted(took) == sorted(tiles) and np.array_equal(sc.film(), want)
    finally:
        dev_lib.kz_debug_grow_delay(0)
    sc.close()


def test_a_call_that_fails_in_the_middle_leaves_nothing_running(dev_lib, kz):
    """Found by scripts/dev/fail_sweep.py: with two passes in flight, a call whose SECOND context could not be allocated returned its error while the first pass was still
    queued on an internal stream - and that pass's film stage then added to tap sums the next call had already cleared (the next film carried 10 samples per pixel instead
    of 8). A failed call now waits for what it has launched. Swept over the allocations of a fresh scene's first call, with every pass mode: each call either goes through
    or fails with KZ_ERR_OOM, and the render after it gives the film."""
    desc = kz.scenes.glass_scene(160, 128, 8)
    sc = kz.Scene(desc, device=0, lib=dev_lib)
    sc.render(shadow_beside=1, pass_halves=1)
    ref = sc.film()
    sc.close()
    failed = 0
    try:
        for kw in (dict(pass_items=160 * 128 * 2, passes_in_flight=2), dict(pass_items=160 * 128 * 2, passes_in_flight=2, pass_halves=2), dict(pass_halves=2, shadow_beside=2)):
            for n in range(36, 60):
                dev_lib.kz_device_trim(0)                                  # a fresh scene on an empty pool: its first call allocates everything
                dev_lib.kz_debug_fail_alloc(n)
                sc = kz.Scene(desc, device=0, lib=dev_lib)
                try:
                    sc.render(**kw)
                    sc.sync()
                    assert np.array_equal(sc.film(), ref), (kw, n)
                except kz.abi.KzError as e:
                    assert e.code == 6, (kw, n, str(e))                     # KZ_ERR_OOM
                    failed += 1
                dev_lib.kz_debug_fail_alloc(0)
                sc.render(**kw)
                assert np.array_equal(sc.film(), ref), (kw, n)
                sc.close()
    finally:
        dev_lib.kz_debug_fail_alloc(0)
    assert failed >= 10, failed
