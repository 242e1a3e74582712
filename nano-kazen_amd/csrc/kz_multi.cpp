// kz_multi.cpp - host code only: the reference's BlockGenerator and its merge of ImageBlocks (block.cpp:87-148) one level up - dealing image tiles over
// devices (kz_deal_tiles), adding packed tile rects into the frame in tile order (kz_film_merge_tiles), and the in-process multi-device driver
// kz_render_multi (renderer.cpp:94-127: one host thread per device instead of one TBB task per block).
#include "kz_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

static size_t packedFloats(const KzParams &P, const KzTile *tiles, uint32_t nTiles) {
    size_t n = 0;
    for (uint32_t t = 0; t < nTiles; ++t) n += (size_t)(tiles[t].w + 2 * P.border) * (size_t)(tiles[t].h + 2 * P.border) * 4;
    return n;
}

extern "C" {

int kz_deal_tiles(int32_t width, int32_t height, int32_t tileSize, uint32_t nParts, uint32_t part, KzTile *out, uint32_t cap, uint32_t *count) {
    if (tileSize == 0) tileSize = 64;
    if (width <= 0 || height <= 0 || tileSize < 32 || tileSize % 32 || nParts == 0 || part >= nParts || !count)
        return kz_fail(KZ_ERR_INVALID_ARG, "kz_deal_tiles: bad argument (tile size must be a positive multiple of 32, part < nParts)");
    std::vector<KzTile> tiles;
    for (int y = 0; y < height; y += tileSize)
        for (int x = 0; x < width; x += tileSize) tiles.push_back(KzTile{x, y, std::min(tileSize, width - x), std::min(tileSize, height - y)});
    // largest first (stable: row-major order among equals), each to the part with the least area so far (ties: the lower part)
    std::vector<uint32_t> order(tiles.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (int64_t)tiles[a].w * tiles[a].h > (int64_t)tiles[b].w * tiles[b].h; });
    std::vector<int64_t> area(nParts, 0);
    std::vector<uint32_t> mine;
    for (uint32_t i : order) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < nParts; ++p) if (area[p] < area[best]) best = p;
        area[best] += (int64_t)tiles[i].w * tiles[i].h;
        if (best == part) mine.push_back(i);
    }
    std::sort(mine.begin(), mine.end());                                // back to row-major order within the part
    *count = (uint32_t)mine.size();
    if (mine.size() > cap || (!out && !mine.empty())) return kz_fail(KZ_ERR_INVALID_ARG, "kz_deal_tiles: %zu tiles, room for %u", mine.size(), cap);
    for (size_t i = 0; i < mine.size(); ++i) out[i] = tiles[mine[i]];
    return KZ_OK;
}

int kz_film_merge(float *dst, const float *src, size_t nFloats) {
    if (!dst || !src) return kz_fail(KZ_ERR_INVALID_ARG, "null film");
    for (size_t i = 0; i < nFloats; ++i) dst[i] += src[i];
    return KZ_OK;
}

// ImageBlock::put(ImageBlock&) (block.cpp:87-96) for a LIST of blocks: rect[t] (the packed (h + 2b) x (w + 2b) x 4 floats of tiles[t]) is added to the film in list
// order. Rows of the film are cut into bands, one host thread per band (disjoint destinations: no lock, and every film texel still receives
// its rects in list order, so the result does not depend on the number of threads). `clear`: every band zeroes itself first.
static void mergeRects(float *film, int width, int height, int border, const KzTile *tiles, const float *const *rect, uint32_t nTiles, int nThreads, bool clear) {
    const int cols = width + 2 * border, rows = height + 2 * border;
    int nt = nThreads > 0 ? nThreads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::max(1, std::min(nt, rows / 8 + 1));
    auto band = [&](int r0, int r1) {
        if (clear) std::memset(film + (size_t)r0 * cols * 4, 0, (size_t)(r1 - r0) * cols * 4 * sizeof(float));      // (every thread clears its own band: 133 MB at C5)
        for (uint32_t t = 0; t < nTiles; ++t) {
            const KzTile &tl = tiles[t];
            const int rw = tl.w + 2 * border, y0 = std::max(tl.y0, r0), y1 = std::min(tl.y0 + tl.h + 2 * border, r1);
            for (int y = y0; y < y1; ++y) {
                float *d = film + ((size_t)y * cols + tl.x0) * 4;
                const float *s = rect[t] + (size_t)(y - tl.y0) * rw * 4;
                for (int i = 0; i < rw * 4; ++i) d[i] += s[i];
            }
        }
    };
    if (nt == 1) { band(0, rows); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(band, (int)((int64_t)rows * i / nt), (int)((int64_t)rows * (i + 1) / nt));
    for (auto &t : th) t.join();
}

static int checkMergeArgs(const float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, uint32_t nTiles) {
    if (!film || (nTiles && !tiles) || width <= 0 || height <= 0 || border < 0) return kz_fail(KZ_ERR_INVALID_ARG, "film merge: null or bad argument");
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > width || tl.y0 + tl.h > height) return kz_fail(KZ_ERR_INVALID_ARG, "tile %u outside the %dx%d image", t, width, height);
    }
    return KZ_OK;
}

int kz_film_merge_tiles(float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, uint32_t nTiles, const float *packed, size_t nFloats, int32_t nThreads) {
    if (nTiles && !packed) return kz_fail(KZ_ERR_INVALID_ARG, "kz_film_merge_tiles: null packed buffer");
    if (const int rc = checkMergeArgs(film, width, height, border, tiles, nTiles)) return rc;
    std::vector<const float *> rect(nTiles);
    size_t off = 0;
    for (uint32_t t = 0; t < nTiles; ++t) { rect[t] = packed + off; off += (size_t)(tiles[t].w + 2 * border) * (size_t)(tiles[t].h + 2 * border) * 4; }
    if (off != nFloats) return kz_fail(KZ_ERR_INVALID_ARG, "packed buffer holds %zu floats, the tiles need %zu", nFloats, off);
    mergeRects(film, width, height, border, tiles, rect.data(), nTiles, nThreads, false);
    return KZ_OK;
}

// The same merge for rects that lie in DIFFERENT buffers (a multi-process launcher: every rank's packed rects in that rank's shared-memory file): rects[t] points
// at the rect of tiles[t]. The caller orders the list - row-major tile order gives the film every other path of the library gives (H10).
int kz_film_merge_rects(float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, const float *const *rects, uint32_t nTiles, int32_t nThreads) {
    if (nTiles && !rects) return kz_fail(KZ_ERR_INVALID_ARG, "kz_film_merge_rects: null rect list");
    if (const int rc = checkMergeArgs(film, width, height, border, tiles, nTiles)) return rc;
    for (uint32_t t = 0; t < nTiles; ++t) if (!rects[t]) return kz_fail(KZ_ERR_INVALID_ARG, "kz_film_merge_rects: rect %u is null", t);
    mergeRects(film, width, height, border, tiles, rects, nTiles, nThreads, false);
    return KZ_OK;
}

int kz_render_multi(KzScene *scene, const KzRenderOpts *opts, const int32_t *devices, uint32_t nDevices, int32_t tileSize, float *film, size_t nFloats,
                    float *deviceMs) {
    if (!scene || !devices || nDevices == 0 || !film) return kz_fail(KZ_ERR_INVALID_ARG, "kz_render_multi: null argument");
    const KzParams &P = scene->prm;
    const size_t filmFloats = (size_t)(P.width + 2 * P.border) * (size_t)(P.height + 2 * P.border) * 4;
    if (nFloats != filmFloats) return kz_fail(KZ_ERR_INVALID_ARG, "film buffer must hold %zu floats", filmFloats);
    for (uint32_t i = 0; i < nDevices; ++i) for (uint32_t j = 0; j < i; ++j) if (devices[i] == devices[j]) return kz_fail(KZ_ERR_INVALID_ARG, "device %d listed twice", devices[i]);
    // the frame's tiles in row-major order: the unit of dealing AND of the merge. A tile's rect holds what the TILE'S OWN pixels add to each of its texels
    // (kz_film_tile_rects, an ImageBlock of the tile), and a film texel receives the rects that reach it in TILE order - so the film does not depend on which
    // device rendered which tile: static dealing, dynamic dealing and any number of devices give the same bits (H10; tests/test_gpu_multi.py, and
    // tests/host_cpp/multi_tsan_test.cpp runs this very file under ThreadSanitizer)
    uint32_t nAll = 0;
    (void)kz_deal_tiles(P.width, P.height, tileSize, 1, 0, nullptr, 0, &nAll);
    std::vector<KzTile> all(nAll);
    int rc = nAll ? kz_deal_tiles(P.width, P.height, tileSize, 1, 0, all.data(), nAll, &nAll) : KZ_OK;
    if (rc) return rc;
    // replicas come up BEFORE the clocks start (deviceMs is render + gather; a first call pays the upload outside it) - side by side, one host thread per device: every
    // device has its own link to the host, and eight uploads of a C4-sized scene one after the other were 0.36 s in front of a 2.4 s job (round 6)
    {
        std::vector<int> urc(nDevices, KZ_OK);
        std::vector<std::string> uerr(nDevices);
        std::vector<std::thread> up;
        for (uint32_t i = 0; i < nDevices; ++i)
            up.emplace_back([&, i]() { urc[i] = kz_scene_upload(scene, devices[i]); if (urc[i]) uerr[i] = kz_last_error(); });
        for (auto &t : up) t.join();
        for (uint32_t i = 0; i < nDevices; ++i) if (urc[i]) return kz_fail(urc[i], "device %d: %s", devices[i], uerr[i].c_str());
    }
    const bool dynamic = opts && opts->tileDealing == 1;
    // (packed: new float[n] leaves the buffer uninitialised - a std::vector would zero 150 MB at C5 just to have the download overwrite them)
    struct Job { std::vector<KzTile> tiles; std::unique_ptr<float[]> packed; size_t nPacked = 0; int rc = KZ_OK; std::string err; float ms = 0.f; };
    std::vector<Job> jobs(nDevices);
    if (!dynamic) {
        for (uint32_t i = 0; i < nDevices; ++i) {
            uint32_t n = 0;
            (void)kz_deal_tiles(P.width, P.height, tileSize, nDevices, i, nullptr, 0, &n);
            jobs[i].tiles.resize(n);
            if (n && (rc = kz_deal_tiles(P.width, P.height, tileSize, nDevices, i, jobs[i].tiles.data(), n, &n))) return rc;
        }
    }
    // dynamic dealing (the reference's BlockGenerator::next under a mutex, block.cpp:117-148): ONE kz_render_tiles call per device with a KzTileDealer
    // on a shared counter. Every device prepares the frame's whole tile list as its tile set once and takes batches of it - about two passes' worth of
    // (pixel, sample) items each - whenever one of its pass contexts comes free; its passes stay in flight across batch boundaries (no per-batch
    // synchronisation, tile-set change or beam rebuild), and a slow device simply comes back to the counter less often.
    volatile uint32_t nextTile = 0, agreedWord = 0;
    // one host thread per device (renderer.cpp:94-127 runs one TBB task per block; here a task is a GPU's share of the tiles)
    std::vector<std::thread> threads;
    for (uint32_t i = 0; i < nDevices; ++i) {
        threads.emplace_back([&, i]() {
            Job &j = jobs[i];
            const auto t0 = std::chrono::steady_clock::now();
            KzRenderOpts o{};
            if (opts) o = *opts;
            o.stream = nullptr; o.accumulate = 0; o.packedOutput = 0; o.dealer = nullptr;
            if (dynamic) {
                std::vector<uint32_t> taken(2 * (size_t)nAll + 2);
                uint32_t nTaken = 0;
                KzTileDealer dl{};
                dl.counter = &nextTile; dl.batchTiles = 0; dl.takers = nDevices; dl.taken = taken.data(); dl.takenCap = (uint32_t)taken.size(); dl.nTaken = &nTaken; dl.agreed = &agreedWord;
                o.dealer = &dl;
                j.rc = kz_render_tiles(scene, &o, all.data(), nAll, devices[i], nullptr, 0);
                for (uint32_t k = 0; !j.rc && k + 1 < nTaken; k += 2) j.tiles.insert(j.tiles.end(), all.begin() + taken[k], all.begin() + taken[k + 1]);
            } else if (!j.tiles.empty()) j.rc = kz_render_tiles(scene, &o, j.tiles.data(), (uint32_t)j.tiles.size(), devices[i], nullptr, 0);
            if (!j.rc && !j.tiles.empty()) {
                j.nPacked = packedFloats(P, j.tiles.data(), (uint32_t)j.tiles.size());
                j.packed.reset(new float[j.nPacked]);
                j.rc = kz_film_download_tiles(scene, devices[i], j.tiles.data(), (uint32_t)j.tiles.size(), j.packed.get(), j.nPacked);
            }
            if (j.rc) j.err = kz_last_error();                             // the message is thread-local: carry it to the caller's thread
            j.ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    for (auto &t : threads) t.join();
    for (uint32_t i = 0; i < nDevices; ++i) {
        if (deviceMs) deviceMs[i] = jobs[i].ms;
        if (jobs[i].rc) return kz_fail(jobs[i].rc, "device %d: %s", devices[i], jobs[i].err.c_str());
    }
    // ImageBlock::put(ImageBlock&) (block.cpp:87-96) in TILE order: the rects of every device gathered into one list, sorted row-major, merged over row bands
    struct Rect { KzTile t; const float *src; };
    std::vector<Rect> tab;
    for (uint32_t i = 0; i < nDevices; ++i) {
        size_t off = 0;
        for (const KzTile &t : jobs[i].tiles) { tab.push_back(Rect{t, jobs[i].packed.get() + off}); off += (size_t)(t.w + 2 * P.border) * (size_t)(t.h + 2 * P.border) * 4; }
    }
    std::sort(tab.begin(), tab.end(), [](const Rect &a, const Rect &b) { return a.t.y0 != b.t.y0 ? a.t.y0 < b.t.y0 : a.t.x0 < b.t.x0; });
    std::vector<KzTile> mt(tab.size());
    std::vector<const float *> mr(tab.size());
    for (size_t k = 0; k < tab.size(); ++k) { mt[k] = tab[k].t; mr[k] = tab[k].src; }
    mergeRects(film, P.width, P.height, P.border, mt.data(), mr.data(), (uint32_t)tab.size(), 0, true);
    return KZ_OK;
}

} // extern "C"
