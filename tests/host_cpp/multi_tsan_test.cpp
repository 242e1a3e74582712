// multi_tsan_test.cpp - the in-process multi-device driver under ThreadSanitizer, without a GPU.
// Linked here, UNCHANGED: nano-kazen_amd/csrc/kz_multi.cpp (kz_render_multi: one host thread per device, static and dynamic dealing, the
// tile-ordered merge over row bands; kz_deal_tiles; kz_film_merge_tiles) and kz_plan.cpp / kz_plan.h (the planner that sizes a dealer's batches, and
// the two operations on the words the takers share: kzDealerAgree, kzDealerTake). Faked: the four device entry points those threads call
// (kz_scene_upload, kz_render_tiles, kz_film_download_tiles, kz_last_error) - a "device" here remembers which tiles it was given or took from the
// counter and hands back rects whose texels are a function of (tile, texel), with random delays so that the threads interleave differently every run.
// Checked: the merged film is the serial tile-order sum bit for bit for 1 / 2 / 4 / 8 devices and both dealings; every tile is rendered exactly once;
// deviceMs are positive; a failing device fails the call with its message while the other threads finish. ThreadSanitizer reports any data race in
// the driver, the dealer words or the merge as a failure of the process (TSAN_OPTIONS=halt_on_error=1).
#include "kz_internal.h"
#include "kz_plan.h"

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[512];
int kz_fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); std::vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}

struct FakeDevice { std::vector<KzTile> rendered; int uploads = 0; };
static FakeDevice g_dev[64];
static std::atomic<int> g_failDevice{-1};
static std::atomic<int> g_renderCalls{0};

static void jitter(int device, int salt) {
    static thread_local std::mt19937 rng((unsigned)std::chrono::steady_clock::now().time_since_epoch().count() ^ (unsigned)(device * 7919 + salt));
    std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
}
// what tile (x0, y0) adds to film texel (fx, fy): thirds and sevenths, so that the order of the additions where aprons overlap shows in the last bits
static float texelValue(const KzTile &t, int fx, int fy, int c) {
    const uint32_t h = (uint32_t)(t.x0 * 73856093) ^ (uint32_t)(t.y0 * 19349663) ^ (uint32_t)(fx * 83492791) ^ (uint32_t)(fy * 2654435761u) ^ (uint32_t)(c * 40503);
    return (float)(h % 1021u + 1u) / 3.0f + (float)(h % 7u) / 7.0f;
}

extern "C" {

const char *kz_last_error(void) { return g_err; }

int kz_scene_upload(KzScene *, int device) { g_dev[device].uploads++; return KZ_OK; }      // (kz_render_multi calls this from ITS thread, before the device threads start)

int kz_render_tiles(KzScene *scene, const KzRenderOpts *opts, const KzTile *tiles, uint32_t nTiles, int device, float *film, size_t) {
    ++g_renderCalls;
    if (film) return kz_fail(KZ_ERR_INVALID_ARG, "the driver never asks kz_render_tiles for a film");
    jitter(device, 1);
    if (device == g_failDevice.load()) return kz_fail(KZ_ERR_OOM, "injected failure on fake device %d", device);
    FakeDevice &d = g_dev[device];
    d.rendered.clear();
    if (opts && opts->dealer) {
        // as renderOn does: the planner resolves the batch size from the list, the sample range and `takers`; the takers agree; batches come from the counter
        std::vector<uint32_t> offs(nTiles + 1, 0);
        for (uint32_t t = 0; t < nTiles; ++t) offs[t + 1] = offs[t] + (uint32_t)(tiles[t].w * tiles[t].h);
        KzPlanIn in;
        in.nPix = offs[nTiles]; in.s0 = 0; in.s1 = scene->prm.sampleCount; in.limit = (size_t)64 << 30; in.perItem = 176;
        in.passItems = opts->passItems; in.passesInFlight = opts->passesInFlight;
        in.dealer = true; in.takers = opts->dealer->takers; in.dealerBatchTiles = opts->dealer->batchTiles; in.nTiles = nTiles; in.tilePixOffset = offs.data();
        in.heldBefore = (size_t)device << 27;                             // every fake device has another history: the batch size must not depend on it
        KzPlan pl; std::string why;
        if (const int rc = kzPlanCall(in, pl, why)) return kz_fail(rc, "%s", why.c_str());
        if (opts->dealer->nTaken) *opts->dealer->nTaken = 0;
        if (!kzDealerAgree(opts->dealer, pl.batchTiles, nTiles)) return kz_fail(KZ_ERR_INVALID_ARG, "takers disagree on the batch size (%u tiles here)", pl.batchTiles);
        for (uint32_t tb, te; kzDealerTake(opts->dealer, pl.batchTiles, nTiles, tb, te);) {
            d.rendered.insert(d.rendered.end(), tiles + tb, tiles + te);
            jitter(device, 2);
        }
    } else {
        d.rendered.assign(tiles, tiles + nTiles);
        jitter(device, 3);
    }
    return KZ_OK;
}

int kz_film_download_tiles(KzScene *scene, int device, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats) {
    const int b = scene->prm.border;
    const FakeDevice &d = g_dev[device];
    size_t off = 0;
    for (uint32_t t = 0; t < nTiles; ++t) {
        bool mine = false;
        for (const KzTile &r : d.rendered) if (std::memcmp(&r, &tiles[t], sizeof r) == 0) mine = true;
        if (!mine) return kz_fail(KZ_ERR_STATE, "device %d asked for the rect of a tile it did not render", device);
        const int rw = tiles[t].w + 2 * b, rh = tiles[t].h + 2 * b;
        if (off + (size_t)rw * rh * 4 > nFloats) return kz_fail(KZ_ERR_INVALID_ARG, "packed buffer too small");
        for (int y = 0; y < rh; ++y) for (int x = 0; x < rw; ++x) for (int c = 0; c < 4; ++c)
            packed[off + ((size_t)y * rw + x) * 4 + c] = texelValue(tiles[t], tiles[t].x0 + x, tiles[t].y0 + y, c);
        off += (size_t)rw * rh * 4;
    }
    jitter(device, 4);
    return off == nFloats ? KZ_OK : kz_fail(KZ_ERR_INVALID_ARG, "packed buffer holds %zu floats, the tiles need %zu", nFloats, off);
}

} // extern "C"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    KzScene scene;
    std::memset(&scene.prm, 0, sizeof scene.prm);
    const int W = 1000, H = 600, B = 2, T = 64;
    scene.prm.width = W; scene.prm.height = H; scene.prm.border = B; scene.prm.sampleCount = 4096;
    const size_t nFloats = (size_t)(W + 2 * B) * (H + 2 * B) * 4;
    // the serial answer: tiles in row-major order, each adding its rect
    uint32_t nAll = 0;
    (void)kz_deal_tiles(W, H, T, 1, 0, nullptr, 0, &nAll);
    std::vector<KzTile> all(nAll);
    CHECK(kz_deal_tiles(W, H, T, 1, 0, all.data(), nAll, &nAll) == KZ_OK && nAll == 16 * 10);
    std::vector<float> want(nFloats, 0.f);
    for (const KzTile &t : all)
        for (int y = 0; y < t.h + 2 * B; ++y) for (int x = 0; x < t.w + 2 * B; ++x) for (int c = 0; c < 4; ++c)
            want[(((size_t)(t.y0 + y)) * (W + 2 * B) + t.x0 + x) * 4 + c] += texelValue(t, t.x0 + x, t.y0 + y, c);
    std::vector<float> film(nFloats);
    for (int round = 0; round < 3; ++round)
        for (int n : {1, 2, 4, 8})
            for (int dynamic = 0; dynamic < 2; ++dynamic) {
                std::vector<int32_t> devs(n);
                for (int i = 0; i < n; ++i) devs[i] = (i * 5 + round) % 13;                       // (device indices need not be 0 .. n-1, only distinct)
                for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devs[i] == devs[j]) devs[i] = 20 + i;
                KzRenderOpts o{};
                o.tileDealing = dynamic;
                std::vector<float> ms(n, -1.f);
                std::fill(film.begin(), film.end(), -7.f);
                const int rc = kz_render_multi(&scene, &o, devs.data(), (uint32_t)n, T, film.data(), nFloats, ms.data());
                if (rc) std::fprintf(stderr, "kz_render_multi: %s\n", kz_last_error());
                CHECK(rc == KZ_OK);
                CHECK(std::memcmp(film.data(), want.data(), nFloats * sizeof(float)) == 0);       // bit for bit, whoever rendered what
                size_t total = 0;
                std::vector<int> seen(nAll, 0);
                for (int i = 0; i < n; ++i) {
                    CHECK(ms[i] > 0.f);
                    total += g_dev[devs[i]].rendered.size();
                    for (const KzTile &t : g_dev[devs[i]].rendered) for (uint32_t k = 0; k < nAll; ++k) if (std::memcmp(&all[k], &t, sizeof t) == 0) seen[k]++;
                }
                CHECK(total == nAll);
                for (uint32_t k = 0; k < nAll; ++k) CHECK(seen[k] == 1);                          // every tile once
                if (dynamic && n > 1) { int busy = 0; for (int i = 0; i < n; ++i) busy += !g_dev[devs[i]].rendered.empty(); CHECK(busy >= 2); }
            }
    // a device that fails: the call fails with that device's message, the other threads have finished (join), nothing hangs
    for (int dynamic = 0; dynamic < 2; ++dynamic) {
        const int32_t devs[4] = {0, 1, 2, 3};
        KzRenderOpts o{};
        o.tileDealing = dynamic;
        g_failDevice.store(2);
        const int rc = kz_render_multi(&scene, &o, devs, 4, T, film.data(), nFloats, nullptr);
        g_failDevice.store(-1);
        CHECK(rc == KZ_ERR_OOM && std::strstr(kz_last_error(), "device 2") && std::strstr(kz_last_error(), "injected"));
        CHECK(kz_render_multi(&scene, &o, devs, 4, T, film.data(), nFloats, nullptr) == KZ_OK);      // and the next call is whole again
        CHECK(std::memcmp(film.data(), want.data(), nFloats * sizeof(float)) == 0);
    }
    // the merge on its own: the result does not depend on the number of band threads
    {
        std::vector<float> packed;
        for (const KzTile &t : all)
            for (int y = 0; y < t.h + 2 * B; ++y) for (int x = 0; x < t.w + 2 * B; ++x) for (int c = 0; c < 4; ++c) packed.push_back(texelValue(t, t.x0 + x, t.y0 + y, c));
        for (int threads : {1, 3, 16}) {
            std::fill(film.begin(), film.end(), 0.f);
            CHECK(kz_film_merge_tiles(film.data(), W, H, B, all.data(), nAll, packed.data(), packed.size(), threads) == KZ_OK);
            CHECK(std::memcmp(film.data(), want.data(), nFloats * sizeof(float)) == 0);
        }
    }
    std::printf("ok: %d driver calls, every film equal to the serial tile-order sum\n", g_renderCalls.load());
    return 0;
}
