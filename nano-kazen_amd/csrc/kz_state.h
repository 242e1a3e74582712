// kz_state.h - what the translation units of the device library share (not part of the ABI): the path-state arrays of the wavefront pipeline,
// the per-pass contexts, the per-device replica state, error / allocation helpers, and the few host functions that cross a unit boundary.
//   kz_render.hip   replicas + upload, pass schedule (renderOn / wfPass), every path kernel (kz_wavefront.h, the megakernel), kz_render*, stats
//   kz_film.hip     film reconstruction kernels + their launcher, tile packing / download, kz_film_* entry points
//   kz_multi.cpp    tile dealing, host merge of tile rects, kz_render_multi (host code only)
//   kz_debug.hip    function-level query kernels and known-answer entry points of include/kazen_mi355x_dev.h
#pragma once
#include <hip/hip_runtime.h>
#include "kz_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// Path state in HBM: one array per field (SoA; 16-B records, coalesced for the stages that sweep all slots). -DKZ_STATE_AOS=1 builds the
// alternative that round 3 measured and rejected (profiles/r03h_state_layout): one 64-B line per path for (ray origin | direction | hit |
// throughput) and one for (shadow origin | direction | pending radiance | misc). It was meant to cut the lines fetched per path once the
// queues hold a thinning, scattered subset of the slots; shade did not move (19.3 -> 19.4 ms: it is not bound by the bytes of these
// arrays) and the stages that sweep every slot lost (generate 1.5 -> 4.0 ms, camera rays 5.8 -> 6.9): C4 1533-1562 -> 1464-1470 Msamples/s.
#ifndef KZ_STATE_AOS
#define KZ_STATE_AOS 0
#endif
template <class Tp> struct KzField {
    Tp *p;
    __device__ __forceinline__ Tp &operator[](uint32_t i) const { return p[KZ_STATE_AOS ? (size_t)i * 4u : (size_t)i]; }
};
struct KzWf {
    KzField<float4> rayA, rayB;    // o.xyz tmin | d.xyz tmax
    KzField<float4> hit;           // t u v gid(bits) - the shading record of the triangle; t = +inf: miss
    KzField<float4> thr;           // throughput.xyz eta (compact state, see kz_wf_shade: throughput.xyz bsdfPdf)
    KzField<float4> misc;          // bsdfPdf accumulatedRoughness discrete(EDiscrete) -
    uint4 *smp;                    // pcg32 state (.x,.y) + dimension index (.z); the pcg32 stream id is recomputed from the pixel
    KzField<float4> shA, shB, shL; // shadow (or walk-through) ray o.xyz tmax | d.xyz tmin | pending radiance
    uint32_t *queue[3];            // two ping-pong path queues + the shadow queue
    uint32_t *counts;              // [stage][4] zeroed per pass
    float *outJx, *outJy, *outR, *outG, *outB;
    unsigned long long *stats;
};

struct KzTune { int refill, postpone, batch, travBlocksPerCU, shadeBlocksPerCU, ldsStack, packet, filmGather, shadeSplit;
                int wide, keyStack, ldsTop, leafQueue, legacyTrace, mixed;      // kz_experiments.h only
                uint32_t *ovf; uint32_t ovfStride; };


#ifndef KZ_BEAM_CAP
#define KZ_BEAM_CAP 32                // leaves per pixel list (16 / 24 / 48 measured in r03o: 32 stays)
#endif

struct KzTileRect { int32_t x0, y0, w, h; uint32_t offset; };
struct KzTileDesc { int32_t x0, y0, w, h; uint32_t pixOffset; };      // a tile of the current set and the position of its first pixel in the pixel list

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return kz_fail(KZ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// Every device allocation of the library goes through here (kz_debug_fail_alloc can make the nth one fail).
hipError_t kzMalloc(void **p, size_t bytes);          // kz_render.hip
#define KZ_ALLOC(pp, bytes) do { hipError_t e_ = kzMalloc((void **)(pp), (bytes)); if (e_ != hipSuccess) \
    return kz_fail(e_ == hipErrorOutOfMemory ? KZ_ERR_OOM : KZ_ERR_HIP, "device allocation of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString(e_)); } while (0)
// A device buffer that is released on every way out of the call that made it.
struct DevMem {
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete; DevMem &operator=(const DevMem &) = delete;
    ~DevMem() { if (p) (void)hipFree(p); }
    template <class Tp> Tp *as() const { return (Tp *)p; }
};

// kz_debug_trace (kazen_mi355x_dev.h, development builds): a timeline of the allocation / growth / pass-planning events of the calling process on stderr
#ifdef KZ_EXPERIMENTS
extern std::atomic<int> g_kzTrace;
void kzTraceLine(const char *fmt, ...);
#define KZ_TRACE(...) do { if (g_kzTrace.load(std::memory_order_relaxed)) kzTraceLine(__VA_ARGS__); } while (0)
int kzPhysicalDevice(int logical);            // kz_debug_alias_devices: the HIP device behind the index a replica is addressed by
#else
#define KZ_TRACE(...) do { } while (0)
static inline int kzPhysicalDevice(int logical) { return logical; }
#endif
int kzLogicalDeviceCount();                   // kz_arena.cpp: devices the library presents (= hipGetDeviceCount unless a development build aliases them)

struct EventPair { hipEvent_t a, b; };
// The path-state memory of one pass context (kz_arena.cpp). Up to 2^23 items: hipMalloc arrays of the size asked for. Beyond: one reserved virtual range
// per array, physical memory mapped into them in levels of 2^23 items on a side thread; `mapped` items of EVERY array are usable at any moment, and only
// ever more (until shrinkTo / releaseAll, which the owner calls on an idle device).
struct KzArena {
    static constexpr int kArrays = 17;        // rayA rayB hit thr misc shA shB shL | smp | queue 0 1 2 | jx jy r g b
    size_t levelItems = (size_t)1 << 23;      // items of one level, fixed while the ranges are reserved (every chunk of an array has the same size): 2^23 (128 MB chunks for the
                                              // 16-B arrays, 32 MB for the 4-B ones) for a context asked to hold up to 2^27 items, 2^25 (512 / 128 MB) for a larger one
    static constexpr size_t kSmallMax = (size_t)1 << 23;
    static constexpr size_t kRuntimeReserve = (size_t)2 << 30;     // bytes a growing context leaves free for the HIP runtime's own device allocations (growOneLevel)
    int device;
    size_t capItems = 0;                      // items the virtual ranges (one per array) are reserved for; 0: no reservation (a small or empty context)
    size_t smallItems = 0;                    // items of the hipMalloc arrays of a small context
    char *base[kArrays]; size_t elem[kArrays];
    struct Level { size_t firstItem, items; hipMemGenericAllocationHandle_t h[kArrays]; int mappedArrays; };
    std::vector<Level> levels;
    std::atomic<size_t> mapped{0};
    std::mutex m; std::condition_variable cvProgress; std::thread th;
    bool stop = false, busy = false, growthFailed = false; size_t target = 0;
    std::atomic<int> failCountdown{0};        // kz_debug_fail_alloc (development builds): the nth physical allocation from now on fails
    bool injectedFailure() {
#ifdef KZ_EXPERIMENTS
        int fc = failCountdown.load();
        return fc > 0 && failCountdown.compare_exchange_strong(fc, fc - 1) && fc == 1;
#else
        return false;
#endif
    }
    int err = 0; std::string errMsg;
    std::chrono::steady_clock::time_point lastProgress;
    explicit KzArena(int dev);
    ~KzArena();
    KzArena(const KzArena &) = delete; KzArena &operator=(const KzArena &) = delete;
    static size_t bytesPerItem();
    size_t bytes() const { return mapped.load() * bytesPerItem(); }
    bool wouldReallocate(size_t items) const;
    int request(size_t items, size_t minItems, double graceMs, size_t *got);
    void shrinkTo(size_t items);
    void lowerTarget(size_t items);
    void releaseAll();
    template <class Tp> Tp *array(int a) const { return (Tp *)base[a]; }
private:
    int reserve(size_t cap, size_t firstTarget); int requestSmall(size_t items, size_t *got); void freeSmall();
    void growLoop(); bool growOneLevel(size_t first); void dropLevels(size_t keepLevels); void stopThread();
};
// path state + sample records + stage events of one pass in flight
struct PassCtx {
    KzArena *arena = nullptr;                                    // the path-state arrays and the five sample planes (jx | jy | r | g | b)
    KzWf wf{};                                                   // (pointers into the arena, set by ctxEnsure)
    float *plane[5] = {};                                        // the five sample planes jx | jy | r | g | b (each its own range of the arena)
    uint32_t *counts = nullptr;                                  // queue counters of a pass (8 x 520 words)
    uint32_t *litQueue = nullptr; size_t litCap = 0;             // kz_wf_trace_dq<2>: shadow rays that need the literal walk-through
    uint32_t *ovf = nullptr; size_t ovfCap = 0;
    hipStream_t side = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr;      // small passes: the shadow rays of a bounce beside its closest-hit rays (wfPass)
    // A pass run as two HALVES of its pixels side by side (renderOn: KzRenderOpts::passHalves): two views of this context's arrays - the first and the second part
    // of every array - each with its own counters, overflow stacks, side stream and stage clock, the second on a stream of its own. A view owns no arena.
    PassCtx *view[2] = {nullptr, nullptr}; hipStream_t halfStream = nullptr; hipEvent_t evHalfFork = nullptr, evHalfJoin = nullptr;
    size_t wanted = 0;                                           // items the last call with the default schedule asked this context to hold (kz_render.hip: `earned`)
    uint64_t beamSeen = 0;                                       // the last beam-list build (KzDeviceState::beamSeq) this context's stream has waited for
    std::vector<hipEvent_t> stageEv; std::vector<int> stageKind; size_t stageUsed = 0;
    size_t items() const { return arena ? arena->mapped.load() : 0; }
    size_t bytes() const { return (arena ? arena->bytes() : 0) + ovfCap * sizeof(uint32_t) + litCap * 4 + (view[0] ? view[0]->bytes() : 0) + (view[1] ? view[1]->bytes() : 0); }
    // gives the memory back (the context stays usable: it grows again on demand); the caller has synchronised the device
    void release() {
        if (arena) arena->shrinkTo(0);
        wf = KzWf{}; for (float *&q : plane) q = nullptr; wanted = 0;
        if (ovf) (void)hipFree(ovf); ovf = nullptr; ovfCap = 0;
        if (litQueue) (void)hipFree(litQueue); litQueue = nullptr; litCap = 0;
        for (PassCtx *v : view) if (v) v->release();
    }
    // buffers sized for another frame (a pooled context): given back when a call is short of memory (the caller has synchronised the device)
    void trimAux() {
        if (ovf) { (void)hipFree(ovf); ovf = nullptr; ovfCap = 0; }
        if (litQueue) { (void)hipFree(litQueue); litQueue = nullptr; litCap = 0; }
        for (PassCtx *v : view) if (v) v->trimAux();
    }
    void destroy() {
        release();
        for (PassCtx *&v : view) if (v) { v->destroy(); delete v; v = nullptr; }
        if (halfStream) { (void)hipStreamDestroy(halfStream); halfStream = nullptr; }
        if (evHalfFork) { (void)hipEventDestroy(evHalfFork); evHalfFork = nullptr; }
        if (evHalfJoin) { (void)hipEventDestroy(evHalfJoin); evHalfJoin = nullptr; }
        if (counts) (void)hipFree(counts); counts = nullptr;
        delete arena; arena = nullptr;
        for (auto &e : stageEv) (void)hipEventDestroy(e);
        stageEv.clear(); stageKind.clear();
        if (side) { (void)hipStreamDestroy(side); side = nullptr; }
        if (evFork) { (void)hipEventDestroy(evFork); evFork = nullptr; }
        if (evJoin) { (void)hipEventDestroy(evJoin); evJoin = nullptr; }
    }
};
// Pass contexts live in a per-device pool between replicas (kz_arena.cpp): a replica takes them on first use and hands them back when it goes.
PassCtx *kzCtxAcquire(int device);
void kzCtxRelease(int device, PassCtx *c);
size_t kzCtxPoolBytes(int device);
size_t kzCtxPoolMaxItems(int device);
size_t kzCtxPoolTrim(int device, size_t keepBytes);
size_t kzCtxPoolTrimPhysical(int hipDevice);      // every idle pooled context that lives on that HIP device (kzMalloc's answer to an out-of-memory)
struct KzDeviceState {
    int device = -1;                                             // the index the caller addresses this replica by
    int hipDevice = -1;                                          // the HIP device it lives on (the same number, unless a development build aliases devices: kz_debug_alias_devices)
    KzDevTables T{};
    std::vector<void *> allocs;
    bool bvh2Resident = false;                                   // T.nodes holds the BVH2 (uploaded on first use: kzEnsureBvh2)
    float4 *film = nullptr; size_t filmPixels = 0;
    float4 *tapSums = nullptr; size_t tapSumsBytes = 0;          // the running tap sums of every pixel of the frame, [tap][y * width + x] (kz_film.hip): what the film is resolved from
    uint8_t *srgb = nullptr;                                     // staging raster of kz_film_to_srgb8 (allocated on first use)
    float4 *packDev = nullptr; size_t packCap = 0; KzTileRect *rectsDev = nullptr; size_t rectsCap = 0;      // kz_film_download_tiles: packed tile rects + their table
    float4 *packHost = nullptr; size_t packHostCap = 0;           // pinned staging of the same (D2H at link rate)
    // The tile set: pixList = its pixels (tile after tile, 8x8 blocks row-major inside a tile, row-major inside a block), written on the device from the
    // tile descriptors (kz_tiles_expand).
    uint32_t *pixList = nullptr; size_t pixCap = 0; uint32_t nPix = 0;
    KzTileDesc *tileDev = nullptr; size_t tileDevCap = 0; KzTileDesc *tileHost = nullptr; size_t tileHostCap = 0; hipEvent_t evTiles = nullptr;
    std::vector<KzTile> curTiles; std::vector<uint32_t> tilePixOffset;      // tilePixOffset[t]: first list position of tile t (+ the total at the end)
    bool tilesValid = false; uint64_t tileGen = 0;                          // tileGen: bumped whenever the pixel list changes
    unsigned long long *stats = nullptr; bool statsOn = false;
    hipStream_t lastStream = nullptr;
    int numCU = 256; size_t totalMem = 0;
    PassCtx *ctx[KZ_MAX_PASSES_IN_FLIGHT] = {};                  // taken from the device's pool on first use (ctxAt), handed back by releaseReplica
    PassCtx &ctxAt(int i) { if (!ctx[i]) ctx[i] = kzCtxAcquire(device); return *ctx[i]; }
    std::vector<EventPair> events; size_t eventsUsed = 0;
    hipStream_t passStream[KZ_MAX_PASSES_IN_FLIGHT] = {}; hipEvent_t evFork = nullptr, evFilm[KZ_MAX_PASSES_IN_FLIGHT] = {}, evCallA = nullptr, evCallB = nullptr;
    // LARGE passes with KzRenderOpts::shadowBeside = passHalves = 0: the replica times four passes of one size - one stream [0], the shadow rays beside the closest-hit
    // rays [1], two halves of its pixels side by side [2], one stream again [3] - then keeps the fastest way for its scene (renderOn).
    // largeMode: -1 = not known yet, 0 = one stream, 1 = shadow rays beside, 2 = halves.
    hipEvent_t evProbe[4][2] = {}; size_t probeItems[4] = {}; int probeLaunched = 0; int largeMode = -1; float probeMs[4] = {};
    int lastCtx = 0; bool lastDual = false; int streamMode = 0;
    PassCtx *lastStageCtx = nullptr;                             // whose stage clock kz_last_stage_ms reads (a view, when the last pass ran as halves)
    // Beam lists (kz_wf_beam), one per pixel of the FRAME, built at most once per pixel and replica - the camera belongs to the scene - whatever tile
    // sets and pixel chunks the pixel is rendered in. They are built on the call's stream (evBeam / beamSeq: the passes wait for the latest build);
    // beamDone remembers the ranges of the CURRENT pixel list that have been handed to the kernel (it skips pixels that already have a list).
    uint2 *beamEntries = nullptr, *beamCount = nullptr; size_t beamCap = 0; hipEvent_t evBeam = nullptr; uint64_t beamSeq = 0;
    std::vector<std::pair<uint32_t, uint32_t>> beamDone; uint64_t beamDoneGen = 0;
    size_t beamBytes() const { return beamCap * (KZ_BEAM_CAP + 1) * sizeof(uint2); }
    size_t ctxBytes() const { size_t b = 0; for (const PassCtx *c : ctx) if (c) b += c->bytes(); return b; }
    KzPassInfo lastInfo{}; std::string growNote;
};
struct KzReplicaSet { std::mutex m; std::vector<KzDeviceState *> v; };

static inline KzReplicaSet *replicaSet(const KzScene *scene) { return (KzReplicaSet *)scene->dev; }
int findReplica(const KzScene *scene, int device, KzDeviceState **out);                                   // kz_render.hip
int kzEnsureBvh2(KzScene *scene, KzDeviceState *ds);                                                      // kz_render.hip: the BVH2 table, on first use
static inline int requireDevice(KzScene *scene, KzDeviceState **out) { return findReplica(scene, -1, out); }
// kz_film.hip
size_t packedFloats(const KzParams &P, const KzTile *tiles, uint32_t nTiles);
int checkTiles(const KzParams &P, const KzTile *tiles, uint32_t nTiles);
int downloadTiles(KzScene *scene, KzDeviceState *ds, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats, hipStream_t stream);
// The film (kz_film.hip): running tap sums per frame pixel, fed by every pass (kzFilmStage: ImageBlock::put for the sample records of the pass, launched on `pst` behind
// `waitFilm`), resolved into the film texels once per call (kzFilmResolve).
#define KZ_FILM_GRID 64                      // the canonical tile grid of the resolve = kz_deal_tiles' default tile: the film of one device equals the host merge of its tiles' rects bit for bit
int kzFilmEnsureTapSums(KzScene *scene, KzDeviceState *ds, hipStream_t stream);
int kzFilmClear(KzDeviceState *ds, hipStream_t stream);
int kzFilmStage(KzScene *scene, KzDeviceState *ds, PassCtx &c, hipStream_t pst, const uint32_t *pixList, uint32_t nPixPass, uint32_t Sp, hipEvent_t waitFilm, int lanesPerPixel);
int kzFilmResolve(KzScene *scene, KzDeviceState *ds, hipStream_t stream);
