// kz_render.hip - the device side of kz_render / kz_render_tiles for the path_mis hot path: replicas and their upload, the launch code of a pass (wfPass: the
// wavefront pipeline of kz_wavefront.h - camera rays by pixel beams, persistent BVH4 traversal with Moeller-Trumbore leaf tests, post-intersection, BSDF
// eval + sample, MIS next-event estimation with the invisible-light walk-through, Russian roulette - and the reference-shaped megakernel below), and renderOn:
// plan (kz_plan.cpp) -> make room (kz_arena.cpp) -> launch -> film (kz_film.hip).
// Reference region replaced: src/kazen/renderer.cpp:85-133 and everything it calls.
//
// Execution model (wave64): one lane = one (pixel, sample) path; items are ordered pixel-major so the lanes of a wave share a pixel neighbourhood (coherent
// camera rays, shared top-of-tree node packets in L1 / L2). Per-lane traversal stacks live in LDS ([entry][lane]: conflict-free columns). No MFMA: this is
// branchy pointer chasing, bound by VALU issue and the CU's L1 gather rate, not a contraction.
#include <hip/hip_runtime.h>

#include "kz_internal.h"
#include "kz_devfn.h"
#include "kz_wavefront.h"
#include "kz_plan.h"
#ifdef KZ_EXPERIMENTS
#include "variants/experiments/kz_experiments.h"
#endif
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// a1/a2 renderBlock + renderSample (renderer.cpp:20-69): item = pixLinear * S + sampleOffset
template <bool STATS, int EXT>
__global__ __launch_bounds__(KZ_BLOCK) void kz_path_megakernel(KzParams P, KzDevTables T, const uint32_t *__restrict__ pixList,
                                                               uint32_t nItems, uint32_t S, uint32_t sampleBegin, const uint32_t *__restrict__ itemSample,
                                                               float *__restrict__ outJx, float *__restrict__ outJy, float *__restrict__ outR,
                                                               float *__restrict__ outG, float *__restrict__ outB,
                                                               unsigned long long *__restrict__ stats) {
    __shared__ uint32_t s_stack[KZ_STACK_DEPTH * KZ_BLOCK];
    const uint32_t item = blockIdx.x * KZ_BLOCK + threadIdx.x;
    Counters cn = {0, 0, 0, 0, 0, 0};
    if (item < nItems) {
        const uint32_t pl = item / S, so = item - pl * S;
        const uint32_t pxy = pixList[pl];
        const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
        Sampler smp; smp.type = P.samplerType;
        smp.generateSample(P, T, px, py, itemSample ? itemSample[item] : sampleBegin + so);
        float jx, jy; smp.nextPixel2D(P, T, jx, jy);
        const float sx = (float)px + jx, sy = (float)py + jy;
        float ax, ay; smp.next2D(P, T, ax, ay);                  // aperture sample, always consumed (renderer.cpp:28)
        V3 ro, rd; float mint, maxt;
        cameraRay(P, sx, sy, ax, ay, ro, rd, mint, maxt);
        V3 L = pathLi<STATS, EXT>(P, T, smp, ro, rd, mint, maxt, s_stack + threadIdx.x, cn);
        outJx[item] = jx; outJy[item] = jy; outR[item] = L.x; outG[item] = L.y; outB[item] = L.z;
        if (STATS) {
            bool valid = L.x >= 0 && L.y >= 0 && L.z >= 0 && isfinite(L.x) && isfinite(L.y) && isfinite(L.z);
            if (!valid) cn.dropped++;
        }
    }
    if (STATS) {
        // wave reduction then one atomic per counter per wave
        unsigned long long v[7] = {item < nItems ? 1ull : 0ull, cn.rays, cn.nodes, cn.tris, cn.hits, cn.lsamples, cn.dropped};
        for (int k = 0; k < 7; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k], x);
        }
    }
}

// ============================================================================================
// host side: replicas (one device state per GPU the scene is resident on), upload, passes
// ============================================================================================
// Every device allocation of the library goes through here. Out of memory: the idle pass contexts of the device's pool (up to 3/4 of the card, parked there by
// replicas that have gone) are given back and the allocation is tried once more - tables, film and beam lists of the NEXT scene must not fail because the
// last one's path state is still mapped. (Development builds: kz_debug_fail_alloc makes the nth allocation of the calling thread fail.)
#ifdef KZ_EXPERIMENTS
static thread_local int g_failAlloc = 0;
// kz_debug_fail_device: a countdown per (logical) device, handed to whichever thread addresses that device's replica next (findReplica) - so that a test can fail an
// allocation inside ONE device thread of kz_render_multi, which no caller's thread-local countdown reaches
static std::atomic<int> g_failDevice[64];
#endif
hipError_t kzMalloc(void **p, size_t bytes) {
    *p = nullptr;
#ifdef KZ_EXPERIMENTS
    if (g_failAlloc > 0 && --g_failAlloc == 0) return hipErrorOutOfMemory;
#endif
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && kzCtxPoolTrimPhysical(dev) > 0) e = hipMalloc(p, bytes);
    }
    return e;
}

template <class Tp> static int uploadVec(KzDeviceState *ds, const std::vector<Tp> &v, const Tp **out) {
    *out = nullptr;
    void *p = nullptr;
    const size_t bytes = v.empty() ? 256 : v.size() * sizeof(Tp);    // keep a valid (dummy) pointer so kernels never see null
    KZ_ALLOC(&p, bytes);
    ds->allocs.push_back(p);
    if (v.empty()) HIP_TRY(hipMemset(p, 0, bytes));
    else HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = (const Tp *)p;
    return KZ_OK;
}

static void releaseReplica(KzDeviceState *ds) {
    (void)hipSetDevice(ds->hipDevice);
    (void)hipDeviceSynchronize();
    for (void *p : ds->allocs) (void)hipFree(p);
    for (void *p : {(void *)ds->film, (void *)ds->tapSums, (void *)ds->srgb, (void *)ds->pixList, (void *)ds->stats, (void *)ds->packDev, (void *)ds->rectsDev, (void *)ds->beamEntries, (void *)ds->beamCount, (void *)ds->tileDev}) if (p) (void)hipFree(p);
    if (ds->tileHost) (void)hipHostFree(ds->tileHost);
    if (ds->evTiles) (void)hipEventDestroy(ds->evTiles);
    if (ds->evBeam) (void)hipEventDestroy(ds->evBeam);
    if (ds->packHost) (void)hipHostFree(ds->packHost);
    for (PassCtx *&c : ds->ctx) if (c) { kzCtxRelease(ds->device, c); c = nullptr; }      // (the device is idle: the contexts go back to its pool, memory and all)
    for (auto &e : ds->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipStream_t st : ds->passStream) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : ds->evFilm) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ds->evFork, ds->evCallA, ds->evCallB, ds->evProbe[0][0], ds->evProbe[0][1], ds->evProbe[1][0], ds->evProbe[1][1], ds->evProbe[2][0], ds->evProbe[2][1], ds->evProbe[3][0], ds->evProbe[3][1]}) if (e) (void)hipEventDestroy(e);
    delete ds;
}

void kz_device_init(KzScene *scene) { scene->dev = new KzReplicaSet(); }

void kz_device_release(KzScene *scene) {
    KzReplicaSet *rs = replicaSet(scene);
    if (!rs) return;
    for (KzDeviceState *ds : rs->v) releaseReplica(ds);
    delete rs;
    scene->dev = nullptr;
}

// KzTuning -> the kernels' KzTune. Zero = the library default. (The KZ_* environment overrides of ABI v2 / v3 are gone: nothing in the
// library reads the environment.) Fields that select a kernel of kz_experiments.h are honoured only by a -DKZ_EXPERIMENTS build; the
// default library refuses them loudly instead of ignoring them.
static int resolveTune(const KzTuning &t, KzTune &r) {
    auto pick = [](int a, int d) { return a > 0 ? a : d; };
    r = KzTune{};
    r.refill = pick(t.refill, 0); r.postpone = pick(t.postpone, 24); r.batch = pick(t.batch, 0);                      // 0: the default per ray kind (wfPass)
    r.travBlocksPerCU = std::min(8, pick(t.traceBlocksPerCU, KZ_TRACE_WAVES)); r.shadeBlocksPerCU = std::min(16, pick(t.shadeBlocksPerCU, 0));
    r.ldsStack = pick(t.ldsStack, 16);
    r.packet = pick(t.packetPrimary, 0); r.filmGather = pick(t.filmGather, 0);
#if defined(KZ_EXPERIMENTS) && defined(KZ_SHADE_SPLIT)
    r.shadeSplit = KZ_SHADE_SPLIT;          // (development builds only: -DKZ_EXPERIMENTS -DKZ_SHADE_SPLIT=1, profiles/r04b_shade_split)
#endif
    r.wide = t.KZ_TUNE_BVH2 ? 0 : 1; r.keyStack = pick(t.KZ_TUNE_KEY_STACK, 0); r.ldsTop = pick(t.KZ_TUNE_LDS_TOP, 0); r.leafQueue = pick(t.KZ_TUNE_LEAF_QUEUE, 0);
    r.legacyTrace = pick(t.KZ_TUNE_LEGACY_TRACE, 0); r.mixed = pick(t.KZ_TUNE_MIXED_LAUNCH, 0);
#ifndef KZ_EXPERIMENTS
    if (t.dev0 || t.dev1 > 0 || t.dev2 > 0 || t.dev3 > 1 || t.dev4 > 0 || t.dev5 > 0)
        return kz_fail(KZ_ERR_UNSUPPORTED, "KzTuning.dev0 .. dev5 (kazen_mi355x_dev.h: bvh2 / keyStack / ldsTop / leafQueue / legacyTrace / mixedLaunch) select kernels of rejected experiments: "
                                           "this library was built without -DKZ_EXPERIMENTS (kz_build_flags)");
#endif
    r.ovf = nullptr; r.ovfStride = 0;
    return KZ_OK;
}

int findReplica(const KzScene *scene, int device, KzDeviceState **out) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    KzReplicaSet *rs = replicaSet(scene);
    KzDeviceState *ds = nullptr;
    if (rs) {
        std::lock_guard<std::mutex> g(rs->m);
        if (device < 0) ds = rs->v.empty() ? nullptr : rs->v.front();
        else for (KzDeviceState *d : rs->v) if (d->device == device) { ds = d; break; }
    }
    if (!ds) {
        if (device < 0 || !rs || rs->v.empty()) return kz_fail(KZ_ERR_STATE, "scene is not on a device: call kz_scene_upload first");
        return kz_fail(KZ_ERR_STATE, "scene is not resident on device %d: call kz_scene_upload(scene, %d) first", device, device);
    }
    hipError_t e = hipSetDevice(ds->hipDevice);
    if (e != hipSuccess) return kz_fail(KZ_ERR_HIP, "hipSetDevice(%d): %s", ds->hipDevice, hipGetErrorString(e));
#ifdef KZ_EXPERIMENTS
    { const int n = g_failDevice[ds->device & 63].exchange(0); if (n > 0) g_failAlloc = n; }
#endif
    *out = ds;
    return KZ_OK;
}
// the primary replica (calls without a device argument)

static int uploadReplica(KzScene *scene, KzDeviceState *ds) {
    int rc;
    KZ_TRACE("upload: start (%.1f MB of tables)", (scene->nodes.size() * sizeof(scene->nodes[0]) + scene->nodes4.size() * sizeof(scene->nodes4[0]) + scene->tris.size() * sizeof(scene->tris[0]) + scene->shade.size() * sizeof(scene->shade[0])) / 1e6);
    // The four large tables (C4: 248 MB; their allocations first, in this thread, so that a failure is this call's) travel side by side: a pageable copy is
    // a chain of host memcpys into the runtime's pinned staging buffers and DMA transfers out of them, one chain per calling thread - one thread per table in
    // flight took the upload of C4 from 45 to 36 ms (profiles/r05d_cold_job; finer pieces over eight threads change nothing - round 6: ~32 ms of an upload are
    // fixed costs of a process's first allocations, the 8 MB scene of the reference's own file takes as long -; the FIRST process on a freshly leased box spends 170 ms here either way: the
    // first transfers of a box wake something up that no later process pays for)
    {
        struct Job { const void *src; void *dst; size_t bytes; hipError_t err; };
        Job jobs[4] = {};
        auto prep = [&](auto &vec, auto **out, Job &j) -> int {
            using Tp = typename std::remove_reference<decltype(vec)>::type::value_type;
            *out = nullptr;
            void *p = nullptr;
            const size_t bytes = vec.empty() ? 256 : vec.size() * sizeof(Tp);
            KZ_ALLOC(&p, bytes);
            ds->allocs.push_back(p);
            if (vec.empty()) HIP_TRY(hipMemset(p, 0, bytes));
            *out = (const Tp *)p;
            j = Job{vec.empty() ? nullptr : (const void *)vec.data(), p, bytes, hipSuccess};
            return KZ_OK;
        };
        // (the BVH2 - C4: 62 MB of the 248 - serves the reference-shaped megakernel, kz_trace_rays and kz_render_samples only: it goes up on the first of those
        //  calls, kzEnsureBvh2; a one-frame job through the wavefront pipeline never pays for it)
        static const std::vector<KzNode> noNodes;
        if ((rc = prep(noNodes, &ds->T.nodes, jobs[0]))) return rc;
        if ((rc = prep(scene->nodes4, &ds->T.nodes4, jobs[1]))) return rc;
        if ((rc = prep(scene->tris, &ds->T.tris, jobs[2]))) return rc;
        if ((rc = prep(scene->shade, &ds->T.shade, jobs[3]))) return rc;
        std::vector<std::thread> th;
        const int dev = ds->hipDevice;
        for (Job &j : jobs) if (j.src && j.bytes >= ((size_t)4 << 20)) th.emplace_back([&j, dev] { j.err = hipSetDevice(dev); if (j.err == hipSuccess) j.err = hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice); });
        for (Job &j : jobs) if (j.src && j.bytes < ((size_t)4 << 20)) j.err = hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice);
        for (std::thread &t : th) t.join();
        for (Job &j : jobs) if (j.err != hipSuccess) return kz_fail(KZ_ERR_HIP, "upload of a scene table (%zu bytes) failed: %s", j.bytes, hipGetErrorString(j.err));
    }
    KZ_TRACE("upload: nodes, nodes4, tris, shade there");
    if ((rc = uploadVec(ds, scene->meshRows, &ds->T.meshes))) return rc;
    if ((rc = uploadVec(ds, scene->bsdfs, &ds->T.bsdfs))) return rc;
    if ((rc = uploadVec(ds, scene->lightRows, &ds->T.lights))) return rc;
    if ((rc = uploadVec(ds, scene->cdf, &ds->T.cdf))) return rc;
    if ((rc = uploadVec(ds, scene->pmj, &ds->T.pmj))) return rc;
    if ((rc = uploadVec(ds, scene->bn, &ds->T.bn))) return rc;
    if ((rc = uploadVec(ds, scene->pixelSamples, &ds->T.pixelSamples))) return rc;
    if ((rc = uploadVec(ds, scene->jump, &ds->T.jump))) return rc;
    std::vector<float> ft(scene->filter, scene->filter + KZ_FILTER_RESOLUTION + 1);
    if ((rc = uploadVec(ds, ft, &ds->T.filter))) return rc;
    if ((rc = uploadVec(ds, scene->ilTris, &ds->T.ilTris))) return rc;
    if ((rc = uploadVec(ds, scene->texProgs, &ds->T.texProgs))) return rc;
    if ((rc = uploadVec(ds, scene->texOps, &ds->T.texOps))) return rc;
    if ((rc = uploadVec(ds, scene->images, &ds->T.images))) return rc;
    if ((rc = uploadVec(ds, scene->texels, &ds->T.texels))) return rc;
    ds->T.texPow2 = scene->texPow2;
    const KzParams &P = scene->prm;
    ds->filmPixels = (size_t)(P.width + 2 * P.border) * (size_t)(P.height + 2 * P.border);
    KZ_TRACE("upload: small tables there");
    KZ_ALLOC(&ds->film, ds->filmPixels * sizeof(float4));
    HIP_TRY(hipMemset(ds->film, 0, ds->filmPixels * sizeof(float4)));
    KZ_ALLOC(&ds->stats, 32 * sizeof(unsigned long long));             // 8 counters of KzStats + 16 lane statistics of the -DKZ_LANESTAT development build + 3 beam-list counters
    HIP_TRY(hipMemset(ds->stats, 0, 32 * sizeof(unsigned long long)));
    { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, ds->hipDevice)); ds->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; ds->totalMem = prop.totalGlobalMem; }
    KZ_TRACE("upload: film + counters there");
    HIP_TRY(hipDeviceSynchronize());
    KZ_TRACE("upload: done");
    return KZ_OK;
}

// The BVH2 table of a replica, on first use (see uploadReplica). The caller has made the replica's device current.
int kzEnsureBvh2(KzScene *scene, KzDeviceState *ds) {
    if (ds->bvh2Resident || scene->nodes.empty()) return KZ_OK;
    void *p = nullptr;
    const size_t bytes = scene->nodes.size() * sizeof(KzNode);
    KZ_ALLOC(&p, bytes);
    ds->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, scene->nodes.data(), bytes, hipMemcpyHostToDevice));      // (blocking, behind everything queued on the device: a rare call)
    ds->T.nodes = (const KzNode *)p;
    ds->bvh2Resident = true;
    return KZ_OK;
}

extern "C" {

#ifdef KZ_EXPERIMENTS
void kz_debug_fail_alloc(int nth) { g_failAlloc = nth > 0 ? nth : 0; }
void kz_debug_fail_device(int device, int nth) { g_failDevice[device & 63].store(nth > 0 ? nth : 0); }
#endif

int kz_device_count(void) { return kzLogicalDeviceCount(); }

int kz_device_mem_info(int device, uint64_t *freeBytes, uint64_t *totalBytes) {
    int n = kz_device_count();
    if (device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return KZ_OK;
}

int kz_scene_upload(KzScene *scene, int device) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    const int n = kzLogicalDeviceCount();
    if (n <= 0) return kz_fail(KZ_ERR_NO_DEVICE, "no HIP device visible (the product path has no CPU fallback)");
    if (device < 0 || device >= n) return kz_fail(KZ_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, n);
    KzReplicaSet *rs = replicaSet(scene);
    {
        std::lock_guard<std::mutex> g(rs->m);
        for (KzDeviceState *d : rs->v) if (d->device == device) return KZ_OK;       // already resident
    }
    KZ_TRACE("kz_scene_upload(%d)", device);
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    KZ_TRACE("upload: hipSetDevice done");
    KzDeviceState *ds = new KzDeviceState();
    ds->device = device; ds->hipDevice = kzPhysicalDevice(device);
    const int rc = uploadReplica(scene, ds);
    if (rc) { releaseReplica(ds); return rc; }
    std::lock_guard<std::mutex> g(rs->m);
    for (KzDeviceState *d : rs->v) if (d->device == device) { releaseReplica(ds); return KZ_OK; }     // lost a race for the same device
    rs->v.push_back(ds);
    return KZ_OK;
}

int kz_scene_evict(KzScene *scene, int device) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    KzReplicaSet *rs = replicaSet(scene);
    std::vector<KzDeviceState *> gone;
    {
        std::lock_guard<std::mutex> g(rs->m);
        for (size_t i = 0; i < rs->v.size();) {
            if (device < 0 || rs->v[i]->device == device) { gone.push_back(rs->v[i]); rs->v.erase(rs->v.begin() + i); } else ++i;
        }
    }
    if (gone.empty() && device >= 0) return kz_fail(KZ_ERR_STATE, "scene is not resident on device %d", device);
    for (KzDeviceState *d : gone) releaseReplica(d);
    return KZ_OK;
}

int kz_scene_devices(const KzScene *scene, int32_t *devices, uint32_t cap, uint32_t *count) {
    if (!scene || !count) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    KzReplicaSet *rs = replicaSet(scene);
    std::lock_guard<std::mutex> g(rs->m);
    *count = (uint32_t)rs->v.size();
    for (uint32_t i = 0; i < *count && i < cap && devices; ++i) devices[i] = rs->v[i]->device;
    return KZ_OK;
}

} // extern "C"

// The pixel list of a tile set (tile after tile; 8x8 blocks row-major inside a tile, row-major inside a block: a wave of 64 list entries is an 8x8
// block of the image), written on the device from the tile descriptors: one thread per list entry finds its tile by bisection over the tiles' first positions.
__global__ __launch_bounds__(256) void kz_tiles_expand(const KzTileDesc *__restrict__ tiles, uint32_t nTiles, uint32_t nPix, int width,
                                                       uint32_t *__restrict__ pixList) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nPix) return;
    uint32_t lo = 0, hi = nTiles;                          // the last tile whose pixOffset <= i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tiles[mid].pixOffset <= i) lo = mid; else hi = mid; }
    const KzTileDesc t = tiles[lo];
    const uint32_t local = i - t.pixOffset, w = (uint32_t)t.w, h = (uint32_t)t.h;
    const uint32_t r = local / (8u * w), rem = local - r * 8u * w, hb = min(8u, h - 8u * r);       // block row r holds hb rows of the tile
    const uint32_t c = rem / (8u * hb), rem2 = rem - c * 8u * hb, wb = min(8u, w - 8u * c);        // block c of that row is wb pixels wide
    const uint32_t yl = rem2 / wb, xl = rem2 - yl * wb;
    const uint32_t x = (uint32_t)t.x0 + 8u * c + xl, y = (uint32_t)t.y0 + 8u * r + yl;
    pixList[i] = x | (y << 16);
}

// Makes `tiles` the replica's tile set. Nothing here waits for the device: the descriptors go up through the call's stream, behind the end of the
// previous call (evCallB), and the expansion kernel runs there - a change of tile set costs a few microseconds of host time and ~30 us of device
// time, where it used to synchronise every stream, build the list on the host and copy it with a blocking pageable copy.
static int prepareTiles(KzScene *scene, KzDeviceState *ds, const KzTile *tiles, uint32_t nTiles, hipStream_t stream) {
    const KzParams &P = scene->prm;
    KzTile whole = {0, 0, P.width, P.height};
    if (!tiles || nTiles == 0) { tiles = &whole; nTiles = 1; }
    bool same = ds->tilesValid && ds->curTiles.size() == nTiles && std::memcmp(ds->curTiles.data(), tiles, nTiles * sizeof(KzTile)) == 0;
    if (same) return KZ_OK;
    // bounds, then overlaps: tiles on the 8-px grid (every tile the library deals) are marked on a coarse occupancy map, others compared pairwise
    std::vector<KzTileDesc> desc(nTiles);
    std::vector<uint32_t> offs(nTiles + 1);
    bool grid = true;
    uint64_t total = 0;
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > P.width || tl.y0 + tl.h > P.height)
            return kz_fail(KZ_ERR_INVALID_ARG, "tile %u (%d,%d %dx%d) outside the %dx%d image", t, tl.x0, tl.y0, tl.w, tl.h, P.width, P.height);
        if ((tl.x0 & 7) || (tl.y0 & 7) || ((tl.w & 7) && tl.x0 + tl.w != P.width) || ((tl.h & 7) && tl.y0 + tl.h != P.height)) grid = false;
        desc[t] = KzTileDesc{tl.x0, tl.y0, tl.w, tl.h, (uint32_t)total};
        offs[t] = (uint32_t)total;
        total += (uint64_t)tl.w * (uint64_t)tl.h;
    }
    offs[nTiles] = (uint32_t)total;
    if (total > (uint64_t)P.width * (uint64_t)P.height) return kz_fail(KZ_ERR_INVALID_ARG, "tiles overlap (%llu pixels in a %dx%d image)", (unsigned long long)total, P.width, P.height);
    if (grid) {
        const int gw = (P.width + 7) / 8, gh = (P.height + 7) / 8;
        std::vector<uint8_t> occ((size_t)gw * gh, 0);
        for (uint32_t t = 0; t < nTiles; ++t) {
            const KzTile &tl = tiles[t];
            for (int gy = tl.y0 / 8; gy < (tl.y0 + tl.h + 7) / 8; ++gy)
                for (int gx = tl.x0 / 8; gx < (tl.x0 + tl.w + 7) / 8; ++gx) {
                    uint8_t &o = occ[(size_t)gy * gw + gx];
                    if (o) return kz_fail(KZ_ERR_INVALID_ARG, "tiles overlap at pixel (%d,%d)", gx * 8, gy * 8);
                    o = 1;
                }
        }
    } else {
        for (uint32_t a = 0; a < nTiles; ++a) for (uint32_t b2 = a + 1; b2 < nTiles; ++b2) {
            const KzTile &A = tiles[a], &B = tiles[b2];
            if (A.x0 < B.x0 + B.w && B.x0 < A.x0 + A.w && A.y0 < B.y0 + B.h && B.y0 < A.y0 + A.h)
                return kz_fail(KZ_ERR_INVALID_ARG, "tiles overlap at pixel (%d,%d)", std::max(A.x0, B.x0), std::max(A.y0, B.y0));
        }
    }
    const size_t framePix = (size_t)P.width * P.height;
    if (ds->pixCap < framePix || ds->tileDevCap < nTiles) {          // first use (or a longer tile list than ever before): allocate
        HIP_TRY(hipDeviceSynchronize());
        if (ds->pixCap < framePix) {
            if (ds->pixList) (void)hipFree(ds->pixList);
            ds->pixList = nullptr; ds->pixCap = 0;
            KZ_ALLOC(&ds->pixList, framePix * sizeof(uint32_t));
            ds->pixCap = framePix;
        }
        if (ds->tileDevCap < nTiles) {
            if (ds->tileDev) (void)hipFree(ds->tileDev);
            ds->tileDev = nullptr; ds->tileDevCap = 0;
            const size_t cap = std::max<size_t>(nTiles, 4096);
            KZ_ALLOC(&ds->tileDev, cap * sizeof(KzTileDesc));
            ds->tileDevCap = cap;
        }
    }
    ds->tilesValid = false;
    if (ds->evCallB) HIP_TRY(hipStreamWaitEvent(stream, ds->evCallB, 0));        // behind everything the previous call (on whatever stream) queued
    // the descriptors travel through a pinned buffer of the replica (the copy of the previous tile set has left it: evTiles)
    if (ds->tileHostCap < nTiles) {
        if (ds->evTiles) HIP_TRY(hipEventSynchronize(ds->evTiles));
        if (ds->tileHost) (void)hipHostFree(ds->tileHost);
        ds->tileHost = nullptr; ds->tileHostCap = 0;
        const size_t cap = std::max<size_t>(nTiles, 4096);
        HIP_TRY(hipHostMalloc((void **)&ds->tileHost, cap * sizeof(KzTileDesc), hipHostMallocDefault));
        ds->tileHostCap = cap;
    }
    if (!ds->evTiles) HIP_TRY(hipEventCreateWithFlags(&ds->evTiles, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(ds->evTiles));
    std::memcpy(ds->tileHost, desc.data(), nTiles * sizeof(KzTileDesc));
    HIP_TRY(hipMemcpyAsync(ds->tileDev, ds->tileHost, nTiles * sizeof(KzTileDesc), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(ds->evTiles, stream));
    hipLaunchKernelGGL(kz_tiles_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const KzTileDesc *)ds->tileDev, nTiles, (uint32_t)total, P.width, ds->pixList);
    HIP_TRY(hipGetLastError());
    ds->nPix = (uint32_t)total;
    ds->curTiles.assign(tiles, tiles + nTiles);
    ds->tilePixOffset = std::move(offs);
    ds->tilesValid = true; ++ds->tileGen;
    return KZ_OK;
}

static int stageMark(PassCtx &c, hipStream_t stream, int kind) {
    if (c.stageUsed == c.stageEv.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c.stageEv.push_back(e); c.stageKind.push_back(0); }
    c.stageKind[c.stageUsed] = kind;
    HIP_TRY(hipEventRecord(c.stageEv[c.stageUsed++], stream));
    return KZ_OK;
}

// Path state per (pixel, sample) item of a pass in flight: 8 float4 + uint4 + 3 queue words (wavefront) + 5 sample floats.
static constexpr size_t KZ_STATE_BYTES_PER_ITEM = 8 * sizeof(float4) + sizeof(uint4) + 3 * sizeof(uint32_t);
static constexpr size_t KZ_SAMPLE_BYTES_PER_ITEM = 5 * sizeof(float);

// ---- buffers of one pass context. The path-state arrays and the sample planes live in the context's arena (kz_arena.cpp), which GROWS on a side
// thread: `want` items are asked for, the call returns as soon as `minItems` are there and the arena has stopped making quick progress (graceMs; < 0: wait
// for everything), and tells how many items a pass may use now (*usable, never more than `want`). Nothing is left half-allocated on failure. ----
static int ctxEnsure(PassCtx &c, size_t want, size_t minItems, double graceMs, hipStream_t stream, size_t *usable) {
    if (!c.counts) { KZ_ALLOC(&c.counts, 8 * 520 * sizeof(uint32_t)); }
    KzArena &A = *c.arena;
    if (A.wouldReallocate(want)) HIP_TRY(hipDeviceSynchronize());      // (a small context outgrown, or a pass beyond the reserved ranges: what is there is given up first)
#ifdef KZ_EXPERIMENTS
    if (g_failAlloc > 0) { A.failCountdown.store(g_failAlloc); g_failAlloc = 0; }      // (kz_debug_fail_alloc counts the arena's physical allocations as this thread's)
#endif
    size_t got = 0;
    const int rc = A.request(want, minItems, graceMs, &got);
#ifdef KZ_EXPERIMENTS
    // what the call did not use up stays armed ON THE ARENA while its growth thread is still mapping levels (the thread is where the allocations of a growing
    // context happen: ADVICE r05 - the count used to be taken back here, so a failure injected beyond the first levels never fired)
    { std::lock_guard<std::mutex> g(A.m); if (!A.busy) g_failAlloc = A.failCountdown.exchange(0); }
#endif
    if (rc) return rc;
    KzWf W{};
    if (KZ_STATE_AOS) {                                             // (development build: two 64-B records per slot laid over arrays 0-3 and 4-7 - not supported by the arena's SoA ranges)
        return kz_fail(KZ_ERR_UNSUPPORTED, "-DKZ_STATE_AOS=1 predates the growing pass context (round 5): build a revision before it to repeat profiles/r03h_state_layout");
    }
    W.rayA.p = A.array<float4>(0); W.rayB.p = A.array<float4>(1); W.hit.p = A.array<float4>(2); W.thr.p = A.array<float4>(3);
    W.misc.p = A.array<float4>(4); W.shA.p = A.array<float4>(5); W.shB.p = A.array<float4>(6); W.shL.p = A.array<float4>(7);
    W.smp = A.array<uint4>(8);
    for (int q = 0; q < 3; ++q) W.queue[q] = A.array<uint32_t>(9 + q);
    W.counts = c.counts;
    c.wf = W;
    for (int k = 0; k < 5; ++k) c.plane[k] = A.array<float>(12 + k);
    *usable = std::min(got, want);
    return KZ_OK;
}

#ifdef KZ_EXPERIMENTS
// Launch code of the kernels of kz_experiments.h (development builds only): takes over a traversal launch of wfPass when the caller's
// KzTuning selects one of the rejected experiments. The films stay bit-identical to the product kernels' (tests/test_gpu_configs.py).
struct KzExpLaunch {
    KzScene *scene; KzDeviceState *ds; PassCtx *c; hipStream_t stream; KzWf W; KzTune tune; dim3 gTrav; size_t traceLds; int stackBound; bool st; uint32_t items;
    bool keys = false, dq = false, any = false; size_t ldsX = 0, dqLds = 0, stackBytes = 0; KzTune tuneDq{};
    int prepare() {
        const KzParams &P = scene->prm;
        keys = tune.wide && tune.keyStack == 2;
        dq = tune.wide && tune.leafQueue == 2;
        tune.ldsTop = tune.wide ? (int)std::min<size_t>((size_t)std::max(0, tune.ldsTop), std::min<size_t>(scene->nodes4.size(), 1536)) : 0;
        any = !tune.wide || keys || tune.ldsTop > 0 || dq || tune.legacyTrace || tune.mixed;
        if (!any) return KZ_OK;
        stackBytes = (size_t)P.stackDepth * KZ_BLOCK * sizeof(uint32_t);
        ldsX = (size_t)(tune.ldsStack + 1) * KZ_BLOCK * sizeof(uint32_t) * (keys ? 2 : 1) + (size_t)tune.ldsTop * sizeof(KzNode4);
        if (ldsX > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void *)kz_wf_trace_x<0, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute((const void *)kz_wf_trace_x<2, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (dq) {
            const int dqLS = std::max(2, std::min(tune.ldsStack, std::min(stackBound, 9)));             // 9 rows + queue + results = 19.4 KB per workgroup: 8 per CU
            dqLds = (size_t)4 * ((size_t)(dqLS + 1) * 64 + 128 + 192 + 2 * KZ_DQ_JOBS) * sizeof(uint32_t);
            tuneDq = tune; tuneDq.ldsStack = dqLS;
            const size_t stride = (size_t)gTrav.x * KZ_BLOCK, needOvf = stride * (size_t)std::max(1, stackBound - dqLS);
            if (needOvf > c->ovfCap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->ovf) (void)hipFree(c->ovf);
                c->ovf = nullptr; c->ovfCap = 0;
                KZ_ALLOC(&c->ovf, needOvf * sizeof(uint32_t));
                c->ovfCap = needOvf;
                tune.ovf = c->ovf;
            }
            tuneDq.ovf = c->ovf; tuneDq.ovfStride = (uint32_t)stride;
            if (items > c->litCap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->litQueue) (void)hipFree(c->litQueue);
                c->litQueue = nullptr; c->litCap = 0;
                KZ_ALLOC(&c->litQueue, (size_t)items * sizeof(uint32_t));
                c->litCap = items;
            }
        }
        return KZ_OK;
    }
    bool allowsPacket() const { return tune.wide && !tune.legacyTrace; }
    // the round-2 kernel with its options; MODE 0, 1, 2, 3
    template <int MODE> void traceX(const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, const uint32_t *qb, const uint32_t *cb) {
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        constexpr int M = MODE;
        if (tune.wide && tune.ldsTop > 0 && !st && (M == 0 || M == 2)) hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 2) ? M : 0, false, true, false, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        else if (tune.wide && keys && (M == 0 || M == 1)) {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 1) ? M : 0, true, true, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 1) ? M : 0, false, true, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        } else if (tune.wide) {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<M, true, true>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<M, false, true>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        } else {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<M, true, false>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<M, false, false>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        }
    }
    // closest-hit launches outside the bounce loop (camera rays, first-hit walk-through) and the lit-ray walk-through
    bool trace(int mode, const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, uint32_t *qb, uint32_t *cb) {
        if (!any) return false;
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        if (tune.legacyTrace && (mode == 0 || mode == 1)) {
            if (mode == 0) { if (st) hipLaunchKernelGGL((kz_wf_extend<true, false>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); else hipLaunchKernelGGL((kz_wf_extend<false, false>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); }
            else { if (st) hipLaunchKernelGGL((kz_wf_extend<true, true>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); else hipLaunchKernelGGL((kz_wf_extend<false, true>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); }
            return true;
        }
        if (mode == 0) { traceX<0>(q, cptr, cimm, head, qb, cb); return true; }
        if (mode == 1) { traceX<1>(q, cptr, cimm, head, qb, cb); return true; }
        if (mode == 2) { traceX<2>(q, cptr, cimm, head, qb, cb); return true; }
        return false;
    }
    // the traversal launches of one bounce (shadow rays of this bounce, closest-hit rays of the next)
    bool bounce(int iter, bool needExtend, uint32_t *nextQ, uint32_t *nextCount, uint32_t *shQ, uint32_t *shCount) {
        if (!any) return false;
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        if (!tune.legacyTrace && tune.mixed && P.nLights > 0 && needExtend) {
            traceX<3>(nextQ, nextCount, 0u, nextCount + 2, shQ, shCount);
            (void)stageMark(*c, stream, 1);
            return true;
        }
        if (P.nLights > 0) {
            if (dq && P.shadowFast) {
                uint32_t *litCount = W.counts + 4 * 520 + 2 * (iter + 1), *litHead = litCount + 1;
                if (st) hipLaunchKernelGGL((kz_wf_trace_dq<2, true>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount, 0u, nextCount + 3, tuneDq, c->litQueue, litCount);
                else hipLaunchKernelGGL((kz_wf_trace_dq<2, false>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount, 0u, nextCount + 3, tuneDq, c->litQueue, litCount);
                traceX<2>(c->litQueue, litCount, 0u, litHead, nullptr, nullptr);      // the few rays that cross an invisible light
            } else if (tune.legacyTrace) {
                if (st) hipLaunchKernelGGL(kz_wf_shadow<true>, gTrav, blk, stackBytes, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount);
                else hipLaunchKernelGGL(kz_wf_shadow<false>, gTrav, blk, stackBytes, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount);
            } else traceX<2>(shQ, shCount, 0u, nextCount + 3, nullptr, nullptr);
        }
        (void)stageMark(*c, stream, 3);
        if (needExtend) {
            if (dq && st) hipLaunchKernelGGL((kz_wf_trace_dq<0, true>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)nextQ, (const uint32_t *)nextCount, 0u, nextCount + 2, tuneDq, (uint32_t *)nullptr, (uint32_t *)nullptr);
            else if (dq) hipLaunchKernelGGL((kz_wf_trace_dq<0, false>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)nextQ, (const uint32_t *)nextCount, 0u, nextCount + 2, tuneDq, (uint32_t *)nullptr, (uint32_t *)nullptr);
            else trace(0, nextQ, nextCount, 0u, nextCount + 2, nullptr, nullptr);
            (void)stageMark(*c, stream, 1);
        }
        return true;
    }
};
#endif

// kz_wf_trace instantiations by (mode, stats): the launch code picks from this table instead of a ladder of macros
typedef void (*KzTraceFn)(KzParams, KzDevTables, KzWf, const uint32_t *, const uint32_t *, uint32_t, uint32_t *, KzTune, uint32_t *, uint32_t *);
static KzTraceFn traceFn(int mode, bool stats) {
    static const KzTraceFn tab[5][2] = {{kz_wf_trace<0, false>, kz_wf_trace<0, true>}, {kz_wf_trace<1, false>, kz_wf_trace<1, true>},
                                        {kz_wf_trace<2, false>, kz_wf_trace<2, true>}, {nullptr, nullptr}, {kz_wf_trace<4, false>, kz_wf_trace<4, true>}};
    return tab[mode][stats ? 1 : 0];
}


// Beam lists for pixels [p0, p0 + n) of the current pixel list: launched on `stream` (the call's stream) unless that range of this list has been
// handed to the kernel before; the kernel itself skips pixels that already have a list (from another tile set or chunk). evBeam / beamSeq tell the
// pass streams what to wait for.
static int ensureBeamBuffers(KzScene *scene, KzDeviceState *ds, hipStream_t stream) {
    const KzParams &P = scene->prm;
    const size_t framePix = (size_t)P.width * P.height;
    if (!ds->evBeam) HIP_TRY(hipEventCreateWithFlags(&ds->evBeam, hipEventDisableTiming));
    if (!ds->beamEntries || !ds->beamCount) {
        if (!ds->beamEntries) KZ_ALLOC(&ds->beamEntries, framePix * KZ_BEAM_CAP * sizeof(uint2));
        if (!ds->beamCount) KZ_ALLOC(&ds->beamCount, framePix * sizeof(uint2));
        ds->beamCap = framePix;
        HIP_TRY(hipMemsetAsync(ds->beamCount, 0xFF, framePix * sizeof(uint2), stream));       // every pixel: KZ_BEAM_UNBUILT
        ds->beamDone.clear(); ds->beamDoneGen = ds->tileGen;
    }
    return KZ_OK;
}
static int ensureBeams(KzScene *scene, KzDeviceState *ds, hipStream_t stream, uint32_t p0, uint32_t n) {
    const KzParams &P = scene->prm;
    { const int rc_ = ensureBeamBuffers(scene, ds, stream); if (rc_) return rc_; }
    if (ds->beamDoneGen != ds->tileGen) { ds->beamDone.clear(); ds->beamDoneGen = ds->tileGen; }
    for (const auto &r : ds->beamDone) if (r.first <= p0 && p0 + n <= r.first + r.second) return KZ_OK;
    const int LS = std::max(1, std::min(P.stackBound4, KZ_BEAM_STACK));      // a beam whose open set would grow beyond this leaves the rest unexplored (t_valid)
    const dim3 gBeam((n + KZ_BLOCK - 1) / KZ_BLOCK);
    const size_t beamLds = (size_t)2 * LS * KZ_BLOCK * sizeof(uint32_t);
    hipLaunchKernelGGL(kz_wf_beam, gBeam, dim3(KZ_BLOCK), beamLds, stream, P, ds->T, (const uint32_t *)(ds->pixList + p0), n, LS, ds->beamEntries, ds->beamCount);
    if (ds->statsOn) hipLaunchKernelGGL(kz_wf_beam_count, gBeam, dim3(KZ_BLOCK), 0, stream, (const uint2 *)ds->beamCount, (const uint32_t *)(ds->pixList + p0), P.width, n, ds->stats + 24);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ds->evBeam, stream));
    ++ds->beamSeq;
    ds->beamDone.emplace_back(p0, n);
    return KZ_OK;
}

// The global overflow area of the traversal stacks (entries beyond tune.ldsStack per lane, sized from the builder's worst-case bound) of one pass context.
static int ensureOverflow(KzScene *scene, KzDeviceState *ds, PassCtx &c, const KzTune &tune, hipStream_t stream) {
    const KzParams &P = scene->prm;
    const int stackBound = std::max(tune.wide ? P.stackBound4 : P.stackDepth, 2);
    const int ldsStack = std::max(2, std::min(tune.ldsStack, stackBound));
    const size_t stride = (size_t)(ds->numCU * tune.travBlocksPerCU) * KZ_BLOCK, needOvf = stride * (size_t)std::max(1, stackBound - ldsStack) * 3;      // (x 3: the key stack of kz_experiments.h; the shadow kernels of a small pass on the side stream, wfPass)
    if (needOvf > c.ovfCap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.ovf) (void)hipFree(c.ovf);
        c.ovf = nullptr; c.ovfCap = 0;
        KZ_ALLOC(&c.ovf, needOvf * sizeof(uint32_t));
        c.ovfCap = needOvf;
    }
    return KZ_OK;
}

// Passes of at most this many items put the shadow rays of a bounce beside its closest-hit rays by default (wfPass; KzRenderOpts::shadowBeside).
#ifndef KZ_BESIDE_ITEMS
#define KZ_BESIDE_ITEMS (1u << 27)
#endif
#ifndef KZ_BESIDE_ITEMS_IN_FLIGHT
#define KZ_BESIDE_ITEMS_IN_FLIGHT (1u << 24)      // ... with passes in flight or a dealer (renderOn)
#endif

// One pass of the wavefront pipeline over `items` = nPixPass x Sp (pixel, sample) items: pixels pixList[0 .. nPixPass), sample indices
// [sBegin, sBegin + Sp). Every launch goes to `stream`; queue counts stay on the device.
static int wfPass(KzScene *scene, KzDeviceState *ds, PassCtx &c, hipStream_t stream, const uint32_t *pixList, uint32_t p0, uint32_t nPixPass, uint32_t sBegin, uint32_t Sp, uint32_t items, KzTune tune, bool beams, int shadowBeside) {
    const KzParams &P = scene->prm;
    KzWf W = c.wf;
    W.outJx = c.plane[0]; W.outJy = c.plane[1]; W.outR = c.plane[2]; W.outG = c.plane[3]; W.outB = c.plane[4]; W.stats = ds->stats;
    const bool st = ds->statsOn;
    const dim3 blk(KZ_BLOCK);
    // shade: the lean variant runs 4 workgroups per CU at once (its launch bounds), so the default grid is exactly those, each looping over
    // its share: with 6 per CU the kernel ran 1.5 rounds, the second with half the CUs idle (same-call sweep, profiles/r02h_shade: shade alone
    // 24.1 ms at 6, 23.0 at 4, 22.6 at 8, 22.3 at 12 - and with two passes in flight the small grid leaves the most room for the other pass:
    // 1246 Msamples/s at 4 against 1214-1218 at 6 / 8 / 12). The EXT variant (3 resident) shows no preference on C3 and stays at 6.
    const int shadeBlocks = tune.shadeBlocksPerCU > 0 ? tune.shadeBlocksPerCU : (P.bsdfExt ? 6 : KZ_SHADE_WAVES);
    const dim3 gTrav((unsigned)(ds->numCU * tune.travBlocksPerCU)), gShade((unsigned)(ds->numCU * shadeBlocks));
    const dim3 gPacket((unsigned)(ds->numCU * 8));                  // the packet kernel is compiled for 8 waves per SIMD whatever KZ_TRACE_WAVES is
    // stack: tune.ldsStack entries per lane in LDS, the rest of the worst case (known from the builder) in a global overflow area
    const int stackBound = std::max(tune.wide ? P.stackBound4 : P.stackDepth, 2);
    tune.ldsStack = std::max(2, std::min(tune.ldsStack, stackBound));
    const size_t traceLds = (size_t)(tune.ldsStack + 1) * KZ_BLOCK * sizeof(uint32_t);      // + one scratch slot per lane (branch-free pushes)
    {
        const int rc_ = ensureOverflow(scene, ds, c, tune, stream);
        if (rc_) return rc_;
        tune.ovf = c.ovf; tune.ovfStride = (uint32_t)((size_t)gTrav.x * KZ_BLOCK);
    }
    const int maxDepth = P.maxDepth;
    c.stageUsed = 0;
    HIP_TRY(hipMemsetAsync(W.counts, 0, 8 * 520 * sizeof(uint32_t), stream));
    { int rc_ = stageMark(c, stream, -1); if (rc_) return rc_; }
    hipLaunchKernelGGL(kz_wf_generate, dim3((items + KZ_BLOCK - 1) / KZ_BLOCK), blk, 0, stream, P, ds->T, W, pixList, items, Sp, sBegin);
    { int rc_ = stageMark(c, stream, 0); if (rc_) return rc_; }
    if (maxDepth <= 0) return KZ_OK;           // Li returns 0 before the loop contributes anything
#ifdef KZ_EXPERIMENTS
    KzTune tuneX = tune; if (tuneX.batch <= 0) tuneX.batch = 128; if (tuneX.refill <= 0) tuneX.refill = 40;
    KzExpLaunch X{scene, ds, &c, stream, W, tuneX, gTrav, traceLds, stackBound, st, items};
    if (int rc_ = X.prepare()) return rc_;
#endif
    // mode 0 / 1 / 2 / 4 of kz_wf_trace on queue q (nullptr: identity) of *cptr (nullptr: cimm) entries
    const size_t ovfPart = (size_t)tune.ovfStride * (size_t)std::max(1, stackBound - tune.ldsStack);
    auto trace = [&](int mode, const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, uint32_t *qb, uint32_t *cb, hipStream_t on = nullptr) {
#ifdef KZ_EXPERIMENTS
        if (X.trace(mode, q, cptr, cimm, head, qb, cb)) return;
#endif
        KzTune t = tune;
        const bool side = on != nullptr;                 // (a launch beside the pass's own stream walks its own part of the overflow area)
        if (side) t.ovf = tune.ovf + 2 * ovfPart; else on = stream;
        if (t.batch <= 0) t.batch = 128;                 // (the least a wave reserves per global atomic: kz_wf_trace asks for more while much is left)
        // idle lanes are refilled once fewer than this many are busy. A refill of the shadow kernel is the dearer one (the invisible-light test of every new
        // ray), so it waits for more idle lanes: same-call sweep, shadow stage 21.7 / 20.9 / 20.9 ms at 40 / 32 / 28 on C3, 20.3 / 20.0 / 20.3 on C4; the
        // closest-hit kernel 32.85 / 33.3 / 34.2 on C4
        if (t.refill <= 0) t.refill = (mode == 2 || mode == 4) ? 32 : 40;
        hipLaunchKernelGGL(traceFn(mode, st), gTrav, blk, traceLds, on, P, ds->T, W, q, cptr, cimm, head, t, qb, cb);
    };
    // camera rays: pixel beams + per-sample triangle tests (kz_wf_beam / kz_wf_trace_list), the wave-level packet traversal
    // (kz_wf_trace_packet) for what the beams cannot take, or the per-lane kernel on request
    bool packet = tune.packet != 1 && P.stackBound4 <= 128;
#ifdef KZ_EXPERIMENTS
    if (!X.allowsPacket()) packet = false;
#endif
#define KZ_PACKET(ST, FX, q, cptr, cimm, headp) hipLaunchKernelGGL((kz_wf_trace_packet<ST, FX>), gPacket, blk, 0, stream, P, ds->T, W, q, cptr, cimm, headp, 8, W.queue[2], W.counts + 0)
#define KZ_PACKET4(q, cptr, cimm, headp) do { if (P.anyInvisibleLight) { if (st) KZ_PACKET(true, true, q, cptr, cimm, headp); else KZ_PACKET(false, true, q, cptr, cimm, headp); } \
                                              else { if (st) KZ_PACKET(true, false, q, cptr, cimm, headp); else KZ_PACKET(false, false, q, cptr, cimm, headp); } } while (0)
    if (packet && beams) {
        // (the lists of this pass's pixels were handed to kz_wf_beam on the call's stream - ensureBeams - and this stream has waited for it)
        const uint2 *lstEntries = ds->beamEntries, *lstHeads = ds->beamCount;
        uint32_t *fbQ = W.queue[0], *fbCount = W.counts + 8 * 520 - 8, *fbHead = fbCount + 1;      // rays of pixels whose list overflowed: the packet kernel's
        const dim3 gList((items + KZ_BLOCK - 1) / KZ_BLOCK);
#define KZ_LIST(ST, FX) hipLaunchKernelGGL((kz_wf_trace_list<ST, FX>), gList, blk, 0, stream, P, ds->T, W, pixList, items, Sp, lstEntries, lstHeads, fbQ, fbCount, W.queue[2], W.counts + 0)
        if (P.anyInvisibleLight) { if (st) KZ_LIST(true, true); else KZ_LIST(false, true); }
        else { if (st) KZ_LIST(true, false); else KZ_LIST(false, false); }
#undef KZ_LIST
        KZ_PACKET4((const uint32_t *)fbQ, (const uint32_t *)fbCount, 0u, fbHead);
    } else if (packet) {
        // (scenes with an invisible light: the epilogue queues the first hits on such a light for the walk-through launch below)
        KZ_PACKET4((const uint32_t *)nullptr, (const uint32_t *)nullptr, items, W.counts + 2);
    } else {
        trace(0, nullptr, nullptr, items, W.counts + 2, nullptr, nullptr);
        if (P.anyInvisibleLight) hipLaunchKernelGGL(kz_wf_primary_fix, gShade, blk, 0, stream, P, ds->T, W, items, W.queue[2], W.counts + 0);
    }
#undef KZ_PACKET4
#undef KZ_PACKET
    if (P.anyInvisibleLight) trace(1, W.queue[2], W.counts + 0, 0u, W.counts + 3, nullptr, nullptr);      // H6 walk-through of the first hit
    { int rc_ = stageMark(c, stream, 5); if (rc_) return rc_; }
    const uint32_t *cur = nullptr, *curCount = nullptr;
#ifdef KZ_SHADE_CONST_ARGS
    int shadeArgSlot = 0;
    for (int i = 0; i < KZ_MAX_PASSES_IN_FLIGHT; ++i) if (ds->ctx[i] == &c) shadeArgSlot = i;
    { KzShadeArgs a{P, ds->T, W}; HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_kzShadeArgs), &a, sizeof a, (size_t)shadeArgSlot * sizeof a, hipMemcpyHostToDevice, stream)); }
#endif
    const bool split = tune.shadeSplit != 0;
    const dim3 gClassify((unsigned)(ds->numCU * 8));
    // A pass of a small job (C1: 1 M items on a chip of 524 288 lanes) is a chain of ~25 dependent launches, each of which lasts as long as its slowest ray
    // whatever the number of rays: there the shadow rays of a bounce run BESIDE its closest-hit rays (the context's side stream) instead of in front of them
    // (C1 2.47 -> 2.03 ms; the gain fades with the pass size: -4 .. -6 % at 2^22 - 2^23 items of C3 / C4-like scenes, -1 % at 2^25, nothing either way at 2^27 - where the q1 asset still gains 7 %, so the rule ends there: profiles/r06v_shadow_beside).
    // A LARGE pass is another matter: on C4 each kernel saturates the VALUs by itself and sharing the chip costs 1 % (bench 1 798 / 1 812 one stream, 1 788 / 1 782
    // beside) - but a scene whose shadow rays are short-lived (the reference's own q1 asset: an object on a backdrop under three lights, shadow kernel at 0.67 VALU
    // busy) gains 6 - 10 % at EVERY pass size, all 22 scene files of the reference do. Nothing known at upload tells the two kinds apart, so renderOn MEASURES it
    // (one large pass each way per replica) and passes 1 or 2 down here; 0 reaches this function for small passes only. The film is the same bits either way.
    bool beside = !split && P.nLights > 0 && maxDepth > 1 && (shadowBeside == 2 || (shadowBeside == 0 && items <= KZ_BESIDE_ITEMS));
#ifdef KZ_EXPERIMENTS
    if (X.any) beside = false;              // (an experiment kernel of kz_experiments.h is selected: one stream)
#endif
    ds->lastInfo.shadowBeside = beside ? 1u : 0u;      // (what the last pass did: kz_last_pass_info)
    if (beside && !c.side) {
        HIP_TRY(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c.evFork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c.evJoin, hipEventDisableTiming));
    }
    for (int iter = 0; iter < maxDepth; ++iter) {
        // One kernel per bounce: the path queues ping-pong (W.queue[iter & 1] is written, the other one read). Two kernels (tune.shadeSplit): kz_wf_classify
        // reads the path queue and writes the survivors to W.queue[0]; kz_wf_shade_b reads those and writes the next path queue to W.queue[1], which
        // the classification has finished reading by then (one stream).
        uint32_t *nextQ = split ? W.queue[1] : W.queue[iter & 1], *nextCount = W.counts + 4 * (iter + 1), *shQ = W.queue[2], *shCount = W.counts + 4 * (iter + 1) + 1;
#ifdef KZ_EXPERIMENTS
        if (split) {
            uint32_t *svQ = W.queue[0], *svCount = W.counts + 6 * 520 + (iter + 1);
#define KZ_SHADE2(ST, EX) do { hipLaunchKernelGGL((kz_wf_classify<ST, EX>), gClassify, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, cur, curCount, items, svQ, svCount); \
                               hipLaunchKernelGGL((kz_wf_shade_b<ST, EX>), gShade, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, (const uint32_t *)svQ, (const uint32_t *)svCount, nextQ, nextCount, shQ, shCount); } while (0)
            if (st) { if (P.bsdfExt) KZ_SHADE2(true, KZ_X_ALL); else KZ_SHADE2(true, 0); }
            else { if (P.bsdfExt) KZ_SHADE2(false, KZ_X_ALL); else KZ_SHADE2(false, 0); }
#undef KZ_SHADE2
        } else
#endif
        {
#ifdef KZ_SHADE_CONST_ARGS
#define KZ_SHADE(ST, EX) hipLaunchKernelGGL((kz_wf_shade<ST, EX>), gShade, blk, 0, stream, shadeArgSlot, pixList, Sp, sBegin, iter, cur, curCount, items, nextQ, nextCount, shQ, shCount)
#else
#define KZ_SHADE(ST, EX) hipLaunchKernelGGL((kz_wf_shade<ST, EX>), gShade, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, cur, curCount, items, nextQ, nextCount, shQ, shCount)
#endif
            // the kernel compiled for what the scene's BSDF rows need (KzParams::bsdfExt): nothing beyond diffuse / kiss; other models only (four workgroups per CU,
            // no texture machinery); everything (textures, normal maps)
            const int xsel = P.bsdfExt == 0 ? 0 : (P.bsdfExt == KZ_X_MODELS ? KZ_X_MODELS : KZ_X_ALL);
            if (st) { if (xsel == KZ_X_ALL) KZ_SHADE(true, KZ_X_ALL); else if (xsel) KZ_SHADE(true, KZ_X_MODELS); else KZ_SHADE(true, 0); }
            else { if (xsel == KZ_X_ALL) KZ_SHADE(false, KZ_X_ALL); else if (xsel) KZ_SHADE(false, KZ_X_MODELS); else KZ_SHADE(false, 0); }
#undef KZ_SHADE
        }
        const bool lastIter = iter == maxDepth - 1;
        const bool needExtend = !lastIter || P.bgPresent;
#ifdef KZ_SORT_EXPERIMENT
        const uint32_t *trQ = nextQ, *trShQ = shQ;            // what the traversal launches read
        if (g_sortExp.on) {
            if (int rc_ = g_sortExp.ensure(scene, c.wfCap)) return rc_;
            uint32_t nB = 0, nS = 0;
            if (needExtend) { if (int rc_ = g_sortExp.sort(stream, W, 0, nextQ, nextCount, 0, &nB)) return rc_; trQ = g_sortExp.qOut[0]; }
            if (P.nLights > 0) { if (int rc_ = g_sortExp.sort(stream, W, 1, shQ, shCount, 1, &nS)) return rc_; trShQ = g_sortExp.qOut[1]; }
            if (iter == maxDepth - 1) { std::fprintf(stderr, "sortexp: bits %d dir %d: sorts of this pass %.3f ms\n", g_sortExp.bits, g_sortExp.useDir, g_sortExp.sortMs); g_sortExp.sortMs = 0; }
        }
#else
        const uint32_t *trQ = nextQ, *trShQ = shQ;
#endif
        { int rc_ = stageMark(c, stream, 2); if (rc_) return rc_; }
#ifdef KZ_EXPERIMENTS
        if (X.bounce(iter, needExtend, nextQ, nextCount, shQ, shCount)) { cur = nextQ; curCount = nextCount; continue; }
#endif
        // `beside` (above): this bounce's shadow rays go to the context's side stream and run beside its closest-hit rays; the next shade
        // waits for both. (What the two kernels write is disjoint: the shadow kernels add to the sample sums and use the path queue the shade has just
        // consumed; the closest-hit kernel stores hit records and reads the other path queue.)
        hipStream_t shOn = nullptr;
        if (beside && needExtend) {
            HIP_TRY(hipEventRecord(c.evFork, stream));
            HIP_TRY(hipStreamWaitEvent(c.side, c.evFork, 0));
            shOn = c.side;
        }
        if (P.nLights > 0) {
            if (P.shadowFast) {
                // any-hit kernel without the walk-through machinery; the (rare) rays whose segment crosses an invisible-light triangle go to a
                // queue - the ping-pong path queue this bounce's shade has just consumed - and are walked through by the general kernel
                uint32_t *litQ = split ? W.queue[0] : W.queue[(iter & 1) ^ 1], *litCount = W.counts + 4 * 520 + 2 * (iter + 1), *litHead = litCount + 1;
                trace(4, trShQ, shCount, 0u, nextCount + 3, litQ, litCount, shOn);
                if (P.anyInvisibleLight) trace(2, litQ, litCount, 0u, litHead, nullptr, nullptr, shOn);
            } else trace(2, trShQ, shCount, 0u, nextCount + 3, nullptr, nullptr, shOn);
        }
        if (shOn) HIP_TRY(hipEventRecord(c.evJoin, c.side));
        { int rc_ = stageMark(c, stream, 3); if (rc_) return rc_; }
        if (needExtend) {
            trace(0, trQ, nextCount, 0u, nextCount + 2, nullptr, nullptr);
            if (shOn) HIP_TRY(hipStreamWaitEvent(stream, c.evJoin, 0));          // (the stage clock then shows the pair under "trace_bounce" and nothing under "trace_shadow")
            int rc_ = stageMark(c, stream, 1); if (rc_) return rc_;
        }
        cur = nextQ; curCount = nextCount;
    }
    if (P.bgPresent) hipLaunchKernelGGL(kz_wf_final, gShade, blk, 0, stream, P, ds->T, W, cur, curCount);
    if (st) hipLaunchKernelGGL(kz_wf_count, dim3((items + KZ_BLOCK - 1) / KZ_BLOCK), blk, 0, stream, W, items);
    HIP_TRY(hipGetLastError());
    return KZ_OK;
}

// kz_render / kz_render_tiles on one replica: plan (kz_plan.cpp: pure arithmetic, tabulated by tests/test_plan_cpu.py) -> make room -> launch.
static int renderOn(KzScene *scene, KzDeviceState *ds, const KzRenderOpts *opts) {
    int rc;
    const KzParams &P = scene->prm;
    if (opts->pipeline < 0 || opts->pipeline > 2) return kz_fail(KZ_ERR_INVALID_ARG, "pipeline %d (0 = default, 1 = megakernel, 2 = wavefront)", opts->pipeline);
    if (opts->passesInFlight < 0 || opts->passesInFlight > KZ_MAX_PASSES_IN_FLIGHT)
        return kz_fail(KZ_ERR_INVALID_ARG, "passesInFlight %d (0 = default, 1 .. %d)", opts->passesInFlight, KZ_MAX_PASSES_IN_FLIGHT);
    if (opts->shadowBeside < 0 || opts->shadowBeside > 2) return kz_fail(KZ_ERR_INVALID_ARG, "shadowBeside %d (0 = default, 1 = never, 2 = always)", opts->shadowBeside);
    if (opts->passHalves < 0 || opts->passHalves > 2) return kz_fail(KZ_ERR_INVALID_ARG, "passHalves %d (0 = default, 1 = never, 2 = always)", opts->passHalves);
    const int pipeline = opts->pipeline ? opts->pipeline : 2;
#ifdef KZ_EXPERIMENTS
    if ((rc = kzEnsureBvh2(scene, ds))) return rc;          // (development builds: several experiment kernels walk the BVH2)
#else
    if (pipeline == 1 && (rc = kzEnsureBvh2(scene, ds))) return rc;
#endif
    uint32_t s0 = opts->sampleBegin, s1 = opts->sampleEnd;
    if (s0 == 0 && s1 == 0) s1 = P.sampleCount;
    if (s0 >= s1 || s1 > P.sampleCount) return kz_fail(KZ_ERR_INVALID_ARG, "sample range [%u,%u) outside [0,%u)", s0, s1, P.sampleCount);
    hipStream_t stream = (hipStream_t)opts->stream;
    // a call is ordered behind the previous call on this replica WHATEVER stream that one was given: it clears or extends the tap sums the previous call's last
    // kernels read (a caller that alternates streams used to be ordered only when the tile set changed)
    if (ds->evCallB && stream != ds->lastStream) HIP_TRY(hipStreamWaitEvent(stream, ds->evCallB, 0));
    ds->lastStream = stream;
    KzTune tune;
    if ((rc = resolveTune(opts->tune, tune))) return rc;
    // (a development build asked for one of kz_experiments.h's kernels: such a call runs its passes plainly)
    const bool expTune = !tune.wide || tune.keyStack || tune.ldsTop || tune.leafQueue || tune.legacyTrace || tune.mixed || tune.shadeSplit;
    if (tune.filmGather != 0 && tune.filmGather != 3) return kz_fail(KZ_ERR_UNSUPPORTED, "KzTuning.filmGather %d: the staged gather kernel of round 1 is gone (round 6: the film is resolved from running tap sums for every filter); 0 = default, 3 = one lane per pixel", tune.filmGather);
    if ((rc = prepareTiles(scene, ds, opts->tiles, opts->nTiles, stream))) return rc;
    // the film: running tap sums of the frame's pixels (kz_film.hip), part of the replica like the film itself - cleared unless the call accumulates
    { const bool fresh = !ds->tapSums; if ((rc = kzFilmEnsureTapSums(scene, ds, stream))) return rc; if (!opts->accumulate && !fresh) HIP_TRY(hipMemsetAsync(ds->tapSums, 0, ds->tapSumsBytes, stream)); }
    const KzTileDealer *dealer = opts->dealer;
    if (dealer) {
        if (!dealer->counter || (dealer->takenCap && (!dealer->taken || !dealer->nTaken))) return kz_fail(KZ_ERR_INVALID_ARG, "KzTileDealer: null counter / taken buffer");
        if (pipeline != 2) return kz_fail(KZ_ERR_UNSUPPORTED, "dynamic tile dealing needs the wavefront pipeline");
        if (dealer->nTaken) *dealer->nTaken = 0;
    }
    // ---- plan. The pass contexts never take more than the caller's limit; without one, not more than 3/4 of the device and not more than what is free now plus
    // what this replica already holds for the purpose (another process or replica may own the rest).
    KzPlanIn in;
    in.pipeline = pipeline; in.nPix = ds->nPix; in.s0 = s0; in.s1 = s1;
    in.passItems = opts->passItems; in.passesInFlight = opts->passesInFlight; in.sppPerPass = opts->tune.sppPerPass;
    in.perItem = (pipeline == 2 ? KZ_STATE_BYTES_PER_ITEM : 0) + KZ_SAMPLE_BYTES_PER_ITEM;
    in.limit = opts->maxStateBytes;
    if (!in.limit) {
        size_t freeB = 0, totalB = 0;
        const size_t held = ds->ctxBytes() + kzCtxPoolBytes(ds->device);          // what this replica's contexts and the device's idle pooled contexts hold (the beam lists - one per frame pixel, 264 B each - and the film's tap sums are the replica's, not part of this budget)
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB > 0) in.limit = std::min(totalB / 4 * 3, freeB + held - std::min(freeB + held, KzArena::kRuntimeReserve + ((size_t)256 << 20)));      // (what the arena will not map anyway: its reserve for the HIP runtime)
        else in.limit = (size_t)32 << 30;
        if (in.limit == 0) return kz_fail(KZ_ERR_OOM, "the device has %.2f GB free and this replica holds no pass context: nothing is left for path state beyond the %.2f GB a context leaves to the HIP runtime's own allocations",
                                          freeB / 1e9, (KzArena::kRuntimeReserve + ((size_t)256 << 20)) / 1e9);
    }
    const uint32_t nTilesSet = (uint32_t)ds->curTiles.size();
    in.dealer = dealer != nullptr;
    if (dealer) { in.takers = dealer->takers; in.dealerBatchTiles = dealer->batchTiles; in.nTiles = nTilesSet; in.tilePixOffset = ds->tilePixOffset.data(); }
    // (what the first context was last ASKED to hold, not what it holds: a context that is still growing - or growing slowly - earns the same as a complete one;
    //  a fresh replica takes the pool's largest context and goes on from what that holds)
    in.heldBefore = ds->ctx[0] ? std::max(ds->ctx[0]->wanted, ds->ctx[0]->items()) : kzCtxPoolMaxItems(ds->device);
    KzPlan pl;
    { std::string why; if ((rc = kzPlanCall(in, pl, why))) return kz_fail(rc, "%s", why.c_str()); }
    const int nCtx = pl.nCtx;
    const bool multi = pl.multi;
    const size_t need = pl.need, limit = in.limit, perItem = in.perItem;
    const uint32_t batchTiles = pl.batchTiles;
    if (dealer && !kzDealerAgree(dealer, batchTiles, nTilesSet))
        return kz_fail(KZ_ERR_INVALID_ARG, "KzTileDealer: this taker resolved batches of %u tiles of a list of %u, another taker of the same counter resolved something else "
                                           "(all takers must pass the same tile list, sample range, pass options and `takers`)", batchTiles, nTilesSet);
    // camera rays by pixel beams: a pinhole camera with an affine sample map, a stack that fits LDS twice, unless the caller asks otherwise
    const bool beams = pipeline == 2 && P.beamOk && tune.packet != 1 && tune.packet != 2 && P.maxDepth > 0;
    {   // memory this call does not use is given back when the call needs it: contexts beyond nCtx, what the device's pool holds idle, and - under a
        // limit below what an earlier call grew them to - the tails of the contexts it does use (a growth thread still mapping towards an earlier, larger
        // target is told the new one first: ADVICE r05)
        const size_t perCtx = need * perItem;
        size_t keep = 0;
        for (int i = 0; i < nCtx; ++i) { PassCtx &c = ds->ctxAt(i); if (c.arena) c.arena->lowerTarget(need); keep += std::max(c.bytes(), perCtx); }      // (taken from the device's pool here, with whatever they hold)
        for (int i = KZ_MAX_PASSES_IN_FLIGHT - 1; i >= nCtx; --i) {
            if (!ds->ctx[i] || !ds->ctx[i]->bytes()) continue;
            if (keep + ds->ctx[i]->bytes() > limit) { HIP_TRY(hipDeviceSynchronize()); ds->ctx[i]->release(); } else keep += ds->ctx[i]->bytes();
        }
        if (keep + kzCtxPoolBytes(ds->device) > limit) (void)kzCtxPoolTrim(ds->device, limit > keep ? limit - keep : 0);
        if (keep > limit)
            for (int i = 0; i < nCtx; ++i) if (ds->ctx[i] && ds->ctx[i]->bytes() > perCtx) {
                HIP_TRY(hipDeviceSynchronize());
                // (a context that has grown into mapped levels cannot hold fewer than one of them - 1.5 GB -: under a cap below that it starts over as a small one)
                if (need <= KzArena::kSmallMax) ds->ctx[i]->arena->releaseAll(); else ds->ctx[i]->arena->shrinkTo(need);
                ds->ctx[i]->trimAux();                                                             // (a pooled context may carry another frame's overflow stacks)
            }
    }
    if (!ds->evCallA) { HIP_TRY(hipEventCreate(&ds->evCallA)); HIP_TRY(hipEventCreate(&ds->evCallB)); }
    if (multi) {
        // Different priorities put streams on different hardware queues whatever other streams the process has created (streams of one
        // priority share a small round-robin pool of queues and two of them may end up serialised on one): the pass streams cycle
        // through the priority levels the device has.
        int prLeast = 0, prGreatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
        const int mode = opts->tune.streamPriority > 0 ? opts->tune.streamPriority : 3;
        if (mode != ds->streamMode) {                                  // another policy than the streams were made with: make them again
            HIP_TRY(hipDeviceSynchronize());
            for (hipStream_t &st : ds->passStream) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            ds->streamMode = mode;
        }
        const int nPr = std::max(1, prLeast - prGreatest + 1);
        for (int i = 0; i < nCtx; ++i) {
            if (!ds->evFilm[i]) HIP_TRY(hipEventCreateWithFlags(&ds->evFilm[i], hipEventDisableTiming));
            if (ds->passStream[i]) continue;
            const int pr = mode == 1 ? (prLeast + prGreatest) / 2 : mode == 2 ? ((i & 1) ? prGreatest : prLeast) : prLeast - (i % nPr);
            HIP_TRY(hipStreamCreateWithPriority(&ds->passStream[i], hipStreamNonBlocking, pr));
        }
        if (!ds->evFork) HIP_TRY(hipEventCreateWithFlags(&ds->evFork, hipEventDisableTiming));
    }
    if (dealer && !ds->evFilm[0]) HIP_TRY(hipEventCreateWithFlags(&ds->evFilm[0], hipEventDisableTiming));
    ds->eventsUsed = 0;
    HIP_TRY(hipEventRecord(ds->evCallA, stream));
    if (multi) {
        HIP_TRY(hipEventRecord(ds->evFork, stream));
        for (int i = 0; i < nCtx; ++i) HIP_TRY(hipStreamWaitEvent(ds->passStream[i], ds->evFork, 0));
    }
    // Everything ELSE the call allocates is allocated now, before the first context is asked to grow: behind a wipe the growth thread takes whatever
    // clean memory there is the moment it appears, and a hipMalloc of this thread issued after that waits for the wipe like any other (round 5: the beam
    // lists, allocated inside the first pass, held the first call of a job for seconds while its context was already 80 GB large).
    if (pl.autoShape) for (int i = 0; i < nCtx; ++i) ds->ctxAt(i).wanted = need;
    KZ_TRACE("renderOn: %u pixels x samples [%u, %u), target pass %zu items (%u px x %u spp), limit %.1f GB, %d context(s)", ds->nPix, s0, s1, need, pl.pixPerPass, pl.S, limit / 1e9, nCtx);
    if (beams && (rc = ensureBeamBuffers(scene, ds, stream))) return rc;
    if (pipeline == 2) for (int i = 0; i < nCtx; ++i) if ((rc = ensureOverflow(scene, ds, ds->ctxAt(i), tune, multi ? ds->passStream[i] : stream))) return rc;
    KZ_TRACE("renderOn: beam / overflow buffers there");
    // evFilm[i] of a context that has not run a pass in this call must not be waited for: inFlight marks the ones that have
    bool inFlight[KZ_MAX_PASSES_IN_FLIGHT] = {};
    uint32_t pass = 0;
    size_t firstPassItems = 0, largestPassItems = 0;
    // the context the next pass runs in, with what a pass may use of it now
    size_t lastAvail = need;
    auto nextCtx = [&](size_t *usable) -> int {
        const int ci = multi ? (int)(pass % (uint32_t)nCtx) : 0;
        PassCtx &c = ds->ctxAt(ci);
        hipStream_t pst = multi ? ds->passStream[ci] : stream;
        // While the context is still growing the host must not plan the whole job on what is mapped NOW (it queues passes a thousand times faster than
        // the device runs them): it stays one pass ahead - pass k + 1 is planned when pass k - 1 has finished, with what has been mapped by then.
        if (pl.grow && !dealer && lastAvail < need && pass >= 2) HIP_TRY(hipEventSynchronize(ds->events[pass - 2].b));
        // Back-pressure of dynamic dealing: the host takes the next batch only when the context it needs has finished its previous pass, so a device
        // holds at most nCtx passes - never the whole frame - and a slower device simply comes back to the counter less often.
        if (dealer && inFlight[ci]) HIP_TRY(hipEventSynchronize(ds->evFilm[ci]));
        const int rc_ = ctxEnsure(c, need, pl.minStart, pl.graceMs, pst, usable);
        lastAvail = *usable;
        KZ_TRACE("pass %u: context %d holds %zu M items", pass, ci, *usable >> 20);
        return rc_;
    };
    // one pass: pixels [p0, p0 + nPixPass) of the pixel list x sample indices [s, s + Sp), in the context nextCtx has prepared
    auto onePass = [&](uint32_t p0, uint32_t nPixPass, uint32_t s, uint32_t Sp) -> int {
        const uint32_t *pixList = ds->pixList + p0;
        const size_t items = (size_t)nPixPass * Sp;
        const int ci = multi ? (int)(pass % (uint32_t)nCtx) : 0;
        PassCtx &c = ds->ctxAt(ci);
        hipStream_t pst = multi ? ds->passStream[ci] : stream;
        if (items > c.items()) return kz_fail(KZ_ERR_STATE, "pass of %zu items planned on a context that holds %zu", items, c.items());      // (the planner's invariant, kz_plan.h: never a write beyond what is mapped)
        if (!firstPassItems) firstPassItems = items;
        largestPassItems = std::max(largestPassItems, items);
        if (beams) {
            if ((rc = ensureBeams(scene, ds, stream, p0, nPixPass))) return rc;
            if (pst != stream && c.beamSeen != ds->beamSeq) { HIP_TRY(hipStreamWaitEvent(pst, ds->evBeam, 0)); c.beamSeen = ds->beamSeq; }
        }
        if (ds->eventsUsed == ds->events.size()) {
            EventPair ep; HIP_TRY(hipEventCreate(&ep.a)); HIP_TRY(hipEventCreate(&ep.b)); ds->events.push_back(ep);
        }
        EventPair &ep = ds->events[ds->eventsUsed++];
        const dim3 grid((unsigned)((items + KZ_BLOCK - 1) / KZ_BLOCK));
        HIP_TRY(hipEventRecord(ep.a, pst));
        // How a LARGE pass runs is measured, not guessed. Three ways give the same film bit for bit: (0) one stream; (1) the shadow rays of a bounce beside its
        // closest-hit rays (wfPass tells why no rule of thumb separates the scenes that gain 6 - 10 % from those that lose 1 %); (2) the pass as two HALVES of its
        // pixels side by side - two views of the context's arrays, the second on a stream of its own, disjoint pixels and therefore disjoint tap sums -, which lets
        // one half's shade kernel run beside the other half's traversal (materials_scene + 7 %, C4 - 1.4 %). With both options left at 0, one pass at a time and
        // nobody counting, the replica times one large pass each way and one stream twice (equal sizes), waits for the fourth when a fifth comes, and keeps for its
        // scene what was fastest per item - "one stream" unless something beat it by 3 %. (Such a wait is nothing new: a call already stays one pass ahead of a context that still grows.)
        int sb = opts->shadowBeside, probe = -1;
        bool halves = opts->passHalves == 2;
        if (pipeline == 2 && opts->shadowBeside == 0 && opts->passHalves == 0 && items > KZ_BESIDE_ITEMS) {
            if (multi || dealer || ds->statsOn || expTune) sb = 1;
            else {
                if (ds->largeMode < 0 && ds->probeLaunched == 4) {
                    HIP_TRY(hipEventSynchronize(ds->evProbe[3][1]));
                    double per[4];
                    for (int k = 0; k < 4; ++k) { HIP_TRY(hipEventElapsedTime(&ds->probeMs[k], ds->evProbe[k][0], ds->evProbe[k][1])); per[k] = (double)ds->probeMs[k] / (double)ds->probeItems[k]; }
                    // probe 0 ran on one stream, 1 beside, 2 as halves, 3 on one stream again: the yardstick is the better of its two timings - the first pass at a
                    // size is often slow (its context has just grown, behind the driver's wipe), and one slow pass must neither hide a gain nor invent one
                    const double one = std::min(per[0], per[3]);
                    ds->largeMode = 0;
                    if (per[1] < 0.97 * one) ds->largeMode = 1;
                    if (per[2] < 0.97 * one && per[2] < per[1]) ds->largeMode = 2;
                    KZ_TRACE("large passes of %zu items: %.2f / %.2f ms one stream, %.2f ms beside, %.2f ms as halves -> mode %d", ds->probeItems[0], ds->probeMs[0], ds->probeMs[3], ds->probeMs[1], ds->probeMs[2], ds->largeMode);
                }
                int mode = 0;
                if (ds->largeMode >= 0) mode = ds->largeMode;
                else if (ds->probeLaunched == 0 || items > ds->probeItems[0]) { probe = 0; ds->probeLaunched = 0; mode = 0; }      // (a larger pass than the one timed: the context has grown - start over at this size)
                else if (items == ds->probeItems[0]) { probe = ds->probeLaunched; mode = probe == 3 ? 0 : probe; }
                sb = mode == 1 ? 2 : 1; halves = mode == 2;                   // (a remainder pass of another size: one stream, not comparable)
            }
        }
        if (halves && (pipeline != 2 || multi || dealer || expTune || nPixPass < 256)) halves = false;      // (passes in flight and dealt batches overlap already)
        // ... and for the same reason the small-pass rule ends earlier under them: two passes of 2^22 items in flight still gain from "beside" (+1.5 .. +11 %), at
        // 2^24 - 2^25 C3 / C4 are neutral, from 2^26 they lose 2 - 3 % (the q1 asset gains at every size: whoever renders such scenes that way says 2)
        if (sb == 0 && (multi || dealer) && items > KZ_BESIDE_ITEMS_IN_FLIGHT) sb = 1;
        if (halves) {                                                         // (what the halves need is made before the clock of a timed pass starts)
            if (!c.halfStream) {
                HIP_TRY(hipStreamCreateWithFlags(&c.halfStream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&c.evHalfFork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&c.evHalfJoin, hipEventDisableTiming));
            }
            for (int h = 0; h < 2; ++h) {
                if (!c.view[h]) c.view[h] = new PassCtx();
                if (!c.view[h]->counts) KZ_ALLOC(&c.view[h]->counts, 8 * 520 * sizeof(uint32_t));
                if ((rc = ensureOverflow(scene, ds, *c.view[h], tune, h ? c.halfStream : pst))) return rc;
            }
        }
        if (probe >= 0) {
            for (int k = 0; k < 2; ++k) if (!ds->evProbe[probe][k]) HIP_TRY(hipEventCreate(&ds->evProbe[probe][k]));
            HIP_TRY(hipEventRecord(ds->evProbe[probe][0], pst));
        }
        const int prev = (ci + nCtx - 1) % nCtx;
        ds->lastStageCtx = nullptr;
        if (halves) {
            // pixels [0, nA) of the pass in the first part of every array, pixels [nA, nPixPass) behind them (nA a multiple of 64 pixels: aligned sub-arrays)
            const uint32_t nA = std::min(nPixPass - 64u, ((nPixPass + 1u) / 2u + 63u) & ~63u), nB = nPixPass - nA;
            const size_t itemsA = (size_t)nA * Sp, itemsB = (size_t)nB * Sp;
            for (int h = 0; h < 2; ++h) {
                PassCtx &v = *c.view[h];
                const size_t off = h ? itemsA : 0;
                v.wf = c.wf;
                v.wf.rayA.p += off; v.wf.rayB.p += off; v.wf.hit.p += off; v.wf.thr.p += off; v.wf.misc.p += off; v.wf.shA.p += off; v.wf.shB.p += off; v.wf.shL.p += off;
                v.wf.smp += off;
                for (int q = 0; q < 3; ++q) v.wf.queue[q] += off;
                v.wf.counts = v.counts;
                for (int k = 0; k < 5; ++k) v.plane[k] = c.plane[k] + off;
            }
            PassCtx &va = *c.view[0], &vb = *c.view[1];
            HIP_TRY(hipEventRecord(c.evHalfFork, pst));
            HIP_TRY(hipStreamWaitEvent(c.halfStream, c.evHalfFork, 0));
            if ((rc = wfPass(scene, ds, va, pst, pixList, p0, nA, s, Sp, (uint32_t)itemsA, tune, beams, sb))) return rc;
            if ((rc = wfPass(scene, ds, vb, c.halfStream, pixList + nA, p0 + nA, nB, s, Sp, (uint32_t)itemsB, tune, beams, sb))) return rc;
            ds->lastInfo.shadowBeside = 2u;                                   // (kz_last_pass_info: the last pass ran as halves)
            // the halves' pixels are disjoint, so are their tap sums: the two film stages need no order between them
            if ((rc = kzFilmStage(scene, ds, va, pst, pixList, nA, Sp, nullptr, tune.filmGather == 3 ? 1 : 0))) return rc;
            if ((rc = kzFilmStage(scene, ds, vb, c.halfStream, pixList + nA, nB, Sp, nullptr, tune.filmGather == 3 ? 1 : 0))) return rc;
            { int rc_ = stageMark(va, pst, 4); if (rc_) return rc_; }
            HIP_TRY(hipEventRecord(c.evHalfJoin, c.halfStream));
            HIP_TRY(hipStreamWaitEvent(pst, c.evHalfJoin, 0));
            HIP_TRY(hipEventRecord(ep.b, pst));
            HIP_TRY(hipGetLastError());
            ds->lastStageCtx = &va;                                           // (the stage clock of a pass in halves is its first half's)
        } else {
            if (pipeline == 2) { if ((rc = wfPass(scene, ds, c, pst, pixList, p0, nPixPass, s, Sp, (uint32_t)items, tune, beams, sb))) return rc; }
            else {
                float *sJx = c.plane[0], *sJy = c.plane[1], *sR = c.plane[2], *sG = c.plane[3], *sB = c.plane[4];
#define KZ_MEGA(ST, EX) hipLaunchKernelGGL((kz_path_megakernel<ST, EX>), grid, dim3(KZ_BLOCK), 0, pst, P, ds->T, pixList, (uint32_t)items, Sp, s, \
                                       (const uint32_t *)nullptr, sJx, sJy, sR, sG, sB, ds->stats)
                if (ds->statsOn) { if (P.bsdfExt) KZ_MEGA(true, KZ_X_ALL); else KZ_MEGA(true, 0); }
                else { if (P.bsdfExt) KZ_MEGA(false, KZ_X_ALL); else KZ_MEGA(false, 0); }
#undef KZ_MEGA
            }
            HIP_TRY(hipEventRecord(ep.b, pst));
            HIP_TRY(hipGetLastError());
            // the film stage adds this pass's samples to the running tap sums of its pixels, behind the film stage of the pass before it (another stream's)
            if ((rc = kzFilmStage(scene, ds, c, pst, pixList, nPixPass, Sp, (multi && pass > 0) ? ds->evFilm[prev] : nullptr, tune.filmGather == 3 ? 1 : 0))) return rc;
            if (pipeline == 2) { int rc_ = stageMark(c, pst, 4); if (rc_) return rc_; }
        }
        if (probe >= 0) { HIP_TRY(hipEventRecord(ds->evProbe[probe][1], pst)); ds->probeItems[probe] = items; ds->probeLaunched = probe + 1; }      // (film stage included, the same for all three)
        if (multi || dealer) { HIP_TRY(hipEventRecord(ds->evFilm[ci], pst)); inFlight[ci] = true; }      // (a dealer paces itself on this event even with one context)
        ds->lastCtx = ci;
        ++pass;
        return KZ_OK;
    };
    // A call that fails in the middle (an allocation of a later pass, typically) leaves NOTHING running: passes it has already launched on the internal streams - pass
    // streams, a context's side streams - are not joined into `stream` on this path, and their film stages would add to tap sums that the next call has meanwhile
    // cleared (found by injecting allocation failures under two passes in flight: the next film carried the failed call's first pass, scripts/dev/fail_sweep.py).
    auto failed = [&](int code) -> int { (void)hipDeviceSynchronize(); return code; };
    if (!dealer) { if ((rc = kzPlanRun(pl, 0u, ds->nPix, s0, s1, nextCtx, onePass))) return failed(rc); }
    else {
        // BlockGenerator::next (block.cpp:117-148): batches of the tile list from the shared counter (it may live in memory shared between processes)
        for (uint32_t tb, te; kzDealerTake(dealer, batchTiles, nTilesSet, tb, te);)
            if ((rc = kzPlanRun(pl, ds->tilePixOffset[tb], ds->tilePixOffset[te], s0, s1, nextCtx, onePass))) return failed(rc);
    }
    if (multi)                                                         // join: everything after this call on `stream` sees the film
        for (int i = 0; i < nCtx; ++i) if (inFlight[i]) HIP_TRY(hipStreamWaitEvent(stream, ds->evFilm[i], 0));
    if ((rc = kzFilmResolve(scene, ds, stream))) return rc;           // the film texels from the tap sums: once per call
    HIP_TRY(hipEventRecord(ds->evCallB, stream));
    ds->lastDual = multi;
    ds->lastInfo.passes = pass; ds->lastInfo.passesInFlight = (uint32_t)nCtx; ds->lastInfo.itemsPerPass = need; ds->lastInfo.sppPerPass = pl.S;
    ds->lastInfo.pixels = ds->nPix; ds->lastInfo.stateBytes = ds->ctxBytes(); ds->lastInfo.pixelsPerPass = pl.pixPerPass;
    ds->lastInfo.firstPassItems = firstPassItems; ds->lastInfo.largestPassItems = largestPassItems;
    ds->growNote.clear();
    for (int i = 0; i < nCtx; ++i) if (ds->ctx[i] && ds->ctx[i]->arena) { std::lock_guard<std::mutex> g(ds->ctx[i]->arena->m); if (ds->ctx[i]->arena->growthFailed) ds->growNote = ds->ctx[i]->arena->errMsg; }
    return KZ_OK;
}

extern "C" {

int kz_render(KzScene *scene, const KzRenderOpts *opts) {
    if (!opts) return kz_fail(KZ_ERR_INVALID_ARG, "null opts");
    KzDeviceState *ds; int rc;
    // opts->device addresses a replica by HIP device index; a scene resident on ONE device is addressed by any zero-initialised opts
    if ((rc = findReplica(scene, opts->device, &ds))) {
        KzReplicaSet *rs = scene ? replicaSet(scene) : nullptr;
        bool single = false;
        if (rs) { std::lock_guard<std::mutex> g(rs->m); single = rs->v.size() == 1 && opts->device == 0; }
        if (!single || (rc = findReplica(scene, -1, &ds))) return rc;
    }
    return renderOn(scene, ds, opts);
}

int kz_render_tiles(KzScene *scene, const KzRenderOpts *opts, const KzTile *tiles, uint32_t nTiles, int device, float *film, size_t nFloats) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    const size_t full = ds->filmPixels * 4;
    const bool packedOut = opts && opts->packedOutput;
    if (opts && opts->dealer && film) return kz_fail(KZ_ERR_INVALID_ARG, "kz_render_tiles with a dealer hands back no film: download the batches it took (KzTileDealer.taken) with kz_film_download_tiles");
    if (film) {
        // what the buffer receives is SAID by opts->packedOutput, never inferred from its size (a full tiling with a border-less filter packs to exactly the film's size)
        if (packedOut) {
            if ((rc = checkTiles(scene->prm, tiles, nTiles))) return rc;
            const size_t packed = packedFloats(scene->prm, tiles, nTiles);
            if (!nTiles || nFloats != packed) return kz_fail(KZ_ERR_INVALID_ARG, "packedOutput: the buffer must hold the tiles' packed rects (%zu floats, kz_tiles_packed_floats), got %zu", packed, nFloats);
        } else if (nFloats != full) return kz_fail(KZ_ERR_INVALID_ARG, "film buffer must hold the whole film (%zu floats; set opts->packedOutput for the tiles' packed rects), got %zu", full, nFloats);
    }
    KzRenderOpts o{};
    if (opts) o = *opts;
    o.tiles = tiles; o.nTiles = nTiles; o.device = device;
    if ((rc = renderOn(scene, ds, &o))) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)o.stream));
    if (film && !packedOut) HIP_TRY(hipMemcpy(film, ds->film, nFloats * sizeof(float), hipMemcpyDeviceToHost));
    else if (film) return downloadTiles(scene, ds, tiles, nTiles, film, nFloats, (hipStream_t)o.stream);
    return KZ_OK;
}

// Device time of the stages of the LAST pass of the last kz_render (wavefront pipeline), from hipEvents on the launch stream:
// out[0] generate, [1] closest-hit traversal of the BOUNCE rays (the kz_wf_trace<0> launches), [2] shade, [3] shadow traversal, [4] film,
// [5] camera rays (kz_wf_beam when the lists are rebuilt, kz_wf_trace_list, kz_wf_trace_packet, the first-hit walk-through).
int kz_last_stage_ms(KzScene *scene, float *out6) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out6) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    for (int i = 0; i < 6; ++i) out6[i] = 0.f;
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    const PassCtx &c = ds->lastStageCtx ? *ds->lastStageCtx : ds->ctxAt(ds->lastCtx);
    for (size_t i = 1; i < c.stageUsed; ++i) {
        float t = 0; HIP_TRY(hipEventElapsedTime(&t, c.stageEv[i - 1], c.stageEv[i]));
        const int k = c.stageKind[i];
        if (k >= 0 && k < 6) out6[k] += t;
    }
    return KZ_OK;
}

// Why the pass context of the last kz_render stopped growing short of its target, if it did ("" otherwise): the call itself succeeds on what there is.
int kz_last_grow_note(KzScene *scene, char *buf, size_t cap) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!buf || !cap) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    std::snprintf(buf, cap, "%s", ds->growNote.c_str());
    return KZ_OK;
}

int kz_last_pass_info(KzScene *scene, KzPassInfo *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    *out = ds->lastInfo;
    out->contextItems = ds->ctx[0] ? ds->ctx[0]->items() : 0;
    return KZ_OK;
}

// What the replica on `device` (-1: the primary one) has measured about its large passes (renderOn: KzRenderOpts::shadowBeside / passHalves at 0).
int kz_pass_mode_info(KzScene *scene, int device, KzPassModeInfo *out) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    if (!out) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    *out = KzPassModeInfo{};
    out->kept = ds->largeMode; out->timedPasses = (uint32_t)ds->probeLaunched; out->items = ds->probeItems[0];
    if (ds->largeMode >= 0) { out->msOneStream[0] = ds->probeMs[0]; out->msOneStream[1] = ds->probeMs[3]; out->msShadowBeside = ds->probeMs[1]; out->msHalves = ds->probeMs[2]; }
    return KZ_OK;
}

int kz_sync_on(KzScene *scene, int device) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    return KZ_OK;
}
int kz_sync(KzScene *scene) { return kz_sync_on(scene, -1); }

int kz_last_kernel_ms(KzScene *scene, float *ms) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!ms) return kz_fail(KZ_ERR_INVALID_ARG, "null ms");
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    if (ds->lastDual && ds->lastInfo.passes) {          // passes overlap: the per-pass figure is the span of the call over its passes (film included)
        float t = 0; HIP_TRY(hipEventElapsedTime(&t, ds->evCallA, ds->evCallB));
        *ms = t / (float)ds->lastInfo.passes;
        return KZ_OK;
    }
    double tot = 0;
    for (size_t i = 0; i < ds->eventsUsed; ++i) { float t = 0; HIP_TRY(hipEventElapsedTime(&t, ds->events[i].a, ds->events[i].b)); tot += t; }
    *ms = ds->eventsUsed ? (float)(tot / ds->eventsUsed) : 0.f;
    return KZ_OK;
}

int kz_set_stats(KzScene *scene, int enable) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    ds->statsOn = enable != 0;
    return KZ_OK;
}

int kz_get_stats(KzScene *scene, KzStats *out, int reset) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, ds->stats, sizeof h, hipMemcpyDeviceToHost));
    { unsigned long long b[3]; HIP_TRY(hipMemcpy(b, ds->stats + 24, sizeof b, hipMemcpyDeviceToHost)); out->beamPixels = b[0]; out->beamListEntries = b[1]; out->beamCompletePixels = b[2];
      if (reset) HIP_TRY(hipMemset(ds->stats + 24, 0, sizeof b)); }
    out->samples = h[0]; out->rays = h[1]; out->nodeVisits = h[2]; out->triTests = h[3]; out->shadedHits = h[4]; out->lightSamples = h[5]; out->droppedSamples = h[6];
#ifdef KZ_LANESTAT
    {
        unsigned long long ls[16];
        HIP_TRY(hipMemcpy(ls, ds->stats + 8, sizeof ls, hipMemcpyDeviceToHost));
        for (int m = 0; m < 2; ++m) {
            const unsigned long long *a = ls + 8 * m;
            std::fprintf(stderr, "lanestat %s: node iters %llu  active/iter %.2f  inner/iter %.2f | leaf phases %llu  lanes/phase %.2f  tri iters/phase %.2f | refills %llu  lanes/refill %.2f\n",
                         m ? "shadow " : "closest", a[0], a[0] ? (double)a[1] / a[0] : 0.0, a[0] ? (double)a[2] / a[0] : 0.0, a[3], a[3] ? (double)a[4] / a[3] : 0.0,
                         a[3] ? (double)a[7] / a[3] : 0.0, a[5], a[5] ? (double)a[6] / a[5] : 0.0);
        }
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ls));
    }
#endif
#ifdef KZ_TRACESTAT
    {
        unsigned long long ts[10];
        HIP_TRY(hipMemcpy(ts, ds->stats + 8, sizeof ts, hipMemcpyDeviceToHost));
        for (int m = 0; m < 2; ++m) {
            const unsigned long long *a = ts + 4 * m; const double tot = (double)(a[0] + a[1] + a[2] + a[3]);
            std::fprintf(stderr, "tracestat %s: refill %.2f %%  node phase %.2f %%  leaf phase (+ pop / finish) %.2f %% (triangle loop alone %.2f %%)  loop %.2f %%\n", m ? "shadow " : "closest",
                         tot > 0 ? 100.0 * a[0] / tot : 0.0, tot > 0 ? 100.0 * a[1] / tot : 0.0, tot > 0 ? 100.0 * a[2] / tot : 0.0, tot > 0 ? 100.0 * ts[8 + m] / tot : 0.0, tot > 0 ? 100.0 * a[3] / tot : 0.0);
        }
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ts));
    }
#endif
#ifdef KZ_SHADESTAT
    {
        unsigned long long ss[14]; double tot = 0;
        HIP_TRY(hipMemcpy(ss, ds->stats + 8, sizeof ss, hipMemcpyDeviceToHost));
        static const char *nm[14] = {"passA: rest (emitter, classification, roulette)", "compact+barrier", "B:loads+setup", "B:light draws+sample", "B:eval+pdf+shadow stores", "B:bsdf draws", "B:bsdf sample", "B:next-ray stores",
                                     "barrier+staging+flush", "loop head", "passA: queue+hit+ray loads", "passA: shading record arrives", "passA: postIntersect", "-"};
        for (int k = 0; k < 14; ++k) tot += (double)ss[k];
        for (int k = 0; k < 14; ++k) std::fprintf(stderr, "shadestat %-48s %6.2f %%\n", nm[k], tot > 0 ? 100.0 * (double)ss[k] / tot : 0.0);
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ss));
    }
#endif
    if (reset) HIP_TRY(hipMemset(ds->stats, 0, sizeof h));
    return KZ_OK;
}

int kz_render_samples(KzScene *scene, uint32_t n, const int32_t *pxy, const uint32_t *idx, float *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!pxy || !idx || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    if ((rc = kzEnsureBvh2(scene, ds))) return rc;
    const KzParams &P = scene->prm;
    std::vector<uint32_t> pl(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (pxy[2 * i] < 0 || pxy[2 * i] >= P.width || pxy[2 * i + 1] < 0 || pxy[2 * i + 1] >= P.height || idx[i] >= P.sampleCount)
            return kz_fail(KZ_ERR_INVALID_ARG, "sample %u: pixel (%d,%d) index %u out of range", i, pxy[2 * i], pxy[2 * i + 1], idx[i]);
        pl[i] = (uint32_t)pxy[2 * i] | ((uint32_t)pxy[2 * i + 1] << 16);
    }
    DevMem dP, dI, dOut;
    KZ_ALLOC(&dP.p, (size_t)n * 4); KZ_ALLOC(&dI.p, (size_t)n * 4); KZ_ALLOC(&dOut.p, (size_t)n * 20);
    float *dO = dOut.as<float>();
    HIP_TRY(hipMemcpy(dP.p, pl.data(), (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dI.p, idx, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((kz_path_megakernel<false, KZ_X_ALL>), dim3((n + KZ_BLOCK - 1) / KZ_BLOCK), dim3(KZ_BLOCK), 0, 0, P, ds->T, dP.as<uint32_t>(), n, 1u, 0u, dI.as<uint32_t>(),
                       dO, dO + n, dO + 2 * (size_t)n, dO + 3 * (size_t)n, dO + 4 * (size_t)n, ds->stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> h((size_t)n * 5);
    HIP_TRY(hipMemcpy(h.data(), dO, (size_t)n * 20, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        out[5 * i] = (float)pxy[2 * i] + h[i]; out[5 * i + 1] = (float)pxy[2 * i + 1] + h[n + i];
        out[5 * i + 2] = h[2 * (size_t)n + i]; out[5 * i + 3] = h[3 * (size_t)n + i]; out[5 * i + 4] = h[4 * (size_t)n + i];
    }
    return KZ_OK;
}


} // extern "C"
