// kz_arena.cpp - the path-state memory of a pass context: ONE reserved virtual range per context, physical memory mapped into it level by level
// on a side thread while the passes already run on what is there (host code only).
//
// Why (round 5, profiles/r05a_alloc): a hipMalloc of the 13 state arrays of a 2^30-item pass (175 GB) took anything between 6 ms and 5.8 s. The cost is
// not the allocation - a GB of CLEAN memory maps in ~15 us whether it comes from hipMalloc or hipMemCreate, 176 GB in 10 ms - it is the driver's
// asynchronous wipe of memory some process (this one or the one before it) has just released (~33 GB/s): an allocation that needs more than what is
// clean at that moment blocks in ONE call until the wipe has got far enough. A renderer cannot know how much is clean. So the context is not allocated,
// it GROWS: the virtual ranges of its arrays are reserved for the largest pass (reserving is free), a side thread maps physical chunks into all of them
// level by level (hipMemCreate + hipMemMap + hipMemSetAccess; a level = the same item range of every array), and the pass schedule (kz_render.hip)
// sizes each pass by what is mapped at that moment. On clean memory the context is at full size before the first pass has been planned; behind a wipe
// the first passes are small and the job is under way while the stall is served on the side thread. Memory mapped this way is as fast as hipMalloc
// memory (copy 5.3 vs 4.8 TB/s, random 16-B gathers 48.5 vs 49.1 G/s over 32 GB: profiles/r05a_alloc/alloc_grow.json).
// Contexts outlive the replica that grew them: releaseReplica hands them to a per-device pool, the next replica on that device takes them from there
// (a process that renders scene after scene - the reference's 22 parameter pictures - would otherwise release 175 GB per scene and wait for its own wipe).
#include "kz_state.h"

#include <algorithm>
#include <chrono>
#include <cstring>

static constexpr size_t KZ_ARENA_ALIGN_ITEMS = (size_t)1 << 20;      // levels and capacities are multiples of this: every array's chunk is a multiple of 4 MB
static constexpr size_t KZ_ARENA_MAX_LEVEL_ITEMS = (size_t)1 << 25;  // a level maps at most this many items of every array (512 MB chunks for the float4 arrays)

static size_t roundUpItems(size_t n) { return (n + KZ_ARENA_ALIGN_ITEMS - 1) / KZ_ARENA_ALIGN_ITEMS * KZ_ARENA_ALIGN_ITEMS; }

KzArena::KzArena(int dev) : device(dev) {
    // element sizes in the order the pass launcher reads them (kz_render.hip: wfPass): 8 float4 fields, the sampler record, 3 queues, 5 sample planes
    const size_t e[kArrays] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 4, 4, 4, 4, 4};
    std::memcpy(elem, e, sizeof e);
    std::memset(base, 0, sizeof base);
}

KzArena::~KzArena() { releaseAll(); }

size_t KzArena::bytesPerItem() { return 8 * 16 + 16 + 3 * 4 + 5 * 4; }

void KzArena::stopThread() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> g(m); stop = true; }
    cvWork.notify_all();
    th.join();
    stop = false;
}

// Unmaps and releases every level from `keepLevels` on. The caller has made sure that nothing on the device uses them.
void KzArena::dropLevels(size_t keepLevels) {
    while (levels.size() > keepLevels) {
        Level &L = levels.back();
        for (int a = 0; a < L.mappedArrays; ++a) {
            (void)hipMemUnmap(base[a] + L.firstItem * elem[a], L.items * elem[a]);
            (void)hipMemRelease(L.h[a]);
        }
        levels.pop_back();
    }
    size_t mp = 0;
    for (const Level &L : levels) mp = L.firstItem + L.items;
    mapped.store(mp);
}

void KzArena::releaseAll() {
    stopThread();
    (void)hipSetDevice(device);
    dropLevels(0);
    if (va) { (void)hipMemAddressFree(va, vaBytes); va = nullptr; vaBytes = 0; }
    capItems = 0; target = 0; err = 0; errMsg.clear();
}

// Reserves the virtual ranges for `cap` items (everything mapped so far is given up when the reservation has to grow: rare - the default
// reservation covers the largest default pass).
int KzArena::reserve(size_t cap) {
    cap = roundUpItems(std::max<size_t>(cap, KZ_ARENA_ALIGN_ITEMS));
    if (cap <= capItems) return KZ_OK;
    releaseAll();
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = cap * bytesPerItem();
    void *p = nullptr;
    hipError_t e = hipMemAddressReserve(&p, bytes, (size_t)2 << 20, nullptr, 0);
    if (e != hipSuccess) return kz_fail(KZ_ERR_HIP, "hipMemAddressReserve of %zu bytes of virtual address space failed: %s", bytes, hipGetErrorString(e));
    va = (char *)p; vaBytes = bytes; capItems = cap;
    size_t off = 0;
    for (int a = 0; a < kArrays; ++a) { base[a] = va + off; off += cap * elem[a]; }
    return KZ_OK;
}

// One level: the item range [first, first + items) of every array. Runs on the growth thread.
bool KzArena::growOneLevel(size_t first, size_t items) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    hipMemAccessDesc ad{};
    ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    Level L{};
    L.firstItem = first; L.items = items; L.mappedArrays = 0;
    hipError_t e = hipSuccess; const char *what = "";
    for (int a = 0; a < kArrays; ++a) {
        const size_t bytes = items * elem[a];
        int fc = failCountdown.load();
        if (fc > 0 && failCountdown.compare_exchange_strong(fc, fc - 1) && fc == 1) { e = hipErrorOutOfMemory; what = "hipMemCreate (kz_debug_fail_alloc)"; break; }
        if ((e = hipMemCreate(&L.h[a], bytes, &prop, 0)) != hipSuccess) { what = "hipMemCreate"; break; }
        if ((e = hipMemMap(base[a] + first * elem[a], bytes, 0, L.h[a], 0)) != hipSuccess) { (void)hipMemRelease(L.h[a]); what = "hipMemMap"; break; }
        if ((e = hipMemSetAccess(base[a] + first * elem[a], bytes, &ad, 1)) != hipSuccess) { (void)hipMemUnmap(base[a] + first * elem[a], bytes); (void)hipMemRelease(L.h[a]); what = "hipMemSetAccess"; break; }
        L.mappedArrays = a + 1;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int a = 0; a < L.mappedArrays; ++a) { (void)hipMemUnmap(base[a] + first * elem[a], items * elem[a]); (void)hipMemRelease(L.h[a]); }
        std::lock_guard<std::mutex> g(m);
        err = e == hipErrorOutOfMemory ? KZ_ERR_OOM : KZ_ERR_HIP;
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s of a path-state level (%zu items, %zu bytes over %d arrays) failed: %s", what, items, items * bytesPerItem(), kArrays, hipGetErrorString(e));
        errMsg = buf;
        target = mapped.load();                                   // stop growing: the passes keep what there is
        return false;
    }
    std::lock_guard<std::mutex> g(m);
    levels.push_back(L);
    mapped.store(first + items);
    lastProgress = std::chrono::steady_clock::now();
    return true;
}

void KzArena::growLoop() {
    (void)hipSetDevice(device);
    for (;;) {
        size_t first, items;
        {
            std::unique_lock<std::mutex> lk(m);
            busy = false;
            cvProgress.notify_all();
            cvWork.wait(lk, [&] { return stop || mapped.load() < target; });
            if (stop) return;
            busy = true;
            first = mapped.load();
            // levels double from 2^20 items up to 2^25, so that a small job maps a small context and a large one needs few chunks
            items = std::min({std::max(first, KZ_ARENA_ALIGN_ITEMS), KZ_ARENA_MAX_LEVEL_ITEMS, roundUpItems(target - first), capItems - first});
        }
        (void)growOneLevel(first, items);
        cvProgress.notify_all();
    }
}

// Asks for `items` items (the growth thread maps level after level until they are there) and waits
//   - until at least `minItems` are mapped (or growing has failed), and then
//   - while the thread keeps making progress towards `items`: a level on clean memory takes ~0.3 ms, so on a fresh device the whole context is there
//     after a few milliseconds; the wait ends as soon as a level takes longer than `graceMs` (the driver is wiping: go on with what there is) or the
//     context is complete. graceMs < 0: wait for all of it.
// Returns the items mapped; an error only when fewer than minItems can be had.
int KzArena::request(size_t items, size_t minItems, double graceMs, size_t *got) {
    items = std::min(roundUpItems(items), capItems);
    minItems = std::min(minItems, items);
    std::unique_lock<std::mutex> lk(m);
    if (items > target) { target = items; err = 0; errMsg.clear(); }
    if (mapped.load() < target) {
        if (!th.joinable()) { lastProgress = std::chrono::steady_clock::now(); busy = true; th = std::thread([this] { growLoop(); }); }
        else { busy = true; cvWork.notify_all(); }
    }
    for (;;) {
        const size_t mp = mapped.load();
        if (mp >= items || (!busy && mp >= target)) break;                                    // complete, or growing has stopped (failure)
        if (mp >= minItems && graceMs >= 0) {
            const double idle = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lastProgress).count();
            if (idle > graceMs) break;
            cvProgress.wait_for(lk, std::chrono::microseconds((long)((graceMs - idle) * 1000.0) + 50));
        } else cvProgress.wait_for(lk, std::chrono::milliseconds(50));
    }
    const size_t mp = mapped.load();
    if (got) *got = mp;
    if (mp < minItems || mp == 0) {
        const int code = err ? err : KZ_ERR_OOM;
        return kz_fail(code, "%s", errMsg.empty() ? "no path-state memory could be mapped" : errMsg.c_str());
    }
    return KZ_OK;
}

// Gives back everything beyond `items` (the caller has synchronised the device).
void KzArena::shrinkTo(size_t items) {
    {
        std::unique_lock<std::mutex> lk(m);
        target = std::min(target, roundUpItems(items));
        cvProgress.wait(lk, [&] { return !busy || !th.joinable(); });          // the growth thread parks when mapped >= target
    }
    (void)hipSetDevice(device);
    size_t keep = 0;
    while (keep < levels.size() && levels[keep].firstItem < items) ++keep;
    std::lock_guard<std::mutex> g(m);
    dropLevels(keep);
    target = std::min(target, mapped.load());
}

// ---- the per-device pool of pass contexts ----
static std::mutex g_poolMutex;
static std::vector<PassCtx *> g_pool[64];

PassCtx *kzCtxAcquire(int device) {
    {
        std::lock_guard<std::mutex> g(g_poolMutex);
        std::vector<PassCtx *> &v = g_pool[device & 63];
        if (!v.empty()) {                                            // the largest one first
            size_t best = 0;
            for (size_t i = 1; i < v.size(); ++i) if (v[i]->bytes() > v[best]->bytes()) best = i;
            PassCtx *c = v[best];
            v.erase(v.begin() + best);
            return c;
        }
    }
    PassCtx *c = new PassCtx();
    c->arena = new KzArena(device);
    return c;
}

void kzCtxRelease(int device, PassCtx *c) {
    if (!c) return;
    c->beamSeen = 0;
    std::lock_guard<std::mutex> g(g_poolMutex);
    g_pool[device & 63].push_back(c);
}

size_t kzCtxPoolBytes(int device) {
    std::lock_guard<std::mutex> g(g_poolMutex);
    size_t b = 0;
    for (PassCtx *c : g_pool[device & 63]) b += c->bytes();
    return b;
}

// Releases pooled contexts of `device` until at most keepBytes remain (0: all of them); returns what was released.
size_t kzCtxPoolTrim(int device, size_t keepBytes) {
    std::vector<PassCtx *> gone;
    {
        std::lock_guard<std::mutex> g(g_poolMutex);
        std::vector<PassCtx *> &v = g_pool[device & 63];
        size_t have = 0;
        for (PassCtx *c : v) have += c->bytes();
        while (!v.empty() && have > keepBytes) { have -= std::min(have, v.back()->bytes()); gone.push_back(v.back()); v.pop_back(); }
        if (keepBytes == 0) { gone.insert(gone.end(), v.begin(), v.end()); v.clear(); }
    }
    size_t freed = 0;
    if (!gone.empty()) { (void)hipSetDevice(device); (void)hipDeviceSynchronize(); }
    for (PassCtx *c : gone) { freed += c->bytes(); c->destroy(); delete c; }
    return freed;
}

extern "C" int kz_device_trim(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    (void)kzCtxPoolTrim(device, 0);
    return KZ_OK;
}
