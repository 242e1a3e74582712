// kz_arena.cpp - the path-state memory of a pass context (host code only). A context that holds more than 2^23 items lives in reserved virtual
// ranges - one per array - into which a side thread maps physical memory level by level while the passes already run on what is there.
//
// Why (round 5, profiles/r05a_alloc): a hipMalloc of the 13 state arrays of a 2^30-item pass (175 GB) took anything between 6 ms and 5.8 s. The cost is
// not the allocation - a GB of CLEAN memory maps in ~15 us whether it comes from hipMalloc or hipMemCreate, 176 GB in 10 ms - it is the driver's
// asynchronous wipe of memory some process (this one or the one before it) has just released (~33 GB/s): an allocation that needs more than what is
// clean at that moment blocks in ONE call until the wipe has got far enough. A renderer cannot know how much is clean. So the context is not allocated,
// it GROWS: the virtual ranges of its arrays are reserved for the largest pass (reserving is free), a side thread maps physical chunks into all of them
// level by level (hipMemCreate + hipMemMap + hipMemSetAccess; a level = the same 2^23 items of every array), and the pass schedule (kz_render.hip)
// sizes each pass by what is mapped at that moment. On clean memory the context is at full size before the first pass has been planned; behind a wipe
// the first passes are small and the job is under way while the stall is served on the side thread. Memory mapped this way is as fast as hipMalloc
// memory (copy 5.3 vs 4.8 TB/s, random 16-B gathers 48.5 vs 49.1 G/s over 32 GB: profiles/r05a_alloc/alloc_grow.json).
// All chunks of one array have ONE size: on this ROCm (7.2) hipMemSetAccess answers "invalid argument" for some sequences of chunks of different sizes
// in one reservation (scripts/micro/vmm_probe.hip -> profiles/r05a_alloc/vmm_probe.json: 4 MB behind 16 MB, 256 KB behind 4 MB), while thousands of equal
// chunks map, unmap and map again without a fault (alloc_grow.hip). Contexts of up to 2^23 items (1.5 GB: every C1 / C2-sized job, every call under a
// tight maxStateBytes) are plain hipMalloc arrays of exactly the size asked for - allocations of that size never waited in any measurement.
// Contexts outlive the replica that grew them: releaseReplica hands them to a per-device pool, the next replica on that device takes them from there
// (a process that renders scene after scene - the reference's 22 parameter pictures - would otherwise release 175 GB per scene and wait for its own wipe).
#include "kz_state.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>

static constexpr size_t KZ_ARENA_SMALL_ALIGN = (size_t)1 << 12;      // small contexts: arrays of a multiple of 4096 items

static size_t roundUp(size_t n, size_t a) { return (n + a - 1) / a * a; }

// ---- development builds only (-DKZ_EXPERIMENTS, kazen_mi355x_dev.h): the hooks that ARE process-global state. The product library contains none of them
// (nm -D libkazen_mi355x.so | grep kz_debug is empty): nothing behind its ABI depends on state outside the objects the caller holds. ----
#ifdef KZ_EXPERIMENTS
// kz_debug_grow_delay: the growth thread sleeps this long before every level - a test hook that makes "the context is still growing while the passes run"
// happen on demand (on a quiet device the memory is there before the first pass is planned)
static std::atomic<int> g_growDelayMs{0};
extern "C" void kz_debug_grow_delay(int ms) { g_growDelayMs.store(ms > 0 ? ms : 0); }
static inline int growDelayMs() { return g_growDelayMs.load(); }

std::atomic<int> g_kzTrace{0};
static const std::chrono::steady_clock::time_point g_traceT0 = std::chrono::steady_clock::now();
extern "C" void kz_debug_trace(int on) { g_kzTrace.store(on ? 1 : 0); }
void kzTraceLine(const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); std::vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::fprintf(stderr, "[kz %9.3f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_traceT0).count(), buf);
}

// kz_debug_alias_devices(n): the library then presents n LOGICAL devices, logical d living on physical device d % (devices really there). Every replica, pass
// context, pool and growth thread is keyed by the logical index, every HIP call goes to the physical one: kz_render_multi's one-host-thread-per-device driver
// runs with real concurrency on a box with ONE GPU (tests/test_gpu_multi.py) - the only way that path can be executed before an 8-GPU node exists.
static std::atomic<int> g_aliasCount{0};
extern "C" void kz_debug_alias_devices(int n) { g_aliasCount.store(n > 0 ? std::min(n, 64) : 0); }
int kzLogicalDeviceCount() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    const int a = g_aliasCount.load();
    return a > 0 ? a : n;
}
int kzPhysicalDevice(int logical) {
    if (g_aliasCount.load() <= 0) return logical;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return logical;
    return logical % n;
}
#else
static inline int growDelayMs() { return 0; }
int kzLogicalDeviceCount() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
#endif

// Every arena of the process, so that ALL mapped path state is unmapped and released - and every growth thread joined - before the HIP runtime tears
// itself down at exit (this library's static destructors run before those of libamdhip64, which it depends on): a process that exits with live
// mappings in reserved ranges crashed inside the runtime's own exit handlers (round 5: bench.py wrote its record, then segfaulted).
static std::mutex g_registryMutex;
static std::vector<KzArena *> g_registry;
static struct KzArenaJanitor {
    ~KzArenaJanitor() {
        std::vector<KzArena *> all;
        { std::lock_guard<std::mutex> g(g_registryMutex); all = g_registry; }
        for (KzArena *a : all) { (void)hipSetDevice(a->device); (void)hipDeviceSynchronize(); a->releaseAll(); }
    }
} g_janitor;

KzArena::KzArena(int dev) : device(dev) {
    // element sizes in the order the pass launcher reads them (kz_render.hip: ctxEnsure): 8 float4 fields, the sampler record, 3 queues, 5 sample planes
    const size_t e[kArrays] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 4, 4, 4, 4, 4};
    std::memcpy(elem, e, sizeof e);
    std::memset(base, 0, sizeof base);
    std::lock_guard<std::mutex> g(g_registryMutex);
    g_registry.push_back(this);
}

KzArena::~KzArena() {
    releaseAll();
    std::lock_guard<std::mutex> g(g_registryMutex);
    g_registry.erase(std::remove(g_registry.begin(), g_registry.end(), this), g_registry.end());
}

size_t KzArena::bytesPerItem() { return 8 * 16 + 16 + 3 * 4 + 5 * 4; }

void KzArena::stopThread() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> g(m); stop = true; }
    th.join();
    stop = false; busy = false;
}

// Unmaps and releases every level from `keepLevels` on. The caller has made sure that nothing on the device uses them and that the growth thread is parked.
void KzArena::dropLevels(size_t keepLevels) {
    while (levels.size() > keepLevels) {
        Level &L = levels.back();
        for (int a = 0; a < L.mappedArrays; ++a) {
            (void)hipMemUnmap(base[a] + L.firstItem * elem[a], L.items * elem[a]);
            (void)hipMemRelease(L.h[a]);
        }
        levels.pop_back();
    }
    mapped.store(levels.empty() ? 0 : levels.back().firstItem + levels.back().items);
}

void KzArena::freeSmall() {
    if (!smallItems) return;
    for (int a = 0; a < kArrays; ++a) { if (base[a]) (void)hipFree(base[a]); base[a] = nullptr; }
    smallItems = 0; mapped.store(0);
}

void KzArena::releaseAll() {
    stopThread();
    (void)hipSetDevice(device);
    freeSmall();
    dropLevels(0);
    for (int a = 0; a < kArrays; ++a) if (base[a]) { (void)hipMemAddressFree(base[a], capItems * elem[a]); base[a] = nullptr; }
    capItems = 0; target = 0; err = 0; errMsg.clear(); growthFailed = false;
}

// True when serving `items` would free or re-reserve what the context holds now: the caller then waits for the device first.
bool KzArena::wouldReallocate(size_t items) const {
    if (items <= kSmallMax && !capItems) return smallItems > 0 && roundUp(items, KZ_ARENA_SMALL_ALIGN) > smallItems;
    return smallItems > 0 || (capItems > 0 && items > capItems);
}

// hipMalloc arrays of exactly roundUp(items, 4096) items (a context of at most 2^23 items that has never been larger).
int KzArena::requestSmall(size_t items, size_t *got) {
    items = roundUp(items, KZ_ARENA_SMALL_ALIGN);
    if (items > smallItems) {
        freeSmall();
        for (int a = 0; a < kArrays; ++a) {
            void *p = nullptr;
            hipError_t e = injectedFailure() ? hipErrorOutOfMemory : hipMalloc(&p, items * elem[a]);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int b = 0; b < a; ++b) { (void)hipFree(base[b]); base[b] = nullptr; }
                return kz_fail(e == hipErrorOutOfMemory ? KZ_ERR_OOM : KZ_ERR_HIP, "device allocation of %zu bytes of path state failed: %s", items * elem[a], hipGetErrorString(e));
            }
            base[a] = (char *)p;
        }
        smallItems = items; mapped.store(items);
    }
    *got = smallItems;
    return KZ_OK;
}

// Reserves the virtual ranges for `cap` items (everything mapped so far is given up when the reservation has to grow: rare - the default
// reservation covers the largest default pass).
int KzArena::reserve(size_t cap, size_t firstTarget) {
    if (cap <= capItems) return KZ_OK;
    releaseAll();
    // few, large chunks for a context that is going to be large (a 2^30-item context: 32 levels = 544 chunks, ~25 ms on clean memory; in levels of 2^23
    // items it took ~100 ms), small ones where the caller's budget is small (a level is the granularity a context can be held to)
    levelItems = firstTarget > ((size_t)1 << 27) ? (size_t)1 << 25 : (size_t)1 << 23;
    cap = roundUp(std::max<size_t>(cap, levelItems), (size_t)1 << 25);
    HIP_TRY(hipSetDevice(device));
    // one range per array: the chunks of an array are mapped one behind the other from the start of ITS reservation, all of one size
    for (int a = 0; a < kArrays; ++a) {
        void *p = nullptr;
        hipError_t e = hipMemAddressReserve(&p, cap * elem[a], (size_t)2 << 20, nullptr, 0);
        if (e != hipSuccess) {
            for (int b = 0; b < a; ++b) { (void)hipMemAddressFree(base[b], cap * elem[b]); base[b] = nullptr; }
            return kz_fail(KZ_ERR_HIP, "hipMemAddressReserve of %zu bytes of virtual address space failed: %s", cap * elem[a], hipGetErrorString(e));
        }
        base[a] = (char *)p;
    }
    capItems = cap;
    return KZ_OK;
}

// One level: the item range [first, first + levelItems) of every array. Runs on the growth thread.
bool KzArena::growOneLevel(size_t first) {
    const size_t items = levelItems;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    hipMemAccessDesc ad{};
    ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    Level L{};
    L.firstItem = first; L.items = items; L.mappedArrays = 0;
    hipError_t e = hipSuccess; const char *what = "";
    // The HIP runtime allocates device memory of its own while kernels are being dispatched - the scratch of a queue's first kernel that spills (the packet kernel's
    // 12 B, the EXT shade kernels' 32 - 240 B per lane), the signals of a new stream - and when THAT fails there is no error code to hand back: the queue aborts the
    // process (HSA_STATUS_ERROR_OUT_OF_RESOURCES; seen with several replicas sharing one card, each growing into what the others had just released). A level is
    // therefore only mapped while kRuntimeReserve bytes stay free behind it; otherwise the context stops growing and the passes run on what there is.
    // Memory that is being WIPED (released a moment ago, by anybody) is not reported free yet, and a hipMemCreate would simply wait for it: "too little free" only
    // counts once the figure has stopped rising (the wipe moves ~33 GB/s, though not evenly: a second without 64 MB more is a full card, not a wipe).
    {
        const size_t needB = items * bytesPerItem() + kRuntimeReserve;
        size_t freeB = 0, totalB = 0, best = 0; int still = 0;
        while (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB < needB) {
            still = freeB <= best + ((size_t)64 << 20) ? still + 1 : 0;
            best = std::max(best, freeB);
            if (still >= 100) { e = hipErrorOutOfMemory; what = "the device is full (2 GB are left to the HIP runtime's own allocations): mapping"; break; }
            { std::lock_guard<std::mutex> g(m); if (stop || first >= target) return false; }      // (nobody wants this level any more)
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    for (int a = 0; a < kArrays && e == hipSuccess; ++a) {
        const size_t bytes = items * elem[a];
        if (injectedFailure()) { e = hipErrorOutOfMemory; what = "hipMemCreate (kz_debug_fail_alloc)"; break; }
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
        std::snprintf(buf, sizeof buf, "%s of a path-state level (items %zu .. %zu, %zu bytes over %d arrays) failed: %s", what, first, first + items, items * bytesPerItem(), kArrays, hipGetErrorString(e));
        errMsg = buf;
        target = mapped.load();                                   // stop growing: the passes keep what there is, and nobody asks again until the context
        growthFailed = true;                                      // has been shrunk or released (a retry per pass would cost every pass a failed attempt)
        return false;
    }
    std::lock_guard<std::mutex> g(m);
    levels.push_back(L);
    mapped.store(first + items);
    lastProgress = std::chrono::steady_clock::now();
    KZ_TRACE("arena %p: level %zu mapped (%zu M items, %.1f GB)", (void *)this, levels.size(), (first + items) >> 20, (first + items) * bytesPerItem() / 1e9);
    return true;
}

// The growth thread: level after level until the target is reached (or lowered to what is there: a failure, shrinkTo), then it ENDS - no thread is
// parked while the process renders on a complete context, idles or exits; request() starts a new one when a larger context is asked for.
void KzArena::growLoop() {
    (void)hipSetDevice(device);
    for (;;) {
        size_t first;
        {
            std::lock_guard<std::mutex> lk(m);
            if (stop || mapped.load() >= target) { busy = false; cvProgress.notify_all(); return; }
            first = mapped.load();
        }
        if (const int d = growDelayMs()) std::this_thread::sleep_for(std::chrono::milliseconds(d));
        (void)growOneLevel(first);
        cvProgress.notify_all();
    }
}

// Asks for `items` items and tells how many there are (*got; whole levels, possibly more than asked for).
//   up to 2^23 items, for a context that has never been larger: hipMalloc arrays of that size, at once;
//   otherwise the growth thread maps level after level until they are there, and the call waits
//   - until at least `minItems` are mapped (or growing has failed), and then
//   - while the thread keeps making progress towards `items`: a level on clean memory takes well under a millisecond, so on a quiet device the whole
//     context is there after a few milliseconds; the wait ends as soon as a level takes longer than `graceMs` (the driver is wiping: go on with what
//     there is) or the context is complete. graceMs < 0: wait for all of it.
// An error only when fewer than minItems can be had. The caller has waited for the device if wouldReallocate(items).
int KzArena::request(size_t items, size_t minItems, double graceMs, size_t *got) {
    if (items <= kSmallMax && !capItems) return requestSmall(items, got);
    if (smallItems) freeSmall();
    if (items > capItems) { const int rc = reserve(std::max<size_t>(items, (size_t)1 << 30), items); if (rc) return rc; }
    items = std::min(roundUp(items, levelItems), capItems);
    minItems = std::min(minItems, items);
    std::unique_lock<std::mutex> lk(m);
    if (items > target && !growthFailed) target = items;
    if (mapped.load() < target && !busy) {                            // (a running thread sees the new target under the mutex before it decides to end)
        if (th.joinable()) { lk.unlock(); th.join(); lk.lock(); }     // the previous one has ended (busy is false): reap it
        lastProgress = std::chrono::steady_clock::now(); busy = true;
        th = std::thread([this] { growLoop(); });
    }
    for (;;) {
        const size_t mp = mapped.load();
        if (mp >= items || (!busy && mp >= target)) break;                                    // complete, or growing has stopped (failure)
        if (mp >= minItems && mp > 0 && graceMs >= 0) {
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

// A call that needs fewer items than an earlier one asked for: a growth thread that is still mapping towards the old target stops at the new one
// (what is mapped already stays: shrinkTo gives memory back). Without this a context asked for 2^28 items behind a wipe kept mapping towards them
// under a later call's smaller maxStateBytes.
void KzArena::lowerTarget(size_t items) {
    if (!capItems) return;
    std::lock_guard<std::mutex> g(m);
    target = std::min(target, std::max(mapped.load(), roundUp(items, levelItems)));
}

// Gives back everything beyond `items` (the caller has synchronised the device).
void KzArena::shrinkTo(size_t items) {
    if (smallItems) { if (items == 0) freeSmall(); return; }
    if (!capItems) return;
    {
        std::unique_lock<std::mutex> lk(m);
        target = std::min(target, roundUp(items, levelItems));
        cvProgress.wait(lk, [&] { return !busy; });                            // the growth thread ends when mapped >= target
    }
    (void)hipSetDevice(device);
    const size_t keep = std::min(levels.size(), (items + levelItems - 1) / levelItems);
    std::lock_guard<std::mutex> g(m);
    dropLevels(keep);
    target = std::min(target, mapped.load());
    growthFailed = false; err = 0; errMsg.clear();
}

// ---- the per-device pool of pass contexts ----
static std::mutex g_poolMutex;
static std::vector<PassCtx *> g_pool[64];

// (`device` is the index the caller addresses the replica by; the arena lives on the physical device behind it)
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
    c->arena = new KzArena(kzPhysicalDevice(device));
    return c;
}

void kzCtxRelease(int device, PassCtx *c) {
    if (!c) return;
    c->beamSeen = 0;
    std::lock_guard<std::mutex> g(g_poolMutex);
    g_pool[device & 63].push_back(c);
}

// items of the largest pooled context of `device` (what the next replica's first context will hold when it takes it)
size_t kzCtxPoolMaxItems(int device) {
    std::lock_guard<std::mutex> g(g_poolMutex);
    size_t n = 0;
    for (PassCtx *c : g_pool[device & 63]) n = std::max(n, c->items());
    return n;
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
    if (!gone.empty()) { (void)hipSetDevice(kzPhysicalDevice(device)); (void)hipDeviceSynchronize(); }
    for (PassCtx *c : gone) { freed += c->bytes(); c->destroy(); delete c; }
    return freed;
}

size_t kzCtxPoolTrimPhysical(int hipDevice) {
    size_t freed = 0;
    for (int l = 0; l < 64; ++l) {
        if (kzPhysicalDevice(l) != hipDevice) continue;
        bool any;
        { std::lock_guard<std::mutex> g(g_poolMutex); any = !g_pool[l].empty(); }
        if (any) freed += kzCtxPoolTrim(l, 0);
    }
    return freed;
}

extern "C" int kz_device_trim(int device) {
    const int n = kzLogicalDeviceCount();
    if (device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    (void)kzCtxPoolTrim(device, 0);
    return KZ_OK;
}
