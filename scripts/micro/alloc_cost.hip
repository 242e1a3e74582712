// Microbenchmark: where do the ~2.5 s of a 175 GB path-state allocation go, and which way of getting the memory lets a
// renderer start on a small context and grow while it renders?
//   A  hipMalloc of 13 arrays (the shape of a pass context: 8 x 16 B, 1 x 16 B, 3 x 4 B, 1 x 20 B per item) for 2^24 .. 2^30 items:
//      time of the hipMalloc calls, of the FIRST touch (a store to every 4 KB), of a second touch, of hipFree, and of the same hipMalloc again
//   B  one hipMalloc of the same total
//   C  virtual memory: hipMemAddressReserve of the whole range, then hipMemCreate + hipMemMap + hipMemSetAccess per chunk (chunk sizes 256 MB .. 4 GB):
//      time per chunk = what a side thread pays to grow a context by that much
//   D  hipMallocAsync from the default pool (release threshold = max), first and second time
//   E  interference: a VALU-bound kernel (fixed work, ~10 ms) launched back to back on the main thread while a side thread does A or C;
//      its mean / max duration with and without the side thread
// Build: hipcc -O3 --offload-arch=gfx950 alloc_cost.hip -o alloc_cost -pthread    (binary not tracked)        Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(char *p, size_t bytes, size_t stride) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    const size_t step = (size_t)gridDim.x * blockDim.x * stride;
    for (; i < bytes; i += step) p[i] = 1;
}
__global__ __launch_bounds__(256) void spin(float *sink, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
    for (int i = 0; i < iters; ++i) { a = a * b + c; a = a * b + c; a = a * b + c; a = a * b + c; }
    if (a == 12345.f) sink[0] = a;
}
static const size_t kPerItem[13] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 20};

static double touchAll(const std::vector<void *> &ptrs, const std::vector<size_t> &bytes) {
    const double t0 = now();
    for (size_t k = 0; k < ptrs.size(); ++k) hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (char *)ptrs[k], bytes[k], (size_t)4096);
    CK(hipDeviceSynchronize());
    return now() - t0;
}

int main(int argc, char **argv) {
    const int maxLog = argc > 1 ? atoi(argv[1]) : 30;
    CK(hipSetDevice(0));
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("{\"free_gb\": %.1f, \"total_gb\": %.1f", fr / 1e9, tot / 1e9);
    { float *s; CK(hipMalloc(&s, 256)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, s, 1); CK(hipDeviceSynchronize()); CK(hipFree(s)); }
    // ---- A0: ONE hipMalloc by size: where does the cost per byte jump? ----
    printf(",\n \"A0_single_malloc\": [");
    {
        bool first = true;
        for (size_t mb : {64, 256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384, 32768}) {
            void *p = nullptr; const size_t bytes = mb << 20;
            double t0 = now(); hipError_t e = hipMalloc(&p, bytes); const double tM = now() - t0;
            if (e != hipSuccess) { printf("%s\n  {\"mb\": %zu, \"error\": \"%s\"}", first ? "" : ",", mb, hipGetErrorString(e)); first = false; (void)hipGetLastError(); continue; }
            std::vector<void *> pv{p}; std::vector<size_t> bv{bytes};
            const double tT = touchAll(pv, bv);
            t0 = now(); CK(hipFree(p)); const double tF = now() - t0;
            size_t f2 = 0, t2 = 0; CK(hipMemGetInfo(&f2, &t2));
            t0 = now(); CK(hipMalloc(&p, bytes)); const double tM2 = now() - t0; CK(hipFree(p));
            printf("%s\n  {\"mb\": %zu, \"malloc_ms\": %.2f, \"touch_ms\": %.2f, \"free_ms\": %.2f, \"free_gb_after\": %.1f, \"malloc_again_ms\": %.2f, \"gb_per_s\": %.1f}", first ? "" : ",", mb, tM * 1e3, tT * 1e3, tF * 1e3, f2 / 1e9, tM2 * 1e3, bytes / 1e9 / tM);
            first = false; fflush(stdout);
        }
    }
    printf("]");
    // ---- A1: the same total (2^28 items x 176 B = 47 GB) as allocations of one size ----
    printf(",\n \"A1_47gb_in_pieces\": [");
    {
        bool first = true;
        const size_t total = ((size_t)1 << 28) * 176;
        for (size_t mb : {256, 1024, 2048, 4096}) {
            const size_t bytes = mb << 20; const int n = (int)(total / bytes);
            std::vector<void *> p(n, nullptr); int got = 0;
            double t0 = now(); for (int k = 0; k < n; ++k) { if (hipMalloc(&p[k], bytes) != hipSuccess) { (void)hipGetLastError(); break; } ++got; } const double tM = now() - t0;
            t0 = now(); for (int k = 0; k < got; ++k) CK(hipFree(p[k])); const double tF = now() - t0;
            size_t f2 = 0, t2 = 0; CK(hipMemGetInfo(&f2, &t2));
            printf("%s\n  {\"piece_mb\": %zu, \"pieces\": %d, \"allocated\": %d, \"malloc_ms\": %.1f, \"free_ms\": %.1f, \"free_gb_after\": %.1f}", first ? "" : ",", mb, n, got, tM * 1e3, tF * 1e3, f2 / 1e9);
            first = false; fflush(stdout);
        }
    }
    printf("]");
    // ---- A: 13 arrays ----
    printf(",\n \"A_13_arrays\": [");
    for (int lg = 24; lg <= maxLog; lg += 1) {
        const size_t items = (size_t)1 << lg;
        std::vector<void *> p(13); std::vector<size_t> b(13); size_t total = 0;
        for (int k = 0; k < 13; ++k) { b[k] = items * kPerItem[k]; total += b[k]; }
        double t0 = now();
        bool ok = true;
        for (int k = 0; k < 13 && ok; ++k) if (hipMalloc(&p[k], b[k]) != hipSuccess) { (void)hipGetLastError(); ok = false; for (int j = 0; j < k; ++j) CK(hipFree(p[j])); }
        const double tMalloc = now() - t0;
        if (!ok) { size_t f2 = 0, t2 = 0; CK(hipMemGetInfo(&f2, &t2)); printf(",\n  {\"log2_items\": %d, \"gb\": %.2f, \"error\": \"out of memory\", \"failed_after_ms\": %.1f, \"free_gb_now\": %.1f}", lg, total / 1e9, tMalloc * 1e3, f2 / 1e9); break; }
        const double tTouch1 = touchAll(p, b), tTouch2 = touchAll(p, b);
        t0 = now(); for (int k = 0; k < 13; ++k) CK(hipFree(p[k])); const double tFree = now() - t0;
        t0 = now(); for (int k = 0; k < 13; ++k) CK(hipMalloc(&p[k], b[k])); const double tMalloc2 = now() - t0;
        const double tTouch3 = touchAll(p, b);
        t0 = now(); for (int k = 0; k < 13; ++k) CK(hipFree(p[k])); const double tFree2 = now() - t0;
        printf("%s\n  {\"log2_items\": %d, \"gb\": %.2f, \"malloc_ms\": %.1f, \"touch1_ms\": %.1f, \"touch2_ms\": %.1f, \"free_ms\": %.1f, \"malloc_again_ms\": %.1f, \"touch_again_ms\": %.1f, \"free_again_ms\": %.1f, \"gb_per_s\": %.1f}",
               lg == 24 ? "" : ",", lg, total / 1e9, tMalloc * 1e3, tTouch1 * 1e3, tTouch2 * 1e3, tFree * 1e3, tMalloc2 * 1e3, tTouch3 * 1e3, tFree2 * 1e3, total / 1e9 / tMalloc);
        fflush(stdout);
    }
    printf("]");
    // ---- B: one allocation ----
    printf(",\n \"B_one_array\": [");
    for (int lg = 24; lg <= maxLog; lg += 2) {
        size_t total = 0; for (int k = 0; k < 13; ++k) total += ((size_t)1 << lg) * kPerItem[k];
        void *p; double t0 = now(); if (hipMalloc(&p, total) != hipSuccess) { (void)hipGetLastError(); printf("%s\n  {\"log2_items\": %d, \"gb\": %.2f, \"error\": \"out of memory\"}", lg == 24 ? "" : ",", lg, total / 1e9); break; } const double tM = now() - t0;
        std::vector<void *> pv{p}; std::vector<size_t> bv{total};
        const double tT = touchAll(pv, bv);
        t0 = now(); CK(hipFree(p)); const double tF = now() - t0;
        printf("%s\n  {\"log2_items\": %d, \"gb\": %.2f, \"malloc_ms\": %.1f, \"touch1_ms\": %.1f, \"free_ms\": %.1f}", lg == 24 ? "" : ",", lg, total / 1e9, tM * 1e3, tT * 1e3, tF * 1e3);
        fflush(stdout);
    }
    printf("]");
    // ---- C: reserve + map in chunks ----
    printf(",\n \"C_vmm\": [");
    {
        hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t granMin = 0; CK(hipMemGetAllocationGranularity(&granMin, &prop, hipMemAllocationGranularityMinimum));
        const size_t range = (size_t)64 << 30;
        bool first = true;
        for (size_t chunk : {(size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30}) {
            void *va = nullptr; double t0 = now(); CK(hipMemAddressReserve(&va, range, gran, nullptr, 0)); const double tRes = now() - t0;
            const int n = (int)std::min<size_t>(16, range / chunk);
            std::vector<hipMemGenericAllocationHandle_t> h(n);
            double tCreate = 0, tMap = 0, tAccess = 0, tMaxChunk = 0;
            hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            for (int i = 0; i < n; ++i) {
                const double a = now(); CK(hipMemCreate(&h[i], chunk, &prop, 0));
                const double b = now(); CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
                const double c = now(); CK(hipMemSetAccess((char *)va + i * chunk, chunk, &ad, 1));
                const double d = now(); tCreate += b - a; tMap += c - b; tAccess += d - c; tMaxChunk = std::max(tMaxChunk, d - a);
            }
            std::vector<void *> pv{va}; std::vector<size_t> bv{(size_t)n * chunk};
            const double tT = touchAll(pv, bv), tT2 = touchAll(pv, bv);
            t0 = now();
            for (int i = 0; i < n; ++i) { CK(hipMemUnmap((char *)va + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
            CK(hipMemAddressFree(va, range));
            const double tRel = now() - t0;
            printf("%s\n  {\"chunk_mb\": %zu, \"chunks\": %d, \"gran_mb\": %.1f, \"gran_min_kb\": %.0f, \"reserve_64gb_ms\": %.2f, \"create_ms_per_chunk\": %.2f, \"map_ms_per_chunk\": %.2f, \"access_ms_per_chunk\": %.2f, \"max_chunk_ms\": %.2f, "
                   "\"gb_per_s\": %.1f, \"touch1_ms\": %.1f, \"touch2_ms\": %.1f, \"release_ms\": %.1f}", first ? "" : ",", chunk >> 20, n, gran / 1048576.0, granMin / 1024.0, tRes * 1e3, tCreate * 1e3 / n, tMap * 1e3 / n, tAccess * 1e3 / n,
                   tMaxChunk * 1e3, n * chunk / 1e9 / (tCreate + tMap + tAccess), tT * 1e3, tT2 * 1e3, tRel * 1e3);
            first = false; fflush(stdout);
        }
    }
    printf("]");
    // ---- D: hipMallocAsync ----
    printf(",\n \"D_malloc_async\": [");
    {
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = UINT64_MAX; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; ++rep) {
            const size_t bytes = (size_t)16 << 30; void *p;
            double t0 = now(); CK(hipMallocAsync(&p, bytes, 0)); CK(hipStreamSynchronize(0)); const double tM = now() - t0;
            std::vector<void *> pv{p}; std::vector<size_t> bv{bytes};
            const double tT = touchAll(pv, bv);
            t0 = now(); CK(hipFreeAsync(p, 0)); CK(hipStreamSynchronize(0)); const double tF = now() - t0;
            printf("%s\n  {\"rep\": %d, \"gb\": 17.2, \"malloc_ms\": %.1f, \"touch_ms\": %.1f, \"free_ms\": %.1f}", rep ? "," : "", rep, tM * 1e3, tT * 1e3, tF * 1e3);
        }
        uint64_t zero = 0; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &zero)); CK(hipMemPoolTrimTo(pool, 0));
    }
    printf("]");
    // ---- E: interference with running kernels ----
    printf(",\n \"E_interference\": [");
    {
        float *sink; CK(hipMalloc(&sink, 256));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        int iters = 20000;
        auto one = [&]() { CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, st, sink, iters); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms; };
        for (int i = 0; i < 5; ++i) one();
        { const double ms = one(); iters = (int)(iters * 10.0 / ms); }
        auto series = [&](std::atomic<int> *stop, int minN) { std::vector<double> v; while ((int)v.size() < minN || (stop && !stop->load())) v.push_back(one()); return v; };
        auto stat = [&](const char *name, const std::vector<double> &v, double sideMs, bool comma) {
            double s = 0, mx = 0; for (double x : v) { s += x; mx = std::max(mx, x); }
            printf("%s\n  {\"side\": \"%s\", \"kernels\": %zu, \"mean_ms\": %.3f, \"max_ms\": %.3f, \"side_thread_ms\": %.1f}", comma ? "," : "", name, v.size(), s / v.size(), mx, sideMs); fflush(stdout);
        };
        stat("none", series(nullptr, 50), 0, false);
        {   // side thread: hipMalloc of the 13 arrays for 2^28 items (47 GB), then free
            std::atomic<int> stop{0}; double side = 0;
            std::thread th([&]() { CK(hipSetDevice(0)); std::vector<void *> p(13); const double t0 = now(); for (int k = 0; k < 13; ++k) CK(hipMalloc(&p[k], ((size_t)1 << 28) * kPerItem[k])); side = now() - t0; stop = 1; for (void *q : p) CK(hipFree(q)); });
            auto v = series(&stop, 10); th.join(); stat("hipMalloc 13 arrays x 2^28 items", v, side * 1e3, true);
        }
        {   // side thread: reserve + map 47 GB in 1 GB chunks
            std::atomic<int> stop{0}; double side = 0;
            std::thread th([&]() {
                CK(hipSetDevice(0));
                hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
                size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
                const size_t chunk = (size_t)1 << 30; const int n = 44; void *va; CK(hipMemAddressReserve(&va, n * chunk, gran, nullptr, 0));
                std::vector<hipMemGenericAllocationHandle_t> h(n); hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
                const double t0 = now();
                for (int i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0)); CK(hipMemSetAccess((char *)va + i * chunk, chunk, &ad, 1)); }
                side = now() - t0; stop = 1;
                for (int i = 0; i < n; ++i) { CK(hipMemUnmap((char *)va + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
                CK(hipMemAddressFree(va, n * chunk));
            });
            auto v = series(&stop, 10); th.join(); stat("vmm 44 x 1 GB chunks", v, side * 1e3, true);
        }
        CK(hipFree(sink));
    }
    printf("]}\n");
    return 0;
}
