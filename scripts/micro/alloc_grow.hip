// Microbenchmark (second step; alloc_cost.hip found hipMalloc bimodal: microseconds, or ~20-30 GB/s for allocations of >= 2 GB):
// can a pass context of up to 189 GB be put together from SMALL physical chunks mapped into one reserved virtual range - fast, every time -
// and is memory mapped that way as fast to use as hipMalloc memory?
//   1  for chunk sizes 64 MB / 256 MB / 1 GB: reserve 176 GB of VA, hipMemCreate + hipMemMap + hipMemSetAccess chunk after chunk until 176 GB are
//      mapped: total time, slowest chunk, time by which 12 / 47 / 94 / 176 GB were mapped; then a streaming copy and a random 16-B gather over the
//      first 32 GB (GB/s, G gathers/s); then unmap + release: time. Twice per chunk size (the second round allocates what the first released).
//   2  the same streaming copy and gather over 32 GB from ONE hipMalloc (what the library does today), for comparison
//   3  (argument "sleep") after the last release: sleep 5 s and map 176 GB of 256 MB chunks once more - does a background clear make it cheaper?
// Build: hipcc -O3 --offload-arch=gfx950 alloc_grow.hip -o alloc_grow        Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void copyK(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) b[i] = a[i];
}
__global__ __launch_bounds__(256) void gatherK(const float4 *__restrict__ a, float *__restrict__ out, size_t n, uint32_t mul) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t step = (size_t)gridDim.x * 256;
    float s = 0;
    for (int k = 0; k < 64; ++k, i += step) { const size_t j = ((i * 2654435761ull) ^ (i >> 13) * mul) % n; s += a[j].x; }
    if (s == 1234.5f) out[0] = s;
}
// GB/s of a copy of `bytes` (read + write counted) and G gathers/s of random 16-B reads over the first `bytes` of p
static void useIt(void *p, size_t bytes, double *copyGBs, double *gatherG) {
    const size_t n = bytes / 32;                               // copy the first half onto the second
    float4 *a = (float4 *)p, *b = a + n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(copyK, dim3(256 * 16), dim3(256), 0, 0, a, b, n); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(copyK, dim3(256 * 16), dim3(256), 0, 0, a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); *copyGBs = 3.0 * 2.0 * n * 16 / 1e9 / (ms * 1e-3);
    float *out; CK(hipMalloc(&out, 256));
    const int blocks = 256 * 32;
    hipLaunchKernelGGL(gatherK, dim3(blocks), dim3(256), 0, 0, a, out, bytes / 16, 7u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gatherK, dim3(blocks), dim3(256), 0, 0, a, out, bytes / 16, 11u + r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); *gatherG = 3.0 * blocks * 256.0 * 64 / 1e9 / (ms * 1e-3);
    CK(hipFree(out)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

static void vmmRound(size_t chunk, size_t total, bool use, bool first) {
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr; CK(hipMemAddressReserve(&va, total, (size_t)2 << 20, nullptr, 0));
    const int n = (int)(total / chunk);
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    double tMax = 0, tAt[4] = {0, 0, 0, 0}; const double marks[4] = {12e9, 47e9, 94e9, 176e9};
    const double t0 = now(); int got = 0;
    for (int i = 0; i < n; ++i) {
        const double a = now();
        if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemMap((char *)va + (size_t)i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char *)va + (size_t)i * chunk, chunk, &ad, 1));
        const double b = now(); tMax = std::max(tMax, b - a); ++got;
        for (int m = 0; m < 4; ++m) if (tAt[m] == 0 && (double)(i + 1) * chunk >= marks[m] - 1) tAt[m] = b - t0;
    }
    const double tAll = now() - t0;
    double cp = 0, ga = 0;
    if (use && (size_t)got * chunk >= ((size_t)32 << 30)) useIt(va, (size_t)32 << 30, &cp, &ga);
    const double t1 = now();
    for (int i = 0; i < got; ++i) { CK(hipMemUnmap((char *)va + (size_t)i * chunk, chunk)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(va, total));
    const double tRel = now() - t1;
    printf("%s\n  {\"chunk_mb\": %zu, \"chunks\": %d, \"mapped\": %d, \"map_all_ms\": %.1f, \"slowest_chunk_ms\": %.2f, \"ms_to_12gb\": %.1f, \"ms_to_47gb\": %.1f, \"ms_to_94gb\": %.1f, \"ms_to_176gb\": %.1f, "
           "\"copy_gb_per_s\": %.0f, \"gather_g_per_s\": %.2f, \"release_ms\": %.1f}", first ? "" : ",", chunk >> 20, n, got, tAll * 1e3, tMax * 1e3, tAt[0] * 1e3, tAt[1] * 1e3, tAt[2] * 1e3, tAt[3] * 1e3, cp, ga, tRel * 1e3);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const bool doSleep = argc > 1 && !strcmp(argv[1], "sleep");
    CK(hipSetDevice(0));
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("{\"free_gb\": %.1f, \"total_gb\": %.1f,\n \"vmm_rounds\": [", fr / 1e9, tot / 1e9);
    const size_t total = (size_t)176 << 30;
    bool first = true;
    for (size_t chunk : {(size_t)256 << 20, (size_t)64 << 20, (size_t)1 << 30})
        for (int rep = 0; rep < 2; ++rep) { vmmRound(chunk, total, rep == 0, first); first = false; }
    printf("],\n \"hipMalloc_32gb\": ");
    {
        void *p; const double t0 = now(); CK(hipMalloc(&p, (size_t)32 << 30)); const double tM = now() - t0;
        double cp, ga; useIt(p, (size_t)32 << 30, &cp, &ga);
        const double t1 = now(); CK(hipFree(p)); const double tF = now() - t1;
        printf("{\"malloc_ms\": %.1f, \"copy_gb_per_s\": %.0f, \"gather_g_per_s\": %.2f, \"free_ms\": %.1f}", tM * 1e3, cp, ga, tF * 1e3);
    }
    if (doSleep) {
        printf(",\n \"after_5s_sleep\": [");
        std::this_thread::sleep_for(std::chrono::seconds(5));
        vmmRound((size_t)256 << 20, total, false, true);
        printf("]");
    }
    printf("}\n");
    return 0;
}
