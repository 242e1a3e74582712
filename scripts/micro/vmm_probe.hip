// Which hipMemMap / hipMemSetAccess patterns does this ROCm accept? (round 5: the first form of kz_arena.cpp - ONE reservation for all 17 arrays of a pass
// context, the first chunk of every array mapped at that array's offset, i.e. NOT contiguous with the previous map - failed in hipMemSetAccess with
// "invalid argument"; alloc_grow.hip's pattern - chunks appended contiguously from the start of a reservation - works.)
// Build: hipcc -O2 --offload-arch=gfx950 vmm_probe.hip -o vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const char *S(hipError_t e) { return e == hipSuccess ? "ok" : hipGetErrorString(e); }
int main() {
    hipSetDevice(0);
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    size_t gMin = 0, gRec = 0;
    hipMemGetAllocationGranularity(&gMin, &prop, hipMemAllocationGranularityMinimum); hipMemGetAllocationGranularity(&gRec, &prop, hipMemAllocationGranularityRecommended);
    printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu", gMin, gRec);
    auto one = [&](const char *name, size_t reserve, std::vector<std::pair<size_t, size_t>> maps /* offset, size */) {
        void *va = nullptr; hipError_t e = hipMemAddressReserve(&va, reserve, (size_t)2 << 20, nullptr, 0);
        printf(",\n \"%s\": {\"reserve\": \"%s\", \"steps\": [", name, S(e));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        bool first = true;
        for (auto &m : maps) {
            hipMemGenericAllocationHandle_t h; hipError_t c = hipMemCreate(&h, m.second, &prop, 0), mp = hipErrorUnknown, ac = hipErrorUnknown;
            if (c == hipSuccess) mp = hipMemMap((char *)va + m.first, m.second, 0, h, 0);
            if (mp == hipSuccess) ac = hipMemSetAccess((char *)va + m.first, m.second, &ad, 1);
            hipError_t use = hipErrorUnknown;
            if (ac == hipSuccess) { use = hipMemset((char *)va + m.first, 1, m.second); if (use == hipSuccess) use = hipDeviceSynchronize(); }
            printf("%s{\"offset_mb\": %zu, \"size_kb\": %zu, \"create\": \"%s\", \"map\": \"%s\", \"access\": \"%s\", \"memset\": \"%s\"}", first ? "" : ", ", m.first >> 20, m.second >> 10, S(c), S(mp), S(ac), S(use));
            first = false; (void)hipGetLastError();
            if (mp == hipSuccess) hs.push_back(h); else if (c == hipSuccess) hipMemRelease(h);
        }
        printf("]}");
        // (leak the mappings: the process exits)
    };
    const size_t MB = (size_t)1 << 20;
    one("contiguous_16mb", 256 * MB, {{0, 16 * MB}, {16 * MB, 16 * MB}, {32 * MB, 32 * MB}});
    one("gap_then_back", 256 * MB, {{0, 16 * MB}, {128 * MB, 16 * MB}, {16 * MB, 16 * MB}, {144 * MB, 16 * MB}});
    one("first_map_not_at_start", 256 * MB, {{64 * MB, 16 * MB}, {0, 16 * MB}});
    one("small_chunks", 64 * MB, {{0, 64 << 10}, {64 << 10, 64 << 10}, {128 << 10, 4 * MB}, {(128 << 10) + 4 * MB, 256 << 10}});
    one("big_range_sparse", (size_t)176 << 30, {{0, 16 * MB}, {(size_t)16 << 30, 16 * MB}, {(size_t)144 << 30, 4 * MB}, {16 * MB, 16 * MB}});
    printf("}\n");
    return 0;
}
