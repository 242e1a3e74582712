// Microbenchmark: cost of a per-lane 64-B record gather on the vector L1 (TCP).
//  A  "own":        every lane issues four 16-B loads of ITS record (four instructions, 64 distinct lines each)
//  B  "transposed": in instruction k the four lanes of a quad read the four 16-B quarters of the record of lane 4g+k (one line per
//                   lane quad), then a 4x4 transpose through DPP quad permutes hands every lane its own record
// Same records, same bytes; only the lane -> address pattern of each instruction differs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t nextIdx(uint32_t acc, uint32_t mask) { return (acc * 2654435761u + 12345u) & mask; }

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ uint4 dpp4(const uint4 &g) { uint4 r; r.x = dpp<CTRL>(g.x); r.y = dpp<CTRL>(g.y); r.z = dpp<CTRL>(g.z); r.w = dpp<CTRL>(g.w); return r; }
__device__ __forceinline__ uint4 sel(bool c, const uint4 &a, const uint4 &b) { return c ? a : b; }

template <int MODE>
__device__ __forceinline__ void loadRecord(const uint4 *__restrict__ recs, uint32_t idx, int lane, uint4 &q0, uint4 &q1, uint4 &q2, uint4 &q3) {
    if (MODE == 0) { const uint4 *p = recs + (size_t)idx * 4; q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3]; return; }
    const int j = lane & 3;
    const uint32_t i0 = dpp<0x00>(idx), i1 = dpp<0x55>(idx), i2 = dpp<0xaa>(idx), i3 = dpp<0xff>(idx);      // quad broadcast of lane 0..3's index
    const uint4 a0 = recs[(size_t)i0 * 4 + j], a1 = recs[(size_t)i1 * 4 + j], a2 = recs[(size_t)i2 * 4 + j], a3 = recs[(size_t)i3 * 4 + j];
    // lane j holds quarter j of the quad's records 0..3; lane m wants quarters 0..3 of record m: 4x4 transpose in two butterfly stages
    const bool b0 = lane & 1, b1 = lane & 2;
    uint4 r, s0, s1, s2, s3;
    r = dpp4<0xb1>(sel(b0, a0, a1)); s0 = sel(b0, r, a0); s1 = sel(b0, a1, r);       // partner lane^1 (quad_perm [1,0,3,2])
    r = dpp4<0xb1>(sel(b0, a2, a3)); s2 = sel(b0, r, a2); s3 = sel(b0, a3, r);
    r = dpp4<0x4e>(sel(b1, s0, s2)); q0 = sel(b1, r, s0); q2 = sel(b1, s2, r);       // partner lane^2 (quad_perm [2,3,0,1])
    r = dpp4<0x4e>(sel(b1, s1, s3)); q1 = sel(b1, r, s1); q3 = sel(b1, s3, r);
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ recs, uint32_t mask, int steps, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) * 747796405u) & mask;
    uint32_t acc = idx;
    for (int s = 0; s < steps; ++s) {
        uint4 q0, q1, q2, q3;
        loadRecord<MODE>(recs, idx, lane, q0, q1, q2, q3);
        acc += q0.x ^ q1.y ^ q2.z ^ q3.w;
        acc += q0.y + q1.z + q2.w + q3.x;
        idx = nextIdx(acc + s, mask);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
__global__ void check(const uint4 *recs, const uint32_t *idxs, uint32_t *out) {
    uint4 q0, q1, q2, q3;
    loadRecord<MODE>(recs, idxs[threadIdx.x], threadIdx.x & 63, q0, q1, q2, q3);
    out[threadIdx.x] = q0.x * 3u + q0.y * 5u + q0.z * 7u + q0.w * 11u + q1.x * 13u + q1.y * 17u + q1.z * 19u + q1.w * 23u + q2.x * 29u + q2.y * 31u + q2.z * 37u + q2.w * 41u +
                       q3.x * 43u + q3.y * 47u + q3.z * 53u + q3.w * 59u;
}

int main(int argc, char **argv) {
    const uint32_t logN = argc > 1 ? atoi(argv[1]) : 19;        // 2^19 records x 64 B = 32 MB
    const uint32_t n = 1u << logN, mask = n - 1;
    std::vector<uint32_t> h((size_t)n * 16);
    std::mt19937 rng(1);
    for (auto &v : h) v = rng();
    uint4 *d; CK(hipMalloc(&d, (size_t)n * 64)); CK(hipMemcpy(d, h.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    // correctness
    { std::vector<uint32_t> idx(256); for (auto &v : idx) v = rng() & mask;
      uint32_t *di, *o0, *o1; CK(hipMalloc(&di, 1024)); CK(hipMalloc(&o0, 1024)); CK(hipMalloc(&o1, 1024));
      CK(hipMemcpy(di, idx.data(), 1024, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(check<0>, dim3(1), dim3(256), 0, 0, d, di, o0); hipLaunchKernelGGL(check<1>, dim3(1), dim3(256), 0, 0, d, di, o1);
      std::vector<uint32_t> r0(256), r1(256); CK(hipMemcpy(r0.data(), o0, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), o1, 1024, hipMemcpyDeviceToHost));
      int bad = 0; for (int i = 0; i < 256; ++i) bad += r0[i] != r1[i];
      printf("transpose check: %d mismatches\n", bad); }
    uint32_t *out; const int blocks = 256 * 8; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const int steps = 200;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, d, mask, steps, out);
        else hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, d, mask, steps, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double recsPerS = (double)blocks * 256 * steps / (ms * 1e-3);
        printf("%s  2^%u records: %.3f ms  %.2f G records/s  %.1f GB/s\n", mode ? "transposed" : "own       ", logN, ms, recsPerS * 1e-9, recsPerS * 64e-9);
    }
    return 0;
}
