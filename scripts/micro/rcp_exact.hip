// rcp_exact.hip — EXHAUSTIVE check (all 2^32 bit patterns) of candidate sequences for 1.0f / x against the compiler's IEEE division under
// -fgpu-flush-denormals-to-zero (v_div_scale, v_rcp, Newton, v_div_fmas, v_div_fixup, 42-47 SIMD cycles, scripts/micro/op_cost.hip):
//   A: v_rcp_f32 + two Newton-Raphson steps in FMA form (4 FMAs)
//   B: v_rcp_f32 + one Newton step + two residual corrections of the quotient (the compiler's arithmetic for numerator 1 without scaling)
// Prints, per candidate, the number of inputs with |x| in [2^-100, 2^100] whose result differs in any bit (NaN payloads aside), and the
// first few of them. A candidate with 0 mismatches in the range is bit-identical to the division there BY EXHAUSTION.
//   build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fgpu-flush-denormals-to-zero rcp_exact.hip -o rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float candA(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
    return r;
}
__device__ __forceinline__ float candB(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e0, r, r);
    float q = r;                                                  // 1 * r
    const float e1 = __builtin_fmaf(-x, q, 1.0f); q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-x, q, 1.0f); return __builtin_fmaf(e2, r, q);
}
// C: the square root: v_rsq_f32 + the compiler's own refinement (g = x*y, h = y/2, one coupled Newton step, one residual correction) without
// its input scaling, un-scaling and class test - for 2^-95 <= x < 2^96
__device__ __forceinline__ float candC(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__global__ void checkSqrt(unsigned long long base, unsigned long long *bad, unsigned *firstBad) {
    const unsigned bits = (unsigned)(base + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (!((bits - 0x10000000u) < (0x6F800000u - 0x10000000u))) return;
    const float x = __uint_as_float(bits);
    if (__float_as_uint(candC(x)) != __float_as_uint(sqrtf(x))) { const unsigned long long k = atomicAdd(&bad[0], 1ull); if (k < 8) firstBad[k] = bits; }
    atomicAdd(&bad[1], 1ull);
}
__global__ void check(unsigned long long base, unsigned long long *bad, unsigned *firstBad) {
    const unsigned bits = (unsigned)(base + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float ax = fabsf(x);
    if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) return;
    const float ref = 1.0f / x;
    const float a = candA(x), b = candB(x);
    if (__float_as_uint(a) != __float_as_uint(ref)) { const unsigned long long k = atomicAdd(&bad[0], 1ull); if (k < 8) firstBad[k] = bits; }
    if (__float_as_uint(b) != __float_as_uint(ref)) { const unsigned long long k = atomicAdd(&bad[1], 1ull); if (k < 8) firstBad[8 + k] = bits; }
    atomicAdd(&bad[2], 1ull);
}
int main() {
    unsigned long long *dBad; unsigned *dFirst;
    CK(hipMalloc(&dBad, 24)); CK(hipMalloc(&dFirst, 64)); CK(hipMemset(dBad, 0, 24)); CK(hipMemset(dFirst, 0, 64));
    for (unsigned long long base = 0; base < (1ull << 32); base += (1ull << 28)) {
        hipLaunchKernelGGL(check, dim3(1u << 20), dim3(256), 0, 0, base, dBad, dFirst);
        CK(hipDeviceSynchronize());
    }
    unsigned long long bad[3]; unsigned first[16];
    CK(hipMemcpy(bad, dBad, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(first, dFirst, 64, hipMemcpyDeviceToHost));
    printf("{\"inputs_in_range\": %llu, \"mismatches_A_rcp_2newton\": %llu, \"mismatches_B_div_core\": %llu, \"first_bad_A\": [", bad[2], bad[0], bad[1]);
    for (int i = 0; i < 8 && i < (int)bad[0]; ++i) printf("%s\"0x%08x\"", i ? ", " : "", first[i]);
    printf("], \"first_bad_B\": [");
    for (int i = 0; i < 8 && i < (int)bad[1]; ++i) printf("%s\"0x%08x\"", i ? ", " : "", first[8 + i]);
    printf("]");
    CK(hipMemset(dBad, 0, 24)); CK(hipMemset(dFirst, 0, 64));
    for (unsigned long long base = 0; base < (1ull << 32); base += (1ull << 28)) {
        hipLaunchKernelGGL(checkSqrt, dim3(1u << 20), dim3(256), 0, 0, base, dBad, dFirst);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(bad, dBad, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(first, dFirst, 64, hipMemcpyDeviceToHost));
    printf(", \"sqrt_inputs_in_range\": %llu, \"mismatches_C_rsq_refine\": %llu, \"first_bad_C\": [", bad[1], bad[0]);
    for (int i = 0; i < 8 && i < (int)bad[0]; ++i) printf("%s\"0x%08x\"", i ? ", " : "", first[i]);
    printf("]}\n");
    return 0;
}
