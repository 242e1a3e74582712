// op_cost.hip — issue cost of the opcodes the SHADE kernel is made of (gfx950), 4 waves/SIMD like that kernel and 8 for comparison:
// the pieces of the IEEE fp32 division the compiler emits under -fgpu-flush-denormals-to-zero (v_div_scale / v_rcp / fma / v_div_fmas /
// v_div_fixup, bracketed by two s_setreg of the denormal mode), the correctly rounded sqrt, f64 multiply / convert, the 64-bit
// integer multiply-add of the Murmur hash, and the transcendental unit. Single opcodes are inline asm (32 independent instructions per
// iteration); "ieee_div" / "ieee_sqrt" are C expressions (8 independent ones per iteration), i.e. exactly the compiler's sequences.
//   build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fgpu-flush-denormals-to-zero op_cost.hip -o op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R16(A, B) A "%0" B "\n" A "%1" B "\n" A "%2" B "\n" A "%3" B "\n" A "%4" B "\n" A "%5" B "\n" A "%6" B "\n" A "%7" B "\n" \
                  A "%8" B "\n" A "%9" B "\n" A "%10" B "\n" A "%11" B "\n" A "%12" B "\n" A "%13" B "\n" A "%14" B "\n" A "%15" B "\n"
#define OPS(X, TXT) asm volatile(TXT TXT : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), "+v"(X[8]), "+v"(X[9]), \
                              "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]) : "v"(m), "v"(c), "v"(u0), "v"(u1), "v"(dm) : "vcc", "s20", "s21")


// three numerators over one denominator: the compiler's division arithmetic (rcp, one Newton step on the reciprocal, quotient, two residual
// corrections) with the reciprocal shared, the residuals computed with fp32 denormals enabled (ONE s_setreg pair), no v_div_scale / v_div_fixup:
// bit-identical to a/s for in-range operands, which the caller's guard selects
__device__ __forceinline__ void div3asm(float ax, float ay, float az, float s, float &qx, float &qy, float &qz) {
    float r, e1, e2, e3;
    asm volatile(
        "v_rcp_f32 %3, %10\n"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 3\n"
        "v_fma_f32 %4, -%10, %3, 1.0\n"
        "v_fma_f32 %3, %4, %3, %3\n"
        "v_mul_f32 %0, %7, %3\n v_mul_f32 %1, %8, %3\n v_mul_f32 %2, %9, %3\n"
        "v_fma_f32 %4, -%10, %0, %7\n v_fma_f32 %5, -%10, %1, %8\n v_fma_f32 %6, -%10, %2, %9\n"
        "v_fma_f32 %0, %4, %3, %0\n v_fma_f32 %1, %5, %3, %1\n v_fma_f32 %2, %6, %3, %2\n"
        "v_fma_f32 %4, -%10, %0, %7\n v_fma_f32 %5, -%10, %1, %8\n v_fma_f32 %6, -%10, %2, %9\n"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0\n"
        "v_fma_f32 %0, %4, %3, %0\n v_fma_f32 %1, %5, %3, %1\n v_fma_f32 %2, %6, %3, %2\n"
        : "=&v"(qx), "=&v"(qy), "=&v"(qz), "=&v"(r), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(ax), "v"(ay), "v"(az), "v"(s));
}
__device__ __forceinline__ void div3guarded(float ax, float ay, float az, float s, float &qx, float &qy, float &qz) {
    const float as = fabsf(s), am = fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fabsf(az));
    if (as >= 0x1p-60f && as <= 0x1p60f && am <= 0x1p60f) div3asm(ax, ay, az, s, qx, qy, qz);
    else { qx = ax / s; qy = ay / s; qz = az / s; }
}
template <int V>
__global__ __launch_bounds__(256) void op_loop(int iters, float seed, float *__restrict__ sink) {
    float a[16]; double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x + i; d[i] = a[i]; }
    const float m = 0.999f * seed, c = 1e-3f * seed; const unsigned u0 = __float_as_uint(seed) * 2654435761u, u1 = u0 ^ 0x9e3779b9u; const double dm = 0.999 * seed;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) OPS(a, R16("v_fma_f32 ", ", %16, %17, %17"));
        else if (V == 1) OPS(a, R16("v_rcp_f32 ", ", %16"));
        else if (V == 2) OPS(a, R16("v_rsq_f32 ", ", %16"));
        else if (V == 3) OPS(a, R16("v_sqrt_f32 ", ", %16"));
        else if (V == 4) OPS(a, R16("v_div_scale_f32 ", ", vcc, %16, %17, %16"));
        else if (V == 5) OPS(a, R16("v_div_fmas_f32 ", ", %16, %17, %16"));
        else if (V == 6) OPS(a, R16("v_div_fixup_f32 ", ", %16, %17, %16"));
        else if (V == 7) OPS(d, R16("v_mul_f64 ", ", %20, %20"));
        else if (V == 8) OPS(d, R16("v_fma_f64 ", ", %20, %20, %20"));
        else if (V == 9) OPS(d, R16("v_cvt_f64_f32 ", ", %16"));
        else if (V == 10) OPS(a, R16("v_cvt_f32_f64 ", ", %20"));
        else if (V == 11) OPS(d, R16("v_mad_u64_u32 ", ", vcc, %18, %19, %20"));
        else if (V == 12) OPS(a, R16("v_mul_hi_u32 ", ", %18, %19"));
        else if (V == 13) OPS(a, R16("v_exp_f32 ", ", %16"));
        else if (V == 14) OPS(a, R16("v_log_f32 ", ", %16"));
        else if (V == 15) OPS(a, R16("v_sin_f32 ", ", %16"));
        else if (V == 16) OPS(a, R16("v_ldexp_f32 ", ", %16, %18"));
        else if (V == 17) OPS(a, R16("v_cvt_f32_u32 ", ", %18"));
        else if (V == 18) OPS(a, R16("v_mul_lo_u32 ", ", %18, %19"));
        else if (V == 19) OPS(a, R16("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 3\n v_fma_f32 ", ", %16, %17, %17\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0"));   // 32 fma, each bracketed
        else if (V == 20) {            // 8 independent IEEE divisions per iteration, as compiled
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i + 8] / a[i];
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (V == 21) {          // 8 independent correctly rounded square roots per iteration, as compiled
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = sqrtf(a[i] + a[i + 8]);
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (V == 23) {          // 3 x (three numerators over one denominator) per iteration through the shared-reciprocal sequence + guard: 9 divisions
#pragma unroll
            for (int i = 0; i < 3; ++i) div3guarded(a[3 * i + 4], a[3 * i + 5], a[3 * i + 6], a[i], a[3 * i + 4], a[3 * i + 5], a[3 * i + 6]);
            asm volatile("" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]));
        } else if (V == 24) {          // the same 9 divisions as the compiler emits them
#pragma unroll
            for (int i = 0; i < 3; ++i) { a[3 * i + 4] /= a[i]; a[3 * i + 5] /= a[i]; a[3 * i + 6] /= a[i]; }
            asm volatile("" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]));
        } else {                       // 22: the unscaled core of the division (rcp + 2 fma, then mul + 4 fma): same arithmetic for in-range operands
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float n = a[i + 8], b = a[i];
                float r = __builtin_amdgcn_rcpf(b);
                const float e0 = __builtin_fmaf(-b, r, 1.0f); r = __builtin_fmaf(e0, r, r);
                float q = n * r;
                const float e1 = __builtin_fmaf(-b, q, n); q = __builtin_fmaf(e1, r, q);
                const float e2 = __builtin_fmaf(-b, q, n); a[i] = __builtin_fmaf(e2, r, q);
            }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        }
    }
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i] + (float)d[i];
    if (r == 1234.5678f) sink[0] = r;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *dS; CK(hipMalloc(&dS, 64));
    static const char *name[] = {"v_fma_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mul_f64", "v_fma_f64", "v_cvt_f64_f32",
                                 "v_cvt_f32_f64", "v_mad_u64_u32", "v_mul_hi_u32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_ldexp_f32", "v_cvt_f32_u32", "v_mul_lo_u32",
                                 "v_fma_f32 between two s_setreg MODE", "ieee_div (a/b as compiled, FTZ)", "ieee_sqrt (sqrtf as compiled, FTZ)", "div core without scale/fixup/setreg", "div3 shared reciprocal + guard (per division)", "3 x a/s as compiled (per division)"};
    static const int per[] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 8, 8, 8, 9, 9};
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"clock_assumed_GHz\": 2.4, \"results\": [\n", prop.name, cus, iters);
    for (int v = 0; v < 25; ++v) {
        for (int wps : {4, 8}) {
            const int blocks = cus * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (v) {
#define L(K) case K: hipLaunchKernelGGL(op_loop<K>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dS); break;
                    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24)
                }
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * per[v] * iters);       // SIMD cycles per wave64 unit (instruction, or whole division / sqrt)
            printf("%s{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"units_per_iter\": %d, \"simd_cycles_per_unit_at_2p4GHz\": %.2f}\n", (v || wps != 4) ? "," : " ", name[v], wps, ms, per[v], cyc);
        }
    }
    printf("]}\n");
    return 0;
}
