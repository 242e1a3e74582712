// Microbenchmark: VALU issue ceiling of one gfx950 SIMD, measured the same way profiles/ measures the traversal kernels.
//
// Question ("Roofline"): profiles/bound_latest.json prices the traversal at "VALU busy = SQ_ACTIVE_INST_VALU x 4 /
// SIMD-cycles ~ 100 %", i.e. one wave64 VALU instruction per 4 cycles per SIMD, while MI355X_MICROARCH.md quotes v_fma_f32 at
// 2 cycles per wave64 instruction on the SIMD-32 (4 only for one wave alone). Which ceiling applies to this instruction mix?
//
// Each variant runs a loop of independent VALU instructions (inline asm, so the stream is exactly what is written) on every SIMD of
// the chip at 1, 2, 4 and 8 waves per SIMD, stamps s_memtime around the loop, and reports
//     wave-instructions per cycle per SIMD = waves/SIMD x instructions per wave / median cycles of a wave
// The same binary is run under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE ...`
// (scripts/micro/valu_peak.sh) so that the x4 formula can be read against a stream whose true issue rate is known.
//
//   variant 0  fma      16 independent v_fma_f32
//   variant 1  pkfma    16 independent v_pk_fma_f32 (2 FMAs per lane per instruction)
//   variant 2  node     the BVH4 node step's mix, compiled from the expression forms of kz_devfn.h node4Keys: v_cvt_f32_ubyteN,
//                       v_pk_fma_f32, v_max3_f32 / v_min3_f32, v_mul_f32, v_cmp + v_cndmask, v_and_or (instructions per iteration are
//                       counted from the disassembly by valu_peak.sh and passed as argv[3])
//   variant 3  int64    the sampler's mix (pcg32 step + Murmur mixing): v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 / v_lshrrev_b64 / v_xor
//                       (instructions per iteration counted the same way, argv[4])
//   variant 4  trans    v_rcp_f32 / v_sqrt_f32 / v_exp_f32 (quarter rate)
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak   (binary not tracked)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int V>
__global__ __launch_bounds__(256) void valu_loop(int iters, float seed, unsigned long long *__restrict__ cycles, float *__restrict__ sink) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    const float m = 0.999f, c = 1e-3f;
    f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
    const f2 pm = {m, m}, pc = {c, c};
    unsigned q0 = __float_as_uint(a0) * 2654435761u, q1 = q0 ^ 0x9e3779b9u, q2 = q0 + 0x7f4a7c15u;
    unsigned long long s0 = ((unsigned long long)q0 << 32) | q1, s1 = s0 * 0x5851f42d4c957f2dULL + 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
            asm volatile(
                "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "v"(m), "v"(c));
        } else if (V == 1) {
            asm volatile(
                REP4("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n")
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(pm), "v"(pc));
            asm volatile("" : "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
        } else if (V == 2) {
            // one BVH4 node step's arithmetic, the expression forms of kz_devfn.h node4Keys, on register-resident packed planes
            // (q0,q1,q2 stand for the packet words; the result feeds back into q0 so no iteration can be hoisted or dropped)
            const f2 AX = {m, m}, AY = {c + m, c + m}, AZ = {m - c, m - c}, BX = {a1, a1}, BY = {a2, a2}, BZ = {a3, a3};
            unsigned key[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2 qx = {(float)((q0 >> (8 * i)) & 0xffu), (float)((q1 >> (8 * i)) & 0xffu)};
                const f2 qy = {(float)((q2 >> (8 * i)) & 0xffu), (float)((q0 >> (8 * (3 - i))) & 0xffu)};
                const f2 qz = {(float)((q1 >> (8 * (3 - i))) & 0xffu), (float)((q2 >> (8 * (3 - i))) & 0xffu)};
                const f2 tx = __builtin_elementwise_fma(qx, AX, BX), ty = __builtin_elementwise_fma(qy, AY, BY), tz = __builtin_elementwise_fma(qz, AZ, BZ);
                const float n = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), c);
                const float f = fminf(fminf(fminf(tx.y, ty.y), tz.y) * 1.0000004f, a0);
                key[i] = (n <= f) ? ((__float_as_uint(n) & ~3u) | (unsigned)i) : 0xFFFFFFFFu;
            }
            const unsigned kmin = min(min(key[0], key[1]), min(key[2], key[3]));
            q0 ^= kmin; q1 += key[1] & 0x01010101u; q2 ^= key[2] >> 7;
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
        } else if (V == 3) {
            // pcg32 step + murmur-style 64-bit mixing: what Sampler::nextUInt / hashPixelDimSeed compile to
            unsigned long long x = s0, y = s1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x = x * 0x5851f42d4c957f2dULL + 0x14057b7ef767814fULL; x ^= x >> 47;
                y = y * 0xc6a4a7935bd1e995ULL; y ^= y >> 47;
                asm volatile("" : "+v"(x), "+v"(y));
            }
            s0 = x; s1 = y;
        } else {
            asm volatile(
                REP4("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_exp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
              __uint_as_float(q0) + (float)(s0 ^ s1);
    if (r == 1234.5678f) sink[0] = r;
}

// ---- single-opcode variants (32 independent instructions per iteration): what each VALU opcode of the traversal costs ----
#define OP16(fmt) \
    fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(8) fmt(9) fmt(10) fmt(11) fmt(12) fmt(13) fmt(14) fmt(15)
template <int V>
__global__ __launch_bounds__(256) void op_loop(int iters, float seed, unsigned long long *__restrict__ cycles, float *__restrict__ sink) {
    float a[16]; unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x + i; u[i] = __float_as_uint(a[i]) * 2654435761u; }
    const float m = 0.999f, c = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(TXT) asm volatile(TXT TXT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
                              "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(m), "v"(c), "v"(u[0]), "v"(u[1]) : "vcc", "s20", "s21")
#define R16(A, B) A "%0" B "\n" A "%1" B "\n" A "%2" B "\n" A "%3" B "\n" A "%4" B "\n" A "%5" B "\n" A "%6" B "\n" A "%7" B "\n" \
                  A "%8" B "\n" A "%9" B "\n" A "%10" B "\n" A "%11" B "\n" A "%12" B "\n" A "%13" B "\n" A "%14" B "\n" A "%15" B "\n"
        if (V == 0) OPS(R16("v_fma_f32 ", ", %16, %17, %17"));             // dst = m * c + c style: independent of dst (no chain at all)
        else if (V == 1) OPS(R16("v_add_f32 ", ", %16, %17"));
        else if (V == 2) OPS(R16("v_cvt_f32_ubyte1 ", ", %18"));
        else if (V == 3) OPS(R16("v_max3_f32 ", ", %16, %17, %16"));
        else if (V == 4) OPS(R16("v_max_f32 ", ", %16, %17"));
        else if (V == 5) OPS(R16("v_cndmask_b32 ", ", %16, %17, vcc"));
        else if (V == 6) OPS(R16("v_and_or_b32 ", ", %18, %19, %18"));
        else if (V == 7) OPS(R16("v_mul_lo_u32 ", ", %18, %19"));
        else if (V == 8) OPS(R16("v_cmp_le_f32 vcc, %16, ", ""));
        else if (V == 9) OPS(R16("v_mov_b32 ", ", %16"));
        else if (V == 10) OPS(R16("v_rcp_f32 ", ", %16"));
        else if (V == 11) OPS(R16("v_perm_b32 ", ", %18, %19, %18"));
        else if (V == 12) OPS(R16("v_mul_f32 ", ", %16, %17"));
        else if (V == 13) OPS(R16("v_min3_f32 ", ", %16, %17, %16"));
        else if (V == 14) OPS(R16("v_cndmask_b32_e64 ", ", %16, %17, s[20:21]"));           // select by an SGPR-pair mask
        else if (V == 15) OPS(R16("v_cmp_le_f32 vcc, %16, %17\n v_cndmask_b32 ", ", %16, %17, vcc"));   // compare + select pairs (32 + 32)
        else if (V == 16) OPS(R16("v_fma_f32 ", ", %16, s20, %17"));                         // VOP3 with one SGPR operand
        else if (V == 17) OPS(R16("v_cvt_f32_ubyte0 ", ", s20"));                            // conversion of a wave-uniform byte
        else if (V == 18) OPS(R16("v_lshrrev_b32 ", ", 8, %18"));
        else OPS(R16("v_bfe_u32 ", ", %18, 8, 8"));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i];
    if (r == 1234.5678f) sink[0] = r;
}

// ---- does a wave64 instruction whose upper (or lower) 32 lanes are all masked off issue in one pass of the SIMD-32 instead of two? ----
template <int MASKMODE>
__global__ __launch_bounds__(256) void exec_loop(int iters, float seed, unsigned long long *__restrict__ cycles, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x + i;
    const float m = 0.999f, c = 1e-3f;
    const bool on = MASKMODE == 0 ? true : MASKMODE == 1 ? lane < 32 : MASKMODE == 2 ? (lane & 1) == 0 : MASKMODE == 3 ? lane < 16 : lane >= 32;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int it = 0; it < iters; ++it) {
            asm volatile(
                "v_max3_f32 %0, %0, %16, %17\n v_max3_f32 %1, %1, %16, %17\n v_max3_f32 %2, %2, %16, %17\n v_max3_f32 %3, %3, %16, %17\n"
                "v_max3_f32 %4, %4, %16, %17\n v_max3_f32 %5, %5, %16, %17\n v_max3_f32 %6, %6, %16, %17\n v_max3_f32 %7, %7, %16, %17\n"
                "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                  "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 || lane == 32) cycles[(blockIdx.x * 256 + threadIdx.x) >> 5] = t1 - t0;
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i];
    if (r == 1234.5678f) sink[0] = r;
}

int main(int argc, char **argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;
    int iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"results\": [\n", prop.name, cus, iters);
    unsigned long long *dC; float *dS;
    CK(hipMalloc(&dC, sizeof(unsigned long long) * cus * 8 * 8)); CK(hipMalloc(&dS, 64));
    bool first = true;
    for (int v = 0; v < 5; ++v) {
        if (only >= 0 && only != v) continue;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;            // 256-thread block = 4 waves = one per SIMD; wps blocks per CU
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (v) {
                    case 0: hipLaunchKernelGGL(valu_loop<0>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                    case 1: hipLaunchKernelGGL(valu_loop<1>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                    case 2: hipLaunchKernelGGL(valu_loop<2>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                    case 3: hipLaunchKernelGGL(valu_loop<3>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                    default: hipLaunchKernelGGL(valu_loop<4>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                }
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)blocks * 4);
            CK(hipMemcpy(h.data(), dC, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const int ipi = v == 0 ? 16 : v == 1 ? 16 : (v == 2 || v == 3) ? (argc > 3 + (v == 3) ? atoi(argv[3 + (v == 3)]) : 0) : 16;
            const double perSimd = ipi ? (double)wps * ipi * iters / med : 0.0;
            printf("%s {\"variant\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"median_wave_cycles\": %.0f, \"asm_instr_per_iter\": %d, "
                   "\"wave_instr_per_cycle_per_simd\": %.4f, \"cycles_per_iter\": %.2f}", first ? " " : ",", v, wps, ms, med, ipi, perSimd, med / iters);
            printf("\n");
            first = false;
        }
    }

    static const char *opName[] = {"v_fma_f32", "v_add_f32", "v_cvt_f32_ubyte1", "v_max3_f32", "v_max_f32", "v_cndmask_b32", "v_and_or_b32", "v_mul_lo_u32",
                                   "v_cmp_le_f32", "v_mov_b32", "v_rcp_f32", "v_perm_b32", "v_mul_f32", "v_min3_f32",
                                   "v_cndmask_b32 (sgpr pair)", "v_cmp+v_cndmask pairs", "v_fma_f32 (sgpr operand)", "v_cvt_f32_ubyte0 (sgpr)", "v_lshrrev_b32", "v_bfe_u32"};
    for (int v = 0; v < 20 && only < 0; ++v) {
        const int wps = 8, blocks = cus * wps;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            switch (v) {
#define LAUNCH(K) case K: hipLaunchKernelGGL(op_loop<K>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15) LAUNCH(16) LAUNCH(17) LAUNCH(18) LAUNCH(19)
            }
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        // whole-kernel rate: 8 waves/SIMD x 32 instructions x iters per SIMD over the kernel's duration at the 2.4 GHz maximum clock
        const double perSimdPerCycle = 8.0 * (v == 15 ? 64 : 32) * iters / (ms * 1e-3 * 2.4e9);
        printf(", {\"op\": \"%s\", \"waves_per_simd\": 8, \"ms\": %.3f, \"asm_instr_per_iter\": 32, \"wave_instr_per_cycle_per_simd_at_2p4GHz\": %.4f, "
               "\"cycles_per_instr_at_2p4GHz\": %.3f}\n", opName[v], ms, perSimdPerCycle, 1.0 / perSimdPerCycle);
    }

    static const char *maskName[] = {"all 64 lanes", "lanes 0-31", "even lanes (32 active, both halves)", "lanes 0-15", "lanes 32-63"};
    for (int v = 0; v < 5 && only < 0; ++v) {
        const int wps = 8, blocks = cus * wps;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            switch (v) {
#define LAUNCHX(K) case K: hipLaunchKernelGGL(exec_loop<K>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, dC, dS); break;
                LAUNCHX(0) LAUNCHX(1) LAUNCHX(2) LAUNCHX(3) LAUNCHX(4)
            }
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf(", {\"exec_mask\": \"%s\", \"waves_per_simd\": 8, \"ms\": %.3f, \"asm_instr_per_iter\": 16}\n", maskName[v], ms);
    }
    printf("]}\n");
    return 0;
}
