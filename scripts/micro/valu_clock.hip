// Microbenchmark: is the sustained wave64 v_fma_f32 issue rate of a gfx950 SIMD 2 cycles per instruction
// (MI355X_MICROARCH.md, = 157.3 TFLOP/s at 2.4 GHz) or the 2.418 "cycles at 2.4 GHz" profiles/valu_peak.json derived from kernel wall time?
// The two differ by what the clock is. Every wave stamps BOTH counters around its loop:
//     s_memtime      shader cycles (follows the DVFS clock)
//     s_memrealtime  the constant 100 MHz counter
// and the host times the launches with hipEvents after >= 2 s of back-to-back launches (so the chip sits at its sustained clock):
//     cycles per wave-instruction per SIMD  = median(d memtime) / (waves per SIMD x instructions per wave)      [issue rate, clock independent]
//     in-kernel shader clock                = d memtime / d memrealtime x 100 MHz                                 [DVFS give-back, guide item 6]
//     chip rate                             = all wave-instructions / wall time                                    [what bench.py's roof uses]
// It also reports whether all waves of a launch were co-resident (span of all waves / median wave duration ~ 1) - a launch that runs in two
// rounds halves the apparent rate.
//   variant 0: 16 independent v_fma_f32, KZ_BODY (16) times per loop iteration     variant 1: the same with v_max3_f32 (the "4.3-cycle class")
// (profiles/valu_peak.json of round 2 used 32-instruction loop bodies: the backward branch of so short a body is NOT hidden, even at 8
// waves per SIMD - -DKZ_BODY=1 / 2 reproduce its 16- and 32-instruction figures.)
// Build: hipcc -O3 --offload-arch=gfx950 valu_clock.hip -o valu_clock   (binary not tracked)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#ifndef KZ_BODY
#define KZ_BODY 16                 // asm blocks of 16 instructions per loop iteration (256 instructions between two backward branches)
#endif
struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int V>
__global__ __launch_bounds__(256) void loop(int iters, float seed, Stamp *__restrict__ stamps, float *__restrict__ sink) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    const float m = 0.999f, c = 1e-3f;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < KZ_BODY; ++rep) {
#define KZ_OPS16(OP, TAIL) asm volatile( \
                OP " %0, " TAIL(0) "\n " OP " %1, " TAIL(1) "\n " OP " %2, " TAIL(2) "\n " OP " %3, " TAIL(3) "\n " OP " %4, " TAIL(4) "\n " OP " %5, " TAIL(5) "\n " OP " %6, " TAIL(6) "\n " OP " %7, " TAIL(7) "\n " \
                OP " %8, " TAIL(8) "\n " OP " %9, " TAIL(9) "\n " OP " %10, " TAIL(10) "\n " OP " %11, " TAIL(11) "\n " OP " %12, " TAIL(12) "\n " OP " %13, " TAIL(13) "\n " OP " %14, " TAIL(14) "\n " OP " %15, " TAIL(15) "\n" \
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m), "v"(c))
#define T_D_M_C(i) "%" #i ", %16, %17"          /* dst, m, c: three distinct VGPR sources, one of them the destination (a chain per register) */
#define T_M_C_C(i) "%16, %17, %17"              /* m, c, c: two distinct sources, no chain (the form of round 2's valu_peak.hip) */
#define T_D_M_M(i) "%" #i ", %16, %16"          /* dst, m, m: two distinct sources */
#define T_D_D_D(i) "%" #i ", %" #i ", %" #i     /* one distinct source */
#define T_M_C(i) "%16, %17"
#define T_D_M(i) "%" #i ", %16"
        if (V == 0) KZ_OPS16("v_fma_f32", T_D_M_C);
        else if (V == 1) KZ_OPS16("v_max3_f32", T_D_M_C);
        else if (V == 2) KZ_OPS16("v_fma_f32", T_M_C_C);
        else if (V == 3) KZ_OPS16("v_fma_f32", T_D_M_M);
        else if (V == 4) KZ_OPS16("v_fma_f32", T_D_D_D);
        else if (V == 5) KZ_OPS16("v_mul_f32", T_D_M);
        else if (V == 6) KZ_OPS16("v_mul_f32", T_M_C);
        else KZ_OPS16("v_max3_f32", T_M_C_C);
      }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
    if (s == 12345.678f) sink[0] = s;                       // never true: keeps the chains alive
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const double warmSeconds = argc > 2 ? atof(argv[2]) : 2.0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nCU = prop.multiProcessorCount;
    Stamp *dStamps; float *dSink;
    CK(hipMalloc(&dStamps, sizeof(Stamp) * nCU * 8 * 4)); CK(hipMalloc(&dSink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"results\": [\n", prop.name, nCU, iters);
    bool first = true;
    static const char *opName[] = {"v_fma_f32 d,d,m,c (3 distinct sources)", "v_max3_f32 d,d,m,c (3 distinct)", "v_fma_f32 d,m,c,c (2 distinct, no chain)", "v_fma_f32 d,d,m,m (2 distinct)",
                                   "v_fma_f32 d,d,d,d (1 distinct)", "v_mul_f32 d,d,m (2 distinct)", "v_mul_f32 d,m,c (2 distinct, no chain)", "v_max3_f32 d,m,c,c (2 distinct, no chain)"};
    for (int v = 0; v < 8; ++v)
        for (int wps : {1, 2, 8}) {
            const int grid = nCU * wps;                     // 256-thread workgroups: one wave per SIMD each, wps of them per CU
            auto launch = [&]() { switch (v) {
#define L(K) case K: hipLaunchKernelGGL(loop<K>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, dStamps, dSink); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) default: hipLaunchKernelGGL(loop<7>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, dStamps, dSink); } };
            int occ = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, loop<0>, 256, 0));
            // sustained load first: back-to-back launches for warmSeconds
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float one = 0; CK(hipEventElapsedTime(&one, e0, e1));
            const int nWarm = std::max(1, (int)(warmSeconds * 1e3 / std::max(one, 1e-3f)));
            for (int i = 0; i < nWarm; ++i) launch();
            const int nTimed = 8;
            CK(hipEventRecord(e0));
            for (int i = 0; i < nTimed; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= nTimed;
            std::vector<Stamp> h((size_t)grid * 4);
            CK(hipMemcpy(h.data(), dStamps, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> dc, dr, clk;
            unsigned long long rmin = ~0ull, rmax = 0;
            for (const Stamp &s : h) { dc.push_back((double)(s.c1 - s.c0)); dr.push_back((double)(s.r1 - s.r0)); clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 100e6);
                                       rmin = std::min(rmin, s.r0); rmax = std::max(rmax, s.r1); }
            auto med = [](std::vector<double> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
            const double instrPerWave = 16.0 * KZ_BODY * iters, mc = med(dc), mr = med(dr), mclk = med(clk);
            const double total = instrPerWave * (double)grid * 4.0;
            printf("%s{\"op\": \"%s\", \"waves_per_simd\": %d, \"occupancy_blocks_per_cu\": %d, \"kernel_ms\": %.4f, \"median_wave_shader_cycles\": %.0f, \"median_wave_us\": %.2f, "
                   "\"all_waves_span_over_median_wave\": %.3f, \"shader_clock_GHz_in_kernel\": %.4f, \"cycles_per_wave_instr_per_simd\": %.4f, "
                   "\"wave_instr_per_shader_cycle_per_simd\": %.4f, \"chip_G_wave_instr_per_s_wall\": %.1f, \"chip_G_wave_instr_per_s_at_2.4GHz_and_2cyc\": %.1f}",
                   first ? "" : ",\n", opName[v], wps, occ, ms, mc, mr / 100.0, (double)(rmax - rmin) / mr, mclk / 1e9, mc / (wps * instrPerWave),
                   wps * instrPerWave / mc, total / (ms * 1e-3) / 1e9, nCU * 4 * 0.5 * 2.4);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
