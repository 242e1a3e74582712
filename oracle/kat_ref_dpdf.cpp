// oracle/kat_ref_dpdf.cpp - mints known-answer vectors for DiscretePDF (the area CDF of a light mesh) and for the integer helpers that size
// PMJ02BN's pixel tile (a9) from the REFERENCE'S OWN text. The Makefile target `ref` extracts
//   * `struct DiscretePDF { ... };` of /root/reference/include/kazen/dpdf.h (lines 14-168: it uses nothing but <vector> / <algorithm> / <string>) into
//     oracle/_ref/dpdf_block.inc, and
//   * `isPowerOf4` and the integer `log2i` overloads, `log4i`, `roundUpPow4` of include/kazen/common.h (lines 271-288 and 306-319; the float overload of
//     log2i between them needs the floatingPoint helpers and is not on the path) into oracle/_ref/math_block.inc
// - git-ignored, never committed - and this driver compiles them. It contains no reference source text. Output: JSON on stdout (tests/golden/make_int_kats.sh
// merges it into tests/golden/int_kats.json).
//   dpdf: for each list of pdf values (areas exactly representable as small integers x powers of two from 2^-20 to 2^10, so that a test can BUILD triangles
//   with these areas; zero entries included): after append() x n and normalize() the sum, the normalisation and the whole CDF as float bit patterns; then
//   sample(v) for v = 0, every CDF entry and its two float neighbours, 1 - ulp, 1, and a spread of others - with FTZ | DAZ set as the reference's main() sets them.
//   pow4: isPowerOf4 over 1 .. 65536 (the true ones), and (roundUpPow4, log4i of it, the pixel tile 1 << (log4i(65536) - log4i(roundUpPow4(spp)))) for every spp
//   in 1 .. 65536 at which the triple changes (sampler.cpp:291).
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <xmmintrin.h>

#define NAMESPACE_BEGIN(name) namespace name {
#define NAMESPACE_END(name) }
#define private public                       /* (the CDF itself is a private member: the vectors pin its bits) */
NAMESPACE_BEGIN(kazen)
#include "_ref/dpdf_block.inc"
NAMESPACE_BEGIN(math)
#include "_ref/math_block.inc"
NAMESPACE_END(math)
NAMESPACE_END(kazen)
#undef private

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float fromBits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    using namespace kazen;
    // the reference's arithmetic environment: main.cpp:22-23 turns flush-to-zero and denormals-are-zero on before anything runs, so a
    // denormal draw compares as 0 against the table - as it does on the device (-fgpu-flush-denormals-to-zero) and in the oracle (FtzScope)
    _mm_setcsr(_mm_getcsr() | 0x8040u);
    // areas = m * 2^e: m < 2^11, so (2 * area)^2 is exact in float and a right triangle with legs (area, 2) has exactly this area in the library's arithmetic
    auto A = [](int m, int e) { return std::ldexp((float)m, e); };
    const std::vector<std::vector<float>> sets = {
        {A(1, 0)},
        {A(3, -2), A(1, -20)},
        {A(1, -20), A(3, -18), 0.0f, A(5, -10), A(1, 0)},
        {A(1, 0), A(1, 0), A(1, 0), A(1, 0), A(1, 0), A(1, 0), A(1, 0), A(1, 0)},
        {0.0f, A(7, 0), 0.0f, 0.0f, A(125, 3), A(1, -20), A(9, -7), A(3, 1), 0.0f},
        {A(1, 10), A(1, -20), A(1, 10), A(1, -20), A(1, 10), A(1, -20), A(1, 10), A(1, -20), A(1, 10), A(1, -20), A(3, 8), A(5, -15)},
        {0.0f, 0.0f, 0.0f},
    };
    std::vector<std::vector<float>> all = sets;
    {   // two long tables (the device's binary search): 33 and 200 entries, areas from a small LCG over m * 2^e, some zeros
        uint32_t s = 12345u;
        for (int n : {33, 200}) {
            std::vector<float> v;
            for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const int m = (int)((s >> 8) % 2047u), e = (int)((s >> 20) % 31u) - 20; v.push_back((s >> 29) == 0 ? 0.0f : A(m + 1, e)); }
            all.push_back(v);
        }
    }
    std::printf("{\n \"dpdf\": [\n");
    for (size_t k = 0; k < all.size(); ++k) {
        DiscretePDF d(all[k].size());
        for (float a : all[k]) d.append(a);
        const float sum = d.normalize();
        std::printf("%s  {\"values\": [", k ? ",\n" : "");
        for (size_t i = 0; i < all[k].size(); ++i) std::printf("%s%u", i ? ", " : "", bits(all[k][i]));
        std::printf("], \"sum\": %u, \"normalization\": %u, \"normalized\": %d, \"cdf\": [", bits(sum), bits(d.getNormalization()), d.isNormalized() ? 1 : 0);
        for (size_t i = 0; i < d.m_cdf.size(); ++i) std::printf("%s%u", i ? ", " : "", bits(d.m_cdf[i]));
        std::printf("], \"sample\": [");
        std::vector<uint32_t> vs = {bits(0.0f), bits(1.0f), 0x3f7fffffu, bits(0.5f), bits(0.25f), bits(1e-7f), bits(0.999f), bits(0.3333333f)};
        for (float c : d.m_cdf) { const uint32_t u = bits(c); vs.push_back(u); if (u > 0) vs.push_back(u - 1); if (c < 1.0f) vs.push_back(u + 1); }
        bool first = true;
        for (uint32_t u : vs) {
            const float v = fromBits(u);
            if (!(v >= 0.0f && v <= 1.0f)) continue;
            std::printf("%s[%u, %zu]", first ? "" : ", ", u, d.sample(v));
            first = false;
        }
        std::printf("]}");
    }
    std::printf("\n ],\n \"pow4\": {\"is_power_of_4\": [");
    bool first = true;
    for (int n = 1; n <= 65536; ++n) if (math::isPowerOf4(n)) { std::printf("%s%d", first ? "" : ", ", n); first = false; }
    std::printf("], \"changes\": [");
    int last[3] = {-1, -1, -1};
    first = true;
    for (int spp = 1; spp <= 65536; ++spp) {
        const int r = math::roundUpPow4(spp), l = math::log4i(r), tile = 1 << (math::log4i(65536) - math::log4i(math::roundUpPow4(spp)));
        if (r != last[0] || l != last[1] || tile != last[2]) { std::printf("%s[%d, %d, %d, %d]", first ? "" : ", ", spp, r, l, tile); first = false; last[0] = r; last[1] = l; last[2] = tile; }
    }
    std::printf("]}\n}\n");
    return 0;
}
