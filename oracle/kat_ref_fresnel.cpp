// oracle/kat_ref_fresnel.cpp — mints fp32 known-answer vectors for the Fresnel functions of the dielectric / rough BSDFs from the
// REFERENCE'S OWN text: the Makefile target `ref` extracts lines `float fresnel(float cosThetaI, float extIOR, float intIOR) {` ... up to (not including)
// `Vector3f refract(` of /root/reference/src/kazen/common.cpp (lines 447-523: four functions that use nothing but <cmath> / <algorithm>) into the
// git-ignored oracle/_ref/fresnel_block.inc, and this driver compiles it. It is the ONLY floating-point code of the reference that compiles without Eigen.
// Compiled with -O1 -ffp-contract=off for x86-64 SSE2 (no FMA): IEEE single precision, operation for operation what the reference's own build evaluates.
// This file contains no reference source text. Output: JSON (float BIT PATTERNS as uint32), merged into tests/golden/int_kats.json by make_int_kats.sh.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>

namespace kazen {
#include "_ref/fresnel_block.inc"
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main() {
    using namespace kazen;
    const float cosv[] = {1.0f, 0.999f, 0.9f, 0.75f, 0.5f, 0.3f, 0.1f, 0.02f, 1e-4f, 0.0f, -1e-4f, -0.05f, -0.25f, -0.6f, -0.8f, -0.97f, -1.0f, 0.7071068f, -0.7071068f, 0.6427876f};
    const float ior[][2] = {{1.000277f, 1.5046f}, {1.5046f, 1.000277f}, {1.0f, 1.33f}, {1.0f, 1.0f}, {1.0f, 2.4f}, {1.5f, 1.5f}, {1.2f, 1.1f}};
    const float etas[] = {1.5046f / 1.000277f, 1.000277f / 1.5046f, 1.33f, 1.0f / 1.33f, 2.4f, 1.0f, 1.05f};
    std::printf("{\n \"fresnel_ior\": [\n");
    bool first = true;
    for (float c : cosv) for (auto &p : ior) {
        std::printf("%s  [%u, %u, %u, %u]", first ? "" : ",\n", bits(c), bits(p[0]), bits(p[1]), bits(fresnel(c, p[0], p[1])));
        first = false;
    }
    std::printf("\n ],\n \"fresnel_eta\": [\n");
    first = true;
    for (float c : cosv) for (float e : etas) {
        std::printf("%s  [%u, %u, %u]", first ? "" : ",\n", bits(c), bits(e), bits(fresnel(c, e)));
        first = false;
    }
    std::printf("\n ],\n \"fresnel_dielectric\": [\n");
    first = true;
    for (float c : cosv) for (float e : etas) {
        float ct = 123.0f;
        const float F = fresnelDielectric(c, e, ct);
        std::printf("%s  [%u, %u, %u, %u]", first ? "" : ",\n", bits(c), bits(e), bits(F), bits(ct));
        first = false;
    }
    std::printf("\n ]\n}\n");
    return 0;
}
