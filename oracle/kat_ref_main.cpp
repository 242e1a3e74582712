// oracle/kat_ref_main.cpp — mints integer known-answer vectors from the REFERENCE'S OWN
// headers (include/kazen/define.h, hash.h, pcg32.h), compiled where they lie under
// /root/reference (see oracle/Makefile target `ref`; output goes to oracle/_ref/ only).
// These three headers are the only part of the reference that compiles without Eigen /
// Embree / TBB / OpenImageIO. This file contains no reference source text.
// Output: JSON on stdout, committed as tests/golden/int_kats.json by tests/golden/make_int_kats.sh.
#include <kazen/define.h>
#include <kazen/hash.h>
#include <kazen/pcg32.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct P2i { int32_t x, y; };   // same bytes as Eigen Point2i (two int32)

int main() {
    using namespace kazen;
    std::printf("{\n");
    // Hash(Point2i, uint64 seed): sampler.cpp:44
    std::printf(" \"hash_pixel_seed\": [\n");
    const int32_t px[] = {0, 3, 1, 255, 1919, 7, 100, 65535};
    const int32_t py[] = {0, 5, 0, 255, 1079, 2047, 3, 4095};
    const uint64_t seeds[] = {0ull, 1ull, 42ull, 0xdeadbeefcafef00dull};
    bool first = true;
    for (int i = 0; i < 8; ++i) for (int s = 0; s < 4; ++s) {
        P2i p{px[i], py[i]};
        uint64_t h = Hash(p, seeds[s]);
        std::printf("%s  [%d, %d, \"%llu\", \"%llu\"]", first ? "" : ",\n", p.x, p.y, (unsigned long long)seeds[s], (unsigned long long)h);
        first = false;
    }
    std::printf("\n ],\n \"hash_pixel_dim_seed\": [\n");
    // Hash(Point2i, uint32 dim, uint64 seed): sampler.cpp:341,355
    first = true;
    const uint32_t dims[] = {2u, 3u, 4u, 11u, 47u, 100u};
    for (int i = 0; i < 8; ++i) for (int d = 0; d < 6; ++d) {
        P2i p{px[i], py[i]};
        uint64_t seed = seeds[(i + d) & 3];
        uint64_t h = Hash(p, dims[d], seed);
        std::printf("%s  [%d, %d, %u, \"%llu\", \"%llu\"]", first ? "" : ",\n", p.x, p.y, dims[d], (unsigned long long)seed, (unsigned long long)h);
        first = false;
    }
    std::printf("\n ],\n \"murmur64a\": [\n");
    first = true;
    unsigned char buf[40];
    for (int i = 0; i < 40; ++i) buf[i] = (unsigned char)(i * 37 + 11);
    for (int len = 0; len <= 33; ++len) {
        uint64_t h = MurmurHash64A(buf, (size_t)len, (uint64_t)len * 0x9e3779b97f4a7c15ull);
        std::printf("%s  [%d, \"%llu\", \"%llu\"]", first ? "" : ",\n", len, (unsigned long long)((uint64_t)len * 0x9e3779b97f4a7c15ull), (unsigned long long)h);
        first = false;
    }
    std::printf("\n ],\n \"mixbits\": [\n");
    first = true;
    const uint64_t mv[] = {0ull, 1ull, 42ull, 0xffffffffffffffffull, 0x0123456789abcdefull, 0xd539d46ed3159a89ull};
    for (int i = 0; i < 6; ++i) {
        std::printf("%s  [\"%llu\", \"%llu\"]", first ? "" : ",\n", (unsigned long long)mv[i], (unsigned long long)MixBits(mv[i]));
        first = false;
    }
    std::printf("\n ],\n \"pcg32_stream\": [\n");
    // seed(initseq) [1-arg], advance(delta), 8 x nextUInt + the same 8 as nextFloat bit patterns
    first = true;
    const uint64_t iseq[] = {0ull, 1ull, 0xd539d46ed3159a89ull, 0x853c49e6748fea9bull, 12345678901234567ull};
    const int64_t deltas[] = {0, 1, 65536, 2 * 65536, 1023ll * 65536, 65535ll * 65536 + 17, -1};
    for (int i = 0; i < 5; ++i) for (int d = 0; d < 7; ++d) {
        pcg32 a; a.seed(iseq[i]); a.advance(deltas[d]);
        pcg32 b = a;
        std::printf("%s  {\"initseq\": \"%llu\", \"delta\": \"%lld\", \"u\": [", first ? "" : ",\n", (unsigned long long)iseq[i], (long long)deltas[d]);
        for (int k = 0; k < 8; ++k) std::printf("%s%u", k ? ", " : "", a.nextUInt());
        std::printf("], \"fbits\": [");
        for (int k = 0; k < 8; ++k) { union { float f; uint32_t u; } x; x.f = b.nextFloat(); std::printf("%s%u", k ? ", " : "", x.u); }
        std::printf("]}");
        first = false;
    }
    std::printf("\n ],\n \"pcg32_seed2\": [\n");
    first = true;
    for (int i = 0; i < 5; ++i) {
        pcg32 a(iseq[i], iseq[(i + 1) % 5]);
        std::printf("%s  {\"initstate\": \"%llu\", \"initseq\": \"%llu\", \"u\": [", first ? "" : ",\n", (unsigned long long)iseq[i], (unsigned long long)iseq[(i + 1) % 5]);
        for (int k = 0; k < 4; ++k) std::printf("%s%u", k ? ", " : "", a.nextUInt());
        std::printf("]}");
        first = false;
    }
    std::printf("\n ]\n}\n");
    return 0;
}
