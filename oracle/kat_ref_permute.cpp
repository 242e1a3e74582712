// oracle/kat_ref_permute.cpp — mints known-answer vectors for random::permute / random::sampleTEA32 from the
// REFERENCE'S OWN text: the Makefile target `ref` extracts the `NAMESPACE_BEGIN(random) ... NAMESPACE_END(random)`
// block of /root/reference/src/kazen/common.cpp (lines 300-346: it uses nothing but <cstdint>) into
// oracle/_ref/random_block.inc — git-ignored, never committed — and this driver compiles it next to the reference's
// own include/kazen/define.h, exactly as kat_ref_main.cpp does for hash.h / pcg32.h. This file contains no
// reference source text. Output: JSON on stdout, merged into tests/golden/int_kats.json by tests/golden/make_int_kats.sh.
//
// Keys are formed the way the callers form them: `permute` takes a uint32 p, the samplers pass a
// uint64 hash (sampler.cpp:122,132,342,356 - silently truncated) or hash * constant evaluated in 64 bits and then
// truncated (sampler.cpp:220,234,241-242).
#include <kazen/define.h>
#include <kazen/hash.h>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

NAMESPACE_BEGIN(kazen)
#include "_ref/random_block.inc"
NAMESPACE_END(kazen)

struct P2i { int32_t x, y; };

int main() {
    using namespace kazen;
    const uint32_t ls[] = {1u, 2u, 3u, 4u, 7u, 16u, 64u, 100u, 1024u, 4096u, 65536u, 1000003u};
    const int32_t px[] = {0, 3, 255, 1919, 65535}, py[] = {0, 5, 255, 1079, 4095};
    const uint32_t dims[] = {2u, 5u, 11u};
    const uint64_t seeds[] = {0ull, 1ull, 0xdeadbeefcafef00dull};
    const uint64_t muls[] = {1ull, 0x45fbe943ull, 0x51633e2dull, 0x68bc21ebull, 0x02e5be93ull};
    std::printf("{\n \"permute\": [\n");
    bool first = true;
    for (uint32_t l : ls) for (int k = 0; k < 5; ++k) for (int d = 0; d < 3; ++d) {
        const P2i p{px[k], py[k]};
        const uint64_t hash = Hash(p, dims[d], seeds[(k + d) % 3]);
        const uint64_t key64 = hash * muls[(k + d) % 5];              // the callers' 64-bit expression
        const uint32_t key = (uint32_t)key64;                        // what permute(uint32_t p) receives
        // indices: the first, the last, and a spread in between
        uint32_t idx[6] = {0u, l - 1u, l / 2u, l / 3u, (uint32_t)((7ull * l) / 8ull), (uint32_t)(key % l)};
        for (int j = 0; j < 6; ++j) {
            std::printf("%s  [%u, %u, \"%llu\", %u]", first ? "" : ",\n", idx[j], l, (unsigned long long)key64, random::permute(idx[j], l, key));
            first = false;
        }
    }
    // a whole permutation (it must be a bijection; the test checks that too)
    std::printf("\n ],\n \"permute_full\": [\n");
    first = true;
    for (uint32_t l : {16u, 100u, 1024u}) {
        const uint32_t key = (uint32_t)(Hash(P2i{7, 9}, 4u, (uint64_t)l) * 0x45fbe943ull);
        std::printf("%s  {\"l\": %u, \"key\": %u, \"out\": [", first ? "" : ",\n", l, key);
        for (uint32_t i = 0; i < l; ++i) std::printf("%s%u", i ? ", " : "", random::permute(i, l, key));
        std::printf("]}");
        first = false;
    }
    std::printf("\n ],\n \"tea32\": [\n");
    first = true;
    const uint32_t v0s[] = {0u, 1u, 12345u, 0xffffffffu, 0x9e3779b9u}, v1s[] = {0u, 2u, 54321u, 0xffffffffu, 0x7f4a7c15u};
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) for (int rounds : {0, 1, 4, 16}) {
        std::printf("%s  [%u, %u, %d, \"%llu\"]", first ? "" : ",\n", v0s[a], v1s[b], rounds, (unsigned long long)random::sampleTEA32(v0s[a], v1s[b], rounds));
        first = false;
    }
    std::printf("\n ]\n}\n");
    return 0;
}
