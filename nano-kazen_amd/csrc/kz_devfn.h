// kz_devfn.h — device functions of the path_mis hot path (vector math, sampler, traversal, post-intersection,
// BSDFs, lights, camera, Li). Included by the device translation units (kz_render.hip, kz_film.hip, kz_debug.hip); every function cites the reference lines it follows.
#pragma once
#include <hip/hip_runtime.h>
#include "kz_internal.h"
#include "kz_crmath.h"

#define KZ_BLOCK 256
#define KZ_INF __builtin_huge_valf()

// ============================================================================================
// small vector math (explicit, so operation order is visible; compiled with -ffp-contract=off)
// ============================================================================================
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 mk(float a) { return mk(a, a, a); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalized(V3 a) { float n2 = dot(a, a); return n2 > 0.f ? a / sqrtf(n2) : a; }   // Eigen normalized()
__device__ __forceinline__ float maxCoeff(V3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
__device__ __forceinline__ float sqr(float x) { return x * x; }

// 1.0f / x, bit for bit. The compiler expands an IEEE division into v_div_scale x2, v_rcp, Newton, v_div_fmas, v_div_fixup and two switches
// of the denormal mode (42-47 SIMD cycles, scripts/micro/op_cost.hip). For a numerator of 1 the hardware reciprocal followed by two
// Newton-Raphson steps in FMA form gives THE SAME BITS for every one of the 3 355 443 202 floats with 2^-100 <= |x| <= 2^100 - checked
// exhaustively on the GPU against that division (scripts/micro/rcp_exact.hip -> profiles/rcp_exact.json; 0 mismatches) - in ~20 cycles.
// Everything else (zero, denormal, huge, inf, nan) takes the division.
__device__ __forceinline__ float rcpExact(float x) {
    const float ax = fabsf(x);
    if (__builtin_expect(ax >= 0x1p-100f && ax <= 0x1p100f, 1)) {
        float r = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
        e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
        return r;
    }
    return 1.0f / x;
}
// sqrtf(x), bit for bit: the compiler's own correctly rounded sequence (v_rsq_f32, g = x*y, h = y/2, one coupled Newton step, one residual
// correction) without its input scaling for tiny x, the un-scaling and the class test for 0 / inf, which only matter outside
// 2^-95 <= x < 2^96: 54-58 -> ~36 SIMD cycles. All 1 602 224 128 floats of that range checked against sqrtf on the GPU (rcp_exact.hip).
__device__ __forceinline__ float sqrtExact(float x) {
    if (__builtin_expect((__float_as_uint(x) - 0x10000000u) < (0x6F800000u - 0x10000000u), 1)) {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y, h = 0.5f * y;
        const float r = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
    return sqrtf(x);
}
#define KZ_EPSILON 1e-5f                      // common.h:27
#define KZ_ONE_MINUS_EPS 0x1.fffffep-1f       // common.h:28
#define KZ_INV_PI 0.31830988618379067154f     // common.h:34
#define KZ_PI_F 3.14159265358979323846f

struct Counters { uint32_t rays, nodes, tris, hits, lsamples, dropped; };

// ============================================================================================
// a6/a7/a8 integer sampler plumbing (hash.h:15-65,71-78,100-108; pcg32.h:54-73,108-117; common.cpp:316-344)
// ============================================================================================
#define MURMUR_M 0xc6a4a7935bd1e995ull
__device__ __forceinline__ uint64_t murmurBlock(uint64_t h, uint64_t k) {
    k *= MURMUR_M; k ^= k >> 47; k *= MURMUR_M;
    h ^= k; h *= MURMUR_M;
    return h;
}
__device__ __forceinline__ uint64_t murmurFinish(uint64_t h) { h ^= h >> 47; h *= MURMUR_M; h ^= h >> 47; return h; }
// Hash(Point2i p, uint64 seed): 16-byte key, seed 0 (sampler.cpp:44)
__device__ __forceinline__ uint64_t hashPixelSeed(int px, int py, uint64_t seed) {
    uint64_t h = 16ull * MURMUR_M;
    h = murmurBlock(h, (uint64_t)(uint32_t)px | ((uint64_t)(uint32_t)py << 32));
    h = murmurBlock(h, seed);
    return murmurFinish(h);
}
// Hash(Point2i p, uint32 dim, uint64 seed): 20-byte key (sampler.cpp:341,355)
__device__ __forceinline__ uint64_t hashPixelDimSeed(int px, int py, uint32_t dim, uint64_t seed) {
    uint64_t h = 20ull * MURMUR_M;
    h = murmurBlock(h, (uint64_t)(uint32_t)px | ((uint64_t)(uint32_t)py << 32));
    h = murmurBlock(h, (uint64_t)dim | ((seed & 0xffffffffull) << 32));
    h ^= (seed >> 32);            // the 4 tail bytes
    h *= MURMUR_M;
    return murmurFinish(h);
}
// The same hash with its first block - the only part that depends on the pixel - taken from `hp` = hashPixelBlock(px, py): a path
// draws ~7 samples per bounce at consecutive dimensions of ONE pixel, so the samplers compute that block once per kernel invocation.
__device__ __forceinline__ uint64_t hashPixelBlock(int px, int py) { return murmurBlock(20ull * MURMUR_M, (uint64_t)(uint32_t)px | ((uint64_t)(uint32_t)py << 32)); }
__device__ __forceinline__ uint64_t hashDimSeed(uint64_t hp, uint32_t dim, uint64_t seed) {
    uint64_t h = murmurBlock(hp, (uint64_t)dim | ((seed & 0xffffffffull) << 32));
    h ^= (seed >> 32);            // the 4 tail bytes
    h *= MURMUR_M;
    return murmurFinish(h);
}
__device__ __forceinline__ uint64_t mixBits(uint64_t v) {
    v ^= (v >> 31); v *= 0x7fb5d329728ea185ull; v ^= (v >> 27); v *= 0x81dadef4bc2dd44dull; v ^= (v >> 33);
    return v;
}
#define PCG32_MULT 0x5851f42d4c957f2dULL
__device__ __forceinline__ uint32_t permuteIdx(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
        i *= 0x0929eb3f; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303; i ^= (i & w) >> 2;
        i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    if ((l & w) == 0) return (i + p) & w;         // l a power of two (w = l - 1): the same value as the modulo, without the division
    return (i + p) % l;
}

// a4/a5/a9 Sampler (sampler.cpp:18-71 independent, :273-390 pmj02bn). Draw order H1 (GCC, right-to-left
// argument evaluation): Independent::next2D draws y first; bsdf->sample(bRec, next1D(), next2D()) draws the
// 2-D sample first — both written as sequenced statements at the call sites.
struct Sampler {
    uint64_t state, inc;            // independent: pcg32
    int px, py; uint32_t idx, dim;  // pmj02bn
    int type;
    uint64_t hp;                    // hashPixelBlock(px, py): the pixel's share of Hash(p, dim, seed) (every sampler but independent)

    __device__ __forceinline__ uint32_t nextUInt() {
        uint64_t old = state;
        state = old * PCG32_MULT + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    __device__ __forceinline__ float nextFloat() { return __uint_as_float((nextUInt() >> 9) | 0x3f800000u) - 1.0f; }
    __device__ __forceinline__ void generateSample(const KzParams &P, const KzDevTables &T, int x, int y, uint32_t sampleIndex) {
        px = x; py = y; idx = sampleIndex;
        hp = type != KZ_SAMPLER_INDEPENDENT ? hashPixelBlock(x, y) : 0ull;
        if (type != KZ_SAMPLER_PMJ02BN) {                 // independent, stratified, correlated: same pcg32 seeding
            dim = 0;
            uint64_t h = hashPixelSeed(x, y, P.seed);
            // pcg32::seed(initseq) = seed(MixBits(initseq), initseq)
            inc = (h << 1u) | 1u;
            state = inc;                       // state = 0*MULT + inc
            state += mixBits(h);
            state = state * PCG32_MULT + inc;
            // advance(sampleIndex * 65536 + 0) through the tabulated affine jump
            KzPcgJump j = T.jump[sampleIndex];
            state = j.mult * state + inc * j.plus;
        } else {
            dim = 2;                           // max(2, dimension=0)
        }
    }
    __device__ __forceinline__ float blueNoise(const KzDevTables &T, uint32_t tex) const {            // bluenoise.h:16-23
        uint32_t t = tex % KZ_BLUENOISE_TEXTURES;
        uint32_t x = (uint32_t)px % KZ_BLUENOISE_RES, y = (uint32_t)py % KZ_BLUENOISE_RES;
        return T.bn[(t * KZ_BLUENOISE_RES + x) * KZ_BLUENOISE_RES + y];
    }
    // (forced inline, like next2D: as CALLS they take the kernel's KzParams / KzDevTables by address, which puts a copy of both - ~600 B per lane - into
    // scratch memory; the EXT kernel variants, where the compiler stopped inlining them, carried 960 B of scratch until round 4)
    __device__ __forceinline__ float next1D(const KzParams &P, const KzDevTables &T) {
        if (type == KZ_SAMPLER_INDEPENDENT) return nextFloat();
        if (type == KZ_SAMPLER_STRATIFIED) {                                       // sampler.cpp:119-127
            const uint64_t h = hashDimSeed(hp, dim, P.seed);
            const int stratum = (int)permuteIdx(idx, P.sampleCount, (uint32_t)h);
            ++dim;
            const float delta = nextFloat();
            return ((float)stratum + delta) / (float)P.sampleCount;
        }
        if (type == KZ_SAMPLER_CORRELATED) {                                       // sampler.cpp:215-227
            const uint64_t h = hashDimSeed(hp, dim, P.seed);
            const int p = (int)permuteIdx(idx, P.sampleCount, (uint32_t)h * 0x45fbe943u);
            const float j = nextFloat();
            ++dim;
            return ((float)p + j) / (float)P.sampleCount;
        }
        uint64_t h = hashDimSeed(hp, dim, P.seed);
        int index = (int)permuteIdx(idx, P.sampleCount, (uint32_t)h);
        float delta = blueNoise(T, dim);
        ++dim;
        const float num = (float)index + delta;
        return fminf(P.sppPow2 ? num * P.invSpp : num / (float)P.sampleCount, KZ_ONE_MINUS_EPS);    // exact either way (kz_internal.h)
    }
    __device__ __forceinline__ void next2D(const KzParams &P, const KzDevTables &T, float &x, float &y) {
        if (type == KZ_SAMPLER_INDEPENDENT) { y = nextFloat(); x = nextFloat(); return; }
        if (type == KZ_SAMPLER_STRATIFIED) {                                       // sampler.cpp:129-139
            const uint64_t h = hashDimSeed(hp, dim, P.seed);
            const int stratum = (int)permuteIdx(idx, P.sampleCount, (uint32_t)h);
            dim += 2;
            const int sx = stratum % P.resX, sy = stratum / P.resX;
            const float dx = nextFloat();
            const float dy = nextFloat();
            x = ((float)sx + dx) / (float)P.resX; y = ((float)sy + dy) / (float)P.resX;
            return;
        }
        if (type == KZ_SAMPLER_CORRELATED) {                                       // sampler.cpp:229-251
            const uint32_t h = (uint32_t)hashDimSeed(hp, dim, P.seed);    // permute() takes the low 32 bits of hash * const
            const uint32_t s = permuteIdx(idx, P.sampleCount, h * 0x51633e2du);
            const uint32_t cy = s / (uint32_t)P.resX, cx = s % (uint32_t)P.resX;
            const uint32_t sx = permuteIdx(cx, (uint32_t)P.resX, h * 0x68bc21ebu);
            const uint32_t sy = permuteIdx(cy, (uint32_t)P.resY, h * 0x02e5be93u);
            const float jx = nextFloat();
            const float jy = nextFloat();
            dim += 2;
            x = ((float)cx + ((float)sy + jx) / (float)P.resY) / (float)P.resX;
            y = ((float)cy + ((float)sx + jy) / (float)P.resX) / (float)P.resY;
            return;
        }
        uint32_t index = idx;
        uint32_t inst = dim / 2;
        if (inst >= KZ_PMJ02BN_SETS) {
            uint64_t h = hashDimSeed(hp, dim, P.seed);
            index = permuteIdx(idx, P.sampleCount, (uint32_t)h);
        }
        inst %= KZ_PMJ02BN_SETS; index %= KZ_PMJ02BN_SAMPLES;
        const float2 e = reinterpret_cast<const float2 *>(T.pmj)[(size_t)inst * KZ_PMJ02BN_SAMPLES + index];                 // pmj02table.h:28-29, narrowed on the host
        float ux = e.x, uy = e.y;
        ux += blueNoise(T, dim); uy += blueNoise(T, dim + 1);
        if (ux >= 1) ux -= 1;
        if (uy >= 1) uy -= 1;
        dim += 2;
        x = fminf(ux, KZ_ONE_MINUS_EPS); y = fminf(uy, KZ_ONE_MINUS_EPS);
    }
    __device__ __forceinline__ void nextPixel2D(const KzParams &P, const KzDevTables &T, float &x, float &y) {
        if (type != KZ_SAMPLER_PMJ02BN) { next2D(P, T, x, y); return; }
        int tile = P.pixelTileSize;
        int tx = px % tile, ty = py % tile;
        size_t off = (size_t)(tx + ty * tile) * P.sampleCount + idx;
        const float2 v = *reinterpret_cast<const float2 *>(T.pixelSamples + 2 * off);
        x = v.x; y = v.y;
    }
};

// ============================================================================================
// a13 traversal: closest hit (replaces rtcIntersect1, accel.cpp:98), a15 Moeller-Trumbore leaf test
// (mesh.cpp:55-92), a16 slab node test (bbox.h:316-343; conservative form on padded boxes).
// ============================================================================================
struct RawHit { float t, u, v; uint32_t tri; uint32_t gid; };

// One 64-B node packet: slab tests of both children (a16). Conservative on the builder's padded boxes: fminf/fmaxf drop
// the NaN of 0*inf on a degenerate axis (that slab then does not constrain) and the far side is widened by 2 ulp.
struct NodeTest { bool h0, h1; float n0, n1; uint32_t c0, c1; };
__device__ __forceinline__ NodeTest nodeTest(const KzDevTables &T, uint32_t node, V3 o, float rx, float ry, float rz, float tmin, float tmax) {
    const float4 *np = reinterpret_cast<const float4 *>(T.nodes + node);
    const float4 q0 = np[0], q1 = np[1], q2 = np[2];
    const uint4 q3 = *reinterpret_cast<const uint4 *>(np + 3);
    float t0, t1, n0, f0, n1, f1;
    t0 = (q0.x - o.x) * rx; t1 = (q0.w - o.x) * rx; n0 = fminf(t0, t1); f0 = fmaxf(t0, t1);
    t0 = (q0.y - o.y) * ry; t1 = (q1.x - o.y) * ry; n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
    t0 = (q0.z - o.z) * rz; t1 = (q1.y - o.z) * rz; n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
    t0 = (q1.z - o.x) * rx; t1 = (q2.y - o.x) * rx; n1 = fminf(t0, t1); f1 = fmaxf(t0, t1);
    t0 = (q1.w - o.y) * ry; t1 = (q2.z - o.y) * ry; n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
    t0 = (q2.x - o.z) * rz; t1 = (q2.w - o.z) * rz; n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
    f0 *= 1.0000004f; f1 *= 1.0000004f;
    NodeTest r;
    r.h0 = (fmaxf(n0, tmin) <= fminf(f0, tmax)); r.h1 = (fmaxf(n1, tmin) <= fminf(f1, tmax));
    r.n0 = n0; r.n1 = n1; r.c0 = q3.x; r.c1 = q3.y;
    return r;
}
// Mesh::rayIntersect (mesh.cpp:55-92), operation for operation, on one 48-B leaf triangle.
// (triTestV: the same body on a triangle already in registers - the list kernel fetches the next triangle while it tests this one)
__device__ __forceinline__ bool triTestV(const float4 a, const float4 b, const float4 c, V3 o, V3 d, float tmin, float tmax, float &t, float &u, float &v, uint32_t &gid) {
    const V3 p0 = mk(a.x, a.y, a.z), e1 = mk(a.w, b.x, b.y), e2 = mk(b.z, b.w, c.x);
    gid = __float_as_uint(c.w);
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (det > -1e-8f && det < 1e-8f) return false;
    float inv_det = rcpExact(det);
    V3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, qvec) * inv_det;
    return t >= tmin && t <= tmax;
}
__device__ __forceinline__ bool triTest(const KzTri *tri, V3 o, V3 d, float tmin, float tmax, float &t, float &u, float &v, uint32_t &gid) {
    const float4 *tp = reinterpret_cast<const float4 *>(tri);
    return triTestV(tp[0], tp[1], tp[2], o, d, tmin, tmax, t, u, v, gid);
}
// Streaming (non-temporal) accesses for the path state: every record is read or written once per kernel, 2^27 of them per pass - the `nt` hint
// keeps them from displacing the BVH packets and leaf triangles, which are what the traversal kernels want to find in L2 again.
typedef float kz_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 kzLoadStream(const float4 *p) { const kz_f4v v = __builtin_nontemporal_load(reinterpret_cast<const kz_f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void kzStoreStream(float4 *p, float4 v) { const kz_f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<kz_f4v *>(p)); }
__device__ __forceinline__ uint32_t kzLoadStream(const uint32_t *p) { return __builtin_nontemporal_load(p); }

// LDS through explicit 32-bit byte addresses (address space 3): ds_read_b32 / ds_write_b32 at a VGPR the kernel keeps as state
typedef __attribute__((address_space(3))) uint32_t KzLds32;
__device__ __forceinline__ uint32_t kzLdsAddr(const uint32_t *p) { return (uint32_t)(uintptr_t)(const KzLds32 *)p; }
__device__ __forceinline__ void kzLdsPut(uint32_t a, uint32_t v) { *(KzLds32 *)(uintptr_t)a = v; }
__device__ __forceinline__ uint32_t kzLdsGet(uint32_t a) { return *(const KzLds32 *)(uintptr_t)a; }

// One 64-B BVH4 packet (KzNode4): four quantised child boxes, slab tests in the FMA form t = q * (s * rcp) + (p - o) * rcp
// (s is a power of two, so s * rcp is exact; the leaf boxes carry an absolute pad that covers the rounding of the rest).
// Returns the four children as sortable keys: (bits of max(tnear, tmin) with the two low bits replaced by the child slot),
// 0xFFFFFFFF for a miss, sorted ascending, so key[0] is the nearest hit child.
struct Node4Test { uint32_t k0, k1, k2, k3; uint4 refs; };
typedef float kz_f2 __attribute__((ext_vector_type(2)));
// keys in slot order: (bits of max(tnear, tmin) with the two low bits replaced by the child slot), 0xFFFFFFFF for a miss
template <bool ORDERED = true>
__device__ __forceinline__ void node4KeysOf(const uint4 q0, const uint4 q1, const uint4 q2, V3 o, float rx, float ry, float rz, float tmin, float tmax,
                                            uint32_t (&key)[4]) {
    const float ax = __uint_as_float(q0.w) * rx, ay = __uint_as_float(q2.z) * ry, az = __uint_as_float(q2.w) * rz;      // the packet carries 2^e per axis as floats
    const float bx = (__uint_as_float(q0.x) - o.x) * rx, by = (__uint_as_float(q0.y) - o.y) * ry, bz = (__uint_as_float(q0.z) - o.z) * rz;
    // The sign of the direction says which plane of a slab is entered first: pick the packed near / far words for all four
    // children at once (rcp is finite and non-zero here, see the caller), so each child needs only 6 cvt + 3 packed fma (the
    // near and the far plane of an axis share scale and offset: one v_pk_fma_f32) + max3 + min3.
    // An empty slot (qlo = 255, qhi = 0) comes out with near > far on every axis for either sign: never hit.
    const uint32_t nX = rx >= 0.f ? q1.x : q1.w, fX = rx >= 0.f ? q1.w : q1.x;
    const uint32_t nY = ry >= 0.f ? q1.y : q2.x, fY = ry >= 0.f ? q2.x : q1.y;
    const uint32_t nZ = rz >= 0.f ? q1.z : q2.y, fZ = rz >= 0.f ? q2.y : q1.z;
    const kz_f2 AX = {ax, ax}, AY = {ay, ay}, AZ = {az, az}, BX = {bx, bx}, BY = {by, by}, BZ = {bz, bz};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const kz_f2 qx = {(float)((nX >> (8 * i)) & 0xffu), (float)((fX >> (8 * i)) & 0xffu)};
        const kz_f2 qy = {(float)((nY >> (8 * i)) & 0xffu), (float)((fY >> (8 * i)) & 0xffu)};
        const kz_f2 qz = {(float)((nZ >> (8 * i)) & 0xffu), (float)((fZ >> (8 * i)) & 0xffu)};
        const kz_f2 tx = __builtin_elementwise_fma(qx, AX, BX), ty = __builtin_elementwise_fma(qy, AY, BY), tz = __builtin_elementwise_fma(qz, AZ, BZ);
        const float n = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), tmin);
        const float f = fminf(fminf(fminf(tx.y, ty.y), tz.y) * 1.0000004f, tmax);
        if (ORDERED) key[i] = (n <= f) ? ((__float_as_uint(n) & ~3u) | (uint32_t)i) : 0xFFFFFFFFu;
        else key[i] = (n <= f) ? 0u : 0xFFFFFFFFu;        // any-hit rays only ask WHETHER a child is hit: no entry distance, no slot bits
    }
}
// The packet of node `node`: table base (scalar) + a 32-BIT byte offset, so the fetches take the base from SGPRs and the offset from one
// VGPR (v_lshlrev_b32) instead of a 64-bit shift and a 64-bit add per visit. kz_scene_create refuses trees above 2^26 packets (4 GB).
__device__ __forceinline__ const uint4 *kzNode4Ptr(const KzDevTables &T, uint32_t node) {
    return reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(T.nodes4) + (uint32_t)(node << 6));
}
template <bool ORDERED = true>
__device__ __forceinline__ void node4Keys(const KzDevTables &T, uint32_t node, V3 o, float rx, float ry, float rz, float tmin, float tmax,
                                          uint32_t (&key)[4], uint4 &refs) {
    const uint4 *np = kzNode4Ptr(T, node);
    const uint4 q0 = np[0], q1 = np[1], q2 = np[2];
    refs = np[3];
    node4KeysOf<ORDERED>(q0, q1, q2, o, rx, ry, rz, tmin, tmax, key);
}
__device__ __forceinline__ Node4Test node4Test(const KzDevTables &T, uint32_t node, V3 o, float rx, float ry, float rz, float tmin, float tmax) {
    Node4Test r; uint32_t key[4];
    node4Keys(T, node, o, rx, ry, rz, tmin, tmax, key, r.refs);
    // 5-comparator sorting network on unsigned keys
    uint32_t a = min(key[0], key[1]), b = max(key[0], key[1]), c = min(key[2], key[3]), d = max(key[2], key[3]);
    uint32_t lo = min(a, c), m1 = max(a, c), m2 = min(b, d), hi = max(b, d);
    r.k0 = lo; r.k1 = min(m1, m2); r.k2 = max(m1, m2); r.k3 = hi;
    return r;
}
// branch-free slot select (two bit tests + three v_cndmask; an `i == 0 ? .. : i == 1 ? ..` chain is lowered to a switch with
// divergent branches by the compiler)
__device__ __forceinline__ uint32_t pick4b(const uint4 &v, uint32_t i) {
    const bool b0 = i & 1u, b1 = i & 2u;
    const uint32_t lo = b0 ? v.y : v.x, hi = b0 ? v.w : v.z;
    return b1 ? hi : lo;
}
__device__ __forceinline__ uint32_t pick4(const uint4 &v, uint32_t i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

__device__ __forceinline__ bool rayIsFinite(V3 o, V3 d) { return fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + fabsf(d.x) + fabsf(d.y) + fabsf(d.z) < KZ_INF; }

template <bool STATS>
__device__ __forceinline__ bool closestHit(const KzDevTables &T, uint32_t rootRef, V3 o, V3 d, float tmin, float tmax,
                                           RawHit &best, uint32_t *stk, Counters &cn) {
    bool found = false;
    best.t = KZ_INF; best.u = best.v = 0.f; best.tri = 0; best.gid = 0;
    if (STATS) cn.rays++;
    if (rootRef == 0xFFFFFFFFu) return false;
    // A ray with a non-finite origin or direction can hit nothing (every Moeller-Trumbore comparison fails on NaN),
    // but fminf/fmaxf would let it pass EVERY slab test: one such lane would walk the whole tree. Miss at once.
    if (!(fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + fabsf(d.x) + fabsf(d.y) + fabsf(d.z) < KZ_INF)) return false;
    const float rx = rcpExact(d.x), ry = rcpExact(d.y), rz = rcpExact(d.z);      // ray.h:56-58 cwiseInverse
    uint32_t cur = rootRef;
    int sp = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            uint32_t start = (cur & 0x7fffffffu) >> 3, count = (cur & 7u) + 1;
            for (uint32_t i = 0; i < count; ++i) {
                float t, u, v; uint32_t gid;
                if (STATS) cn.tris++;
                if (!triTest(T.tris + start + i, o, d, tmin, tmax, t, u, v, gid)) continue;
                // ties on t go to the lower global triangle id: order independent (Embree's tie rule is unspecified)
                if (!found || t < best.t || (t == best.t && gid < best.gid)) {
                    found = true; best.t = t; best.u = u; best.v = v; best.tri = start + i; best.gid = gid; tmax = t;
                }
            }
            if (sp == 0) break;
            cur = stk[(--sp) * KZ_BLOCK];
            continue;
        }
        if (STATS) cn.nodes++;
        const NodeTest nt = nodeTest(T, cur, o, rx, ry, rz, tmin, tmax);
        if (nt.h0 && nt.h1) {
            const bool swap = nt.n1 < nt.n0;
            stk[(sp++) * KZ_BLOCK] = swap ? nt.c0 : nt.c1;
            cur = swap ? nt.c1 : nt.c0;
        } else if (nt.h0) cur = nt.c0;
        else if (nt.h1) cur = nt.c1;
        else {
            if (sp == 0) break;
            cur = stk[(--sp) * KZ_BLOCK];
        }
    }
    return found;
}

// Any-hit variant for shadow segments: true at the first triangle with t in [tmin, tmax] (no ordering, no shrinking).
template <bool STATS>
__device__ __forceinline__ bool anyHit(const KzDevTables &T, uint32_t rootRef, V3 o, V3 d, float tmin, float tmax, uint32_t *stk, Counters &cn) {
    if (STATS) cn.rays++;
    if (rootRef == 0xFFFFFFFFu) return false;
    if (!(fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + fabsf(d.x) + fabsf(d.y) + fabsf(d.z) < KZ_INF)) return false;
    const float rx = rcpExact(d.x), ry = rcpExact(d.y), rz = rcpExact(d.z);
    uint32_t cur = rootRef;
    int sp = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            uint32_t start = (cur & 0x7fffffffu) >> 3, count = (cur & 7u) + 1;
            for (uint32_t i = 0; i < count; ++i) {
                float t, u, v; uint32_t gid;
                if (STATS) cn.tris++;
                if (triTest(T.tris + start + i, o, d, tmin, tmax, t, u, v, gid)) return true;
            }
            if (sp == 0) break;
            cur = stk[(--sp) * KZ_BLOCK];
            continue;
        }
        if (STATS) cn.nodes++;
        const NodeTest nt = nodeTest(T, cur, o, rx, ry, rz, tmin, tmax);
        if (nt.h0 && nt.h1) {
            const bool swap = nt.n1 < nt.n0;
            stk[(sp++) * KZ_BLOCK] = swap ? nt.c0 : nt.c1;
            cur = swap ? nt.c1 : nt.c0;
        } else if (nt.h0) cur = nt.c0;
        else if (nt.h1) cur = nt.c1;
        else {
            if (sp == 0) break;
            cur = stk[(--sp) * KZ_BLOCK];
        }
    }
    return false;
}

// The reference's shadow test, literally (integrator.cpp:257-278): closest hits, walking through lights whose
// lightPrimaryVisibility is false; note that the far end moves out by traceBias per walk-through (maxt - t).
// light row of the mesh a triangle belongs to (-1: not an emitter), from the last quad of its shading record
__device__ __forceinline__ int lightOfGid(const KzDevTables &T, uint32_t gid) {
    return (int)(__float_as_uint(reinterpret_cast<const float4 *>(T.shade + gid)[6].w) >> 2) - 1;
}
template <bool STATS>
__device__ __forceinline__ bool shadowOccludedLiteral(const KzParams &P, const KzDevTables &T, V3 so, V3 dir, float smin, float smax,
                                                      uint32_t *stk, Counters &cn) {
    const float eps = P.traceBias;
    for (;;) {
        RawHit sh;
        if (!closestHit<STATS>(T, P.rootRef, so, dir, smin, smax, sh, stk, cn)) return false;
        const int ol = lightOfGid(T, sh.gid);
        if (ol < 0 || T.lights[ol].primaryVisibility) return true;
        so = so + dir * (sh.t + eps); smin = eps; smax = smax - sh.t;
    }
}

// Is a triangle of a light with lightPrimaryVisibility == false hit on the segment? The few such triangles (<= 64, P.nIlTris)
// are tested by brute force behind their common bounding box (a conservative prefilter: rx, ry, rz only have to be finite
// or the IEEE reciprocals of the direction).
__device__ __forceinline__ bool invisibleLightOnSegment(const KzParams &P, const KzDevTables &T, V3 o, V3 d, float rx, float ry, float rz, float tmin, float tmax) {
    if (P.nIlTris == 0) return false;
    float t0 = (P.ilLo[0] - o.x) * rx, t1 = (P.ilHi[0] - o.x) * rx;
    float n = fminf(t0, t1), f = fmaxf(t0, t1);
    t0 = (P.ilLo[1] - o.y) * ry; t1 = (P.ilHi[1] - o.y) * ry; n = fmaxf(n, fminf(t0, t1)); f = fminf(f, fmaxf(t0, t1));
    t0 = (P.ilLo[2] - o.z) * rz; t1 = (P.ilHi[2] - o.z) * rz; n = fmaxf(n, fminf(t0, t1)); f = fminf(f, fmaxf(t0, t1));
    f *= 1.0000004f;
    if (!(fmaxf(n, tmin) <= fminf(f, tmax))) return false;
    for (uint32_t i = 0; i < P.nIlTris; ++i) {
        float t, u, v; uint32_t g;
        if (triTest(T.ilTris + i, o, d, tmin, tmax, t, u, v, g)) return true;
    }
    return false;
}

// Exact fast form of the same test. If no invisible-light triangle is hit on the segment, the reference's loop is a
// single closest-hit query and "occluded" == "any triangle hit in [tmin,tmax]"; only if one is hit (rare) the literal loop runs.
template <bool STATS>
__device__ __forceinline__ bool shadowOccluded(const KzParams &P, const KzDevTables &T, V3 so, V3 dir, float smin, float smax,
                                               uint32_t *stk, Counters &cn) {
    if (P.shadowFast && !invisibleLightOnSegment(P, T, so, dir, rcpExact(dir.x), rcpExact(dir.y), rcpExact(dir.z), smin, smax))
        return anyHit<STATS>(T, P.rootRef, so, dir, smin, smax, stk, cn);
    return shadowOccludedLiteral<STATS>(P, T, so, dir, smin, smax, stk, cn);
}

// ============================================================================================
// a14 post-intersection (accel.cpp:113-236) + a17 Frame (frame.h:14-51, common.cpp:436-445)
// ============================================================================================
struct Frame3 { V3 s, t, n; };
__device__ __forceinline__ Frame3 frameFromNormal(V3 a) {
    Frame3 f; f.n = a;
    V3 c;
    if (fabsf(a.x) > fabsf(a.y)) { float invLen = rcpExact(sqrtExact(a.x * a.x + a.z * a.z)); c = mk(a.z * invLen, 0.0f, -a.x * invLen); }
    else { float invLen = rcpExact(sqrtExact(a.y * a.y + a.z * a.z)); c = mk(0.0f, a.z * invLen, -a.y * invLen); }
    f.t = c; f.s = cross(c, a);
    return f;
}
__device__ __forceinline__ V3 toLocal(const Frame3 &f, V3 v) { return mk(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
__device__ __forceinline__ V3 toWorld(const Frame3 &f, V3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

struct Its {
    V3 p; float t; float uvx, uvy; Frame3 sh; V3 geoN; uint32_t mesh; uint32_t prim; float bu, bv;
    uint32_t bsdf; int32_t light;        // the mesh's BSDF row and light row (-1: none), from the shading record
    V3 dpdu;        // accel.cpp:185,209 — only NormalMap::getFrame reads it (dead code in the kernels without normal maps)
};

template <bool GEO>
__device__ __forceinline__ void postIntersect(const KzDevTables &T, const RawHit &rh, Its &its) {
    const float4 *sp = reinterpret_cast<const float4 *>(T.shade + rh.gid);
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
    const uint32_t lf = __float_as_uint(s6.w);
    its.mesh = __float_as_uint(s6.x); its.prim = __float_as_uint(s6.y); its.bsdf = __float_as_uint(s6.z); its.light = (int32_t)(lf >> 2) - 1;
    its.t = rh.t; its.bu = rh.u; its.bv = rh.v;
    const bool hasN = lf & 1u, hasUV = lf & 2u;
    const V3 p0 = mk(s0.x, s0.y, s0.z), p1 = mk(s0.w, s1.x, s1.y), p2 = mk(s1.z, s1.w, s2.x);
    const V3 n0 = mk(s2.y, s2.z, s2.w), n1 = mk(s3.x, s3.y, s3.z), n2 = mk(s3.w, s4.x, s4.y);
    const float uv0x = s4.z, uv0y = s4.w, uv1x = s5.x, uv1y = s5.y, uv2x = s5.z, uv2y = s5.w;
    const float bx = 1 - (rh.u + rh.v), by = rh.u, bz = rh.v;                 // accel.cpp:122-123
    const V3 orignP = bx * p0 + by * p1 + bz * p2;                            // accel.cpp:142
    if (hasN) {                                                               // Hanika terminator offset, accel.cpp:144-153
        V3 tu = orignP - p0, tv = orignP - p1, tw = orignP - p2;
        float du = fminf(0.f, dot(tu, n0)), dv = fminf(0.f, dot(tv, n1)), dw = fminf(0.f, dot(tw, n2));
        tu = tu - du * n0; tv = tv - dv * n1; tw = tw - dw * n2;
        its.p = orignP + bx * tu + by * tv + bz * tw;
    } else its.p = orignP;                                                    // H4: the reference is UB without normals
    const V3 dp0 = p1 - p0, dp1 = p2 - p0;
    const V3 gx = cross(dp0, dp1);
    V3 geoN = mk(0.f);
    if (GEO || !hasN) geoN = normalized(gx);                                  // accel.cpp:156-158
    its.geoN = geoN;
    its.uvx = rh.u; its.uvy = rh.v;
    if (hasUV) { its.uvx = bx * uv0x + by * uv1x + bz * uv2x; its.uvy = bx * uv0y + by * uv1y + bz * uv2y; }   // accel.cpp:161-164
    bool tangent = false;
    if (hasN) {
        const V3 shN = bx * n0 + by * n1 + bz * n2;
        if (hasUV) {                                                          // accel.cpp:166-217
            const float duv0x = uv1x - uv0x, duv0y = uv1y - uv0y, duv1x = uv2x - uv0x, duv1y = uv2y - uv0y;
            const float length = norm(gx);
            const float determinant = duv0x * duv1y - duv0y * duv1x;
            if (length > 0.f && determinant > 0.f) {
                const float invDet = rcpExact(determinant);
                const V3 dpdu = (duv1y * dp0 - duv0y * dp1) * invDet;
                its.sh.n = normalized(shN);
                its.sh.s = normalized(dpdu - shN * dot(shN, dpdu));
                its.sh.t = normalized(cross(its.sh.n, its.sh.s));
                its.dpdu = dpdu;
                tangent = true;
            }
        }
        if (!tangent) its.sh = frameFromNormal(normalized(shN));              // accel.cpp:203-229
    } else its.sh = frameFromNormal(geoN);                                    // accel.cpp:231-233
    // H12: outside the tangent branch the reference leaves its.dpdu stale / uninitialised except in the degenerate-uv
    // branch, which sets it to shFrame.s (accel.cpp:209); that value is used for every non-tangent branch here.
    if (!tangent) its.dpdu = its.sh.s;
}

// ============================================================================================
// a23 warp, a22 GGX helpers, a20 diffuse, a21 kiss (warp.cpp:85-115; ggx_brdf.h; bsdf.cpp:27-75, :1215-1371)
// ============================================================================================
__device__ __forceinline__ V3 squareToCosineHemisphere(float sx, float sy) {
    float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (KZ_PI_F / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (KZ_PI_F / 2.0f) - (r1 / r2) * (KZ_PI_F / 4.0f); }
    float sinPhi, cosPhi; kzSinCos(phi, &sinPhi, &cosPhi);             // kz_crmath.h: sin and cos as defined double sequences, the oracle's bit for bit
    float px = r * cosPhi, py = r * sinPhi;
    float z = sqrtExact(1.0f - px * px - py * py);
    if (z == 0) z = 1e-10f;
    return mk(px, py, z);
}
struct A2 { float x, y; };
__device__ __forceinline__ V3 schlickFresnel(V3 f0, float cosTheta) {            // ggx_brdf.h:15-24
    // pow(1 - cosTheta, 5.0f) (ggx_brdf.h:23; glibc powf, < 0.53 ulp): the fifth power in double, narrowed once - within 2^-51 of the exact
    // value before the narrowing, so it equals the correctly rounded float except for ~1 argument in 2^27; ocml's powf costs 168 VALU
    // instructions per call and is the less accurate of the two (scripts/micro/shade_cost.sh)
    const double xd = (double)(1.0f - cosTheta), xd2 = xd * xd;
    float t = (float)(xd2 * xd2 * xd);
    return f0 * 1.0f + (mk(1.f) - f0) * t;
}
__device__ __forceinline__ A2 roughnessToAlpha(float roughness, float anisotropy) {   // ggx_brdf.h:28-37
    float alpha = fmaxf(0.001f, sqr(roughness));
    A2 a; a.x = alpha * (1.0f + anisotropy); a.y = alpha * (1.0f - anisotropy);
    return a;
}
__device__ __forceinline__ float ggxLambda(V3 v, A2 a) {                          // ggx_brdf.h:41-45
    float squared = (sqr(a.x) * sqr(v.x) + sqr(a.y) * sqr(v.y)) / sqr(v.z);
    return (-1.0f + sqrtExact(1.0f + squared)) * 0.5f;
}
__device__ __forceinline__ float smithG1(V3 V, V3 H, A2 a) { return dot(V, H) <= 0.0f ? 0.0f : rcpExact(1.0f + ggxLambda(V, a)); }
__device__ __forceinline__ float smithG2(V3 V, V3 L, V3 H, A2 a) {
    if (dot(V, H) <= 0.0f || dot(L, H) < 0.0f) return 0.0f;
    return rcpExact(1.0f + ggxLambda(V, a) + ggxLambda(L, a));
}
__device__ __forceinline__ float ggxNDF(V3 H, A2 a) {                             // ggx_brdf.h:71-75
    float ellipse = sqr(H.x) / sqr(a.x) + sqr(H.y) / sqr(a.y) + sqr(H.z);
    return rcpExact(KZ_PI_F * a.x * a.y * sqr(ellipse));
}
__device__ __forceinline__ float ggxVNDF(V3 V, V3 H, A2 a) {                      // ggx_brdf.h:80-91
    float VDotH = dot(V, H);
    if (VDotH <= 0.0f) return 0.0f;
    return ggxNDF(H, a) * smithG1(V, H, a) * VDotH / V.z;
}
__device__ __forceinline__ V3 sampleGGXVNDF(V3 V, A2 a, float rx, float ry) {     // ggx_brdf.h:96-120
    V3 Vh = normalized(mk(a.x * V.x, a.y * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? mk(-Vh.y, Vh.x, 0.0f) / sqrtExact(lensq) : mk(1.0f, 0.0f, 0.0f);
    V3 T2 = normalized(cross(Vh, T1));
    float r = sqrtExact(rx);
    float phi = 2.0f * KZ_PI_F * ry;
    float sinPhi, cosPhi; kzSinCos(phi, &sinPhi, &cosPhi);
    float t1 = r * cosPhi, t2 = r * sinPhi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtExact(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + sqrtExact(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalized(mk(a.x * Nh.x, a.y * Nh.y, fmaxf(1e-6f, Nh.z)));
}
__device__ __forceinline__ V3 evalGGXSmithBRDF(V3 V, V3 L, V3 f0, float roughness, float anisotropy) {   // ggx_brdf.h:151-170
    if (V.z * L.z < 0.0f) return mk(0.0f);
    A2 a = roughnessToAlpha(roughness, anisotropy);
    V3 H = normalized(V + L);
    float D = ggxNDF(H, a), G = smithG2(V, L, H, a);
    V3 F = schlickFresnel(f0, dot(V, H));
    float denom = 4.0f * fabsf(V.z) * fabsf(L.z);
    return (D * G) * F / denom;
}
__device__ __forceinline__ float lerpf(float t, float a, float b) { return (1.f - t) * a + t * b; }
__device__ __forceinline__ V3 lerp3(V3 a, V3 b, float t) { return (1.f - t) * a + t * b; }
__device__ __forceinline__ float schlickWeight(float x) { x = fminf(fmaxf(1.f - x, 0.f), 1.f); float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
__device__ __forceinline__ V3 reflectV(V3 wi, V3 n) { return 2 * dot(n, wi) * n - wi; }

// The terms of KazenStandard::eval that depend on the material alone (bsdf.cpp:1221-1232, 1244-1245): computed once per hit (the light
// sample and the BSDF sample of a bounce evaluate the same row) instead of once per eval() call; the same operations on the same values.
struct KissMat { V3 Cdlin, Cspec0, Csheen; };
__device__ __forceinline__ KissMat kissMat(const KzBSDF &m) {
    KissMat k;
    k.Cdlin = mk(m.baseColor[0], m.baseColor[1], m.baseColor[2]);
    const float Cdlum = luminance(k.Cdlin);
    const V3 Ctint = Cdlum > 0.f ? k.Cdlin / Cdlum : mk(1.f);
    const V3 Ctintmix = 0.08f * m.specular * lerp3(mk(1.f), Ctint, m.specularTint);
    k.Cspec0 = lerp3(Ctintmix, k.Cdlin, m.metallic);
    k.Csheen = lerp3(mk(1.f), Ctint, m.sheenTint);
    return k;
}
// KazenStandard::eval (bsdf.cpp:1215-1267) and ::pdf (bsdf.cpp:1269-1299) of one (V, L) pair. The integrator always asks for both (the
// light sample: eval + pdf for the MIS weight; sample(): pdf, then eval / pdf), and both are built from the same half vector, the same
// GGX normal distribution D(H, alpha) and the same Smith Lambda(V, alpha) of the specular and of the clearcoat lobe: each of those
// (1 normalisation, 2 x 3 divisions, 2 x division + square root: ~650 SIMD cycles of IEEE division, scripts/micro/op_cost.hip) is computed
// once here and used by both results, operation for operation what the two functions compute on their own. WANT selects the outputs.
template <bool WANT_F, bool WANT_PDF>
__device__ __forceinline__ void kissEvalPdf(const KzBSDF &m, const KissMat &k, V3 V, V3 L, float accRough, V3 &f, float &pdf) {
    f = mk(0.f); pdf = 0.f;
    if (V.z <= 0 || L.z <= 0) return;
    const V3 H = normalized(V + L);
    const float metallic = m.metallic;
    const float roughness = fminf(1.f, m.roughness + accRough);
    const float ccR = lerpf(m.clearcoatRoughness, .01f, .3f);
    const float VdotH = dot(V, H), LdotH = dot(L, H);
    const A2 aS = roughnessToAlpha(roughness, m.anisotropy);
    const A2 aC = roughnessToAlpha(ccR, m.anisotropy);                 // the coat of eval(): with the row's anisotropy (evalGGXSmithBRDF's argument)
    const float DS = ggxNDF(H, aS), lamVS = ggxLambda(V, aS);
    const float DC = ggxNDF(H, aC), lamVC = ggxLambda(V, aC);
    if (WANT_F) {
        const float FL = schlickWeight(L.z), FV = schlickWeight(V.z), FH = schlickWeight(LdotH);
        const float cosThetaD = VdotH;
        const float Lambert = (1.f - 0.5f * FL) * (1.f - 0.5f * FV);
        const float RR = 2.f * roughness * cosThetaD * cosThetaD;
        const float retro = RR * (FL + FV + FL * FV * (RR - 1.f));
        const V3 Fsheen = FH * m.sheen * k.Csheen;
        // evalGGXSmithBRDF (ggx_brdf.h:151-170) twice; V.z * L.z < 0 cannot happen here; smithG2 (ggx_brdf.h:58-66)
        const bool gZero = VdotH <= 0.0f || LdotH < 0.0f;
        const float denom = 4.0f * fabsf(V.z) * fabsf(L.z);
        const float GS = gZero ? 0.0f : rcpExact(1.0f + lamVS + ggxLambda(L, aS));
        const V3 specTerm = (DS * GS) * schlickFresnel(k.Cspec0, VdotH) / denom;
        const float GC = gZero ? 0.0f : rcpExact(1.0f + lamVC + ggxLambda(L, aC));
        const V3 coatTerm = 0.25f * m.clearcoat * ((DC * GC) * schlickFresnel(mk(0.04f), VdotH) / denom);
        f = ((1.f - metallic) * (k.Cdlin * KZ_INV_PI * (Lambert + retro) + Fsheen) + (specTerm + coatTerm)) * L.z;
    }
    if (WANT_PDF) {
        const float diffuse = (1.f - metallic) * 0.5f;
        const float GTR2 = rcpExact(1.f + m.clearcoat);
        const float jacobian = 4.0f * VdotH;
        // ggxVNDF (ggx_brdf.h:80-91) = D * G1 * VdotH / V.z, G1 = 1 / (1 + Lambda(V)) (0 when VdotH <= 0)
        const float specPdf = (VdotH <= 0.0f ? 0.0f : DS * (rcpExact(1.0f + lamVS)) * VdotH / V.z) / jacobian;
        float DCp = DC, lamVCp = lamVC;                                // pdf()'s coat has anisotropy 0: alpha * (1 +- 0) is the same alpha
        if (m.anisotropy != 0.f) { const A2 aP = roughnessToAlpha(ccR, 0.f); DCp = ggxNDF(H, aP); lamVCp = ggxLambda(V, aP); }
        const float coatPdf = (VdotH <= 0.0f ? 0.0f : DCp * (rcpExact(1.0f + lamVCp)) * VdotH / V.z) / jacobian;
        pdf = diffuse * KZ_INV_PI * L.z + (1.f - diffuse) * (GTR2 * specPdf + (1.f - GTR2) * coatPdf);
    }
}
__device__ __forceinline__ V3 kissEval(const KzBSDF &m, const KissMat &k, V3 V, V3 L, float accRough) {           // bsdf.cpp:1215-1267
    V3 f; float pdf; kissEvalPdf<true, false>(m, k, V, L, accRough, f, pdf); return f;
}
__device__ __forceinline__ float kissPdf(const KzBSDF &m, const KissMat &k, V3 wi, V3 wo, float accRough) {      // bsdf.cpp:1269-1299
    V3 f; float pdf; kissEvalPdf<false, true>(m, k, wi, wo, accRough, f, pdf); return pdf;
}
// fresnel (common.cpp:447-475) and refract (common.cpp:526-534)
__device__ __forceinline__ float fresnelIOR(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) { float t = etaI; etaI = etaT; etaT = t; cosThetaI = -cosThetaI; }
    float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = sqrtExact(1.0f - sinThetaTSqr);
    float Rs = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    float Rp = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}
__device__ __forceinline__ V3 refractV(V3 wi, V3 n, float eta) {
    float cosThetaI = dot(wi, n);
    if (cosThetaI < 0) eta = rcpExact(eta);
    float cosThetaT2 = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
    if (cosThetaT2 <= 0.0f) return mk(0.0f);
    float sign = cosThetaI >= 0.0f ? 1.0f : -1.0f;
    return n * (-cosThetaI * eta + sign * sqrtExact(cosThetaT2)) + wi * eta;
}
// ---- Beckmann helpers of roughconductor / roughplastic / roughdielectric (bsdf.cpp:721-750, warp.cpp:120-129, frame.h:63-68)
__device__ __forceinline__ float tanThetaV(V3 v) { float temp = 1 - v.z * v.z; if (temp <= 0.0f) return 0.0f; return sqrtExact(temp) / v.z; }
__device__ __forceinline__ float alphaOf(float x) { return x; }      // the row carries m_alpha = max(0.001, sqr(property)) (kz_scene_create; KzBSDF.alphaResolved)
__device__ __forceinline__ float evalBeckmann(V3 m, float alpha) {
    float temp = tanThetaV(m) / alpha, ct = m.z, ct2 = ct * ct;
    return kzExp(-temp * temp) / (KZ_PI_F * alpha * alpha * ct2 * ct2);
}
__device__ __forceinline__ float smithBeckmannG1(V3 v, V3 m, float alpha) {
    if (dot(v, m) * v.z <= 0.0f) return 0.0f;
    float tt = fabsf(tanThetaV(v));
    if (tt == 0.0f) return 1.0f;
    float a = rcpExact(alpha * tt);
    if (a >= 1.6f) return 1.0f;
    float aSqr = a * a;
    return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
__device__ __forceinline__ V3 squareToBeckmann(float sx, float sy, float alpha) {
    float phi = 2 * KZ_PI_F * sx;
    float theta = kzAtan(alpha * sqrtExact(kzLog(rcpExact(1 - sy))));
    float sinTheta, cosTheta, sinPhi, cosPhi; kzSinCos(theta, &sinTheta, &cosTheta); kzSinCos(phi, &sinPhi, &cosPhi);
    return mk(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
__device__ __forceinline__ float squareToBeckmannPdf(V3 m, float alpha) {
    float theta = kzAcos(m.z / norm(m));
    float ok = (fabsf(norm(m) - 1) < KZ_EPSILON && m.z >= 0) ? 1.f : 0.f;
    const float tt = kzTan(theta);                                                 // pow(x, 2) is the rounded product, pow(x, 3) the rounded cube
    return ok * kzExp(-(tt * tt) / (alpha * alpha)) / (KZ_PI_F * alpha * alpha * kzCube(kzCos(theta)));
}
__device__ __forceinline__ float fresnelDielectricT(float cosThetaI_, float eta, float &cosThetaT_) {        // common.cpp:492-518
    float scale = (cosThetaI_ > 0.f) ? rcpExact(eta) : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_), cosThetaT = sqrtExact(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
__device__ __forceinline__ V3 fresnelCond(float c, V3 eta, V3 k) {                                           // bsdf.cpp:709-717
    V3 tmp_f = eta * eta + k * k;
    V3 tmp = tmp_f * (c * c);
    V3 twoEtaC = 2.f * eta * c;
    V3 a = tmp - twoEtaC + mk(1.f), bq = tmp + twoEtaC + mk(1.f);
    V3 Rparl2 = mk(a.x / bq.x, a.y / bq.y, a.z / bq.z);
    V3 c2 = mk(c * c);
    V3 e = tmp_f - twoEtaC + c2, f = tmp_f + twoEtaC + c2;
    V3 Rperp2 = mk(e.x / f.x, e.y / f.y, e.z / f.z);
    return (Rparl2 + Rperp2) / 2.0f;
}
__device__ __forceinline__ float signf1(float v) { return (v > 0.f) ? 1.f : -1.f; }
// "ggx" (bsdf.cpp:629-689), "roughconductor" (:692-811), "roughplastic" (:814-943), "roughdielectric" (:947-1145).
// These models are real CALLS (each is big, and inlined at every use they cost the extended kernels hundreds of spilled registers), so what they need of a BSDF
// row travels BY VALUE, in registers (round 6): as `const KzBSDF &` the calls forced the caller's 128-byte copy of the row into scratch memory - for every hit of
// an extended kernel, whatever its model - and every field was read back from there.
struct RoughRow { int32_t type; float alpha, anisotropy, intIOR, extIOR; float albedo[3], condEta[3], condK[3]; };
__device__ __forceinline__ RoughRow roughRowOf(const KzBSDF &b) {
    RoughRow r;
    r.type = b.type; r.alpha = b.alpha; r.anisotropy = b.anisotropy; r.intIOR = b.intIOR; r.extIOR = b.extIOR;
    r.albedo[0] = b.albedo[0]; r.albedo[1] = b.albedo[1]; r.albedo[2] = b.albedo[2];
    r.condEta[0] = b.condEta[0]; r.condEta[1] = b.condEta[1]; r.condEta[2] = b.condEta[2];
    r.condK[0] = b.condK[0]; r.condK[1] = b.condK[1]; r.condK[2] = b.condK[2];
    return r;
}
struct RoughEvalPdfOut { V3 f; float pdf; };
struct RoughSampleOut { V3 w, wo; float etaScale, pdfOut; int32_t alive; };
__device__ V3 roughEval(const RoughRow m, V3 wi, V3 wo) {
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) {
        if (wi.z == 0) return mk(0.f);
        const float alpha = alphaOf(m.alpha), mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
        const bool refl = wi.z * wo.z > 0.f;
        const float eta = wi.z > 0.f ? mEta : mInvEta;
        V3 wm = refl ? normalized(wi + wo) : normalized(wi + wo * eta);
        wm = wm * signf1(wm.z);
        float ct; const float F = fresnelDielectricT(dot(wi, wm), mEta, ct);
        const float D = evalBeckmann(wm, alpha);
        const float G = smithBeckmannG1(wo, wm, alpha) * smithBeckmannG1(wi, wm, alpha);
        if (refl) return mk((F * G * D) / (4.f * fabsf(wi.z)));
        const float denom = dot(wi, wm) + eta * dot(wo, wm);
        const float value = ((1 - F) * D * G * eta * eta * dot(wi, wm) * dot(wo, wm)) / (wi.z * sqr(denom));
        return mk(fabsf(value));
    }
    if (wi.z <= 0 || wo.z <= 0) return mk(0.f);
    if (m.type == KZ_BSDF_GGX) return evalGGXSmithBRDF(wi, wo, mk(m.albedo[0], m.albedo[1], m.albedo[2]), m.alpha, m.anisotropy) * wo.z;
    const float alpha = alphaOf(m.alpha);
    const V3 wh = normalized(wi + wo);
    const float D = evalBeckmann(wh, alpha);
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) {
        const V3 F = fresnelCond(dot(wh, wo), mk(m.condEta[0], m.condEta[1], m.condEta[2]), mk(m.condK[0], m.condK[1], m.condK[2]));
        const float G = smithBeckmannG1(wi, wh, alpha) * smithBeckmannG1(wo, wh, alpha);
        return D * F * G / (4.f * wi.z);
    }
    const V3 kd = mk(m.albedo[0], m.albedo[1], m.albedo[2]);                                                 // roughplastic
    const float ks = 1 - maxCoeff(kd);
    const float F = fresnelIOR(dot(wh, wo), m.extIOR, m.intIOR);
    const float G = smithBeckmannG1(wo, wh, alpha) * smithBeckmannG1(wi, wh, alpha);
    return kd * KZ_INV_PI * wo.z + mk(ks * (D * F * G) / (4.f * wi.z));
}
__device__ float roughPdf(const RoughRow m, V3 wi, V3 wo) {
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) {
        const float mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
        const bool refl = wi.z * wo.z > 0.f;
        const float eta = wi.z > 0.f ? mEta : mInvEta;
        V3 wm; float dwm_dwo;
        if (refl) { wm = normalized(wi + wo); dwm_dwo = rcpExact(4.0f * dot(wo, wm)); }
        else { wm = normalized(wi + wo * eta); const float sd = dot(wi, wm) + eta * dot(wo, wm); dwm_dwo = (eta * eta * dot(wo, wm)) / (sd * sd); }
        wm = wm * signf1(wm.z);
        float ct; const float F = fresnelDielectricT(dot(wi, wm), mEta, ct);
        float prob = evalBeckmann(wm, alphaOf(m.alpha)) * wm.z;
        prob *= refl ? F : (1 - F);
        return fabsf(prob * dwm_dwo);
    }
    if (wi.z <= 0 || wo.z <= 0) return 0.f;
    const V3 wh = normalized(wi + wo);
    if (m.type == KZ_BSDF_GGX) return ggxVNDF(wi, wh, roughnessToAlpha(m.alpha, m.anisotropy)) / (4.0f * dot(wi, wh));
    const float D = evalBeckmann(wh, alphaOf(m.alpha));
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) return D * wh.z * (rcpExact(4.f * dot(wh, wo)));
    const float ks = 1 - fmaxf(m.albedo[0], fmaxf(m.albedo[1], m.albedo[2]));
    return ks * D * wh.z * (rcpExact(4.f * fabsf(dot(wh, wo)))) + (1 - ks) * wo.z * KZ_INV_PI;
}
// roughEval and roughPdf of one direction pair in ONE evaluation: the two functions of the reference form the same half vector and the same Beckmann D(wh, alpha)
// (bsdf.cpp:756-780, :870-905) - same arguments, same bits - so they are formed once here (D is an exp through kz_crmath.h: a call and ~40 double operations).
// Returns exactly (roughEval(m, wi, wo), roughPdf(m, wi, wo)).
__device__ __forceinline__ void roughEvalPdfImpl(const RoughRow &m, V3 wi, V3 wo, V3 &f, float &pdf) {
    if (m.type == KZ_BSDF_GGX) { f = roughEval(m, wi, wo); pdf = roughPdf(m, wi, wo); return; }
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) {                                     // (bsdf.cpp:985-1043: the same wm, Fresnel term and D in eval and pdf)
        const float alpha = alphaOf(m.alpha), mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
        const bool refl = wi.z * wo.z > 0.f;
        const float eta = wi.z > 0.f ? mEta : mInvEta;
        V3 wm = refl ? normalized(wi + wo) : normalized(wi + wo * eta);
        float dwm_dwo;                                                            // (pdf forms it from the half vector BEFORE the flip to the upper hemisphere)
        if (refl) dwm_dwo = rcpExact(4.0f * dot(wo, wm));
        else { const float sd = dot(wi, wm) + eta * dot(wo, wm); dwm_dwo = (eta * eta * dot(wo, wm)) / (sd * sd); }
        wm = wm * signf1(wm.z);
        float ct; const float F = fresnelDielectricT(dot(wi, wm), mEta, ct);
        const float D = evalBeckmann(wm, alpha);
        float prob = D * wm.z;
        prob *= refl ? F : (1 - F);
        pdf = fabsf(prob * dwm_dwo);
        if (wi.z == 0) { f = mk(0.f); return; }
        const float G = smithBeckmannG1(wo, wm, alpha) * smithBeckmannG1(wi, wm, alpha);
        if (refl) { f = mk((F * G * D) / (4.f * fabsf(wi.z))); return; }
        const float denom = dot(wi, wm) + eta * dot(wo, wm);
        const float value = ((1 - F) * D * G * eta * eta * dot(wi, wm) * dot(wo, wm)) / (wi.z * sqr(denom));
        f = mk(fabsf(value));
        return;
    }
    if (wi.z <= 0 || wo.z <= 0) { f = mk(0.f); pdf = 0.f; return; }
    const float alpha = alphaOf(m.alpha);
    const V3 wh = normalized(wi + wo);
    const float D = evalBeckmann(wh, alpha);
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) {
        const V3 F = fresnelCond(dot(wh, wo), mk(m.condEta[0], m.condEta[1], m.condEta[2]), mk(m.condK[0], m.condK[1], m.condK[2]));
        const float G = smithBeckmannG1(wi, wh, alpha) * smithBeckmannG1(wo, wh, alpha);
        f = D * F * G / (4.f * wi.z);
        pdf = D * wh.z * (rcpExact(4.f * dot(wh, wo)));
        return;
    }
    const V3 kd = mk(m.albedo[0], m.albedo[1], m.albedo[2]);                                                 // roughplastic
    const float ks = 1 - maxCoeff(kd);
    const float F = fresnelIOR(dot(wh, wo), m.extIOR, m.intIOR);
    const float G = smithBeckmannG1(wo, wh, alpha) * smithBeckmannG1(wi, wh, alpha);
    f = kd * KZ_INV_PI * wo.z + mk(ks * (D * F * G) / (4.f * wi.z));
    pdf = ks * D * wh.z * (rcpExact(4.f * fabsf(dot(wh, wo)))) + (1 - ks) * wo.z * KZ_INV_PI;
}
__device__ RoughEvalPdfOut roughEvalPdf(const RoughRow m, V3 wi, V3 wo) { RoughEvalPdfOut o; roughEvalPdfImpl(m, wi, wo, o.f, o.pdf); return o; }
// pdfOut: roughPdf(m, wi, wo) at the sampled direction where sample() forms it on the way (roughconductor, roughplastic), else -1
__device__ __forceinline__ V3 roughSampleImpl(const RoughRow &m, V3 wi, float s1, float s2x, float s2y, V3 &wo, bool &alive, float &etaScale, float &pdfOut) {
    alive = true; pdfOut = -1.f;
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) {
        const float mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
        const float alpha = alphaOf(m.alpha) * (1.2f - 0.2f * sqrtExact(fabsf(wi.z)));
        const V3 wm = squareToBeckmann(s2x, s2y, alpha);
        const float pdf = squareToBeckmannPdf(wm, alpha);
        if (pdf == 0.f) return mk(0.f);
        float cosThetaT; const float F = fresnelDielectricT(dot(wi, wm), mEta, cosThetaT);
        if (!(s1 > F)) {
            wo = reflectV(wi, wm);
            if (wi.z * wo.z <= 0) return mk(0.f);
        } else {
            if (cosThetaT == 0) return mk(0.f);
            float e = mEta; if (cosThetaT < 0) e = rcpExact(e);
            wo = wm * (dot(wi, wm) * e + cosThetaT) - wi * e;
            etaScale = cosThetaT < 0.f ? mEta : mInvEta;
            if (wi.z * wo.z >= 0) return mk(0.f);
        }
        const float D = evalBeckmann(wm, alpha);
        const float G = smithBeckmannG1(wo, wm, alpha) * smithBeckmannG1(wi, wm, alpha);
        return mk(fabsf(D * G * dot(wi, wm) / (pdf * wi.z)));
    }
    if (wi.z <= 0) { alive = false; return mk(0.f); }
    if (m.type == KZ_BSDF_GGX) {
        const A2 alpha = roughnessToAlpha(m.alpha, m.anisotropy);
        const V3 H = sampleGGXVNDF(wi, alpha, s2x, s2y);
        wo = reflectV(wi, H);                                                                                 // not normalised (ggx_brdf.h:189)
        const float pdf = ggxVNDF(wi, H, alpha) / (4.0f * dot(wi, H));
        const V3 color = evalGGXSmithBRDF(wi, wo, mk(m.albedo[0], m.albedo[1], m.albedo[2]), m.alpha, m.anisotropy);
        if (wo.z <= 0) return mk(0.f);
        return color * wo.z / pdf;
    }
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) {
        const V3 wh = squareToBeckmann(s2x, s2y, alphaOf(m.alpha));
        wo = normalized(reflectV(wi, wh));
    } else {
        const float ks = 1 - fmaxf(m.albedo[0], fmaxf(m.albedo[1], m.albedo[2]));
        if (s1 < ks) { const V3 wh = squareToBeckmann(s2x, s2y, alphaOf(m.alpha)); wo = normalized((2.f * dot(wh, wi) * wh) - wi); }
        else wo = squareToCosineHemisphere(s2x, s2y);
    }
    if (wo.z <= 0) return mk(0.f);
    V3 f; float pdf; roughEvalPdfImpl(m, wi, wo, f, pdf);
    pdfOut = pdf;
    return f / pdf;
}
__device__ RoughSampleOut roughSample(const RoughRow m, V3 wi, float s1, float s2x, float s2y, float etaScaleIn) {
    RoughSampleOut o; bool alive;
    o.wo = mk(0.f, 0.f, 1.f); o.etaScale = etaScaleIn;
    o.w = roughSampleImpl(m, wi, s1, s2x, s2y, o.wo, alive, o.etaScale, o.pdfOut);
    o.alive = alive ? 1 : 0;
    return o;
}

// returns the sample weight; wo is the sampled direction; alive=false when the path contributes nothing further;
// discrete = bRec.measure == EDiscrete; etaScale = bRec.eta
// EXT (KzParams::bsdfExt) says what the SCENE contains beyond constant diffuse / kazenstandard rows - a bit mask, so that a kernel is compiled for what a
// scene needs and nothing else (round 6; it was one flag before):
//   KZ_X_MODELS  rows of other models: mirror, dielectric, ggx, roughconductor, roughplastic, roughdielectric
//   KZ_X_TEX     texture-backed parameters (texture programs: image / ramp / blend)
//   KZ_X_NMAP    normalmap rows (a texture lookup + the perturbed frame; implies KZ_X_TEX)
// EXT = 0 compiles only diffuse + kazenstandard (the BASELINE configs): the shade kernel stays at 123 VGPRs instead of ~168.
#define KZ_X_MODELS 1
#define KZ_X_TEX 2
#define KZ_X_NMAP 4
#define KZ_X_ALL 7
template <int EXT>
// pdfOut: BSDF::pdf at the sampled direction when the model computes it on the way (diffuse, kiss) — the integrator's own
// pdf(bRec) call right after sample() (integrator.cpp:314) is the same function of the same arguments, so it is reused, not
// recomputed; pdfOut < 0 means "not provided".
__device__ __forceinline__ V3 bsdfSample(const KzBSDF &m, const KissMat &km, V3 wi, float accRough, float s1, float s2x, float s2y, V3 &wo, bool &alive, bool &discrete, float &etaScale, float &pdfOut) {
    wo = mk(0.f, 0.f, 1.f); discrete = false; etaScale = 1.0f; pdfOut = -1.f;
    if ((EXT & KZ_X_MODELS) && m.type == KZ_BSDF_DIELECTRIC) {                                            // bsdf.cpp:119-143 (no back-side test)
        alive = true; discrete = true;
        if (s1 < fresnelIOR(wi.z, m.extIOR, m.intIOR)) { wo = mk(-wi.x, -wi.y, wi.z); return mk(1.0f); }
        V3 n = mk(0.0f, 0.0f, 1.0f);
        float factor = m.intIOR / m.extIOR;
        if (wi.z < 0.f) { factor = m.extIOR / m.intIOR; n.z = -1.0f; }
        wo = refractV(-wi, n, factor);
        etaScale = m.intIOR / m.extIOR;
        return mk(1.0f);
    }
    if ((EXT & KZ_X_MODELS) && m.type >= KZ_BSDF_GGX) {                                   // (pdfOut -1 for ggx / roughdielectric: the caller evaluates pdf())
        const RoughSampleOut o = roughSample(roughRowOf(m), wi, s1, s2x, s2y, etaScale);
        wo = o.wo; alive = o.alive != 0; etaScale = o.etaScale; pdfOut = o.pdfOut;
        return o.w;
    }
    if (wi.z <= 0) { alive = false; return mk(0.f); }                              // bsdf.cpp:60-61, :176-177, :1302-1303
    alive = true;
    if ((EXT & KZ_X_MODELS) && m.type == KZ_BSDF_MIRROR) { wo = mk(-wi.x, -wi.y, wi.z); discrete = true; return mk(1.0f); }   // bsdf.cpp:175-191
    if (m.type == KZ_BSDF_DIFFUSE) {                                               // bsdf.cpp:59-75
        wo = squareToCosineHemisphere(s2x, s2y);
        pdfOut = wo.z <= 0 ? 0.f : KZ_INV_PI * wo.z;                              // Diffuse::pdf (bsdf.cpp:40-56), wi.z > 0 here
        return mk(m.albedo[0], m.albedo[1], m.albedo[2]);
    }
    float diffuse = (1.f - m.metallic) * 0.5f;                                     // bsdf.cpp:1301-1371
    if (s1 < diffuse) wo = squareToCosineHemisphere(s2x, s2y);
    else {
        float sample = (s1 - diffuse) / (1.f - diffuse);
        float GTR2 = rcpExact(1.f + m.clearcoat);
        A2 alpha = (sample < GTR2) ? roughnessToAlpha(m.roughness, m.anisotropy)   // H7: un-regularised roughness, anisotropy 0 for the coat
                                   : roughnessToAlpha(lerpf(m.clearcoatRoughness, 0.01f, .3f), 0.f);
        V3 H = sampleGGXVNDF(wi, alpha, s2x, s2y);                                 // wi.z > 0 here: never flipped
        wo = normalized(reflectV(wi, H));
    }
    bool invalid = isnan(wo.x) || isnan(wo.y) || isnan(wo.z);
    V3 f; float pdf;
    kissEvalPdf<true, true>(m, km, wi, wo, accRough, f, pdf);
    if (wo.z <= 0 || pdf <= KZ_EPSILON || invalid) return mk(0.f);
    pdfOut = pdf;
    return f / pdf;
}
template <int EXT>
__device__ __forceinline__ V3 bsdfEval(const KzBSDF &m, const KissMat &km, V3 wi, V3 wo, float accRough) {
    if (m.type == KZ_BSDF_DIFFUSE) {                                               // bsdf.cpp:27-37 (measure is ESolidAngle at every call site)
        if (wi.z <= 0 || wo.z <= 0) return mk(0.f);
        return mk(m.albedo[0], m.albedo[1], m.albedo[2]) * KZ_INV_PI * wo.z;
    }
    if ((EXT & KZ_X_MODELS) && m.type >= KZ_BSDF_GGX) return roughEval(roughRowOf(m), wi, wo);
    if ((EXT & KZ_X_MODELS) && m.type != KZ_BSDF_KAZENSTANDARD) return mk(0.f);                           // discrete BRDFs evaluate to zero
    return kissEval(m, km, wi, wo, accRough);
}
template <int EXT>
__device__ __forceinline__ float bsdfPdf(const KzBSDF &m, const KissMat &km, V3 wi, V3 wo, float accRough) {
    if (m.type == KZ_BSDF_DIFFUSE) { if (wi.z <= 0 || wo.z <= 0) return 0.f; return KZ_INV_PI * wo.z; }   // bsdf.cpp:40-56
    if ((EXT & KZ_X_MODELS) && m.type >= KZ_BSDF_GGX) return roughPdf(roughRowOf(m), wi, wo);
    if ((EXT & KZ_X_MODELS) && m.type != KZ_BSDF_KAZENSTANDARD) return 0.f;
    return kissPdf(m, km, wi, wo, accRough);
}
// eval and pdf of one direction pair (the light sample of a bounce asks for both): the kiss row shares its half-vector terms between the two
template <int EXT>
__device__ __forceinline__ void bsdfEvalPdf(const KzBSDF &m, const KissMat &km, V3 wi, V3 wo, float accRough, V3 &f, float &pdf) {
    if (m.type == KZ_BSDF_DIFFUSE) {                                               // bsdf.cpp:27-37, 40-56
        const bool up = wi.z > 0 && wo.z > 0;
        f = up ? mk(m.albedo[0], m.albedo[1], m.albedo[2]) * KZ_INV_PI * wo.z : mk(0.f);
        pdf = up ? KZ_INV_PI * wo.z : 0.f;
    } else if ((EXT & KZ_X_MODELS) && m.type >= KZ_BSDF_GGX) { const RoughEvalPdfOut o = roughEvalPdf(roughRowOf(m), wi, wo); f = o.f; pdf = o.pdf; }
    else if ((EXT & KZ_X_MODELS) && m.type != KZ_BSDF_KAZENSTANDARD) { f = mk(0.f); pdf = 0.f; }       // discrete BRDFs evaluate to zero
    else kissEvalPdf<true, true>(m, km, wi, wo, accRough, f, pdf);
}


// ============================================================================================
// Texture<Color3f> trees (texture.cpp:10-270) and the NormalMap wrapper (bsdf.cpp:281-417).
// Only reachable from the EXT kernel variants (KzParams::bsdfExt).
// ============================================================================================
__device__ __forceinline__ int wrapPeriodic(int i, int n) {
    // a power-of-two size: the mask is the same residue, for negative i too. The runtime modulo is ~24 instructions, six of them per bilinear lookup; the empty asm keeps it
    // behind a real branch (left alone the compiler evaluates both forms and selects: no gain) that waves whose lanes all look up power-of-two images skip.
    if ((n & (n - 1)) == 0) return i & (n - 1);
    asm volatile("" : "+v"(i));
    i %= n; return i < 0 ? i + n : i;
}
__device__ __forceinline__ float srgbToLinear(float v) {                           // Color3f::toLinearRGB, common.cpp:368-382
    return v <= 0.04045f ? v * (1.0f / 12.92f) : kzPow((v + 0.055f) * (1.0f / 1.055f), 2.4f);
}
// Taps and weights of one axis of an image lookup at continuous texel coordinate x (texel centres at i + 0.5, so x = s * res - 0.5): the two taps of the
// bilinear filter (N = 2) or the four of the cubic B-spline (N = 4; KzTexture.filter). The 2-tap form is the arithmetic the lookup had before the filter
// became a field: (1 - f) * a + f * b. N is a compile-time constant: the weights stay in registers (a run-time tap count put them in scratch memory).
template <int N> struct KzTaps { int first; float w[N]; };
template <int N> __device__ __forceinline__ KzTaps<N> filterTaps(float x) {
    KzTaps<N> t;
    const float x0 = floorf(x), f = x - x0;
    if (N == 4) {
        const float omf = 1.0f - f, f2 = f * f, f3 = f2 * f;
        t.first = (int)x0 - 1;
        t.w[0] = omf * omf * omf * (1.0f / 6.0f);
        t.w[1] = (3.0f * f3 - 6.0f * f2 + 4.0f) * (1.0f / 6.0f);
        t.w[N - 2] = (-3.0f * f3 + 3.0f * f2 + 3.0f * f + 1.0f) * (1.0f / 6.0f);
        t.w[N - 1] = f3 * (1.0f / 6.0f);
    } else { t.first = (int)x0; t.w[0] = 1.0f - f; t.w[1] = f; }
    return t;
}
// The three colour channels of texel (x, y): one address for the texel, the channels behind it (a missing channel reads 0: TextureOpt::fill)
__device__ __forceinline__ void texel3(const KzImageRow &im, const uint8_t *base, int x, int y, float &t0, float &t1, float &t2) {
    const size_t i = ((size_t)y * (size_t)im.width + (size_t)x) * (size_t)im.channels;
    if (im.format == KZ_PIXEL_F32) {
        const float *p = reinterpret_cast<const float *>(base) + i;
        t0 = p[0]; t1 = im.channels > 1 ? p[1] : 0.0f; t2 = im.channels > 2 ? p[2] : 0.0f;
    } else {
        const uint8_t *p = base + i;
        t0 = (float)p[0] * (1.0f / 255.0f); t1 = im.channels > 1 ? (float)p[1] * (1.0f / 255.0f) : 0.0f; t2 = im.channels > 2 ? (float)p[2] * (1.0f / 255.0f) : 0.0f;
    }
}
// The filtered rgb of an image at continuous texel coordinates (x, y): per channel, the taps of a row left to right, then the rows top to bottom (the oracle adds in
// the same order). Round 6: the N x N taps are addressed ONCE for the three channels (it was once per channel: three times the wraps and the address arithmetic), and
// POW2 - every image of the scene has power-of-two sides, known at kz_scene_create - wraps with a mask, no modulo in the code at all (profiles/r04i_ext: the run-time
// modulo was ~24 instructions, six per bilinear lookup).
template <int N, bool CLAMP_Y, bool POW2>
__device__ __forceinline__ V3 filteredRgb(const KzImageRow &im, const uint8_t *base, float x, float y) {
    const KzTaps<N> tx = filterTaps<N>(x), ty = filterTaps<N>(y);
    int xs[N], ys[N];
#pragma unroll
    for (int i = 0; i < N; ++i) xs[i] = POW2 ? ((tx.first + i) & (im.width - 1)) : wrapPeriodic(tx.first + i, im.width);
#pragma unroll
    for (int j = 0; j < N; ++j) ys[j] = CLAMP_Y ? min(max(ty.first + j, 0), im.height - 1) : (POW2 ? ((ty.first + j) & (im.height - 1)) : wrapPeriodic(ty.first + j, im.height));
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float row0 = 0.0f, row1 = 0.0f, row2 = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float t0, t1, t2;
            texel3(im, base, xs[i], ys[j], t0, t1, t2);
            if (i == 0) { row0 = tx.w[0] * t0; row1 = tx.w[0] * t1; row2 = tx.w[0] * t2; }
            else { row0 = row0 + tx.w[i] * t0; row1 = row1 + tx.w[i] * t1; row2 = row2 + tx.w[i] * t2; }
        }
        if (j == 0) { r0 = ty.w[0] * row0; r1 = ty.w[0] * row1; r2 = ty.w[0] * row2; }
        else { r0 = r0 + ty.w[j] * row0; r1 = r1 + ty.w[j] * row1; r2 = r2 + ty.w[j] * row2; }
    }
    return mk(r0, r1, r2);
}
// the 4 x 4 lookup is a CALL: inlined next to the 2 x 2 one it cost the path kernels registers (the lean megakernel went from 139 to 248 VGPRs) for a
// filter that is off by default
template <bool CLAMP_Y>
__device__ __attribute__((noinline)) V3 filteredRgbCubic(KzImageRow im, const uint8_t *base, float x, float y) { return filteredRgb<4, CLAMP_Y, false>(im, base, x, y); }
// ImageTexture::eval (texture.cpp:46-64): s = u*scale, t = (1-v)*scale, periodic wrap; the filter is KzTexture.filter (kazen_mi355x.h).
// The IMAGE op of a texture program carries its image row (kz_scene_create packs it: f1 = texel offset / 16, f2 = (width - 1) | (height - 1) << 16,
// b = srgb | filter << 1 | channels << 8 | format << 16): one load fewer on the dependent chain program -> op -> image row -> texels.
__device__ __forceinline__ V3 imageLookupOp(const uint8_t *texels, const KzTexOp &op, uint32_t pow2, float u, float v) {
    KzImageRow im;
    const uint32_t wh = __float_as_uint(op.f2);
    im.offset = (uint64_t)__float_as_uint(op.f1) << 4; im.width = (int32_t)(wh & 0xffffu) + 1; im.height = (int32_t)(wh >> 16) + 1;
    im.channels = (int32_t)((op.b >> 8) & 0xffu); im.format = (int32_t)((op.b >> 16) & 0xffu);
    const uint8_t *base = texels + im.offset;
    const uint32_t srgb = op.b & 1u; const int filter = (int)((op.b >> 1) & 0x7fu);
    const float scale = op.f0;
    const float s = u * scale, t = (1.0f - v) * scale;
    const float x = s * (float)im.width - 0.5f, y = t * (float)im.height - 0.5f;
    if (!(fabsf(x) < 1.0e9f) || !(fabsf(y) < 1.0e9f)) return mk(0.f);             // non-finite uv: defined as black
    V3 r;
    if (filter == KZ_TEXFILTER_BICUBIC) r = filteredRgbCubic<false>(im, base, x, y);
    else if (pow2) r = filteredRgb<2, false, true>(im, base, x, y);               // (uniform over the launch: a scalar branch)
    else r = filteredRgb<2, false, false>(im, base, x, y);
    if (srgb) r = mk(srgbToLinear(r.x), srgbToLinear(r.y), srgbToLinear(r.z));
    return r;
}
// ImageTexture::eval(Vector3f) (texture.cpp:66-80): the environment lookup, as include/kazen_mi355x.h declares it (y-up latitude-longitude
// map, s periodic, t clamped; no scale, no colour-space conversion; the nested texture's filter)
__device__ V3 envLookup(const KzDevTables &T, uint32_t image, int filter, V3 d) {
    const KzImageRow im = T.images[image];
    const uint8_t *base = T.texels + im.offset;
    float s = kzAtan2(-d.x, d.z) / (2.0f * KZ_PI_F) + 0.5f;
    float t = 0.5f - kzAtan2(d.y, kzHypot(d.z, -d.x)) / KZ_PI_F;
    if (isnan(s)) s = 0.0f;
    if (isnan(t)) t = 0.0f;
    const float x = s * (float)im.width - 0.5f, y = t * (float)im.height - 0.5f;
    return filter == KZ_TEXFILTER_BICUBIC ? filteredRgbCubic<true>(im, base, x, y) : filteredRgb<2, true, false>(im, base, x, y);
}
// Scene::getBackgroundColor (scene.cpp:54-79) -> BackgroundTexture::eval(Vector3f) (texture.cpp:121-126); the caller has checked bgPresent
__device__ __forceinline__ V3 backgroundRadiance(const KzParams &P, const KzDevTables &T, V3 d) {
    if (isnan(d.x) || isnan(d.y) || isnan(d.z)) return mk(0.f);
    if (P.bgImage >= 0) return P.bgIntensity * envLookup(T, (uint32_t)P.bgImage, P.bgFilter, d);
    return mk(P.bgRadiance[0], P.bgRadiance[1], P.bgRadiance[2]);
}
__device__ __forceinline__ float clampRef(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // common.h:237-243
// texId is 1-based (KzBSDF::*Tex); the postfix program was flattened by kz_scene_create
// (a real call - the tree walk is big - so it takes the four tables it reads BY VALUE: a `const KzDevTables &` would force the caller's copy into scratch)
struct KzTexTables { const KzTexProg *texProgs; const KzTexOp *texOps; const uint8_t *texels; uint32_t pow2; };
__device__ V3 texEvalT(const KzTexTables T, int32_t texId, float u, float v);
__device__ __forceinline__ V3 texEval(const KzDevTables &T, int32_t texId, float u, float v) {
    const KzTexTables tt = {T.texProgs, T.texOps, T.texels, T.texPow2};
    return texEvalT(tt, texId, u, v);
}
__device__ __forceinline__ V3 texRamp(const KzTexOp &op, V3 c) {                 // texture.cpp:162-172
    return mk(op.f0 + (op.f1 - op.f0) * clampRef(c.x, 0.0f, 1.0f), op.f0 + (op.f1 - op.f0) * clampRef(c.y, 0.0f, 1.0f), op.f0 + (op.f1 - op.f0) * clampRef(c.z, 0.0f, 1.0f));
}
__device__ __forceinline__ V3 texBlend(const KzTexOp &op, V3 mask, V3 in1, V3 in2) {   // texture.cpp:211-237
    if (op.a == KZ_BLEND_MIX) return mk(lerpf(mask.x, in1.x, in2.x), lerpf(mask.x, in1.y, in2.y), lerpf(mask.x, in1.z, in2.z));
    if (op.a == KZ_BLEND_MULTIPLY) return in1 * in2;
    return mk(0.f);
}
// The operand stack of a program lives in REGISTERS when the program needs at most four entries (kz_scene_create records the depth in the high half of
// KzTexProg::count; every tree of the reference's scenes and of the test scenes does): a stack that shifts - the top is always s0 - so that every access is to a
// named register. A dynamically indexed array (the deeper programs: up to KZ_TEX_MAX_DEPTH) lives in scratch memory, a store and a dependent load per operand.
__device__ V3 texEvalT(const KzTexTables T, int32_t texId, float u, float v) {
    const KzTexProg pr = T.texProgs[texId - 1];
    const uint32_t count = pr.count & 0xffffu, depth = pr.count >> 16;
    if (depth <= 4u) {
        V3 s0 = mk(0.f), s1 = mk(0.f), s2 = mk(0.f), s3 = mk(0.f);
        for (uint32_t i = 0; i < count; ++i) {
            const KzTexOp op = T.texOps[pr.start + i];
            if (op.op == KZ_TOP_CONST) { s3 = s2; s2 = s1; s1 = s0; s0 = mk(op.f0, op.f1, op.f2); }
            else if (op.op == KZ_TOP_IMAGE) { const V3 c = imageLookupOp(T.texels, op, T.pow2, u, v); s3 = s2; s2 = s1; s1 = s0; s0 = c; }
            else if (op.op == KZ_TOP_RAMP) s0 = texRamp(op, s0);
            else { s0 = texBlend(op, s2, s1, s0); s1 = s3; s2 = mk(0.f); s3 = mk(0.f); }      // mask, input1, input2 = the three top entries
        }
        return s0;
    }
    V3 st[KZ_TEX_MAX_DEPTH];
    int sp = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const KzTexOp op = T.texOps[pr.start + i];
        if (op.op == KZ_TOP_CONST) st[sp++] = mk(op.f0, op.f1, op.f2);
        else if (op.op == KZ_TOP_IMAGE) st[sp++] = imageLookupOp(T.texels, op, T.pow2, u, v);
        else if (op.op == KZ_TOP_RAMP) st[sp - 1] = texRamp(op, st[sp - 1]);
        else { const V3 r = texBlend(op, st[sp - 3], st[sp - 2], st[sp - 1]); sp -= 3; st[sp++] = r; }
    }
    return st[0];
}
// fold the texture-backed parameters of a (local copy of a) BSDF row at this hit's uv: bRec.uv is its.uv at every call
// site of the integrator (integrator.cpp:285,305), so eval / pdf / sample / regularize of one hit all see these values
__device__ __forceinline__ void resolveTextures(const KzDevTables &T, KzBSDF &b, float u, float v) {
    if (b.albedoTex) {
        const V3 c = texEval(T, b.albedoTex, u, v);
        if (b.type == KZ_BSDF_KAZENSTANDARD) { b.baseColor[0] = c.x; b.baseColor[1] = c.y; b.baseColor[2] = c.z; }
        else { b.albedo[0] = c.x; b.albedo[1] = c.y; b.albedo[2] = c.z; }
    }
    if (b.roughnessTex) b.roughness = texEval(T, b.roughnessTex, u, v).x;
    if (b.metallicTex) b.metallic = texEval(T, b.metallicTex, u, v).x;
}

// NormalMap (bsdf.cpp:281-417). n = 2*rgb-1 in the local shading frame (not normalised for the dot(n, wi) tests);
// pf = getFrame(its, n.normalized(), wi) (bsdf.cpp:365-374).
struct NMap { bool on; V3 n; Frame3 pf; KissMat km; };        // + the material-only terms of a kiss row (kissMat), computed once per hit
__device__ __forceinline__ void nmapSetup(const KzDevTables &T, const KzBSDF &outer, const Its &its, NMap &nm) {
    const V3 rgb = texEval(T, outer.normalTex, its.uvx, its.uvy);
    nm.on = true;
    nm.n = mk(2 * rgb.x - 1, 2 * rgb.y - 1, 2 * rgb.z - 1);
    nm.pf.n = normalized(toWorld(its.sh, normalized(nm.n)));
    nm.pf.s = normalized(its.dpdu - nm.pf.n * dot(nm.pf.n, its.dpdu));
    nm.pf.t = normalized(cross(nm.pf.n, nm.pf.s));
}
// One hit's BSDF: the row (normalmap unwrapped to its nested row, textures folded) + the perturbed frame
template <int EXT>
__device__ __forceinline__ void surfaceSetup(const KzDevTables &T, const Its &its, KzBSDF &b, NMap &nm) {
    nm.on = false;
    if (EXT & KZ_X_NMAP) { if (b.type == KZ_BSDF_NORMALMAP) { nmapSetup(T, b, its, nm); b = T.bsdfs[b.nested]; } }
    if (EXT & KZ_X_TEX) resolveTextures(T, b, its.uvx, its.uvy);
    if (b.type == KZ_BSDF_KAZENSTANDARD) nm.km = kissMat(b);
    else { nm.km.Cdlin = mk(0.f); nm.km.Cspec0 = mk(0.f); nm.km.Csheen = mk(0.f); }
}
// solid: bRec.measure == ESolidAngle. Only Diffuse checks it (bsdf.cpp:30,43,213,225); it is lost when NormalMap::sample goes
// through the perturbed record, whose measure is never copied back (bsdf.cpp:348-362).
template <int EXT>
__device__ __forceinline__ V3 surfEval(const KzBSDF &b, const NMap &nm, const Its &its, V3 wi, V3 wo, float accRough) {
    if (!(EXT & KZ_X_NMAP)) return bsdfEval<EXT>(b, nm.km, wi, wo, accRough);
    // (ONE call site of the model code for the three cases - no normal map | the map's fall-back to the unperturbed record, bsdf.cpp:295-296 | the perturbed record,
    //  which carries a fresh Intersection: accumulatedRoughness 0 -: the inputs are selected, the code is not duplicated)
    const bool plain = !nm.on || (wi.z > 0 && wo.z > 0 && dot(nm.n, wi) <= 0);
    V3 wiP = wi, woP = wo;
    if (!plain) { wiP = toLocal(nm.pf, toWorld(its.sh, wi)); woP = toLocal(nm.pf, toWorld(its.sh, wo)); if (wo.z * woP.z <= 0) return mk(0.f); }
    return bsdfEval<EXT>(b, nm.km, wiP, woP, plain ? accRough : 0.0f);
}
template <int EXT>
__device__ __forceinline__ float surfPdf(const KzBSDF &b, const NMap &nm, const Its &its, V3 wi, V3 wo, float accRough, bool solid) {
    if (!(EXT & KZ_X_NMAP)) return bsdfPdf<EXT>(b, nm.km, wi, wo, accRough);
    if (nm.on && !solid && b.type == KZ_BSDF_DIFFUSE) return 0.0f;
    const bool plain = !nm.on || (wi.z > 0 && wo.z > 0 && dot(nm.n, wi) <= 0);
    V3 wiP = wi, woP = wo;
    if (!plain) { wiP = toLocal(nm.pf, toWorld(its.sh, wi)); woP = toLocal(nm.pf, toWorld(its.sh, wo)); if (wo.z * woP.z <= 0) return 0.0f; }
    return bsdfPdf<EXT>(b, nm.km, wiP, woP, plain ? accRough : 0.0f);
}
// surfEval and surfPdf (solid angle measure) of one direction pair
template <int EXT>
__device__ __forceinline__ void surfEvalPdf(const KzBSDF &b, const NMap &nm, const Its &its, V3 wi, V3 wo, float accRough, V3 &f, float &pdf) {
    if (!(EXT & KZ_X_NMAP) || !nm.on) { bsdfEvalPdf<EXT>(b, nm.km, wi, wo, accRough, f, pdf); return; }
    f = surfEval<EXT>(b, nm, its, wi, wo, accRough); pdf = surfPdf<EXT>(b, nm, its, wi, wo, accRough, true);
}
template <int EXT>
__device__ __forceinline__ V3 surfSample(const KzBSDF &b, const NMap &nm, const Its &its, V3 wi, float accRough, float s1, float s2x, float s2y,
                                         V3 &wo, bool &alive, bool &discrete, float &etaScale, float &pdfOut, bool &solid) {
    solid = true;
    if (!(EXT & KZ_X_NMAP)) return bsdfSample<EXT>(b, nm.km, wi, accRough, s1, s2x, s2y, wo, alive, discrete, etaScale, pdfOut);
    // one call site of the model code (see surfEval): no map | the map's fall-back (bsdf.cpp:342-345) | the perturbed record
    const bool fallBack = nm.on && wi.z > 0 && dot(nm.n, wi) <= 0, plain = !nm.on || fallBack;
    const V3 wiP = plain ? wi : toLocal(nm.pf, toWorld(its.sh, wi));
    V3 woP; bool discN; float pdfN;
    const V3 w = bsdfSample<EXT>(b, nm.km, wiP, plain ? accRough : 0.0f, s1, s2x, s2y, woP, alive, discN, etaScale, pdfN);
    if (plain) {
        wo = woP; discrete = discN;
        pdfOut = fallBack ? -1.f : pdfN;                 // (fall-back: NormalMap::pdf may take the other branch - the caller evaluates it)
        return w;
    }
    discrete = false; solid = false; pdfOut = -1.f;      // measure stays EUnknownMeasure in the caller's record
    if (!alive || (w.x == 0.f && w.y == 0.f && w.z == 0.f)) { wo = mk(0.f, 0.f, 1.f); return mk(0.f); }
    wo = toLocal(its.sh, toWorld(nm.pf, woP));
    if (wo.z * woP.z <= 0) return mk(0.f);
    return w;
}

// ============================================================================================
// a18/a19 lights (light.cpp:16-51, mesh.cpp:108-133, dpdf.h:99-104)
// ============================================================================================
__device__ __forceinline__ float lightPdfSolidAngle(float meshPdf, V3 n, V3 wi, V3 p, V3 ref) {    // light.cpp:36-51
    float cosTheta = dot(n, -wi);
    if (cosTheta > 0.f) { V3 dd = p - ref; return meshPdf * dot(dd, dd) / cosTheta; }
    return 0.f;
}
__device__ __forceinline__ uint32_t cdfSample(const float *cdf, uint32_t n, float v) {              // dpdf.h:99-104 (n entries, n+1 floats)
    // std::lower_bound over cdf[0..n]: first element >= v
    uint32_t lo = 0;
    if (n <= 7u && v < 1.0f) {
        // Light meshes are mostly a quad or a few triangles. The table is non-decreasing up to its last entry, which is 1 (dpdf.h:85-88) and never
        // below a sample < 1, so the first element >= v is the NUMBER of elements < v: no chain of dependent loads.
        // (a table starts on a 16-B boundary and the array is padded: one or two float4 loads)
        const float4 c0 = reinterpret_cast<const float4 *>(cdf)[0];
        lo = (c0.x < v ? 1u : 0u) + ((1u <= n && c0.y < v) ? 1u : 0u) + ((2u <= n && c0.z < v) ? 1u : 0u) + ((3u <= n && c0.w < v) ? 1u : 0u);
        if (n > 3u) {
            const float4 c1 = reinterpret_cast<const float4 *>(cdf)[1];
            lo += (c1.x < v ? 1u : 0u) + ((5u <= n && c1.y < v) ? 1u : 0u) + ((6u <= n && c1.z < v) ? 1u : 0u) + ((7u <= n && c1.w < v) ? 1u : 0u);
        }
    } else {
        uint32_t len = n + 1;
        while (len > 0) {
            uint32_t half = len >> 1, mid = lo + half;
            if (cdf[mid] < v) { lo = mid + 1; len -= half + 1; } else len = half;
        }
    }
    int idx = (int)lo - 1;
    if (idx < 0) idx = 0;
    return min((uint32_t)idx, n - 1);
}

// Ls / m_lightPdf.getNormalization() (integrator.cpp:252): with a power-of-two number of lights the divisor is a power of two and the three
// IEEE divisions are three multiplications with the same results
__device__ __forceinline__ V3 lightPickDivide(const KzParams &P, V3 Ls) { return P.lightPickScale > 0.f ? Ls * P.lightPickScale : Ls / P.lightPickPdf; }

// Light::sample of an area light (light.cpp:16-34) through Mesh::sample (mesh.cpp:108-133): the triangle by the area cdf, then
// the sqrt warp; `draw()` supplies Mesh::sample's three next1D values IN ORDER (the path kernels hand in the sampler, the
// kz_light_query kernel a table). Ls = eval / pdf (0 when the pdf is 0, nan or inf), not yet divided by the pick pdf.
struct LightSample { V3 p, n, wi, Ls; float dist, pdf; uint32_t tri; };
template <class Draw>
__device__ __forceinline__ LightSample lightSample(const KzDevTables &T, const KzLightRow &lrow, V3 ref, Draw draw) {
    LightSample r;
    r.tri = cdfSample(T.cdf + lrow.cdfOffset, lrow.nF, draw());
    const float su0 = sqrtExact(draw());
    const float u = 1 - su0;
    const float v = draw() * su0;
    const float4 *sp = reinterpret_cast<const float4 *>(T.shade + lrow.triOffset + r.tri);
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    const V3 p0 = mk(s0.x, s0.y, s0.z), p1 = mk(s0.w, s1.x, s1.y), p2 = mk(s1.z, s1.w, s2.x);
    r.p = p0 + u * (p1 - p0) + v * (p2 - p0);
    if (lrow.hasN) {
        const V3 n0 = mk(s2.y, s2.z, s2.w), n1 = mk(s3.x, s3.y, s3.z), n2 = mk(s3.w, s4.x, s4.y);
        r.n = n0 + u * (n1 - n0) + v * (n2 - n0);                                         // H8: not normalised
    } else r.n = normalized(cross(p1 - p0, p2 - p0));
    const V3 toL = r.p - ref;
    r.wi = normalized(toL);
    r.dist = norm(toL);
    r.pdf = lightPdfSolidAngle(lrow.normalization, r.n, r.wi, r.p, ref);
    r.Ls = mk(0.f);
    if (r.pdf > 0.f && !isnan(r.pdf) && !isinf(r.pdf)) {
        const V3 ev = dot(r.n, -r.wi) > 0.f ? mk(lrow.radiance[0], lrow.radiance[1], lrow.radiance[2]) : mk(0.f);
        r.Ls = ev / r.pdf;
    }
    return r;
}

__device__ __forceinline__ float powerHeuristic(float a, float b) { a *= a; b *= b; return a > 0.f ? a / (a + b) : 0.f; }   // integrator.cpp:340-344

// a3 PerspectiveCamera::sampleRay (camera.cpp:70-91, transform.h:49-62)
__device__ __forceinline__ void cameraRay(const KzParams &P, float sx, float sy, float ax, float ay, V3 &o, V3 &d, float &mint, float &maxt) {
    const float *m = P.s2c;
    float x = sx * P.invW, y = sy * P.invH;
    float rx = m[0] * x + m[1] * y + m[2] * 0.0f + m[3];
    float ry = m[4] * x + m[5] * y + m[6] * 0.0f + m[7];
    float rz = m[8] * x + m[9] * y + m[10] * 0.0f + m[11];
    float rw = m[12] * x + m[13] * y + m[14] * 0.0f + m[15];
    const V3 nearP = mk(rx / rw, ry / rw, rz / rw);
    const float *w = P.c2w;
    V3 dl;
    if (P.cameraType == KZ_CAMERA_THINLENS) {                                      // camera.cpp:191-223, warp.cpp:41-50
        const float r = sqrtExact(ax);
        const float ang = 2.0f * KZ_PI_F * ay;
        float sinAng, cosAng; kzSinCos(ang, &sinAng, &cosAng);
        const float tx = cosAng * r * P.apertureRadius, ty = sinAng * r * P.apertureRadius;
        const V3 focusP = nearP * (P.focusDistance / nearP.z);
        dl = normalized(focusP - mk(tx, ty, 0.0f));
        const float pw = w[12] * tx + w[13] * ty + w[14] * 0.0f + w[15];
        o = mk((w[0] * tx + w[1] * ty + w[2] * 0.0f + w[3]) / pw, (w[4] * tx + w[5] * ty + w[6] * 0.0f + w[7]) / pw,
               (w[8] * tx + w[9] * ty + w[10] * 0.0f + w[11]) / pw);
    } else {
        dl = normalized(nearP);
        const float ow = w[15];
        o = mk(w[3] / ow, w[7] / ow, w[11] / ow);
    }
    float invZ = rcpExact(dl.z);
    d = mk(w[0] * dl.x + w[1] * dl.y + w[2] * dl.z, w[4] * dl.x + w[5] * dl.y + w[6] * dl.z, w[8] * dl.x + w[9] * dl.y + w[10] * dl.z);
    mint = P.nearClip * invZ; maxt = P.farClip * invZ;
}

// ============================================================================================
// a10 PathMisIntegrator::Li (integrator.cpp:195-338) — megakernel form, one lane per path
// ============================================================================================
template <bool STATS, int EXT>
__device__ V3 pathLi(const KzParams &P, const KzDevTables &T, Sampler &smp, V3 ro, V3 rd, float rmint, float rmaxt,
                     uint32_t *stk, Counters &cn) {
    const float eps = P.traceBias;
    V3 L = mk(0.f), throughput = mk(1.f);
    float eta = 1.f, bsdfWeight = 1.f, accRough = 0.f;
    RawHit rh; Its its;
    if (!closestHit<STATS>(T, P.rootRef, ro, rd, rmint, rmaxt, rh, stk, cn)) return L;       // H5: primary miss is black
    postIntersect<false>(T, rh, its); if (STATS) cn.hits++;
    {
        const int li = its.light;
        if (li >= 0 && !T.lights[li].primaryVisibility) {                                     // integrator.cpp:214-219 (H6)
            V3 no = its.p + eps * rd;
            if (closestHit<STATS>(T, P.rootRef, no, rd, KZ_EPSILON, KZ_INF, rh, stk, cn)) { postIntersect<false>(T, rh, its); if (STATS) cn.hits++; }
        }
    }
    int depth = 0;
    while (depth < P.maxDepth) {
        if (its.light >= 0) {                                                                 // integrator.cpp:226-231
            const KzLightRow &lr = T.lights[its.light];
            V3 wi = normalized(its.p - ro);
            if (dot(its.sh.n, -wi) > 0.f) L = L + (bsdfWeight * throughput) * mk(lr.radiance[0], lr.radiance[1], lr.radiance[2]);
            break;
        }
        if (depth >= 3) {                                                                     // integrator.cpp:237-244
            float probability = fminf(maxCoeff(throughput) * eta * eta, 0.95f);
            if (probability <= smp.next1D(P, T)) break;
            throughput = throughput / probability;
        }
        KzBSDF bsdf = T.bsdfs[its.bsdf];
        NMap nm; surfaceSetup<EXT>(T, its, bsdf, nm);
        const V3 wiLocal = toLocal(its.sh, -rd);
        // ---- light sampling (integrator.cpp:247-295); the pick is drawn even when there are no lights
        float pick = smp.next1D(P, T);
        if (P.nLights > 0) {
            uint32_t li = min((uint32_t)floorf((float)P.nLights * pick), P.nLights - 1);
            const KzLightRow lrow = T.lights[li];
            if (STATS) cn.lsamples++;
            const LightSample ls = lightSample(T, lrow, its.p, [&]() { return smp.next1D(P, T); });     // Mesh::sample: three 1-D draws
            const V3 lwi = ls.wi; const float dist = ls.dist, lpdf = ls.pdf;
            V3 Ls = ls.Ls;
            Ls = lightPickDivide(P, Ls);
            // shadow ray with the invisible-light walk-through (integrator.cpp:257-278); closest-hit, like the reference
            const bool occluded = shadowOccluded<STATS>(P, T, its.p, lwi, eps, dist - eps, stk, cn);
            if (!occluded) {
                V3 woLocal = toLocal(its.sh, lwi);
                V3 f; float bpdf;
                surfEvalPdf<EXT>(bsdf, nm, its, wiLocal, woLocal, accRough, f, bpdf);
                float lightWeight = powerHeuristic(lpdf, bpdf);
                L = L + throughput * Ls * f * lightWeight;
            }
        }
        if (P.regularization && bsdf.type == KZ_BSDF_KAZENSTANDARD) accRough += bsdf.roughness * P.accumulatedRoughness;   // integrator.cpp:298-301
        // ---- BSDF sampling (integrator.cpp:304-309): next2D BEFORE next1D (H1)
        float s2x, s2y; smp.next2D(P, T, s2x, s2y);
        float s1 = smp.next1D(P, T);
        V3 woLocal; bool alive, discrete, solid; float etaScale, pdfUnused;
        V3 weight = surfSample<EXT>(bsdf, nm, its, wiLocal, accRough, s1, s2x, s2y, woLocal, alive, discrete, etaScale, pdfUnused, solid);
        throughput = throughput * weight;
        eta *= etaScale;
        // zero weight: the reference keeps looping with throughput 0 (contributes exactly 0); terminate instead
        if (!alive || (weight.x == 0.f && weight.y == 0.f && weight.z == 0.f)) break;
        float bpdf = surfPdf<EXT>(bsdf, nm, its, wiLocal, woLocal, accRough, solid);
        ro = its.p; rd = toWorld(its.sh, woLocal);                                            // H9: not re-normalised
        if (!closestHit<STATS>(T, P.rootRef, ro, rd, eps, KZ_INF, rh, stk, cn)) {
            if (P.bgPresent) L = L + throughput * backgroundRadiance(P, T, rd);             // scene.cpp:54-79
            break;
        }
        postIntersect<false>(T, rh, its); if (STATS) cn.hits++;
        const int nl = its.light;
        if (nl >= 0) {                                                                        // integrator.cpp:322-327
            V3 wi = normalized(its.p - ro);
            float lpdf = lightPdfSolidAngle(T.lights[nl].normalization, its.sh.n, wi, its.p, ro);
            bsdfWeight = powerHeuristic(bpdf, lpdf);
        }
        if (discrete) bsdfWeight = 1.f;                                                       // integrator.cpp:329-331
        depth++;
    }
    return L;
}

