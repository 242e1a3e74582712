// =====================================================================================
// kz_oracle.cpp — CPU restatement of nano-kazen's path_mis hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it. The shipped library
// (nano-kazen_amd/csrc) never includes, links or calls anything in this directory.
//
// Parity status: the integer functions (MurmurHash64A / Hash / MixBits / pcg32) are
// PINNED against the reference's own headers compiled where they lie
// (oracle/build_ref.sh -> oracle/_ref/kat_ref -> tests/golden/int_kats.json).
// Everything floating point is a restatement from the source text: the reference
// ships no test, golden vector or fixture for this path (src/kazen/test.cpp:1-6 is an
// empty main) and cannot be built here (Eigen, Embree, TBB, OpenImageIO absent), so
// for those functions the header says it plainly: PARITY UNPINNED (text-only).
// Embree 3.13.0 (CMakeLists.txt:12) does the reference's BVH build + traversal; it is
// replaced by a brute-force Moeller-Trumbore search (mesh.cpp:55-92) and a scalar BVH2
// that is checked against that brute force.
//
// Every function cites the reference file:line it follows. Paths are relative to
// /root/reference. Draw order hazards H1..H11 are listed in SURVEY.md section 7.
// Compile: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
// =====================================================================================
#include "../include/kazen_mi355x_dev.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>
#include "kz_oracle_math.h"

namespace kzo {

// ---------------------------------------------------------------------------------
// constants (include/kazen/common.h:27-40)
// ---------------------------------------------------------------------------------
// H11: the reference runs with flush-to-zero and denormals-are-zero set in MXCSR (main.cpp:22-23; the render threads inherit
// the floating-point environment of the thread that creates them). Every entry point that does fp32 work holds one of these.
struct FtzScope {
#if defined(__SSE2__)
    unsigned saved;
    FtzScope() : saved(__builtin_ia32_stmxcsr()) { __builtin_ia32_ldmxcsr(saved | 0x8040u); }      // FTZ (bit 15) | DAZ (bit 6)
    ~FtzScope() { __builtin_ia32_ldmxcsr(saved); }
#endif
};
static const float Epsilon = 1e-5f;
static const float OneMinusEpsilon = float(0x1.fffffep-1);
static const float kPi = 3.14159265358979323846f;      // M_PI as common.h:31-33 redefines it (a float literal, after <cmath>)
static const float INV_PI = 0.31830988618379067154f;
static const float kInf = std::numeric_limits<float>::infinity();

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
// Eigen's normalized(): v / sqrt(squaredNorm) when squaredNorm > 0, else v.
static inline V3 normalized(V3 a) {
    float n2 = dot(a, a);
    if (n2 > 0.f) return a / std::sqrt(n2);
    return a;
}
static inline float maxCoeff(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }

// ---------------------------------------------------------------------------------
// a6  Hash / MurmurHash64A / MixBits  (include/kazen/hash.h:15-65, :71-78, :100-108)
// ---------------------------------------------------------------------------------
static inline uint64_t MurmurHash64A(const unsigned char *key, size_t len, uint64_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const unsigned char *end = key + 8 * (len / 8);
    while (key != end) {
        uint64_t k;
        std::memcpy(&k, key, 8);
        key += 8;
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
    }
    size_t tail = len & 7;
    if (tail) {
        for (size_t i = tail; i-- > 0;) h ^= uint64_t(key[i]) << (8 * i);
        h *= m;
    }
    h ^= h >> r; h *= m; h ^= h >> r;
    return h;
}
static inline uint64_t MixBits(uint64_t v) {
    v ^= (v >> 31); v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27); v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}
// Hash(Point2i p, uint64_t seed): 16-byte key (sampler.cpp:44)
static inline uint64_t HashPixelSeed(int32_t px, int32_t py, uint64_t seed) {
    unsigned char buf[16];
    std::memcpy(buf, &px, 4); std::memcpy(buf + 4, &py, 4); std::memcpy(buf + 8, &seed, 8);
    return MurmurHash64A(buf, 16, 0);
}
// Hash(Point2i p, uint32_t dim, uint64_t seed): 20-byte key (sampler.cpp:341,355)
static inline uint64_t HashPixelDimSeed(int32_t px, int32_t py, uint32_t dim, uint64_t seed) {
    unsigned char buf[24];
    std::memcpy(buf, &px, 4); std::memcpy(buf + 4, &py, 4); std::memcpy(buf + 8, &dim, 4);
    std::memcpy(buf + 12, &seed, 8);
    return MurmurHash64A(buf, 20, 0);
}

// ---------------------------------------------------------------------------------
// a7  pcg32 (include/kazen/pcg32.h:54-64 seed, :67-73 nextUInt, :108-117 nextFloat,
//            :145-166 advance)
// ---------------------------------------------------------------------------------
struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;
    Pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u; inc = (initseq << 1u) | 1u;
        nextUInt(); state += initstate; nextUInt();
    }
    void seed(uint64_t initseq) { seed(MixBits(initseq), initseq); }
    uint32_t nextUInt() {
        uint64_t old = state;
        state = old * MULT + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    float nextFloat() {
        union { uint32_t u; float f; } x;
        x.u = (nextUInt() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
    void advance(int64_t delta_) {
        uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        uint64_t delta = (uint64_t)delta_;
        while (delta > 0) {
            if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta /= 2;
        }
        state = acc_mult * state + acc_plus;
    }
};

// ---------------------------------------------------------------------------------
// a8  random::permute (src/kazen/common.cpp:316-344) — p is uint32 (callers truncate)
// ---------------------------------------------------------------------------------
static inline uint32_t permute(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
        i *= 0x0929eb3f; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303; i ^= (i & w) >> 2;
        i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
// src/kazen/common.cpp:304-314
static inline uint64_t sampleTEA32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9;
        v0 += ((v1 << 4) + 0xa341316c) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4);
        v1 += ((v0 << 4) + 0xad90777d) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761e);
    }
    return ((uint64_t)v1 << 32) + v0;
}

// include/kazen/common.h:271-320 (isPowerOf4 / log2i / log4i / roundUpPow4)
static inline bool isPowerOf4(int n) {
    if (n <= 0) return false;
    int x = (int)std::sqrt((double)n);
    if (x * x != n) return false;
    return !(n & (n - 1));
}
static inline int log2i(uint32_t v) { return 31 - __builtin_clz(v); }
static inline int log4i(uint32_t v) { return log2i(v) / 2; }
static inline int roundUpPow4(int v) { return isPowerOf4(v) ? v : (1 << (2 * (1 + log4i((uint32_t)v)))); }
static inline int pmjPixelTile(uint32_t spp) { return 1 << (log4i(KZ_PMJ02BN_SAMPLES) - log4i((uint32_t)roundUpPow4((int)spp))); }      // sampler.cpp:291

// ---------------------------------------------------------------------------------
// a17  Frame / coordinateSystem (include/kazen/frame.h:14-51, src/kazen/common.cpp:436-445)
// ---------------------------------------------------------------------------------
static inline void coordinateSystem(const V3 &a, V3 &b, V3 &c) {
    if (std::fabs(a.x) > std::fabs(a.y)) {
        float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = V3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = V3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}
struct Frame {
    V3 s, t, n;
    Frame() {}
    explicit Frame(const V3 &n_) : n(n_) { coordinateSystem(n, s, t); }
    V3 toLocal(const V3 &v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 toWorld(const V3 &v) const { return s * v.x + t * v.y + n * v.z; }
};

struct Ray {
    V3 o, d;
    float mint, maxt;
    Ray() : mint(Epsilon), maxt(kInf) {}                       // ray.h:31-32
    Ray(V3 o_, V3 d_) : o(o_), d(d_), mint(Epsilon), maxt(kInf) {}   // ray.h:35-38
    Ray(V3 o_, V3 d_, float a, float b) : o(o_), d(d_), mint(a), maxt(b) {}
};

// ---------------------------------------------------------------------------------
// scene tables
// ---------------------------------------------------------------------------------
struct MeshData {
    std::vector<float> V, N, UV;
    std::vector<uint32_t> F;
    uint32_t nV = 0, nF = 0;
    int bsdf = -1, light = -1;
    // light CDF (mesh.cpp:24-45, dpdf.h)
    std::vector<float> cdf;
    float normalization = 0.f;   // DiscretePDF::m_normalization == Mesh::pdf()
    uint32_t triOffset = 0;      // global triangle id of face 0
    V3 v(uint32_t i) const { return V3(V[3 * i], V[3 * i + 1], V[3 * i + 2]); }
    V3 nrm(uint32_t i) const { return V3(N[3 * i], N[3 * i + 1], N[3 * i + 2]); }
};

struct Intersection {
    V3 p;
    float t = 0;
    float uvx = 0, uvy = 0;
    Frame shFrame, geoFrame;
    int mesh = -1;
    float accumulatedRoughness = 0.f;   // mesh.h:40
    V3 dpdu;                            // mesh.h:33 (H12: defined as shFrame.s wherever accel.cpp leaves it unset)
    // extra (not in the reference record): prim ids for ray-level tests
    int prim = -1;
    float bu = 0, bv = 0;
};

struct Stats {
    std::atomic<uint64_t> samples{0}, rays{0}, nodeVisits{0}, triTests{0}, shadedHits{0}, lightSamples{0}, dropped{0};
};
struct LocalStats {
    uint64_t samples = 0, rays = 0, nodeVisits = 0, triTests = 0, shadedHits = 0, lightSamples = 0, dropped = 0;
};

// scalar BVH2, node = both children's boxes (the 64-B packet of SURVEY 8d)
struct BNode {
    float lo[2][3], hi[2][3];
    uint32_t child[2];    // bit31 set: leaf -> (start<<3 | count-1... see encode)
};
static inline uint32_t leafRef(uint32_t start, uint32_t count) { return 0x80000000u | (start << 3) | (count - 1); }

struct Tri { V3 p0, e1, e2; uint32_t mesh, prim, gid; };

struct Scene {
    std::vector<MeshData> meshes;
    std::vector<KzBSDF> bsdfs;
    std::vector<KzLight> lights;
    std::vector<KzTexture> textures;         // Texture<Color3f> nodes (texture.cpp)
    struct Image { int w, h, c, fmt; std::vector<uint8_t> px; };
    std::vector<Image> images;
    std::vector<int> lightMeshes;            // Scene::m_lights (scene.cpp:42-46), mesh order
    KzCamera cam;
    KzSampler smp;
    KzIntegrator integ;
    KzBackground bg;
    std::vector<uint32_t> pmjTable;
    std::vector<uint16_t> bnTable;
    // camera (camera.cpp:35-68)
    float s2c[16];
    float invW, invH;
    // filter table (block.cpp:13-21)
    float filterRadius = 2.f; int border = 2; float filter[KZ_FILTER_RESOLUTION + 1]; float lookupFactor = 16.f;
    // pmj02bn pixel samples (sampler.cpp:291-309)
    int pixelTileSize = 0; std::vector<float> pixelSamples;
    uint32_t sampleCount = 1;
    int resX = 1, resY = 1;       // stratified: resolution; correlated: m_resolution (sampler.cpp:181-187)
    // geometry
    std::vector<Tri> tris;        // leaf order
    std::vector<BNode> nodes;
    uint32_t rootRef = 0;         // for degenerate 1-leaf scenes
    bool useBrute = false;
    int tieMode = 0;              // test knob, see kzo_set_tie_mode
    uint32_t maxDepth = 0;
    Stats stats;
};

// ---------------------------------------------------------------------------------
// a15  Mesh::rayIntersect — Moeller-Trumbore (src/kazen/mesh.cpp:55-92)
// edge1/edge2 are the same float values the reference computes per call (p1-p0, p2-p0).
// ---------------------------------------------------------------------------------
static inline bool triIntersect(const Tri &tr, const V3 &o, const V3 &d, float mint, float maxt,
                                float &u, float &v, float &t) {
    V3 pvec = cross(d, tr.e2);
    float det = dot(tr.e1, pvec);
    if (det > -1e-8f && det < 1e-8f) return false;
    float inv_det = 1.0f / det;
    V3 tvec = o - tr.p0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0.0 || u > 1.0) return false;
    V3 qvec = cross(tvec, tr.e1);
    v = dot(d, qvec) * inv_det;
    if (v < 0.0 || u + v > 1.0) return false;
    t = dot(tr.e2, qvec) * inv_det;
    return t >= mint && t <= maxt;
}

// Closest-hit record of rtcIntersect1 (accel.cpp:98-110). Ties on t are broken by the lower
// global triangle id so that the answer does not depend on traversal order (Embree's own
// tie behaviour is unspecified; parity there is unpinned).
struct RawHit { float t, u, v; uint32_t ti; bool hit; };

static inline void considerTri(const Scene &sc, uint32_t ti, const V3 &o, const V3 &d, float mint, RawHit &best,
                               float &maxt, LocalStats &ls) {
    float u, v, t;
    ls.triTests++;
    const Tri &tr = sc.tris[ti];
    if (triIntersect(tr, o, d, mint, maxt, u, v, t)) {
        if (!best.hit || t < best.t || (t == best.t && tr.gid < sc.tris[best.ti].gid)) {
            best.hit = true; best.t = t; best.u = u; best.v = v; best.ti = ti; maxt = t;
        }
    }
}

// a16 TBoundingBox::rayIntersect slab test (include/kazen/bbox.h:316-343), returning nearT.
static inline bool slab(const float lo[3], const float hi[3], const V3 &o, const V3 &d, const V3 &rcp,
                        float mint, float maxt, float &nearOut) {
    float nearT = -kInf, farT = kInf;
    for (int i = 0; i < 3; i++) {
        float origin = o[i], minVal = lo[i], maxVal = hi[i];
        if (d[i] == 0) {
            if (origin < minVal || origin > maxVal) return false;
        } else {
            float t1 = (minVal - origin) * rcp[i];
            float t2 = (maxVal - origin) * rcp[i];
            if (t1 > t2) std::swap(t1, t2);
            nearT = std::max(t1, nearT);
            farT = std::min(t2, farT);
            if (!(nearT <= farT)) return false;
        }
    }
    nearOut = nearT;
    return mint <= farT && nearT <= maxt;
}

static RawHit closestHit(const Scene &sc, const V3 &o, const V3 &d, float mint, float maxt, LocalStats &ls) {
    RawHit best; best.hit = false; best.t = kInf; best.u = best.v = 0; best.ti = 0;
    ls.rays++;
    if (sc.tris.empty()) return best;
    if (sc.useBrute) {
        for (uint32_t i = 0; i < sc.tris.size(); ++i) considerTri(sc, i, o, d, mint, best, maxt, ls);
        return best;
    }
    V3 rcp(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);     // ray.h:56-58 cwiseInverse
    uint32_t stack[128]; int sp = 0;
    uint32_t cur = sc.rootRef;
    for (;;) {
        if (cur & 0x80000000u) {
            uint32_t start = (cur & 0x7fffffffu) >> 3, count = (cur & 7u) + 1;
            for (uint32_t i = 0; i < count; ++i) considerTri(sc, start + i, o, d, mint, best, maxt, ls);
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const BNode &n = sc.nodes[cur];
        ls.nodeVisits++;
        float n0 = 0.f, n1 = 0.f;
        bool h0 = slab(n.lo[0], n.hi[0], o, d, rcp, mint, maxt, n0);
        bool h1 = slab(n.lo[1], n.hi[1], o, d, rcp, mint, maxt, n1);
        if (h0 && h1) {
            int nearIdx = (n1 < n0) ? 1 : 0;
            stack[sp++] = n.child[1 - nearIdx];
            cur = n.child[nearIdx];
        } else if (h0) cur = n.child[0];
        else if (h1) cur = n.child[1];
        else { if (sp == 0) break; cur = stack[--sp]; }
    }
    return best;
}

// ---------------------------------------------------------------------------------
// host BVH build for the oracle: plain binned-SAH BVH2, <= 4 tris per leaf. Boxes are
// padded by a few ulps so the slab test (different rounding than Moeller-Trumbore) can
// never cull a triangle the brute force would report.
// ---------------------------------------------------------------------------------
struct BuildPrim { float lo[3], hi[3], c[3]; uint32_t tri; };
struct Builder {
    Scene &sc; std::vector<BuildPrim> prims; std::vector<Tri> src; uint32_t maxDepth = 0;
    Builder(Scene &s) : sc(s) {}
    static void padBox(float lo[3], float hi[3]) {
        for (int a = 0; a < 3; ++a) {
            float m = std::max(std::fabs(lo[a]), std::fabs(hi[a]));
            float e = m * 4e-7f + 1e-30f;
            lo[a] -= e; hi[a] += e;
        }
    }
    void bounds(uint32_t b, uint32_t e, float lo[3], float hi[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = kInf; hi[a] = -kInf; }
        for (uint32_t i = b; i < e; ++i)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], prims[i].lo[a]); hi[a] = std::max(hi[a], prims[i].hi[a]); }
    }
    static float area(const float lo[3], const float hi[3]) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
    // returns child ref; writes box of the subtree
    uint32_t build(uint32_t b, uint32_t e, float lo[3], float hi[3], uint32_t depth) {
        bounds(b, e, lo, hi);
        maxDepth = std::max(maxDepth, depth);
        uint32_t n = e - b;
        if (n <= 4) {
            uint32_t start = (uint32_t)sc.tris.size();
            for (uint32_t i = b; i < e; ++i) sc.tris.push_back(src[prims[i].tri]);
            padBox(lo, hi);
            return leafRef(start, n);
        }
        // centroid bounds
        float clo[3] = {kInf, kInf, kInf}, chi[3] = {-kInf, -kInf, -kInf};
        for (uint32_t i = b; i < e; ++i)
            for (int a = 0; a < 3; ++a) { clo[a] = std::min(clo[a], prims[i].c[a]); chi[a] = std::max(chi[a], prims[i].c[a]); }
        const int NB = 16;
        float bestCost = kInf; int bestAxis = -1, bestSplit = 0;
        for (int a = 0; a < 3; ++a) {
            float ext = chi[a] - clo[a];
            if (!(ext > 0.f)) continue;
            float blo[NB][3], bhi[NB][3]; uint32_t cnt[NB];
            for (int k = 0; k < NB; ++k) { cnt[k] = 0; for (int c = 0; c < 3; ++c) { blo[k][c] = kInf; bhi[k][c] = -kInf; } }
            float scale = NB / ext;
            for (uint32_t i = b; i < e; ++i) {
                int k = std::min(NB - 1, (int)((prims[i].c[a] - clo[a]) * scale));
                cnt[k]++;
                for (int c = 0; c < 3; ++c) { blo[k][c] = std::min(blo[k][c], prims[i].lo[c]); bhi[k][c] = std::max(bhi[k][c], prims[i].hi[c]); }
            }
            float rA[NB]; uint32_t rC[NB];
            float rlo[3] = {kInf, kInf, kInf}, rhi[3] = {-kInf, -kInf, -kInf}; uint32_t rc = 0;
            for (int k = NB - 1; k > 0; --k) {
                for (int c = 0; c < 3; ++c) { rlo[c] = std::min(rlo[c], blo[k][c]); rhi[c] = std::max(rhi[c], bhi[k][c]); }
                rc += cnt[k]; rA[k] = area(rlo, rhi); rC[k] = rc;
            }
            float llo[3] = {kInf, kInf, kInf}, lhi[3] = {-kInf, -kInf, -kInf}; uint32_t lc = 0;
            for (int k = 0; k < NB - 1; ++k) {
                for (int c = 0; c < 3; ++c) { llo[c] = std::min(llo[c], blo[k][c]); lhi[c] = std::max(lhi[c], bhi[k][c]); }
                lc += cnt[k];
                if (lc == 0 || rC[k + 1] == 0) continue;
                float cost = area(llo, lhi) * lc + rA[k + 1] * rC[k + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
            }
        }
        uint32_t mid;
        if (bestAxis < 0 || depth > 90) {
            mid = b + n / 2;   // all centroids coincide (or runaway depth): split the list in half
        } else {
            float ext = chi[bestAxis] - clo[bestAxis]; float scale = NB / ext; int a = bestAxis;
            auto it = std::partition(prims.begin() + b, prims.begin() + e, [&](const BuildPrim &p) {
                int k = std::min(NB - 1, (int)((p.c[a] - clo[a]) * scale));
                return k <= bestSplit;
            });
            mid = (uint32_t)(it - prims.begin());
            if (mid == b || mid == e) mid = b + n / 2;
        }
        uint32_t idx = (uint32_t)sc.nodes.size();
        sc.nodes.push_back(BNode());
        float l0[3], h0[3], l1[3], h1[3];
        uint32_t c0 = build(b, mid, l0, h0, depth + 1);
        uint32_t c1 = build(mid, e, l1, h1, depth + 1);
        BNode &nd = sc.nodes[idx];
        for (int a = 0; a < 3; ++a) { nd.lo[0][a] = l0[a]; nd.hi[0][a] = h0[a]; nd.lo[1][a] = l1[a]; nd.hi[1][a] = h1[a]; }
        nd.child[0] = c0; nd.child[1] = c1;
        // parents of leaves carry padded boxes already; inner boxes are unions of padded boxes
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(l0[a], l1[a]); hi[a] = std::max(h0[a], h1[a]); }
        return idx;
    }
    void run() {
        // gather triangles in (mesh, face) order = Embree geomID/primID order (accel.cpp:40-55)
        uint32_t gid = 0;
        for (size_t m = 0; m < sc.meshes.size(); ++m) {
            MeshData &md = sc.meshes[m];
            md.triOffset = gid;
            for (uint32_t f = 0; f < md.nF; ++f, ++gid) {
                uint32_t i0 = md.F[3 * f], i1 = md.F[3 * f + 1], i2 = md.F[3 * f + 2];
                Tri t; t.p0 = md.v(i0); V3 p1 = md.v(i1), p2 = md.v(i2);
                t.e1 = p1 - t.p0; t.e2 = p2 - t.p0; t.mesh = (uint32_t)m; t.prim = f; t.gid = gid;
                src.push_back(t);
                BuildPrim bp; bp.tri = gid;
                for (int a = 0; a < 3; ++a) {
                    float x0 = t.p0[a], x1 = p1[a], x2 = p2[a];
                    bp.lo[a] = std::min(x0, std::min(x1, x2)); bp.hi[a] = std::max(x0, std::max(x1, x2));
                    bp.c[a] = 0.5f * (bp.lo[a] + bp.hi[a]);
                }
                prims.push_back(bp);
            }
        }
        sc.tris.clear(); sc.nodes.clear();
        if (prims.empty()) return;
        float lo[3], hi[3];
        sc.tris.reserve(prims.size());
        sc.rootRef = build(0, (uint32_t)prims.size(), lo, hi, 0);
        sc.maxDepth = maxDepth;
    }
};

// ---------------------------------------------------------------------------------
// a13/a14  Accel::rayIntersect (src/kazen/accel.cpp:63-239)
// ---------------------------------------------------------------------------------
static bool rayIntersect(const Scene &sc, const Ray &ray, Intersection &its, bool shadowRay, LocalStats &ls) {
    RawHit rh = closestHit(sc, ray.o, ray.d, ray.mint, ray.maxt, ls);
    if (!rh.hit) return false;
    const Tri &tr = sc.tris[rh.ti];
    if (shadowRay) {                                   // accel.cpp:100-104
        its.t = rh.t; its.mesh = (int)tr.mesh;
        return true;
    }
    ls.shadedHits++;
    its.t = rh.t; its.uvx = rh.u; its.uvy = rh.v; its.mesh = (int)tr.mesh;      // accel.cpp:105-108
    its.prim = (int)tr.prim; its.bu = rh.u; its.bv = rh.v;
    const MeshData &md = sc.meshes[tr.mesh];
    uint32_t f = tr.prim;
    float bx = 1 - (its.uvx + its.uvy), by = its.uvx, bz = its.uvy;              // accel.cpp:122-123
    uint32_t idx0 = md.F[3 * f], idx1 = md.F[3 * f + 1], idx2 = md.F[3 * f + 2];
    V3 p0 = md.v(idx0), p1 = md.v(idx1), p2 = md.v(idx2);
    bool hasN = !md.N.empty(), hasUV = !md.UV.empty();
    V3 orignP = bx * p0 + by * p1 + bz * p2;                                     // accel.cpp:142
    if (hasN) {
        V3 n0 = md.nrm(idx0), n1 = md.nrm(idx1), n2 = md.nrm(idx2);
        V3 tmpu = orignP - p0, tmpv = orignP - p1, tmpw = orignP - p2;           // accel.cpp:144
        float dotu = std::min(0.f, dot(tmpu, n0));
        float dotv = std::min(0.f, dot(tmpv, n1));
        float dotw = std::min(0.f, dot(tmpw, n2));
        tmpu = tmpu - dotu * n0; tmpv = tmpv - dotv * n1; tmpw = tmpw - dotw * n2;
        its.p = orignP + bx * tmpu + by * tmpv + bz * tmpw;                      // accel.cpp:153
    } else {
        its.p = orignP;   // reference reads N.col() of an empty matrix here (UB, H4); defined as "no offset"
    }
    V3 dp0 = p1 - p0, dp1 = p2 - p0;
    its.geoFrame = Frame(normalized(cross(dp0, dp1)));                           // accel.cpp:156-158
    float uv0x = 0, uv0y = 0, uv1x = 0, uv1y = 0, uv2x = 0, uv2y = 0;
    bool tangent = false;
    if (hasUV) {                                                                 // accel.cpp:161-164
        uv0x = md.UV[2 * idx0]; uv0y = md.UV[2 * idx0 + 1];
        uv1x = md.UV[2 * idx1]; uv1y = md.UV[2 * idx1 + 1];
        uv2x = md.UV[2 * idx2]; uv2y = md.UV[2 * idx2 + 1];
        its.uvx = bx * uv0x + by * uv1x + bz * uv2x;
        its.uvy = bx * uv0y + by * uv1y + bz * uv2y;
    }
    if (hasN && hasUV) {                                                         // accel.cpp:166-217
        V3 n0 = md.nrm(idx0), n1 = md.nrm(idx1), n2 = md.nrm(idx2);
        float duv0x = uv1x - uv0x, duv0y = uv1y - uv0y, duv1x = uv2x - uv0x, duv1y = uv2y - uv0y;
        V3 shNormal = bx * n0 + by * n1 + bz * n2;
        float length = norm(cross(dp0, dp1));
        if (length > 0.f) {
            float determinant = duv0x * duv1y - duv0y * duv1x;
            if (determinant > 0.f) {
                float invDet = 1.0f / determinant;
                V3 dpdu = (duv1y * dp0 - duv0y * dp1) * invDet;
                its.shFrame.n = normalized(shNormal);
                its.shFrame.s = normalized(dpdu - shNormal * dot(shNormal, dpdu));
                its.shFrame.t = normalized(cross(its.shFrame.n, its.shFrame.s));
                its.dpdu = dpdu; tangent = true;
            } else {
                its.shFrame = Frame(normalized(shNormal));                       // accel.cpp:203-212 sets dpdu = shFrame.s here
            }
        } else {
            its.shFrame = Frame(normalized(shNormal));
        }
    } else if (hasN) {                                                           // accel.cpp:219-229
        V3 n0 = md.nrm(idx0), n1 = md.nrm(idx1), n2 = md.nrm(idx2);
        its.shFrame = Frame(normalized(bx * n0 + by * n1 + bz * n2));
    } else {
        its.shFrame = its.geoFrame;                                              // accel.cpp:231-233
    }
    // H12: in the branches that never assign its.dpdu the reference reads whatever the record held before (the previous
    // bounce's value, or uninitialised memory). Only NormalMap::getFrame reads it; defined here as shFrame.s, the value
    // the degenerate-uv branch assigns.
    if (!tangent) its.dpdu = its.shFrame.s;
    return true;
}

// ---------------------------------------------------------------------------------
// a23  Warp::squareToCosineHemisphere (src/kazen/warp.cpp:85-115). M_PI is the FLOAT literal of common.h:31-33 (<cmath> is
// included above it, so the #undef / #define there is the definition every kazen source sees): the expressions are float throughout.
// sin / cos: kz_oracle_math.h.
// ---------------------------------------------------------------------------------
static V3 squareToCosineHemisphere(float sx, float sy) {
    float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) { r = phi = 0; }
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (kPi / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (kPi / 2.0f) - (r1 / r2) * (kPi / 4.0f); }
    float sinPhi, cosPhi; kzoSinCos(phi, &sinPhi, &cosPhi);   // common.h:228-231 math::sincosf
    float px = r * cosPhi, py = r * sinPhi;
    float z = std::sqrt(1.0f - px * px - py * py);
    if (z == 0) z = 1e-10f;
    return V3(px, py, z);
}
// Warp::squareToUniformDisk (warp.cpp:41-50) — thin-lens, "next" row
static void squareToUniformDisk(float sx, float sy, float &ox, float &oy) {
    float r = std::sqrt(sx);
    float a = 2.0f * kPi * sy;
    float sinA, cosA; kzoSinCos(a, &sinA, &cosA);
    ox = cosA * r; oy = sinA * r;
}

// ---------------------------------------------------------------------------------
// a22  GGX helpers (include/kazen/ggx_brdf.h). H3: unqualified abs/pow/cos/sin are taken
// with float semantics; M_PI is common.h's float literal.
// ---------------------------------------------------------------------------------
static inline float sqr(float x) { return x * x; }
struct A2 { float x, y; };
static inline V3 schlickFresnel(V3 f0, float cosTheta) {                 // ggx_brdf.h:15-24
    // pow(x, 5.0f): the fifth power in double, narrowed once (the correctly rounded float but for ~1 argument in 2^27)
    const double xd = (double)(1.0f - cosTheta), xd2 = xd * xd;
    float t = (float)(xd2 * xd2 * xd);
    return f0 * 1.0f + (V3(1.f) - f0) * t;
}
static inline A2 roughnessToAlpha(float roughness, float anisotropy) {   // ggx_brdf.h:28-37
    float alpha = std::max(0.001f, sqr(roughness));
    return A2{alpha * (1.0f + anisotropy), alpha * (1.0f - anisotropy)};
}
static inline float lambda(V3 v, A2 a) {                                 // ggx_brdf.h:41-45
    float squared = (sqr(a.x) * sqr(v.x) + sqr(a.y) * sqr(v.y)) / sqr(v.z);
    return (-1.0f + std::sqrt(1.0f + squared)) * 0.5f;
}
static inline float smithG1(V3 V, V3 H, A2 a) {                          // ggx_brdf.h:49-55
    if (dot(V, H) <= 0.0f) return 0.0f;
    return 1.0f / (1.0f + lambda(V, a));
}
static inline float smithG2(V3 V, V3 L, V3 H, A2 a) {                    // ggx_brdf.h:60-67
    if (dot(V, H) <= 0.0f || dot(L, H) < 0.0f) return 0.0f;
    return 1.0f / (1.0f + lambda(V, a) + lambda(L, a));
}
static inline float ggxNDF(V3 H, A2 a) {                                 // ggx_brdf.h:71-75
    float ellipse = sqr(H.x) / sqr(a.x) + sqr(H.y) / sqr(a.y) + sqr(H.z);
    return 1.0f / (kPi * a.x * a.y * sqr(ellipse));
}
static inline float ggxSmithVNDF(V3 V, V3 H, A2 a) {                     // ggx_brdf.h:80-91
    float VDotH = dot(V, H);
    if (VDotH <= 0.0f) return 0.0f;
    float D = ggxNDF(H, a);
    float G1 = smithG1(V, H, a);
    return D * G1 * VDotH / V.z;
}
static V3 sampleGGXSmithVNDF(V3 V, A2 a, float rx, float ry) {           // ggx_brdf.h:96-120
    V3 Vh = normalized(V3(a.x * V.x, a.y * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) / std::sqrt(lensq) : V3(1.0f, 0.0f, 0.0f);
    V3 T2 = normalized(cross(Vh, T1));
    float r = std::sqrt(rx);
    float phi = 2.0f * kPi * ry;
    float sinPhi, cosPhi; kzoSinCos(phi, &sinPhi, &cosPhi);
    float t1 = r * cosPhi;
    float t2 = r * sinPhi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * std::sqrt(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + std::sqrt(std::max(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalized(V3(a.x * Nh.x, a.y * Nh.y, std::max(1e-6f, Nh.z)));
}
static V3 evaluateGGXSmithBRDF(V3 V, V3 L, V3 f0, float roughness, float anisotropy) {   // ggx_brdf.h:151-170
    if (V.z * L.z < 0.0f) return V3(0.0f);
    A2 a = roughnessToAlpha(roughness, anisotropy);
    V3 H = normalized(V + L);
    float D = ggxNDF(H, a);
    float G = smithG2(V, L, H, a);
    V3 F = schlickFresnel(f0, dot(V, H));
    float denom = 4.0f * std::fabs(V.z) * std::fabs(L.z);
    return (D * G) * F / denom;
}

// ---------------------------------------------------------------------------------
// BSDFs. BSDFQueryRecord: wi, wo local; measure; eta (bsdf.h:20-53).
// ---------------------------------------------------------------------------------
enum { EUnknownMeasure = 0, ESolidAngle = 1, EDiscrete = 2 };
// bRec.its (bsdf.h:22) is a COPY of the integrator's record (integrator.cpp:284,306) and default-constructed in the
// records NormalMap builds (accumulatedRoughness 0, bsdf.cpp:301,325,350); uv is set beside it.
struct BRec {
    V3 wi, wo; float eta = 1.f; int measure = EUnknownMeasure; float accumulatedRoughness = 0.f;
    float uvx = 0.f, uvy = 0.f;
    const Scene *sc = nullptr;          // where the Texture children live
    const Intersection *its = nullptr;  // bRec.its: shFrame, dpdu, uv (NormalMap only)
};

static inline float lerpf(float t, float v1, float v2) { return (1.f - t) * v1 + t * v2; }   // common.h:255-257
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float schlickWeight(float x) {                               // bsdf.cpp:1175-1179
    x = clampf(1.f - x, 0.f, 1.f);
    float x2 = x * x;
    return x2 * x2 * x;
}
static inline V3 lerp3(V3 c1, V3 c2, float t) { return (1.f - t) * c1 + t * c2; }   // bsdf.cpp:1181-1183
static inline float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }   // common.cpp:393-395


// ---------------------------------------------------------------------------------
// textures (src/kazen/texture.cpp). ImageTexture::eval hands (u*scale, (1-v)*scale) with zero
// derivatives and periodic wrap to OpenImageIO's TextureSystem::texture (texture.cpp:46-64). OpenImageIO (pinned only
// as "find_package(OpenImageIO)" by the reference, no version, not vendored) is absent from the checkout, so its
// filter cannot be restated from source: it is a FIELD of KzTexture (bilinear, as SURVEY 8f row 4 specifies, by default; cubic
// B-spline = what OpenImageIO's default "smart bicubic" mode most likely evaluates for these magnifying lookups), over the
// full-resolution level with texel centres at (i+0.5)/res. Parity of this one function is unpinned.
// ---------------------------------------------------------------------------------
static inline float texelAt(const Scene::Image &im, int x, int y, int c) {
    if (c >= im.c) return 0.0f;                                                  // missing channels: TextureOpt::fill = 0
    size_t i = ((size_t)y * (size_t)im.w + (size_t)x) * (size_t)im.c + (size_t)c;
    if (im.fmt == KZ_PIXEL_F32) { float v; std::memcpy(&v, im.px.data() + 4 * i, 4); return v; }
    return (float)im.px[i] * (1.0f / 255.0f);
}
static inline int wrapPeriodic(int i, int n) { i %= n; return i < 0 ? i + n : i; }
static inline float srgbToLinear(float v) {                                      // Color3f::toLinearRGB, common.cpp:368-382
    return v <= 0.04045f ? v * (1.0f / 12.92f) : kzoPow((v + 0.055f) * (1.0f / 1.055f), 2.4f);
}
// taps and weights of one axis: the two of the bilinear lookup or the four of the cubic B-spline (KzTexture.filter, include/kazen_mi355x.h)
struct Taps { int first, n; float w[4]; };
static inline Taps filterTaps(int filter, float x) {
    Taps t;
    const float x0 = std::floor(x), f = x - x0;
    if (filter == KZ_TEXFILTER_BICUBIC) {
        const float omf = 1.0f - f, f2 = f * f, f3 = f2 * f;
        t.first = (int)x0 - 1; t.n = 4;
        t.w[0] = omf * omf * omf * (1.0f / 6.0f);
        t.w[1] = (3.0f * f3 - 6.0f * f2 + 4.0f) * (1.0f / 6.0f);
        t.w[2] = (-3.0f * f3 + 3.0f * f2 + 3.0f * f + 1.0f) * (1.0f / 6.0f);
        t.w[3] = f3 * (1.0f / 6.0f);
    } else { t.first = (int)x0; t.n = 2; t.w[0] = 1.0f - f; t.w[1] = f; t.w[2] = t.w[3] = 0.0f; }
    return t;
}
static inline float filteredTexel(const Scene::Image &im, const Taps &tx, const Taps &ty, int c, bool clampY) {
    float r = 0.0f;
    for (int j = 0; j < ty.n; ++j) {
        const int y = clampY ? std::min(std::max(ty.first + j, 0), im.h - 1) : wrapPeriodic(ty.first + j, im.h);
        float row = tx.w[0] * texelAt(im, wrapPeriodic(tx.first, im.w), y, c);
        for (int i = 1; i < tx.n; ++i) row = row + tx.w[i] * texelAt(im, wrapPeriodic(tx.first + i, im.w), y, c);
        r = j == 0 ? ty.w[0] * row : r + ty.w[j] * row;
    }
    return r;
}
static V3 imageLookup(const Scene &sc, const KzTexture &k, float u, float v) {
    const Scene::Image &im = sc.images[k.image];
    float s = u * k.scale, t = (1.0f - v) * k.scale;                             // texture.cpp:55
    float x = s * (float)im.w - 0.5f, y = t * (float)im.h - 0.5f;
    if (!(std::fabs(x) < 1.0e9f) || !(std::fabs(y) < 1.0e9f)) return V3(0.f);
    const Taps tx = filterTaps(k.filter, x), ty = filterTaps(k.filter, y);
    float r[3];
    for (int c = 0; c < 3; ++c) {
        r[c] = filteredTexel(im, tx, ty, c, false);
        if (k.srgb) r[c] = srgbToLinear(r[c]);                                   // texture.cpp:60-61
    }
    return V3(r[0], r[1], r[2]);
}
// Texture<Color3f>::eval(uv) by recursion over the node table, like the virtual calls of texture.cpp
static V3 textureEval(const Scene &sc, int t, float u, float v) {
    const KzTexture &k = sc.textures[t];
    switch (k.type) {
    case KZ_TEX_CONSTANT: return V3(k.color[0], k.color[1], k.color[2]);        // texture.cpp:16-18
    case KZ_TEX_IMAGE: return imageLookup(sc, k, u, v);
    case KZ_TEX_COLORRAMP: {                                                     // texture.cpp:162-172
        if (k.child[0] < 0) return V3(0.f);
        V3 c = textureEval(sc, k.child[0], u, v);
        auto ramp = [&](float in) { in = clampf(in, 0.0f, 1.0f); return k.rampMin + (k.rampMax - k.rampMin) * in; };
        return V3(ramp(c.x), ramp(c.y), ramp(c.z)); }
    default: {                                                                   // blend, texture.cpp:211-237
        V3 mask(0.5f), input1(0.f), input2(1.f);
        if (k.child[0] >= 0) mask = textureEval(sc, k.child[0], u, v);
        if (k.child[1] >= 0) input1 = textureEval(sc, k.child[1], u, v);
        if (k.child[2] >= 0) input2 = textureEval(sc, k.child[2], u, v);
        if (k.blendMode == KZ_BLEND_MIX) return V3(lerpf(mask.x, input1.x, input2.x), lerpf(mask.x, input1.y, input2.y), lerpf(mask.x, input1.z, input2.z));
        if (k.blendMode == KZ_BLEND_MULTIPLY) return V3(input1.x * input2.x, input1.y * input2.y, input1.z * input2.z);
        return V3(0.f); }
    }
}
// m_albedo->eval(bRec.uv) etc.: a texture id of 0 is the folded constanttexture
static inline V3 albedoAt(const KzBSDF &m, const BRec &b) { return m.albedoTex ? textureEval(*b.sc, m.albedoTex - 1, b.uvx, b.uvy) : V3(m.albedo[0], m.albedo[1], m.albedo[2]); }
static inline V3 baseColorAt(const KzBSDF &m, const BRec &b) { return m.albedoTex ? textureEval(*b.sc, m.albedoTex - 1, b.uvx, b.uvy) : V3(m.baseColor[0], m.baseColor[1], m.baseColor[2]); }
static inline float roughnessAt(const KzBSDF &m, const BRec &b) { return m.roughnessTex ? textureEval(*b.sc, m.roughnessTex - 1, b.uvx, b.uvy).x : m.roughness; }
static inline float metallicAt(const KzBSDF &m, const BRec &b) { return m.metallicTex ? textureEval(*b.sc, m.metallicTex - 1, b.uvx, b.uvy).x : m.metallic; }

// a20 Diffuse (bsdf.cpp:27-75); with a texture child: Lambertian (bsdf.cpp:202-276)
static V3 diffuseEval(const KzBSDF &m, const BRec &b) {
    if (b.measure != ESolidAngle || b.wi.z <= 0 || b.wo.z <= 0) return V3(0.f);
    return albedoAt(m, b) * INV_PI * b.wo.z;
}
static float diffusePdf(const KzBSDF &, const BRec &b) {
    if (b.measure != ESolidAngle || b.wi.z <= 0 || b.wo.z <= 0) return 0.f;
    return INV_PI * b.wo.z;
}
static V3 diffuseSample(const KzBSDF &m, BRec &b, float, float s2x, float s2y, bool &ok) {
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    b.measure = ESolidAngle;
    b.wo = squareToCosineHemisphere(s2x, s2y);
    b.eta = 1.0f;
    return albedoAt(m, b);
}

// a21 KazenStandardSurface (bsdf.cpp:1215-1267 eval, :1269-1299 pdf, :1301-1371 sample)
static V3 kissEval(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return V3(0.f);
    V3 V = b.wi, L = b.wo, H = normalized(V + L);
    V3 Cdlin = baseColorAt(m, b);
    float metallic = metallicAt(m, b);
    float roughness = std::min(1.f, roughnessAt(m, b) + b.accumulatedRoughness);
    float Cdlum = luminance(Cdlin);
    V3 Ctint = Cdlum > 0.f ? Cdlin / Cdlum : V3(1.f);
    V3 Ctintmix = 0.08f * m.specular * lerp3(V3(1.f), Ctint, m.specularTint);
    V3 Cspec0 = lerp3(Ctintmix, Cdlin, metallic);
    float FL = schlickWeight(L.z), FV = schlickWeight(V.z), FH = schlickWeight(dot(L, H));
    float cosThetaD = dot(V, H);
    float Lambert = (1.f - 0.5f * FL) * (1.f - 0.5f * FV);
    float RR = 2.f * roughness * cosThetaD * cosThetaD;
    float retro = RR * (FL + FV + FL * FV * (RR - 1.f));
    V3 Csheen = lerp3(V3(1.f), Ctint, m.sheenTint);
    V3 Fsheen = FH * m.sheen * Csheen;
    V3 specTerm = evaluateGGXSmithBRDF(V, L, Cspec0, roughness, m.anisotropy);
    float ccR = lerpf(m.clearcoatRoughness, .01f, .3f);
    V3 coatTerm = 0.25f * m.clearcoat * evaluateGGXSmithBRDF(V, L, V3(0.04f), ccR, m.anisotropy);
    return ((1.f - metallic) * (Cdlin * INV_PI * (Lambert + retro) + Fsheen) + (specTerm + coatTerm)) * b.wo.z;
}
static float kissPdf(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return 0.f;
    float diffuse = (1.f - metallicAt(m, b)) * 0.5f;
    float GTR2 = 1.f / (1.f + m.clearcoat);
    V3 H = normalized(b.wi + b.wo);
    float jacobian = 4.0f * dot(b.wi, H);
    float roughness = std::min(1.f, roughnessAt(m, b) + b.accumulatedRoughness);
    A2 alpha = roughnessToAlpha(roughness, m.anisotropy);
    float specPdf = ggxSmithVNDF(b.wi, H, alpha) / jacobian;
    A2 coatalpha = roughnessToAlpha(lerpf(m.clearcoatRoughness, .01f, .3f), 0.f);
    float coatPdf = ggxSmithVNDF(b.wi, H, coatalpha) / jacobian;
    return diffuse * INV_PI * b.wo.z + (1.f - diffuse) * (GTR2 * specPdf + (1.f - GTR2) * coatPdf);
}
static inline V3 reflect(V3 wi, V3 n) { return 2 * dot(n, wi) * n - wi; }    // common.cpp:536-538
static V3 kissSample(const KzBSDF &m, BRec &b, float sample1, float s2x, float s2y, bool &ok) {
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    b.measure = ESolidAngle; b.eta = 1.0f;
    float diffuse = (1.f - metallicAt(m, b)) * 0.5f;
    if (sample1 < diffuse) {
        b.wo = squareToCosineHemisphere(s2x, s2y);
    } else {
        float sample = (sample1 - diffuse) / (1.f - diffuse);
        float GTR2 = 1.f / (1.f + m.clearcoat);
        V3 H; bool flip = b.wi.z <= 0.f;
        A2 alpha = (sample < GTR2) ? roughnessToAlpha(roughnessAt(m, b), m.anisotropy)       // H7: no accumulatedRoughness
                                   : roughnessToAlpha(lerpf(m.clearcoatRoughness, 0.01f, .3f), 0.f);
        H = sampleGGXSmithVNDF(flip ? -b.wi : b.wi, alpha, s2x, s2y);
        H = flip ? -H : H;
        b.wo = normalized(reflect(b.wi, H));
    }
    bool invalid = std::isnan(b.wo.x) || std::isnan(b.wo.y) || std::isnan(b.wo.z);
    float pdf = kissPdf(m, b);
    if (b.wo.z <= 0 || pdf <= Epsilon || invalid) return V3(0.f);
    return kissEval(m, b) / pdf;
}

// fresnel (src/kazen/common.cpp:447-475) and refract (:526-534)
static float fresnelIOR(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) { std::swap(etaI, etaT); cosThetaI = -cosThetaI; }
    float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = std::sqrt(1.0f - sinThetaTSqr);
    float Rs = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    float Rp = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}
static V3 refractV(V3 wi, V3 n, float eta) {
    float cosThetaI = dot(wi, n);
    if (cosThetaI < 0) eta = 1.0f / eta;
    float cosThetaT2 = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
    if (cosThetaT2 <= 0.0f) return V3(0.0f);
    float sign = cosThetaI >= 0.0f ? 1.0f : -1.0f;
    return n * (-cosThetaI * eta + sign * std::sqrt(cosThetaT2)) + wi * eta;
}
// Mirror (bsdf.cpp:161-196) and Dielectric (bsdf.cpp:98-155): discrete lobes, eval = pdf = 0
static V3 mirrorSample(BRec &b, bool &ok) {
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    b.wo = V3(-b.wi.x, -b.wi.y, b.wi.z); b.measure = EDiscrete; b.eta = 1.0f;
    return V3(1.0f);
}
static V3 dielectricSample(const KzBSDF &m, BRec &b, float sample1, bool &ok) {
    ok = true;
    b.measure = EDiscrete;
    float cosThetaI = b.wi.z;
    float fresnelTerm = fresnelIOR(cosThetaI, m.extIOR, m.intIOR);
    if (sample1 < fresnelTerm) { b.wo = V3(-b.wi.x, -b.wi.y, b.wi.z); b.eta = 1.f; return V3(1.0f); }
    V3 n(0.0f, 0.0f, 1.0f);
    float factor = m.intIOR / m.extIOR;
    if (b.wi.z < 0.f) { factor = m.extIOR / m.intIOR; n.z = -1.0f; }
    b.wo = refractV(-b.wi, n, factor);
    b.eta = m.intIOR / m.extIOR;
    return V3(1.0f);
}

// ---- Beckmann helpers shared by roughconductor / roughplastic / roughdielectric --------------------------------------
static inline float tanTheta(V3 v) { float temp = 1 - v.z * v.z; if (temp <= 0.0f) return 0.0f; return std::sqrt(temp) / v.z; }   // frame.h:63-68
static inline float alphaOf(float x) { return x; }      // the row carries m_alpha = max(0.001, sqr(property)), formed at kzo_scene_create as the constructors form it: bsdf.cpp:699-701, :820-823, :958-960
static float evalBeckmann(V3 m, float alpha) {                                                                                    // bsdf.cpp:721-727
    float temp = tanTheta(m) / alpha, ct = m.z, ct2 = ct * ct;
    return kzoExp(-temp * temp) / (kPi * alpha * alpha * ct2 * ct2);
}
static float smithBeckmannG1(V3 v, V3 m, float alpha) {                                                                           // bsdf.cpp:730-750
    if (dot(v, m) * v.z <= 0.0f) return 0.0f;
    float tt = std::fabs(tanTheta(v));
    if (tt == 0.0f) return 1.0f;
    float a = 1.0f / (alpha * tt);
    if (a >= 1.6f) return 1.0f;
    float aSqr = a * a;
    return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
static V3 squareToBeckmann(float sx, float sy, float alpha) {                                                                     // warp.cpp:120-124
    float phi = 2 * kPi * sx;
    float theta = kzoAtan(alpha * std::sqrt(kzoLog(1 / (1 - sy))));
    float sinTheta, cosTheta, sinPhi, cosPhi; kzoSinCos(theta, &sinTheta, &cosTheta); kzoSinCos(phi, &sinPhi, &cosPhi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
static float squareToBeckmannPdf(V3 m, float alpha) {                                                                             // warp.cpp:126-129
    float theta = kzoAcos(m.z / norm(m));
    float ok = (std::fabs(norm(m) - 1) < Epsilon && m.z >= 0) ? 1.f : 0.f;
    const float tt = kzoTan(theta);                                       // pow(x, 2) is the rounded product, pow(x, 3) the rounded cube
    return ok * kzoExp(-(tt * tt) / (alpha * alpha)) / (kPi * alpha * alpha * kzoCube(kzoCos(theta)));
}
static float fresnelDielectricT(float cosThetaI_, float eta, float &cosThetaT_) {                                                 // common.cpp:492-518
    float scale = (cosThetaI_ > 0.f) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = std::fabs(cosThetaI_), cosThetaT = std::sqrt(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
// GGX ("ggx", bsdf.cpp:629-689) with a constant albedo; alpha field = "roughness"
static V3 ggxEval(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return V3(0.f);
    return evaluateGGXSmithBRDF(b.wi, b.wo, albedoAt(m, b), m.alpha, m.anisotropy) * b.wo.z;
}
static float ggxPdf(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return 0.f;
    V3 H = normalized(b.wi + b.wo);
    return ggxSmithVNDF(b.wi, H, roughnessToAlpha(m.alpha, m.anisotropy)) / (4.0f * dot(b.wi, H));
}
static V3 ggxSample(const KzBSDF &m, BRec &b, float s2x, float s2y, bool &ok) {                                                    // bsdf.cpp:658-669 + ggx_brdf.h:175-203
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    A2 alpha = roughnessToAlpha(m.alpha, m.anisotropy);
    V3 H = sampleGGXSmithVNDF(b.wi, alpha, s2x, s2y);            // wi.z > 0: no flip
    b.wo = reflect(b.wi, H);                                     // NOT normalised here (ggx_brdf.h:189)
    float pdf = ggxSmithVNDF(b.wi, H, alpha) / (4.0f * dot(b.wi, H));
    V3 color = evaluateGGXSmithBRDF(b.wi, b.wo, albedoAt(m, b), m.alpha, m.anisotropy);
    if (b.wo.z <= 0) return V3(0.f);
    return color * b.wo.z / pdf;                                 // measure stays EUnknownMeasure, eta stays 1
}
// RoughConductor (bsdf.cpp:692-811)
static V3 fresnelCond(float c, V3 eta, V3 k) {                                                                                    // bsdf.cpp:709-717
    V3 tmp_f = eta * eta + k * k;
    V3 tmp = tmp_f * (c * c);
    V3 twoEtaC = 2.f * eta * c;
    V3 a = tmp - twoEtaC + V3(1.f), bq = tmp + twoEtaC + V3(1.f);
    V3 Rparl2(a.x / bq.x, a.y / bq.y, a.z / bq.z);
    V3 c2(c * c);
    V3 e = tmp_f - twoEtaC + c2, f = tmp_f + twoEtaC + c2;
    V3 Rperp2(e.x / f.x, e.y / f.y, e.z / f.z);
    return (Rparl2 + Rperp2) / 2.0f;
}
static V3 rcondEval(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return V3(0.f);
    float alpha = alphaOf(m.alpha);
    V3 wh = normalized(b.wi + b.wo);
    V3 F = fresnelCond(dot(wh, b.wo), V3(m.condEta[0], m.condEta[1], m.condEta[2]), V3(m.condK[0], m.condK[1], m.condK[2]));
    float D = evalBeckmann(wh, alpha);
    float G = smithBeckmannG1(b.wi, wh, alpha) * smithBeckmannG1(b.wo, wh, alpha);
    return D * F * G / (4.f * b.wi.z);
}
static float rcondPdf(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return 0.f;
    V3 wh = normalized(b.wi + b.wo);
    float D = evalBeckmann(wh, alphaOf(m.alpha));
    float Jh = 1.f / (4.f * dot(wh, b.wo));
    return D * wh.z * Jh;
}
static V3 rcondSample(const KzBSDF &m, BRec &b, float s2x, float s2y, bool &ok) {
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    V3 wh = squareToBeckmann(s2x, s2y, alphaOf(m.alpha));
    b.wo = normalized(reflect(b.wi, wh));
    if (b.wo.z <= 0) return V3(0.f);
    return rcondEval(m, b) / rcondPdf(m, b);
}
// RoughPlastic (bsdf.cpp:814-943): ks = 1 - max(kd)
static V3 rplasEval(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return V3(0.f);
    float alpha = alphaOf(m.alpha);
    V3 kd(m.albedo[0], m.albedo[1], m.albedo[2]);
    float ks = 1 - maxCoeff(kd);
    V3 wh = normalized(b.wi + b.wo);
    float D = evalBeckmann(wh, alpha);
    float F = fresnelIOR(dot(wh, b.wo), m.extIOR, m.intIOR);
    float G = smithBeckmannG1(b.wo, wh, alpha) * smithBeckmannG1(b.wi, wh, alpha);
    return kd * INV_PI * b.wo.z + V3(ks * (D * F * G) / (4.f * b.wi.z));
}
static float rplasPdf(const KzBSDF &m, const BRec &b) {
    if (b.wi.z <= 0 || b.wo.z <= 0) return 0.f;
    V3 kd(m.albedo[0], m.albedo[1], m.albedo[2]);
    float ks = 1 - maxCoeff(kd);
    V3 wh = normalized(b.wi + b.wo);
    float D = evalBeckmann(wh, alphaOf(m.alpha));
    float Jh = 1.f / (4.f * std::fabs(dot(wh, b.wo)));           // H3: float abs
    return ks * D * wh.z * Jh + (1 - ks) * b.wo.z * INV_PI;
}
static V3 rplasSample(const KzBSDF &m, BRec &b, float s1, float s2x, float s2y, bool &ok) {
    if (b.wi.z <= 0) { ok = false; return V3(0.f); }
    ok = true;
    float ks = 1 - std::max(m.albedo[0], std::max(m.albedo[1], m.albedo[2]));
    if (s1 < ks) {
        V3 wh = squareToBeckmann(s2x, s2y, alphaOf(m.alpha));
        b.wo = normalized((2.f * dot(wh, b.wi) * wh) - b.wi);
    } else b.wo = squareToCosineHemisphere(s2x, s2y);
    if (b.wo.z <= 0) return V3(0.f);
    return rplasEval(m, b) / rplasPdf(m, b);
}
// RoughDielectric (bsdf.cpp:947-1145)
static inline float signf(float v) { return (v > 0.f) ? 1.f : -1.f; }                                                             // common.h:266-268
static V3 rdielEval(const KzBSDF &m, const BRec &b) {
    if (b.wi.z == 0) return V3(0.f);
    float alpha = alphaOf(m.alpha), mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
    float cosThetaI = b.wi.z, cosThetaO = b.wo.z;
    bool reflectI = cosThetaI * cosThetaO > 0.f;
    float eta = cosThetaI > 0.f ? mEta : mInvEta;
    V3 wm = reflectI ? normalized(b.wi + b.wo) : normalized(b.wi + b.wo * eta);
    wm = wm * signf(wm.z);
    float ct; float F = fresnelDielectricT(dot(b.wi, wm), mEta, ct);
    float D = evalBeckmann(wm, alpha);
    float G = smithBeckmannG1(b.wo, wm, alpha) * smithBeckmannG1(b.wi, wm, alpha);
    if (reflectI) return V3((F * G * D) / (4.f * std::fabs(cosThetaI)));
    float denom = dot(b.wi, wm) + eta * dot(b.wo, wm);
    float value = ((1 - F) * D * G * eta * eta * dot(b.wi, wm) * dot(b.wo, wm)) / (cosThetaI * sqr(denom));
    return V3(std::fabs(value));
}
static float rdielPdf(const KzBSDF &m, const BRec &b) {
    float mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
    float cosThetaI = b.wi.z, cosThetaO = b.wo.z;
    bool reflectI = cosThetaI * cosThetaO > 0.f;
    float eta = cosThetaI > 0.f ? mEta : mInvEta;
    V3 wm; float dwm_dwo;
    if (reflectI) { wm = normalized(b.wi + b.wo); dwm_dwo = 1.0f / (4.0f * dot(b.wo, wm)); }
    else { wm = normalized(b.wi + b.wo * eta); float sd = dot(b.wi, wm) + eta * dot(b.wo, wm); dwm_dwo = (eta * eta * dot(b.wo, wm)) / (sd * sd); }
    wm = wm * signf(wm.z);
    float ct; float F = fresnelDielectricT(dot(b.wi, wm), mEta, ct);
    float D = evalBeckmann(wm, alphaOf(m.alpha));
    float prob = D * wm.z;
    prob *= reflectI ? F : (1 - F);
    return std::fabs(prob * dwm_dwo);
}
static V3 rdielSample(const KzBSDF &m, BRec &b, float s1, float s2x, float s2y, bool &ok) {
    ok = true;
    float mEta = m.intIOR / m.extIOR, mInvEta = m.extIOR / m.intIOR;
    float alpha = alphaOf(m.alpha) * (1.2f - 0.2f * std::sqrt(std::fabs(b.wi.z)));
    V3 wm = squareToBeckmann(s2x, s2y, alpha);
    float pdf = squareToBeckmannPdf(wm, alpha);
    if (pdf == 0.f) return V3(0.f);
    float cosThetaT; float F = fresnelDielectricT(dot(b.wi, wm), mEta, cosThetaT);
    bool sampleReflection = !(s1 > F);
    if (sampleReflection) {
        b.wo = reflect(b.wi, wm); b.eta = 1.0f;
        if (b.wi.z * b.wo.z <= 0) return V3(0.f);
    } else {
        if (cosThetaT == 0) return V3(0.f);
        float e = mEta; if (cosThetaT < 0) e = 1.f / e;                           // RoughDielectric::refract, bsdf.cpp:1127-1132
        b.wo = wm * (dot(b.wi, wm) * e + cosThetaT) - b.wi * e;
        b.eta = cosThetaT < 0.f ? mEta : mInvEta;
        if (b.wi.z * b.wo.z >= 0) return V3(0.f);
    }
    float D = evalBeckmann(wm, alpha);
    float G = smithBeckmannG1(b.wo, wm, alpha) * smithBeckmannG1(b.wi, wm, alpha);
    return V3(std::fabs(D * G * dot(b.wi, wm) / (pdf * b.wi.z)));
}

static const KzBSDF &meshBsdf(const Scene &sc, int mesh) {
    static const KzBSDF dflt = {KZ_BSDF_DIFFUSE, {0.5f, 0.5f, 0.5f}, {0, 0, 0}, 0, 0, 0, 0.5f, 0.5f, 0, 0.5f, 0, 0.5f, 1.5046f, 1.000277f, 0.1f, {0, 0, 0}, {0, 0, 0}, 0, 0, 0, 0, 0, 0, 0};
    int b = sc.meshes[mesh].bsdf;
    return b < 0 ? dflt : sc.bsdfs[b];
}
static V3 nestedEval(const KzBSDF &m, const BRec &b) {
    if (m.type == KZ_BSDF_DIFFUSE) return diffuseEval(m, b);
    if (m.type == KZ_BSDF_KAZENSTANDARD) return kissEval(m, b);
    if (m.type == KZ_BSDF_GGX) return ggxEval(m, b);
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) return rcondEval(m, b);
    if (m.type == KZ_BSDF_ROUGHPLASTIC) return rplasEval(m, b);
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) return rdielEval(m, b);
    return V3(0.f);                                                     // discrete BRDFs evaluate to zero (bsdf.cpp:109-112,165-168)
}
static float nestedPdf(const KzBSDF &m, const BRec &b) {
    if (m.type == KZ_BSDF_DIFFUSE) return diffusePdf(m, b);
    if (m.type == KZ_BSDF_KAZENSTANDARD) return kissPdf(m, b);
    if (m.type == KZ_BSDF_GGX) return ggxPdf(m, b);
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) return rcondPdf(m, b);
    if (m.type == KZ_BSDF_ROUGHPLASTIC) return rplasPdf(m, b);
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) return rdielPdf(m, b);
    return 0.f;
}
static V3 nestedSample(const KzBSDF &m, BRec &b, float s1, float s2x, float s2y, bool &ok) {
    if (m.type == KZ_BSDF_DIFFUSE) return diffuseSample(m, b, s1, s2x, s2y, ok);
    if (m.type == KZ_BSDF_KAZENSTANDARD) return kissSample(m, b, s1, s2x, s2y, ok);
    if (m.type == KZ_BSDF_MIRROR) return mirrorSample(b, ok);
    if (m.type == KZ_BSDF_GGX) return ggxSample(m, b, s2x, s2y, ok);
    if (m.type == KZ_BSDF_ROUGHCONDUCTOR) return rcondSample(m, b, s2x, s2y, ok);
    if (m.type == KZ_BSDF_ROUGHPLASTIC) return rplasSample(m, b, s1, s2x, s2y, ok);
    if (m.type == KZ_BSDF_ROUGHDIELECTRIC) return rdielSample(m, b, s1, s2x, s2y, ok);
    return dielectricSample(m, b, s1, ok);
}

// NormalMap (bsdf.cpp:281-417): a BSDF wrapping a nested BSDF behind a frame perturbed by an RGB normal texture.
static Frame normalMapFrame(const Intersection &its, V3 n) {             // getFrame, "naive implementation" (bsdf.cpp:365-374)
    Frame result;
    result.n = normalized(its.shFrame.toWorld(n));
    result.s = normalized(its.dpdu - result.n * dot(result.n, its.dpdu));
    result.t = normalized(cross(result.n, result.s));
    return result;
}
static inline V3 normalMapN(const KzBSDF &m, const BRec &b) {            // bsdf.cpp:292-293: its.uv
    V3 rgb = textureEval(*b.sc, m.normalTex - 1, b.its->uvx, b.its->uvy);
    return V3(2 * rgb.x - 1, 2 * rgb.y - 1, 2 * rgb.z - 1);
}
static V3 normalMapEval(const Scene &sc, const KzBSDF &m, const BRec &b) {          // bsdf.cpp:290-312
    const KzBSDF &nested = sc.bsdfs[m.nested];
    const Intersection &its = *b.its;
    V3 n = normalMapN(m, b);
    if (b.wi.z > 0 && b.wo.z > 0 && dot(n, b.wi) <= 0) return nestedEval(nested, b);
    Frame pf = normalMapFrame(its, normalized(n));
    BRec q; q.wi = pf.toLocal(its.shFrame.toWorld(b.wi)); q.wo = pf.toLocal(its.shFrame.toWorld(b.wo)); q.measure = b.measure;
    if (b.wo.z * q.wo.z <= 0) return V3(0.0f);
    q.uvx = b.uvx; q.uvy = b.uvy; q.eta = b.eta; q.sc = b.sc;          // q.its stays default: accumulatedRoughness 0
    return nestedEval(nested, q);
}
static float normalMapPdf(const Scene &sc, const KzBSDF &m, const BRec &b) {        // bsdf.cpp:314-336
    const KzBSDF &nested = sc.bsdfs[m.nested];
    const Intersection &its = *b.its;
    V3 n = normalMapN(m, b);
    if (b.wi.z > 0 && b.wo.z > 0 && dot(n, b.wi) <= 0) return nestedPdf(nested, b);
    Frame pf = normalMapFrame(its, normalized(n));
    BRec q; q.wi = pf.toLocal(its.shFrame.toWorld(b.wi)); q.wo = pf.toLocal(its.shFrame.toWorld(b.wo)); q.measure = b.measure;
    if (b.wo.z * q.wo.z <= 0) return 0.0f;
    q.uvx = b.uvx; q.uvy = b.uvy; q.eta = b.eta; q.sc = b.sc;
    return nestedPdf(nested, q);
}
static V3 normalMapSample(const Scene &sc, const KzBSDF &m, BRec &b, float s1, float s2x, float s2y, bool &ok) {   // bsdf.cpp:338-363
    const KzBSDF &nested = sc.bsdfs[m.nested];
    const Intersection &its = *b.its;
    V3 n = normalMapN(m, b);
    if (b.wi.z > 0 && dot(n, b.wi) <= 0) { b.eta = 1.0f; return nestedSample(nested, b, s1, s2x, s2y, ok); }
    Frame pf = normalMapFrame(its, normalized(n));
    BRec q; q.wi = pf.toLocal(its.shFrame.toWorld(b.wi));
    q.uvx = its.uvx; q.uvy = its.uvy; q.measure = b.measure; q.eta = b.eta; q.sc = b.sc;
    V3 result = nestedSample(nested, q, s1, s2x, s2y, ok);
    if (ok && !(result.x == 0.f && result.y == 0.f && result.z == 0.f)) {
        b.wo = its.shFrame.toLocal(pf.toWorld(q.wo));
        b.eta = q.eta;                                                   // measure is NOT copied back: it stays EUnknownMeasure
        if (b.wo.z * q.wo.z <= 0) return V3(0.0f);
    }
    return result;
}

static V3 bsdfEval(const KzBSDF &m, const BRec &b) { return m.type == KZ_BSDF_NORMALMAP ? normalMapEval(*b.sc, m, b) : nestedEval(m, b); }
static float bsdfPdf(const KzBSDF &m, const BRec &b) { return m.type == KZ_BSDF_NORMALMAP ? normalMapPdf(*b.sc, m, b) : nestedPdf(m, b); }
static V3 bsdfSample(const KzBSDF &m, BRec &b, float s1, float s2x, float s2y, bool &ok) {
    return m.type == KZ_BSDF_NORMALMAP ? normalMapSample(*b.sc, m, b, s1, s2x, s2y, ok) : nestedSample(m, b, s1, s2x, s2y, ok);
}
// BSDF::regularize(uv): kiss returns m_roughness->eval(uv).r() (bsdf.cpp:1397-1399), NormalMap forwards to its nested BSDF
// (bsdf.cpp:411), every other model inherits 0 (bsdf.h:125)
static float bsdfRegularize(const Scene &sc, const KzBSDF &m_, float u, float v) {
    const KzBSDF &m = m_.type == KZ_BSDF_NORMALMAP ? sc.bsdfs[m_.nested] : m_;
    if (m.type != KZ_BSDF_KAZENSTANDARD) return 0.f;
    return m.roughnessTex ? textureEval(sc, m.roughnessTex - 1, u, v).x : m.roughness;
}

// ---------------------------------------------------------------------------------
// a4/a5/a9 samplers (src/kazen/sampler.cpp). H1: GCC evaluates call arguments right to
// left, so Independent::next2D draws y first, and bsdf->sample(bRec, next1D(), next2D())
// draws the 2-D sample first. Written here as sequenced statements.
// ---------------------------------------------------------------------------------
struct Sampler {
    const Scene *sc;
    int type;
    Pcg32 rng;
    int32_t px = 0, py = 0; uint32_t sampleIndex = 0, dim = 0;
    void generateSample(int32_t x, int32_t y, uint32_t idx) {
        px = x; py = y; sampleIndex = idx;
        if (type == KZ_SAMPLER_PMJ02BN) { dim = 2; return; }             // sampler.cpp:333-337 max(2, 0)
        rng.seed(HashPixelSeed(x, y, sc->smp.seed));                     // sampler.cpp:43-46, :111-117, :207-213
        rng.advance((int64_t)(idx * 65536ull + 0));
        dim = 0;
    }
    float bluenoise(uint32_t texIndex) const {                           // bluenoise.h:16-23
        int t = (int)texIndex % KZ_BLUENOISE_TEXTURES;
        int x = px % KZ_BLUENOISE_RES, y = py % KZ_BLUENOISE_RES;
        return sc->bnTable[((size_t)t * KZ_BLUENOISE_RES + x) * KZ_BLUENOISE_RES + y] / 65535.f;
    }
    void pmjSample(int setIndex, int sampleIdx, float &x, float &y) const {   // pmj02table.h:17-30
        setIndex %= KZ_PMJ02BN_SETS; sampleIdx %= KZ_PMJ02BN_SAMPLES;
        const uint32_t *e = &sc->pmjTable[((size_t)setIndex * KZ_PMJ02BN_SAMPLES + sampleIdx) * 2];
        x = (float)(e[0] * 0x1p-32); y = (float)(e[1] * 0x1p-32);
    }
    float next1D() {
        if (type == KZ_SAMPLER_INDEPENDENT) return rng.nextFloat();      // sampler.cpp:48-50
        if (type == KZ_SAMPLER_STRATIFIED) {                             // sampler.cpp:119-127
            uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);
            int stratum = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)hash);
            ++dim;
            float delta = rng.nextFloat();
            return (stratum + delta) / sc->sampleCount;
        }
        if (type == KZ_SAMPLER_CORRELATED) {                             // sampler.cpp:215-227
            uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);
            int p = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)(hash * 0x45fbe943));
            float j = rng.nextFloat();
            ++dim;
            return (p + j) / sc->sampleCount;
        }
        uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);     // sampler.cpp:339-347
        int index = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)hash);
        float delta = bluenoise(dim);
        ++dim;
        return std::min((index + delta) / sc->sampleCount, OneMinusEpsilon);
    }
    void next2D(float &x, float &y) {
        if (type == KZ_SAMPLER_INDEPENDENT) {                            // sampler.cpp:52-57, H1
            y = rng.nextFloat();
            x = rng.nextFloat();
            return;
        }
        if (type == KZ_SAMPLER_STRATIFIED) {                             // sampler.cpp:129-139 (dx, dy are sequenced statements)
            uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);
            int stratum = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)hash);
            dim += 2;
            int res = sc->resX;
            int sx = stratum % res, sy = stratum / res;
            float dx = rng.nextFloat();
            float dy = rng.nextFloat();
            x = (sx + dx) / res; y = (sy + dy) / res;
            return;
        }
        if (type == KZ_SAMPLER_CORRELATED) {                             // sampler.cpp:229-251 (Kensler CMJ)
            uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);
            int s = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)(hash * 0x51633e2d));
            uint32_t cy = (uint32_t)s / (uint32_t)sc->resX;
            uint32_t cx = (uint32_t)s % (uint32_t)sc->resX;
            uint32_t sx = permute(cx, (uint32_t)sc->resX, (uint32_t)(hash * 0x68bc21eb));
            uint32_t sy = permute(cy, (uint32_t)sc->resY, (uint32_t)(hash * 0x02e5be93));
            float jx = rng.nextFloat();
            float jy = rng.nextFloat();
            dim += 2;
            x = (cx + (sy + jx) / sc->resY) / sc->resX;
            y = (cy + (sx + jy) / sc->resX) / sc->resY;
            return;
        }
        int index = (int)sampleIndex;                                    // sampler.cpp:349-371
        int pmjInstance = (int)(dim / 2);
        if (pmjInstance >= KZ_PMJ02BN_SETS) {
            uint64_t hash = HashPixelDimSeed(px, py, dim, sc->smp.seed);
            index = (int)permute(sampleIndex, sc->sampleCount, (uint32_t)hash);
        }
        float ux, uy; pmjSample(pmjInstance, index, ux, uy);
        ux += bluenoise(dim); uy += bluenoise(dim + 1);
        if (ux >= 1) ux -= 1;
        if (uy >= 1) uy -= 1;
        dim += 2;
        x = std::min(ux, OneMinusEpsilon); y = std::min(uy, OneMinusEpsilon);
    }
    void nextPixel2D(float &x, float &y) {
        if (type != KZ_SAMPLER_PMJ02BN) { next2D(x, y); return; }        // sampler.cpp:59-61, :141-143, :253-255
        int tile = sc->pixelTileSize;                                    // sampler.cpp:373-377
        int tx = px % tile, ty = py % tile;
        size_t offset = (size_t)(tx + ty * tile) * sc->sampleCount + sampleIndex;
        x = sc->pixelSamples[2 * offset]; y = sc->pixelSamples[2 * offset + 1];
    }
};

// PMJ02BN ctor (sampler.cpp:275-315)
static int preparePmj(Scene &sc) {
    uint32_t spp = sc.smp.sampleCount;
    if (spp > KZ_PMJ02BN_SAMPLES) spp = KZ_PMJ02BN_SAMPLES;              // sampler.cpp:284-287
    sc.sampleCount = spp;
    int tile = pmjPixelTile(spp);
    sc.pixelTileSize = tile;
    size_t nPix = (size_t)tile * tile * spp;
    sc.pixelSamples.assign(nPix * 2, 0.f);
    std::vector<int> nStored((size_t)tile * tile, 0);
    Sampler tmp; tmp.sc = &sc; tmp.type = KZ_SAMPLER_PMJ02BN;
    for (int i = 0; i < KZ_PMJ02BN_SAMPLES; ++i) {
        float x, y; tmp.pmjSample(0, i, x, y);
        x *= tile; y *= tile;
        int ix = (int)x, iy = (int)y;
        if (ix >= tile || iy >= tile) return KZ_ERR_INVALID_ARG;        // table value rounds to 1.0f: the reference would index out of range
        int pixelOffset = ix + iy * tile;
        if (nStored[pixelOffset] == (int)spp) continue;
        size_t so = (size_t)pixelOffset * spp + nStored[pixelOffset];
        sc.pixelSamples[2 * so] = x - std::floor(x);
        sc.pixelSamples[2 * so + 1] = y - std::floor(y);
        ++nStored[pixelOffset];
    }
    return KZ_OK;
}

// ---------------------------------------------------------------------------------
// a3 camera (src/kazen/camera.cpp:35-68, :70-91; transform.h:49-62)
// ---------------------------------------------------------------------------------
static void mat4mul(const double *a, const double *b, double *c) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
        c[i * 4 + j] = s;
    }
}
static bool mat4inv(const double *m, double *out) {
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][j + 4] = (i == j); }
    for (int c = 0; c < 4; ++c) {
        int piv = c; for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
        double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = a[i][j + 4];
    return true;
}
static int prepareCamera(Scene &sc) {
    const KzCamera &c = sc.cam;
    sc.invW = 1.0f / (float)c.width; sc.invH = 1.0f / (float)c.height;         // camera.cpp:20
    if (c.sampleToCamera) { std::memcpy(sc.s2c, c.sampleToCamera, 64); return KZ_OK; }
    float aspect = c.width / (float)c.height;
    float recip = 1.0f / (c.farClip - c.nearClip);
    float cot = 1.0f / std::tan((c.fov / 2.0f) * (kPi / 180.0f));               // common.h:222 degToRad, M_PI the float of common.h:33
    // Eigen's Matrix4f::inverse() is not available here; the product and inverse are formed in
    // double and narrowed once (differences to Eigen's float cofactor inverse are last-ulp).
    double P[16] = {cot, 0, 0, 0, 0, cot, 0, 0, 0, 0, (double)(c.farClip * recip), (double)(-c.nearClip * c.farClip * recip), 0, 0, 1, 0};
    double T[16] = {1, 0, 0, -1, 0, 1, 0, (double)(-1.0f / aspect), 0, 0, 1, 0, 0, 0, 0, 1};
    double D[16] = {-0.5, 0, 0, 0, 0, (double)(-0.5f * aspect), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double TP[16], M[16], Mi[16];
    mat4mul(T, P, TP); mat4mul(D, TP, M);
    if (!mat4inv(M, Mi)) return KZ_ERR_INVALID_ARG;
    for (int i = 0; i < 16; ++i) sc.s2c[i] = (float)Mi[i];
    return KZ_OK;
}
static void cameraSampleRay(const Scene &sc, float sx, float sy, float ax, float ay, Ray &ray) {
    const float *m = sc.s2c;
    float x = sx * sc.invW, y = sy * sc.invH;
    float rx = m[0] * x + m[1] * y + m[2] * 0.0f + m[3];                         // transform.h:59-62
    float ry = m[4] * x + m[5] * y + m[6] * 0.0f + m[7];
    float rz = m[8] * x + m[9] * y + m[10] * 0.0f + m[11];
    float rw = m[12] * x + m[13] * y + m[14] * 0.0f + m[15];
    V3 nearP(rx / rw, ry / rw, rz / rw);
    const float *w = sc.cam.toWorld;
    V3 d;
    if (sc.cam.type == KZ_CAMERA_THINLENS) {                                     // camera.cpp:191-223
        float tx, ty; squareToUniformDisk(ax, ay, tx, ty);
        tx *= sc.cam.apertureRadius; ty *= sc.cam.apertureRadius;
        V3 apertureP(tx, ty, 0.0f);
        V3 focusP = nearP * (sc.cam.focusDistance / nearP.z);
        d = normalized(focusP - apertureP);
        float pw = w[12] * tx + w[13] * ty + w[14] * 0.0f + w[15];
        ray.o = V3((w[0] * tx + w[1] * ty + w[2] * 0.0f + w[3]) / pw, (w[4] * tx + w[5] * ty + w[6] * 0.0f + w[7]) / pw,
                   (w[8] * tx + w[9] * ty + w[10] * 0.0f + w[11]) / pw);
    } else {
        d = normalized(nearP);
        float ow = w[15];
        ray.o = V3(w[3] / ow, w[7] / ow, w[11] / ow);                            // toWorld * Point3f(0,0,0)
    }
    float invZ = 1.0f / d.z;
    ray.d = V3(w[0] * d.x + w[1] * d.y + w[2] * d.z, w[4] * d.x + w[5] * d.y + w[6] * d.z,
               w[8] * d.x + w[9] * d.y + w[10] * d.z);                           // transform.h:49-51
    ray.mint = sc.cam.nearClip * invZ; ray.maxt = sc.cam.farClip * invZ;
}

// ---------------------------------------------------------------------------------
// a26 filters (src/kazen/rfilter.cpp) + a25 table (block.cpp:13-21)
// ---------------------------------------------------------------------------------
static float filterEval(const KzFilter &f, float x) {
    switch (f.type) {
    case KZ_FILTER_GAUSSIAN: {
        float alpha = -1.0f / (2.0f * f.stddev * f.stddev);
        return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * f.radius * f.radius));
    }
    case KZ_FILTER_MITCHELL: {
        float B = f.B, C = f.C;
        x = std::fabs(2.0f * x / f.radius);
        float x2 = x * x, x3 = x2 * x;
        if (x < 1) return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
        else if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
        return 0.0f;
    }
    case KZ_FILTER_TENT: return std::max(0.0f, 1.0f - std::fabs(x));
    default: return 1.0f;
    }
}
static void prepareFilter(Scene &sc) {
    const KzFilter &f = sc.cam.rfilter;
    sc.filterRadius = f.radius;
    sc.border = (int)std::ceil(f.radius - 0.5f);
    for (int i = 0; i < KZ_FILTER_RESOLUTION; ++i) {
        float pos = (f.radius * i) / KZ_FILTER_RESOLUTION;
        sc.filter[i] = filterEval(f, pos);
    }
    sc.filter[KZ_FILTER_RESOLUTION] = 0.0f;
    sc.lookupFactor = KZ_FILTER_RESOLUTION / f.radius;
}

// a25 ImageBlock::put (block.cpp:56-85) on a film with its border. `film` is
// (h+2b) x (w+2b) float4 at offset (0,0); block-relative and image-relative positions
// are the same exact floats (see DESIGN.md "film").
struct Film { int w, h, b; std::vector<float> px; int cols() const { return w + 2 * b; } int rows() const { return h + 2 * b; } };
static inline bool colorValid(V3 c) {                                            // common.cpp:384-391
    for (int i = 0; i < 3; ++i) { float v = c[i]; if (v < 0 || !std::isfinite(v)) return false; }
    return true;
}
static bool filmPut(const Scene &sc, float *film, int cols, int rows, int offx, int offy, float sx, float sy, V3 value) {
    if (!colorValid(value)) return false;
    float posx = sx - 0.5f - (float)(offx - sc.border);
    float posy = sy - 0.5f - (float)(offy - sc.border);
    int x0 = (int)std::ceil(posx - sc.filterRadius), y0 = (int)std::ceil(posy - sc.filterRadius);
    int x1 = (int)std::floor(posx + sc.filterRadius), y1 = (int)std::floor(posy + sc.filterRadius);
    x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, cols - 1); y1 = std::min(y1, rows - 1);
    float wx[32], wy[32];
    for (int x = x0, i = 0; x <= x1; ++x) wx[i++] = sc.filter[(int)(std::fabs(x - posx) * sc.lookupFactor)];
    for (int y = y0, i = 0; y <= y1; ++y) wy[i++] = sc.filter[(int)(std::fabs(y - posy) * sc.lookupFactor)];
    for (int y = y0, yr = 0; y <= y1; ++y, ++yr)
        for (int x = x0, xr = 0; x <= x1; ++x, ++xr) {
            float *p = film + ((size_t)y * cols + x) * 4;
            // Color4f(value) * wx * wy : ((c * wx) * wy), block.cpp:84
            p[0] += value.x * wx[xr] * wy[yr];
            p[1] += value.y * wx[xr] * wy[yr];
            p[2] += value.z * wx[xr] * wy[yr];
            p[3] += 1.0f * wx[xr] * wy[yr];
        }
    return true;
}

// ---------------------------------------------------------------------------------
// a18/a19 lights (src/kazen/light.cpp, src/kazen/mesh.cpp:24-53,108-133, dpdf.h)
// ---------------------------------------------------------------------------------
static float dpdfNormalize(std::vector<float> &cdf, float *sumOut) {             // DiscretePDF::normalize, dpdf.h:77-89: returns m_normalization
    float sum = cdf.back();
    if (sumOut) *sumOut = sum;
    if (sum > 0) {
        float normalization = 1.0f / sum;
        for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
        cdf.back() = 1.0f;
        return normalization;
    }
    return 0.f;
}
static size_t dpdfSampleTable(const std::vector<float> &cdf, float v) {          // DiscretePDF::sample, dpdf.h:99-104
    auto entry = std::lower_bound(cdf.begin(), cdf.end(), v);
    ptrdiff_t idx = std::max((ptrdiff_t)0, (ptrdiff_t)(entry - cdf.begin()) - 1);
    return std::min((size_t)idx, cdf.size() - 2);
}
static void prepareLightMesh(MeshData &md) {
    md.cdf.clear(); md.cdf.push_back(0.0f);                                      // dpdf.h:23-27
    for (uint32_t i = 0; i < md.nF; ++i) {
        uint32_t i0 = md.F[3 * i], i1 = md.F[3 * i + 1], i2 = md.F[3 * i + 2];
        V3 p0 = md.v(i0), p1 = md.v(i1), p2 = md.v(i2);
        float area = 0.5f * norm(cross(p1 - p0, p2 - p0));                       // mesh.cpp:47-53
        md.cdf.push_back(md.cdf.back() + area);                                  // dpdf.h:35-37
    }
    md.normalization = dpdfNormalize(md.cdf, nullptr);
}
static size_t dpdfSample(const MeshData &md, float v) { return dpdfSampleTable(md.cdf, v); }
struct LRec { V3 ref, wi, p, n; Ray shadowRay; float pdf = 0; };
static inline V3 lightRadiance(const KzLight &l) { return l.intensity * V3(l.color[0], l.color[1], l.color[2]); }   // light.cpp:13
static V3 lightEval(const KzLight &l, const LRec &r) {                            // light.cpp:16-19
    float cosTheta = dot(r.n, -r.wi);
    return cosTheta > 0.f ? lightRadiance(l) : V3(0.f);
}
static float lightPdf(const MeshData &md, const LRec &r) {                        // light.cpp:36-51
    float pdf = md.normalization;
    float cosTheta = dot(r.n, -r.wi);
    if (cosTheta > 0.f) {
        V3 dd = r.p - r.ref;
        return pdf * dot(dd, dd) / cosTheta;
    }
    return 0.f;
}
static V3 lightSample(const Scene &sc, const KzLight &l, const MeshData &md, LRec &r, Sampler &s, LocalStats &ls) {   // light.cpp:21-34
    (void)sc;
    ls.lightSamples++;
    size_t index = dpdfSample(md, s.next1D());                                   // mesh.cpp:109
    float su0 = std::sqrt(s.next1D());
    float u = 1 - su0;
    float v = s.next1D() * su0;
    uint32_t i0 = md.F[3 * index], i1 = md.F[3 * index + 1], i2 = md.F[3 * index + 2];
    V3 p0 = md.v(i0), p1 = md.v(i1), p2 = md.v(i2);
    r.p = p0 + u * (p1 - p0) + v * (p2 - p0);
    if (!md.N.empty()) {
        V3 n0 = md.nrm(i0), n1 = md.nrm(i1), n2 = md.nrm(i2);
        r.n = n0 + u * (n1 - n0) + v * (n2 - n0);                                // H8: NOT normalised (mesh.cpp:129)
    } else r.n = normalized(cross(p1 - p0, p2 - p0));
    r.wi = normalized(r.p - r.ref);
    r.shadowRay = Ray(r.ref, r.wi, 0.f, norm(r.p - r.ref));
    r.pdf = lightPdf(md, r);
    if (r.pdf > 0.f && !std::isnan(r.pdf) && !std::isinf(r.pdf)) return lightEval(l, r) / r.pdf;
    return V3(0.f);
}

// ImageTexture::eval(Vector3f) (texture.cpp:66-80): OpenImageIO's TextureSystem::environment is not part of the checkout; the lookup is
// the one include/kazen_mi355x.h declares (y-up latitude-longitude map, bilinear, s periodic, t clamped; no scale, no toLinearRGB)
static V3 envLookup(const Scene &sc, int image, int filter, V3 d) {
    const Scene::Image &im = sc.images[image];
    float s = kzoAtan2(-d.x, d.z) / (2.0f * kPi) + 0.5f;
    float t = 0.5f - kzoAtan2(d.y, kzoHypot(d.z, -d.x)) / kPi;
    if (std::isnan(s)) s = 0.0f;
    if (std::isnan(t)) t = 0.0f;
    float x = s * (float)im.w - 0.5f, y = t * (float)im.h - 0.5f;
    const Taps tx = filterTaps(filter, x), ty = filterTaps(filter, y);
    float r[3];
    for (int c = 0; c < 3; ++c) r[c] = filteredTexel(im, tx, ty, c, true);
    return V3(r[0], r[1], r[2]);
}
static V3 backgroundColor(const Scene &sc, V3 dir) {                              // scene.cpp:54-79, texture.cpp:121-126
    if (!sc.bg.present) return V3(0.f);
    if (std::isnan(dir.x) || std::isnan(dir.y) || std::isnan(dir.z)) return V3(0.f);
    if (sc.bg.texture != 0) {                                                    // nested->eval(Vector3f) by texture class
        const KzTexture &k = sc.textures[sc.bg.texture - 1];
        if (k.type == KZ_TEX_CONSTANT) return sc.bg.intensity * V3(k.color[0], k.color[1], k.color[2]);       // texture.cpp:20-22
        if (k.type == KZ_TEX_IMAGE) return sc.bg.intensity * envLookup(sc, k.image, k.filter, dir);                    // texture.cpp:66-80
        return V3(0.f);                                                                                      // texture.h:13 (colorramp, blend)
    }
    return sc.bg.intensity * V3(sc.bg.color[0], sc.bg.color[1], sc.bg.color[2]);
}

static inline float powerHeuristic(float a, float b) {                            // integrator.cpp:340-344
    a *= a; b *= b;
    return a > 0.f ? a / (a + b) : 0.f;
}

// ---------------------------------------------------------------------------------
// a10  PathMisIntegrator::Li (src/kazen/integrator.cpp:195-338)
// ---------------------------------------------------------------------------------
static V3 Li(const Scene &sc, Sampler &sampler, const Ray &ray_, LocalStats &ls) {
    const int maxDepth = std::min(512, sc.integ.maxDepth);
    const float eps = sc.integ.traceBias;
    Ray ray = ray_;
    V3 L(0.f), throughput(1.f);
    float eta = 1.f, bsdfWeight = 1.f;
    Intersection its;
    if (!rayIntersect(sc, ray, its, false, ls)) return L;                          // H5
    if (sc.meshes[its.mesh].light >= 0 && !sc.lights[sc.meshes[its.mesh].light].primaryVisibility) {
        Ray newRay(its.p + eps * ray.d, ray.d);                                    // H6: result ignored on a miss
        rayIntersect(sc, newRay, its, false, ls);
    }
    int depth = 0;
    while (depth < maxDepth) {
        const MeshData &hitMesh = sc.meshes[its.mesh];
        if (hitMesh.light >= 0) {                                                  // integrator.cpp:226-231
            LRec lr; lr.ref = ray.o; lr.p = its.p; lr.n = its.shFrame.n; lr.wi = normalized(its.p - ray.o);
            L = L + bsdfWeight * throughput * lightEval(sc.lights[hitMesh.light], lr);
            break;
        }
        if (depth >= 3) {                                                          // integrator.cpp:237-244
            float probability = std::min(maxCoeff(throughput) * eta * eta, 0.95f);
            if (probability <= sampler.next1D()) break;
            throughput = throughput / probability;
        }
        const KzBSDF &bsdf = meshBsdf(sc, its.mesh);
        // ---- light sampling (integrator.cpp:247-295); the pick is drawn even with no lights (scene.h:45-53)
        float pick = sampler.next1D();
        size_t nl = sc.lightMeshes.size();
        if (nl > 0) {
            size_t li = std::min((size_t)std::floor(nl * pick), nl - 1);
            const MeshData &lm = sc.meshes[sc.lightMeshes[li]];
            const KzLight &light = sc.lights[lm.light];
            LRec lr; lr.ref = its.p;
            float lightPickPdf = 1.f / nl;                                         // scene.h:56
            V3 Ls = lightSample(sc, light, lm, lr, sampler, ls) / lightPickPdf;
            float lpdf = lightPdf(lm, lr);
            lr.shadowRay.mint = eps; lr.shadowRay.maxt -= eps;
            bool occluded = false;
            Ray tempRay = lr.shadowRay;
            bool walked = false;
            for (;;) {                                                             // integrator.cpp:262-278
                Intersection sh;
                // After a walk-through the far end is maxt - t while the origin moved on by t + eps: in real arithmetic the
                // segment now ends exactly ON the sampled light, so whether that light itself is reported is decided by the
                // rounding of everything upstream (a tie in the reference, not a property of the scene). tieMode +1 / -1 move the
                // far end out / in by 1e-5 of its length and so decide every such tie one way: the two renders bracket
                // every rounding of the literal loop (tests only; 0 = the literal loop).
                Ray q = tempRay;
                if (walked && sc.tieMode != 0) q.maxt *= sc.tieMode > 0 ? 1.00001f : 0.99999f;
                if (rayIntersect(sc, q, sh, true, ls)) {
                    const MeshData &om = sc.meshes[sh.mesh];
                    if (om.light < 0) { occluded = true; break; }
                    if (sc.lights[om.light].primaryVisibility) { occluded = true; break; }
                    tempRay = Ray(tempRay.o + tempRay.d * (sh.t + eps), tempRay.d, eps, tempRay.maxt - sh.t);
                    walked = true;
                } else break;
            }
            if (!occluded) {
                BRec b; b.wi = its.shFrame.toLocal(-ray.d); b.wo = its.shFrame.toLocal(lr.wi); b.measure = ESolidAngle;
                b.accumulatedRoughness = its.accumulatedRoughness; b.its = &its; b.uvx = its.uvx; b.uvy = its.uvy; b.sc = &sc;   // integrator.cpp:284-285
                V3 f = bsdfEval(bsdf, b);
                float bpdf = bsdfPdf(bsdf, b);
                float lightWeight = powerHeuristic(lpdf, bpdf);
                L = L + throughput * Ls * f * lightWeight;
            }
        }
        if (sc.integ.regularization)                                               // integrator.cpp:298-301
            its.accumulatedRoughness += bsdfRegularize(sc, bsdf, its.uvx, its.uvy) * sc.integ.accumulatedRoughness;
        // ---- BSDF sampling (integrator.cpp:304-309). H1: next2D is drawn before next1D.
        BRec b; b.wi = its.shFrame.toLocal(-ray.d); b.accumulatedRoughness = its.accumulatedRoughness;
        b.its = &its; b.uvx = its.uvx; b.uvy = its.uvy; b.sc = &sc;                  // integrator.cpp:305-306
        float s2x, s2y; sampler.next2D(s2x, s2y);
        float s1 = sampler.next1D();
        bool ok;
        V3 bsdfColor = bsdfSample(bsdf, b, s1, s2x, s2y, ok);
        throughput = throughput * bsdfColor;
        eta *= b.eta;
        // A zero weight kills the path: the reference keeps iterating with throughput 0 (and, when
        // sample() bailed out early, an uninitialised bRec.wo), which contributes exactly 0 and stops
        // at the next roulette. Terminating here is the defined equivalent.
        if (!ok || (bsdfColor.x == 0.f && bsdfColor.y == 0.f && bsdfColor.z == 0.f)) break;
        ray = Ray(its.p, its.shFrame.toWorld(b.wo));                                // H9: not re-normalised
        ray.mint = eps;
        float bpdf = bsdfPdf(bsdf, b);
        if (!rayIntersect(sc, ray, its, false, ls)) {
            L = L + throughput * backgroundColor(sc, ray.d);
            break;
        }
        const MeshData &nm = sc.meshes[its.mesh];
        if (nm.light >= 0) {                                                       // integrator.cpp:322-327
            LRec lr; lr.ref = ray.o; lr.p = its.p; lr.n = its.shFrame.n; lr.wi = normalized(its.p - ray.o);
            float lpdf = lightPdf(nm, lr);
            bsdfWeight = powerHeuristic(bpdf, lpdf);
        }
        if (b.measure == EDiscrete) bsdfWeight = 1.f;
        depth++;
    }
    return L;
}

// a2 renderSample (renderer.cpp:20-40) -> (pixelSample, value)
static V3 renderSample(const Scene &sc, Sampler &sampler, int px, int py, uint32_t j, float &sx, float &sy, LocalStats &ls) {
    sampler.generateSample(px, py, j);
    float jx, jy; sampler.nextPixel2D(jx, jy);
    sx = (float)px + jx; sy = (float)py + jy;
    float ax, ay; sampler.next2D(ax, ay);      // aperture sample: always drawn (renderer.cpp:28)
    Ray ray; cameraSampleRay(sc, sx, sy, ax, ay, ray);
    ls.samples++;
    return Li(sc, sampler, ray, ls);           // camera weight is Color3f(1) (camera.cpp:90)
}

static void mergeStats(Scene &sc, const LocalStats &ls) {
    sc.stats.samples += ls.samples; sc.stats.rays += ls.rays; sc.stats.nodeVisits += ls.nodeVisits;
    sc.stats.triTests += ls.triTests; sc.stats.shadedHits += ls.shadedHits; sc.stats.lightSamples += ls.lightSamples;
    sc.stats.dropped += ls.dropped;
}

} // namespace kzo

// =====================================================================================
// C entry points (prefix kzo_). Loaded with ctypes by tests/ and bench.py only.
// =====================================================================================
using namespace kzo;

static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) { std::snprintf(g_err, sizeof g_err, "%s", msg); return code; }

extern "C" {

const char *kzo_last_error() { return g_err; }

int kzo_scene_create(const KzSceneDesc *d, int useBrute, void **out) { FtzScope ftz_;
    if (!d || !out) return fail(KZ_ERR_INVALID_ARG, "null argument");
    if (d->abiVersion != KZ_ABI_VERSION) return fail(KZ_ERR_INVALID_ARG, "abi version mismatch");
    if (d->camera.type != KZ_CAMERA_PERSPECTIVE && d->camera.type != KZ_CAMERA_THINLENS) return fail(KZ_ERR_UNSUPPORTED, "camera type");
    if (d->integrator.type != KZ_INTEGRATOR_PATH_MIS) return fail(KZ_ERR_UNSUPPORTED, "integrator type");
    if (d->sampler.type < KZ_SAMPLER_INDEPENDENT || d->sampler.type > KZ_SAMPLER_CORRELATED) return fail(KZ_ERR_UNSUPPORTED, "sampler type");
    Scene *sc = new Scene();
    sc->cam = d->camera; sc->smp = d->sampler; sc->integ = d->integrator; sc->bg = d->background;
    sc->bsdfs.assign(d->bsdfs, d->bsdfs + d->nBsdfs);
    sc->lights.assign(d->lights, d->lights + d->nLights);
    if (d->nTextures) sc->textures.assign(d->textures, d->textures + d->nTextures);
    for (uint32_t i = 0; i < d->nImages; ++i) {
        const KzImage &im = d->images[i];
        if (!im.pixels || im.width <= 0 || im.height <= 0 || im.channels <= 0 || (im.format != KZ_PIXEL_U8 && im.format != KZ_PIXEL_F32)) { delete sc; return fail(KZ_ERR_INVALID_ARG, "image"); }
        Scene::Image o; o.w = im.width; o.h = im.height; o.c = im.channels; o.fmt = im.format;
        size_t bytes = (size_t)im.width * im.height * im.channels * (im.format == KZ_PIXEL_F32 ? 4 : 1);
        o.px.assign((const uint8_t *)im.pixels, (const uint8_t *)im.pixels + bytes);
        sc->images.push_back(std::move(o));
    }
    for (auto &t : sc->textures) {
        if (t.type < KZ_TEX_CONSTANT || t.type > KZ_TEX_BLEND) { delete sc; return fail(KZ_ERR_UNSUPPORTED, "texture type"); }
        if (t.type == KZ_TEX_IMAGE && (t.image < 0 || t.image >= (int)d->nImages)) { delete sc; return fail(KZ_ERR_INVALID_ARG, "texture image index"); }
        for (int c = 0; c < 3; ++c) if (t.child[c] >= (int)d->nTextures) { delete sc; return fail(KZ_ERR_INVALID_ARG, "texture child index"); }
    }
    for (auto &b : sc->bsdfs) {
        if (b.type < KZ_BSDF_DIFFUSE || b.type > KZ_BSDF_NORMALMAP) { delete sc; return fail(KZ_ERR_UNSUPPORTED, "bsdf type"); }
        const bool rough = b.type == KZ_BSDF_ROUGHCONDUCTOR || b.type == KZ_BSDF_ROUGHPLASTIC || b.type == KZ_BSDF_ROUGHDIELECTRIC;
        if ((b.alphaResolved != 0 && b.alphaResolved != 1) || (b.alphaResolved && !rough)) { delete sc; return fail(KZ_ERR_INVALID_ARG, "alphaResolved"); }
        if (rough && !b.alphaResolved) { b.alpha = std::max(0.001f, sqr(b.alpha)); b.alphaResolved = 1; }                              // bsdf.cpp:696-700, :818-822, :956-959
        const int ids[4] = {b.albedoTex, b.roughnessTex, b.metallicTex, b.normalTex};
        for (int id : ids) if (id < 0 || id > (int)d->nTextures) { delete sc; return fail(KZ_ERR_INVALID_ARG, "bsdf texture id"); }
        if (b.type == KZ_BSDF_NORMALMAP && (b.normalTex == 0 || b.nested < 0 || b.nested >= (int)d->nBsdfs || d->bsdfs[b.nested].type == KZ_BSDF_NORMALMAP)) { delete sc; return fail(KZ_ERR_INVALID_ARG, "normalmap row"); }
    }
    sc->meshes.resize(d->nMeshes);
    for (uint32_t m = 0; m < d->nMeshes; ++m) {
        const KzMesh &km = d->meshes[m]; MeshData &md = sc->meshes[m];
        if (!km.V || !km.F) { delete sc; return fail(KZ_ERR_INVALID_ARG, "mesh without V/F"); }
        md.nV = km.nV; md.nF = km.nF; md.bsdf = km.bsdf; md.light = km.light;
        if (km.bsdf >= (int)d->nBsdfs || km.light >= (int)d->nLights) { delete sc; return fail(KZ_ERR_INVALID_ARG, "bsdf/light index"); }
        md.V.assign(km.V, km.V + 3 * (size_t)km.nV);
        if (km.N) md.N.assign(km.N, km.N + 3 * (size_t)km.nV);
        if (km.UV) md.UV.assign(km.UV, km.UV + 2 * (size_t)km.nV);
        md.F.assign(km.F, km.F + 3 * (size_t)km.nF);
        for (uint32_t i = 0; i < 3 * km.nF; ++i) if (md.F[i] >= km.nV) { delete sc; return fail(KZ_ERR_INVALID_ARG, "face index out of range"); }
        if (md.light >= 0) { prepareLightMesh(md); sc->lightMeshes.push_back((int)m); }
    }
    sc->sampleCount = d->sampler.sampleCount;
    if (d->sampler.type == KZ_SAMPLER_STRATIFIED) {                                       // sampler.cpp:84-92
        int res = d->sampler.resolution;
        while ((uint32_t)(res * res) < sc->sampleCount) res++;
        sc->resX = sc->resY = res; sc->sampleCount = (uint32_t)(res * res);
    }
    if (d->sampler.type == KZ_SAMPLER_CORRELATED) {                                       // sampler.cpp:179-187
        int r1 = (int)std::sqrt((double)sc->sampleCount);
        int r0 = (int)((sc->sampleCount + r1 - 1) / r1);
        sc->resX = r0; sc->resY = r1; sc->sampleCount = (uint32_t)(r0 * r1);
    }
    if (d->sampler.type == KZ_SAMPLER_PMJ02BN) {
        if (!d->sampler.pmj02bnSamples || !d->sampler.blueNoise) { delete sc; return fail(KZ_ERR_INVALID_ARG, "pmj02bn tables missing"); }
        sc->pmjTable.assign(d->sampler.pmj02bnSamples, d->sampler.pmj02bnSamples + (size_t)KZ_PMJ02BN_SETS * KZ_PMJ02BN_SAMPLES * 2);
        sc->bnTable.assign(d->sampler.blueNoise, d->sampler.blueNoise + (size_t)KZ_BLUENOISE_TEXTURES * KZ_BLUENOISE_RES * KZ_BLUENOISE_RES);
        int rc = preparePmj(*sc);
        if (rc) { delete sc; return fail(rc, "pmj02bn table value rounds to 1.0f"); }
    }
    sc->smp.pmj02bnSamples = nullptr; sc->smp.blueNoise = nullptr; sc->cam.sampleToCamera = d->camera.sampleToCamera;
    int rc = prepareCamera(*sc);
    sc->cam.sampleToCamera = nullptr;
    if (rc) { delete sc; return fail(rc, "singular camera matrix"); }
    prepareFilter(*sc);
    sc->useBrute = useBrute != 0;
    Builder b(*sc); b.run();
    *out = sc;
    return KZ_OK;
}
void kzo_scene_destroy(void *s) { delete (Scene *)s; }
void kzo_set_brute(void *s, int brute) { ((Scene *)s)->useBrute = brute != 0; }
void kzo_set_tie_mode(void *s, int mode) { ((Scene *)s)->tieMode = mode; }
// the transcendental functions of kz_oracle_math.h on arrays (same numbering as kz_kat_math of the dev header)
void kzo_math(int fn, uint32_t n, const float *x, const float *y, float *out) { FtzScope ftz_;
    for (uint32_t i = 0; i < n; ++i) {
        const float a = x[i], b = y ? y[i] : x[i];
        float r, s, c;
        switch (fn) {
        case 0: kzoSinCos(a, &s, &c); r = s; break;
        case 1: kzoSinCos(a, &s, &c); r = c; break;
        case 2: r = kzoExp(a); break;
        case 3: r = kzoLog(a); break;
        case 4: r = kzoAtan(a); break;
        case 5: r = kzoAtan2(a, b); break;
        case 6: r = kzoAcos(a); break;
        case 7: r = kzoTan(a); break;
        case 8: r = kzoPow(a, b); break;
        case 9: r = kzoHypot(a, b); break;
        case 10: r = kzoCube(a); break;
        case 11: r = kzoCos(a); break;
        // 100 + fn: the libm float function the reference's text calls, on this machine (for the census of tests/test_oracle_cpu.py)
        case 100: r = std::sin(a); break;
        case 101: r = std::cos(a); break;
        case 102: r = std::exp(a); break;
        case 103: r = std::log(a); break;
        case 104: r = std::atan(a); break;
        case 105: r = std::atan2(a, b); break;
        case 106: r = std::acos(a); break;
        case 107: r = std::tan(a); break;
        case 108: r = std::pow(a, b); break;
        case 109: r = std::hypot(a, b); break;
        case 110: r = std::pow(a, 3.0f); break;
        default: r = 0.f; break;
        }
        out[i] = r;
    }
}

unsigned kzo_sample_count(void *s) { return ((Scene *)s)->sampleCount; }
int kzo_film_dims(void *s, int *w, int *h, int *b) { Scene *sc = (Scene *)s; *w = sc->cam.width; *h = sc->cam.height; *b = sc->border; return 0; }

int kzo_bvh_info(void *s, KzBvhInfo *o) {
    Scene *sc = (Scene *)s; std::memset(o, 0, sizeof *o);
    o->nNodes = (uint32_t)sc->nodes.size(); o->nTris = (uint32_t)sc->tris.size(); o->maxDepth = sc->maxDepth;
    return 0;
}

// Render sample indices [s0,s1) of the tiles (or the whole image) into `film`
// ((h+2b) x (w+2b) float4, ADDED to its contents). Decomposition follows
// renderer.cpp:94-127: 32x32 blocks, one local ImageBlock each, rendered by `threads`
// std::threads; blocks are merged in row-major block order (H10: the reference's merge
// order is nondeterministic).
int kzo_render(void *s, uint32_t s0, uint32_t s1, const KzTile *tiles, uint32_t nTiles, int threads, float *film) {
    Scene *scp = (Scene *)s; if (!scp || !film) return fail(KZ_ERR_INVALID_ARG, "null");
    Scene &sc = *scp;
    if (s0 == 0 && s1 == 0) s1 = sc.sampleCount;
    const int W = sc.cam.width, H = sc.cam.height, B = sc.border;
    const int cols = W + 2 * B, rows = H + 2 * B;
    struct Blk { int x0, y0, w, h; };
    std::vector<Blk> blocks;
    KzTile whole = {0, 0, W, H};
    if (!tiles) { tiles = &whole; nTiles = 1; }
    const int BS = 32;
    for (uint32_t t = 0; t < nTiles; ++t) {
        KzTile tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.x0 + tl.w > W || tl.y0 + tl.h > H) return fail(KZ_ERR_INVALID_ARG, "tile out of image");
        for (int by = tl.y0; by < tl.y0 + tl.h; by += BS)
            for (int bx = tl.x0; bx < tl.x0 + tl.w; bx += BS)
                blocks.push_back(Blk{bx, by, std::min(BS, tl.x0 + tl.w - bx), std::min(BS, tl.y0 + tl.h - by)});
    }
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    threads = std::max(1, std::min(threads, (int)blocks.size()));
    const int bc = BS + 2 * B;
    std::vector<std::vector<float>> local(blocks.size());
    std::atomic<size_t> next{0};
    auto work = [&]() {
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);          // main.cpp:22-23 (H11)
        _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
        Sampler sampler; sampler.sc = &sc; sampler.type = sc.smp.type;
        LocalStats ls;
        for (;;) {
            size_t bi = next.fetch_add(1);
            if (bi >= blocks.size()) break;
            const Blk &bk = blocks[bi];
            std::vector<float> &lf = local[bi];
            lf.assign((size_t)bc * bc * 4, 0.f);
            uint32_t pixelCount = (uint32_t)(bk.w * bk.h);
            for (uint32_t i = 0; i < pixelCount; ++i) {                        // renderer.cpp:52-68
                int px = (int)(i % bk.w) + bk.x0, py = (int)(i / bk.w) + bk.y0;
                for (uint32_t j = s0; j < s1; ++j) {
                    float sx, sy;
                    V3 value = renderSample(sc, sampler, px, py, j, sx, sy, ls);
                    if (!filmPut(sc, lf.data(), bc, bc, bk.x0, bk.y0, sx, sy, value)) ls.dropped++;
                }
            }
        }
        mergeStats(sc, ls);
    };
    FtzScope ftz;                                                           // worker threads inherit it
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (size_t bi = 0; bi < blocks.size(); ++bi) {                             // ImageBlock::put(ImageBlock&), block.cpp:87-96
        const Blk &bk = blocks[bi];
        int sw = bk.w + 2 * B, sh = bk.h + 2 * B;
        for (int y = 0; y < sh; ++y)
            for (int x = 0; x < sw; ++x) {
                const float *src = &local[bi][((size_t)y * bc + x) * 4];
                float *dst = &film[((size_t)(bk.y0 + y) * cols + (bk.x0 + x)) * 4];
                dst[0] += src[0]; dst[1] += src[1]; dst[2] += src[2]; dst[3] += src[3];
            }
        (void)rows;
    }
    return KZ_OK;
}

// The same render with the film's float additions in the order the HIP build FIXES (the reference's own order depends on its thread timing, so any fixed
// order is a legitimate one): per pixel and filter tap the weighted samples are added in sample order - ImageBlock::put's weights and products, block.cpp:64-84,
// with positions relative to the reference's 32 x 32 block - and a film texel is the sum, over the cells of a `grid`-pixel tile grid in row-major order, of each cell's
// partial sum (taps in (row, column) order: ImageBlock::put(ImageBlock&) of the grid's blocks in tile order, block.cpp:87-96). With this order the HIP film and the
// oracle's are the same BITS, not merely close (tests/test_gpu_parity.py::test_whole_films_equal_the_oracle_bit_for_bit). `film` is overwritten.
int kzo_render_canonical(void *s, uint32_t s0, uint32_t s1, const KzTile *tiles, uint32_t nTiles, int threads, int grid, float *film) {
    Scene *scp = (Scene *)s; if (!scp || !film || grid <= 0) return fail(KZ_ERR_INVALID_ARG, "null");
    Scene &sc = *scp;
    if (s0 == 0 && s1 == 0) s1 = sc.sampleCount;
    const int W = sc.cam.width, H = sc.cam.height, B = sc.border;
    const int cols = W + 2 * B, rows = H + 2 * B;
    const float r = sc.filterRadius, lf = sc.lookupFactor;
    const int tapLo = (int)std::floor(-r - 0.5f) + 1, tapHi = (int)std::floor(r + 0.5f), taps = tapHi - tapLo + 1;
    if (taps < 1 || taps > 9) return fail(KZ_ERR_UNSUPPORTED, "filter taps");
    const size_t framePix = (size_t)W * H;
    std::vector<float> tapSums((size_t)taps * taps * framePix * 4, 0.f);           // [tap][y * W + x][rgb w, w]
    struct Blk { int x0, y0, w, h; };
    std::vector<Blk> blocks;
    KzTile whole = {0, 0, W, H};
    if (!tiles) { tiles = &whole; nTiles = 1; }
    const int BS = 32;
    for (uint32_t t = 0; t < nTiles; ++t) {
        KzTile tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.x0 + tl.w > W || tl.y0 + tl.h > H) return fail(KZ_ERR_INVALID_ARG, "tile out of image");
        for (int by = tl.y0; by < tl.y0 + tl.h; by += BS)
            for (int bx = tl.x0; bx < tl.x0 + tl.w; bx += BS)
                blocks.push_back(Blk{bx, by, std::min(BS, tl.x0 + tl.w - bx), std::min(BS, tl.y0 + tl.h - by)});
    }
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    threads = std::max(1, std::min(threads, (int)blocks.size()));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);          // main.cpp:22-23 (H11)
        _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
        Sampler sampler; sampler.sc = &sc; sampler.type = sc.smp.type;
        LocalStats ls;
        std::vector<float> acc((size_t)taps * taps * 4);
        for (;;) {
            size_t bi = next.fetch_add(1);
            if (bi >= blocks.size()) break;
            const Blk &bk = blocks[bi];
            for (int i = 0; i < bk.w * bk.h; ++i) {
                const int px = i % bk.w + bk.x0, py = i / bk.w + bk.y0;
                const int bx0 = px & ~31, by0 = py & ~31;                       // the reference block this pixel is rendered in (KAZEN_BLOCK_SIZE)
                std::fill(acc.begin(), acc.end(), 0.f);
                for (uint32_t j = s0; j < s1; ++j) {
                    float sx, sy;
                    const V3 value = renderSample(sc, sampler, px, py, j, sx, sy, ls);
                    if (!colorValid(value)) { ls.dropped++; continue; }
                    const float posx = sx - 0.5f - (float)(bx0 - B), posy = sy - 0.5f - (float)(by0 - B);                     // block.cpp:64-67
                    const float lox = std::ceil(posx - r), hix = std::floor(posx + r), loy = std::ceil(posy - r), hiy = std::floor(posy + r);   // block.cpp:70-73
                    float wx[9], wy[9];
                    for (int t = 0; t < taps; ++t) {                            // tap t reaches the (block-relative) film coordinate px + B - tapLo - t
                        const float xb = (float)(px + B - tapLo - t - bx0), yb = (float)(py + B - tapLo - t - by0);
                        wx[t] = !(xb < lox || xb > hix) ? sc.filter[(int)(std::fabs(xb - posx) * lf)] : 0.f;               // block.cpp:77-80
                        wy[t] = !(yb < loy || yb > hiy) ? sc.filter[(int)(std::fabs(yb - posy) * lf)] : 0.f;
                    }
                    for (int ty = 0; ty < taps; ++ty)
                        for (int tx = 0; tx < taps; ++tx) {
                            float *a = &acc[(size_t)(ty * taps + tx) * 4];
                            a[0] += value.x * wx[tx] * wy[ty]; a[1] += value.y * wx[tx] * wy[ty]; a[2] += value.z * wx[tx] * wy[ty]; a[3] += 1.0f * wx[tx] * wy[ty];   // block.cpp:84
                        }
                }
                for (int k = 0; k < taps * taps; ++k) std::memcpy(&tapSums[((size_t)k * framePix + (size_t)py * W + px) * 4], &acc[(size_t)k * 4], 4 * sizeof(float));
            }
        }
        mergeStats(sc, ls);
    };
    FtzScope ftz;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    // the resolve: texel = sum over grid cells (row-major) of the cell's partial sum, taps in (row, column) order
    for (int fy = 0; fy < rows; ++fy)
        for (int fx = 0; fx < cols; ++fx) {
            const int x0s = fx - B + tapLo, y0s = fy - B + tapLo;                // the source pixel that reaches this texel through tap (0, 0)
            const int xlo = std::max(x0s, 0), xhi = std::min(x0s + taps - 1, W - 1), ylo = std::max(y0s, 0), yhi = std::min(y0s + taps - 1, H - 1);
            float total[4] = {0.f, 0.f, 0.f, 0.f};
            if (xlo <= xhi && ylo <= yhi)
                for (int tr = ylo / grid; tr <= yhi / grid; ++tr)
                    for (int tc = xlo / grid; tc <= xhi / grid; ++tc) {
                        float part[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int y = std::max(ylo, tr * grid); y <= std::min(yhi, tr * grid + grid - 1); ++y)
                            for (int x = std::max(xlo, tc * grid); x <= std::min(xhi, tc * grid + grid - 1); ++x) {
                                const float *t = &tapSums[((size_t)((y - y0s) * taps + (x - x0s)) * framePix + (size_t)y * W + x) * 4];
                                part[0] += t[0]; part[1] += t[1]; part[2] += t[2]; part[3] += t[3];
                            }
                        total[0] += part[0]; total[1] += part[1]; total[2] += part[2]; total[3] += part[3];
                    }
            std::memcpy(&film[((size_t)fy * cols + fx) * 4], total, sizeof total);
        }
    return KZ_OK;
}

// ImageBlock::toBitmap (block.cpp:39-45) + Color4f::divideByFilterWeight (color.h:94-99)
int kzo_film_to_rgb(const float *film, int w, int h, int b, float *rgb) { FtzScope ftz_;
    int cols = w + 2 * b;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const float *p = &film[((size_t)(y + b) * cols + (x + b)) * 4];
        float *o = &rgb[((size_t)y * w + x) * 3];
        if (p[3] != 0) { o[0] = p[0] / p[3]; o[1] = p[1] / p[3]; o[2] = p[2] / p[3]; }
        else { o[0] = o[1] = o[2] = 0.f; }
    }
    return 0;
}

// Bitmap::savePNG's raster (bitmap.cpp:45-52) from a normalised bitmap: Color3f::toSRGB (common.cpp:351-366), clamp, truncate
int kzo_rgb_to_srgb8(const float *rgb, int w, int h, uint8_t *out) { FtzScope ftz_;
    for (size_t i = 0; i < (size_t)w * h * 3; ++i) {
        float value = rgb[i];
        float t = value <= 0.0031308f ? 12.92f * value : (1.0f + 0.055f) * kzoPow(value, 1.0f / 2.4f) - 0.055f;
        out[i] = (uint8_t)clampf(255.f * t, 0.f, 255.f);
    }
    return 0;
}

int kzo_get_stats(void *s, KzStats *o, int reset) {
    Scene *sc = (Scene *)s;
    o->samples = sc->stats.samples; o->rays = sc->stats.rays; o->nodeVisits = sc->stats.nodeVisits; o->triTests = sc->stats.triTests;
    o->shadedHits = sc->stats.shadedHits; o->lightSamples = sc->stats.lightSamples; o->droppedSamples = sc->stats.dropped;
    if (reset) { sc->stats.samples = 0; sc->stats.rays = 0; sc->stats.nodeVisits = 0; sc->stats.triTests = 0; sc->stats.shadedHits = 0; sc->stats.lightSamples = 0; sc->stats.dropped = 0; }
    return 0;
}

// Accel::rayIntersect(ray, its, false) for n rays.
int kzo_trace_rays(void *s, uint32_t n, const float *o, const float *d, const float *tmin, const float *tmax, KzHit *hits) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; LocalStats ls;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin[i], tmax[i]);
        Intersection its; KzHit &h = hits[i]; std::memset(&h, 0, sizeof h);
        if (!rayIntersect(sc, r, its, false, ls)) { h.t = kInf; h.mesh = -1; h.prim = -1; continue; }
        h.t = its.t; h.u = its.bu; h.v = its.bv; h.mesh = its.mesh; h.prim = its.prim;
        h.p[0] = its.p.x; h.p[1] = its.p.y; h.p[2] = its.p.z; h.uv[0] = its.uvx; h.uv[1] = its.uvy;
        h.sh_s[0] = its.shFrame.s.x; h.sh_s[1] = its.shFrame.s.y; h.sh_s[2] = its.shFrame.s.z;
        h.sh_t[0] = its.shFrame.t.x; h.sh_t[1] = its.shFrame.t.y; h.sh_t[2] = its.shFrame.t.z;
        h.sh_n[0] = its.shFrame.n.x; h.sh_n[1] = its.shFrame.n.y; h.sh_n[2] = its.shFrame.n.z;
        h.geo_n[0] = its.geoFrame.n.x; h.geo_n[1] = its.geoFrame.n.y; h.geo_n[2] = its.geoFrame.n.z;
    }
    mergeStats(sc, ls);
    return 0;
}

// ---- function-level known-answer entry points ------------------------------------
uint64_t kzo_hash_pixel_seed(int32_t x, int32_t y, uint64_t seed) { return HashPixelSeed(x, y, seed); }
uint64_t kzo_hash_pixel_dim_seed(int32_t x, int32_t y, uint32_t dim, uint64_t seed) { return HashPixelDimSeed(x, y, dim, seed); }
uint64_t kzo_murmur64a(const unsigned char *key, size_t len, uint64_t seed) { return MurmurHash64A(key, len, seed); }
uint64_t kzo_mixbits(uint64_t v) { return MixBits(v); }
uint32_t kzo_permute(uint32_t i, uint32_t l, uint32_t p) { return permute(i, l, p); }
// the Fresnel functions of the dielectric / rough BSDFs (common.cpp:447-475, :492-518), for the vectors minted from the reference's own text
float kzo_fresnel_ior(float cosThetaI, float extIOR, float intIOR) { FtzScope ftz_; return fresnelIOR(cosThetaI, extIOR, intIOR); }
float kzo_fresnel_dielectric(float cosThetaI, float eta, float *cosThetaT) { FtzScope ftz_; float ct = 0.f; const float F = fresnelDielectricT(cosThetaI, eta, ct); if (cosThetaT) *cosThetaT = ct; return F; }
uint64_t kzo_tea32(uint32_t v0, uint32_t v1, int rounds) { return sampleTEA32(v0, v1, rounds); }
// pcg32: seed(initseq) [1-arg form], advance(delta), then n draws -> uints and floats
void kzo_pcg32_stream(uint64_t initseq, int64_t delta, int n, uint32_t *u, float *f, uint64_t *stateOut) {
    Pcg32 a; a.seed(initseq); a.advance(delta);
    Pcg32 b = a;
    for (int i = 0; i < n; ++i) { u[i] = a.nextUInt(); f[i] = b.nextFloat(); }
    if (stateOut) { stateOut[0] = a.state; stateOut[1] = a.inc; }
}
void kzo_pcg32_seed2(uint64_t initstate, uint64_t initseq, int n, uint32_t *u) {
    Pcg32 a; a.seed(initstate, initseq);
    for (int i = 0; i < n; ++i) u[i] = a.nextUInt();
}
// sampler stream of one (pixel, sample): nextPixel2D, next2D, then n1 x next1D
void kzo_sampler_stream(void *s, int32_t px, int32_t py, uint32_t idx, int n1, float *out) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; Sampler sm; sm.sc = &sc; sm.type = sc.smp.type;
    sm.generateSample(px, py, idx);
    sm.nextPixel2D(out[0], out[1]); sm.next2D(out[2], out[3]);
    for (int i = 0; i < n1; ++i) out[4 + i] = sm.next1D();
}
// Scene::getBackgroundColor(dir) (scene.cpp:54-79)
void kzo_background(void *s, const float *dir, float *rgb) { FtzScope ftz_;
    V3 c = backgroundColor(*(Scene *)s, V3(dir[0], dir[1], dir[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void kzo_camera_ray(void *s, float sx, float sy, float *o6, float *mint, float *maxt) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; Ray r; cameraSampleRay(sc, sx, sy, 0.5f, 0.5f, r);
    o6[0] = r.o.x; o6[1] = r.o.y; o6[2] = r.o.z; o6[3] = r.d.x; o6[4] = r.d.y; o6[5] = r.d.z; *mint = r.mint; *maxt = r.maxt;
}
// the same with the aperture sample given (ThinLensCamera::sampleRay, camera.cpp:191-223; a pinhole ignores it)
void kzo_camera_ray_lens(void *s, float sx, float sy, float ax, float ay, float *o6, float *mint, float *maxt) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; Ray r; cameraSampleRay(sc, sx, sy, ax, ay, r);
    o6[0] = r.o.x; o6[1] = r.o.y; o6[2] = r.o.z; o6[3] = r.d.x; o6[4] = r.d.y; o6[5] = r.d.z; *mint = r.mint; *maxt = r.maxt;
}
void kzo_filter_table(void *s, float *tab33, float *radius, int *border) {
    Scene &sc = *(Scene *)s; std::memcpy(tab33, sc.filter, sizeof sc.filter); *radius = sc.filterRadius; *border = sc.border;
}
void kzo_cosine_hemisphere(float sx, float sy, float *o3) { FtzScope ftz_; V3 v = squareToCosineHemisphere(sx, sy); o3[0] = v.x; o3[1] = v.y; o3[2] = v.z; }
void kzo_uniform_disk(float sx, float sy, float *o2) { FtzScope ftz_; squareToUniformDisk(sx, sy, o2[0], o2[1]); }
void kzo_frame(const float *n, float *s3, float *t3) { FtzScope ftz_; Frame f(V3(n[0], n[1], n[2])); s3[0] = f.s.x; s3[1] = f.s.y; s3[2] = f.s.z; t3[0] = f.t.x; t3[1] = f.t.y; t3[2] = f.t.z; }
// BSDF: which = 0 eval (3 floats), 1 pdf (1 float), 2 sample (weight 3 + wo 3 + ok 1)
// known answers against vectors minted from the reference's own dpdf.h / common.h (oracle/kat_ref_dpdf.cpp): the functions the light set-up and the
// PMJ02BN constructor above call
void kzo_debug_dpdf(uint32_t n, const float *values, float *cdf, float *sumAndNormalization) { FtzScope ftz_;
    std::vector<float> t(1, 0.0f);
    for (uint32_t i = 0; i < n; ++i) t.push_back(t.back() + values[i]);                                                                // dpdf.h:35-37
    sumAndNormalization[1] = dpdfNormalize(t, &sumAndNormalization[0]);
    std::memcpy(cdf, t.data(), t.size() * sizeof(float));
}
uint32_t kzo_debug_dpdf_sample(uint32_t nCdf, const float *cdf, float v) { FtzScope ftz_; return (uint32_t)dpdfSampleTable(std::vector<float>(cdf, cdf + nCdf), v); }
void kzo_debug_pow4(int spp, int *out4) { out4[0] = isPowerOf4(spp) ? 1 : 0; out4[1] = roundUpPow4(spp); out4[2] = log4i((uint32_t)out4[1]); out4[3] = pmjPixelTile((uint32_t)spp); }

void kzo_bsdf(const KzBSDF *row, int which, const float *wi, const float *wo, float accRough, float s1, float s2x, float s2y, float *out) { FtzScope ftz_;
    KzBSDF resolved = *row, *m = &resolved;                // a bare row: resolve the rough BSDFs' alpha as kzo_scene_create does (the constructors' m_alpha)
    if ((m->type == KZ_BSDF_ROUGHCONDUCTOR || m->type == KZ_BSDF_ROUGHPLASTIC || m->type == KZ_BSDF_ROUGHDIELECTRIC) && !m->alphaResolved) { m->alpha = std::max(0.001f, sqr(m->alpha)); m->alphaResolved = 1; }
    BRec b; b.wi = V3(wi[0], wi[1], wi[2]); b.accumulatedRoughness = accRough;
    if (which == 2) {
        bool ok; V3 w = bsdfSample(*m, b, s1, s2x, s2y, ok);
        out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = b.wo.x; out[4] = b.wo.y; out[5] = b.wo.z; out[6] = ok ? 1.f : 0.f;
        return;
    }
    b.wo = V3(wo[0], wo[1], wo[2]); b.measure = ESolidAngle;
    if (which == 0) { V3 f = bsdfEval(*m, b); out[0] = f.x; out[1] = f.y; out[2] = f.z; }
    else out[0] = bsdfPdf(*m, b);
}
// Scene-aware form (texture children, normalmap rows): the intersection record is the identity frame with dpdu = +x at uv.
// which = 2 returns 8 floats: weight 3, wo 3, ok, and pdf(bRec) right after sample() (integrator.cpp:314; 0 when the path ends).
void kzo_scene_bsdf(void *s, int bsdfIdx, int which, const float *wi, const float *wo, float accRough, float s1, float s2x, float s2y, float u, float v, float *out) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; const KzBSDF &m = sc.bsdfs[bsdfIdx];
    Intersection its; its.uvx = u; its.uvy = v; its.accumulatedRoughness = accRough;
    its.shFrame.s = V3(1, 0, 0); its.shFrame.t = V3(0, 1, 0); its.shFrame.n = V3(0, 0, 1); its.geoFrame = its.shFrame; its.dpdu = V3(1, 0, 0);
    BRec b; b.wi = V3(wi[0], wi[1], wi[2]); b.accumulatedRoughness = accRough; b.its = &its; b.uvx = u; b.uvy = v; b.sc = &sc;
    if (which == 2) {
        bool ok; V3 w = bsdfSample(m, b, s1, s2x, s2y, ok);
        bool zero = w.x == 0.f && w.y == 0.f && w.z == 0.f;
        if (!ok || zero) b.wo = (ok ? b.wo : V3(0, 0, 1));
        out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = b.wo.x; out[4] = b.wo.y; out[5] = b.wo.z; out[6] = ok ? 1.f : 0.f;
        out[7] = (!ok || zero) ? 0.f : bsdfPdf(m, b);
        return;
    }
    b.wo = V3(wo[0], wo[1], wo[2]); b.measure = ESolidAngle;
    if (which == 0) { V3 f = bsdfEval(m, b); out[0] = f.x; out[1] = f.y; out[2] = f.z; }
    else out[0] = bsdfPdf(m, b);
}
void kzo_texture(void *s, int tex, float u, float v, float *out) { FtzScope ftz_;
    V3 c = textureEval(*(Scene *)s, tex, u, v); out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void kzo_ggx_sample_vndf(const float *V, float ax, float ay, float rx, float ry, float *H) { FtzScope ftz_;
    V3 h = sampleGGXSmithVNDF(V3(V[0], V[1], V[2]), A2{ax, ay}, rx, ry); H[0] = h.x; H[1] = h.y; H[2] = h.z;
}
// light: sample the light mesh `lightIdx` (index into Scene::m_lights order) from `ref` with the
// three Mesh::sample draws given explicitly: out = p(3) n(3) wi(3) pdf(1) Ls(3) (Ls = eval/pdf)
void kzo_light_sample(void *s, int lightIdx, const float *ref, float u0, float u1, float u2, float *out) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; const MeshData &md = sc.meshes[sc.lightMeshes[lightIdx]]; const KzLight &l = sc.lights[md.light];
    LRec r; r.ref = V3(ref[0], ref[1], ref[2]);
    size_t index = dpdfSample(md, u0);
    float su0 = std::sqrt(u1); float u = 1 - su0; float v = u2 * su0;
    uint32_t i0 = md.F[3 * index], i1 = md.F[3 * index + 1], i2 = md.F[3 * index + 2];
    V3 p0 = md.v(i0), p1 = md.v(i1), p2 = md.v(i2);
    r.p = p0 + u * (p1 - p0) + v * (p2 - p0);
    if (!md.N.empty()) { V3 n0 = md.nrm(i0), n1 = md.nrm(i1), n2 = md.nrm(i2); r.n = n0 + u * (n1 - n0) + v * (n2 - n0); }
    else r.n = normalized(cross(p1 - p0, p2 - p0));
    r.wi = normalized(r.p - r.ref);
    r.pdf = lightPdf(md, r);
    V3 Ls = (r.pdf > 0.f && !std::isnan(r.pdf) && !std::isinf(r.pdf)) ? lightEval(l, r) / r.pdf : V3(0.f);
    out[0] = r.p.x; out[1] = r.p.y; out[2] = r.p.z; out[3] = r.n.x; out[4] = r.n.y; out[5] = r.n.z;
    out[6] = r.wi.x; out[7] = r.wi.y; out[8] = r.wi.z; out[9] = r.pdf; out[10] = Ls.x; out[11] = Ls.y; out[12] = Ls.z;
    out[13] = (float)index;
}
// Radiance of single samples (no film): out = n x (sx, sy, r, g, b)
void kzo_render_samples(void *s, uint32_t n, const int32_t *pxy, const uint32_t *idx, float *out) { FtzScope ftz_;
    Scene &sc = *(Scene *)s; Sampler sm; sm.sc = &sc; sm.type = sc.smp.type; LocalStats ls;
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON); _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    for (uint32_t i = 0; i < n; ++i) {
        float sx, sy; V3 v = renderSample(sc, sm, pxy[2 * i], pxy[2 * i + 1], idx[i], sx, sy, ls);
        out[5 * i] = sx; out[5 * i + 1] = sy; out[5 * i + 2] = v.x; out[5 * i + 3] = v.y; out[5 * i + 4] = v.z;
    }
    mergeStats(sc, ls);
}

} // extern "C"
