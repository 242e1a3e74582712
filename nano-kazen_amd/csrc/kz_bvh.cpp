// kz_bvh.cpp — host BVH builder of the MI355X core (replaces Embree's build behind Accel::build,
// src/kazen/accel.cpp:25-61; Embree 3.13.0 itself is not vendored in the reference).
//
// Binned-SAH BVH2 (32 bins), <= KZ_MAX_LEAF triangles per leaf, built top-down with std::async tasks
// on the upper levels, then flattened breadth-first into 64-B node packets that carry BOTH children's
// boxes (one packet fetch per traversal step) and 48-B Moeller-Trumbore leaf triangles (p0, e1, e2:
// the same float differences Mesh::rayIntersect forms per call, src/kazen/mesh.cpp:60).
// The tree depth is capped at KZ_STACK_DEPTH-2 (median splits take over when the SAH tree gets too
// deep) because the traversal kernels keep a fixed per-lane stack in LDS.
#include "kz_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <future>
#include <limits>
#include <array>

namespace {

const float kInf = std::numeric_limits<float>::infinity();
const int NBINS = 32;
const int DEPTH_CAP = KZ_STACK_DEPTH - 2;


struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = kInf; hi[a] = -kInf; } }
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
struct Ref { Box b; float c[3]; uint32_t tri; };
struct Tmp { Box b; uint32_t left, right; uint32_t start, count; uint32_t depth; };   // count>0: leaf

struct Ctx {
    std::vector<Ref> refs;
    std::vector<Tmp> tmp;
    std::atomic<uint32_t> nTmp{0};
    std::atomic<uint32_t> maxDepth{0};
    uint32_t maxLeaf = KZ_MAX_LEAF;
    float nodeCost = 0.7f;
    uint32_t alloc() { return nTmp.fetch_add(1); }
};

static int ceilLog2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

static void build(Ctx &cx, uint32_t node, uint32_t b, uint32_t e, uint32_t depth) {
    Tmp &t = cx.tmp[node];
    t.depth = depth;
    t.b.reset();
    Box cb; cb.reset();
    for (uint32_t i = b; i < e; ++i) { t.b.grow(cx.refs[i].b); cb.grow(cx.refs[i].c); }
    uint32_t n = e - b;
    auto makeLeaf = [&]() {
        t.start = b; t.count = n; t.left = t.right = 0;
        uint32_t md = cx.maxDepth.load();
        while (depth > md && !cx.maxDepth.compare_exchange_weak(md, depth)) {}
    };
    if (n == 1) { makeLeaf(); return; }
    t.count = 0;
    uint32_t mid = b;
    bool forceMedian = (int)depth + ceilLog2(n) >= DEPTH_CAP;
    int bestAxis = -1, bestSplit = 0;
    float bestCost = kInf;
    int sweepAxis = -1; uint32_t sweepMid = 0;
    if ((!forceMedian || n <= cx.maxLeaf) && n <= 64) {
        // exact SAH sweep for small ranges: every split position along every axis (binning is too coarse down here)
        std::vector<Ref> tmp(cx.refs.begin() + b, cx.refs.begin() + e), bestOrder;
        std::vector<float> rArea(n);
        for (int a = 0; a < 3; ++a) {
            std::sort(tmp.begin(), tmp.end(), [a](const Ref &x, const Ref &y) { return x.c[a] < y.c[a]; });
            Box acc; acc.reset();
            for (uint32_t i = n; i-- > 1;) { acc.grow(tmp[i].b); rArea[i] = acc.area(); }
            acc.reset();
            for (uint32_t i = 0; i + 1 < n; ++i) {
                acc.grow(tmp[i].b);
                float cost = acc.area() * (float)(i + 1) + rArea[i + 1] * (float)(n - i - 1);
                if (cost < bestCost) { bestCost = cost; sweepAxis = a; sweepMid = i + 1; bestOrder = tmp; }
            }
        }
        if (sweepAxis >= 0) { std::copy(bestOrder.begin(), bestOrder.end(), cx.refs.begin() + b); bestAxis = sweepAxis; }
    } else if (!forceMedian || n <= cx.maxLeaf) {
        for (int a = 0; a < 3; ++a) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            Box bb[NBINS]; uint32_t cnt[NBINS];
            for (int k = 0; k < NBINS; ++k) { bb[k].reset(); cnt[k] = 0; }
            float scale = NBINS / ext;
            for (uint32_t i = b; i < e; ++i) {
                int k = std::min(NBINS - 1, (int)((cx.refs[i].c[a] - cb.lo[a]) * scale));
                cnt[k]++; bb[k].grow(cx.refs[i].b);
            }
            float rA[NBINS]; uint32_t rC[NBINS];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int k = NBINS - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; rA[k] = acc.area(); rC[k] = c; }
            acc.reset(); c = 0;
            for (int k = 0; k < NBINS - 1; ++k) {
                acc.grow(bb[k]); c += cnt[k];
                if (c == 0 || rC[k + 1] == 0) continue;
                float cost = acc.area() * c + rA[k + 1] * rC[k + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
            }
        }
    }
    // SAH leaf termination for small ranges: a leaf of n triangles costs n triangle tests; a split costs one node packet
    // (two slab tests, about one triangle test on CDNA4: ~60 vs ~55 VALU ops and 2 vs 3 16-B gathers) plus the expected
    // tests in the children. Random soups split down to single triangles, coherent meshes keep pairs / quads together.
    if (n <= cx.maxLeaf) {
        const float A = t.b.area();
        if (bestAxis < 0 || !(A > 0.f) || (float)n * A <= cx.nodeCost * A + bestCost) { makeLeaf(); return; }
    }
    if (!forceMedian && sweepAxis >= 0) mid = b + sweepMid;
    else if (!forceMedian && bestAxis >= 0) {
        int a = bestAxis; float lo = cb.lo[a], scale = NBINS / (cb.hi[a] - cb.lo[a]);
        auto it = std::partition(cx.refs.begin() + b, cx.refs.begin() + e, [&](const Ref &r) {
            return std::min(NBINS - 1, (int)((r.c[a] - lo) * scale)) <= bestSplit;
        });
        mid = (uint32_t)(it - cx.refs.begin());
    }
    if (mid == b || mid == e) {
        // median object split along the widest centroid axis (also the depth-cap fallback)
        int a = 0; float best = -1.f;
        for (int k = 0; k < 3; ++k) { float ext = cb.hi[k] - cb.lo[k]; if (ext > best) { best = ext; a = k; } }
        mid = b + n / 2;
        std::nth_element(cx.refs.begin() + b, cx.refs.begin() + mid, cx.refs.begin() + e,
                         [a](const Ref &x, const Ref &y) { return x.c[a] < y.c[a]; });
    }
    uint32_t l = cx.alloc(), r = cx.alloc();
    cx.tmp[node].left = l; cx.tmp[node].right = r;
    if (n > 32768 && depth < 8) {
        auto fut = std::async(std::launch::async, [&cx, l, b, mid, depth]() { build(cx, l, b, mid, depth + 1); });
        build(cx, r, mid, e, depth + 1);
        fut.get();
    } else {
        build(cx, l, b, mid, depth + 1);
        build(cx, r, mid, e, depth + 1);
    }
}

// A few ulps of slack: the slab test rounds differently from the triangle test, and the parity target is
// "never cull a triangle the brute-force Moeller-Trumbore search would report".
// `absPad` = 1e-6 x the scene extent: the slab arithmetic (and the FMA form used on the quantised BVH4) has an error
// proportional to |box - ray origin|, not to the box coordinates, so boxes near the world origin need an absolute term.
static void padBox(Box &b, float absPad) {
    for (int a = 0; a < 3; ++a) {
        float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
        float e = std::max(m * 4e-7f, absPad) + 1e-30f;
        b.lo[a] -= e; b.hi[a] += e;
    }
}

#ifdef KZ_EXPERIMENTS
// (development build only: measured on C4 and C3 and rejected, profiles/r03j_presplit)
// ---- pre-splitting (spatial splits before the build; Embree's RTC_BUILD_QUALITY_HIGH, which the reference asks for at accel.cpp:36, builds with
// spatial splits too). A reference = (box, triangle): the references with the largest boxes are cut in two at the middle of their box's longest
// axis - the triangle's polygon is clipped against the plane, each side gets the box of its part - until `factor` x the triangle count
// extra references exist. The tree is then built over the references; a leaf may name a triangle that another leaf names too (its record
// is simply stored twice: the same Moeller-Trumbore test, the same hit). Boxes only ever get tighter around the same surface.
struct Poly { int n; float v[10][3]; };
static void polyBox(const Poly &p, Box &b) { b.reset(); for (int i = 0; i < p.n; ++i) b.grow(p.v[i]); }
static void clipPoly(const Poly &p, int axis, float pos, bool keepLow, Poly &out) {
    out.n = 0;
    for (int i = 0; i < p.n; ++i) {
        const float *a = p.v[i], *b = p.v[(i + 1) % p.n];
        const bool ina = keepLow ? a[axis] <= pos : a[axis] >= pos, inb = keepLow ? b[axis] <= pos : b[axis] >= pos;
        if (ina && out.n < 10) { for (int k = 0; k < 3; ++k) out.v[out.n][k] = a[k]; out.n++; }
        if (ina != inb && out.n < 10) {
            const float t = (pos - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; ++k) out.v[out.n][k] = a[k] + t * (b[k] - a[k]);
            out.v[out.n][axis] = pos;
            out.n++;
        }
    }
}
static void presplit(const std::vector<KzBuildTri> &in, std::vector<Ref> &refs, float factor) {
    if (!(factor > 0.f) || refs.empty()) return;
    struct Item { float prio; Poly poly; Box box; uint32_t tri; };
    auto cmp = [](const Item &x, const Item &y) { return x.prio < y.prio; };
    std::vector<Item> heap; heap.reserve(refs.size());
    for (const Ref &r : refs) {
        Item it; it.tri = r.tri; it.box = r.b; it.poly.n = 3;
        for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) it.poly.v[v][k] = in[r.tri].v[v][k];
        it.prio = r.b.area();
        heap.push_back(it);
    }
    std::make_heap(heap.begin(), heap.end(), cmp);
    size_t budget = (size_t)((double)factor * (double)refs.size());
    std::vector<Item> done;
    while (budget > 0 && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        Item it = heap.back(); heap.pop_back();
        int a = 0; float ext = -1.f;
        for (int k = 0; k < 3; ++k) { const float e = it.box.hi[k] - it.box.lo[k]; if (e > ext) { ext = e; a = k; } }
        const float pos = 0.5f * (it.box.lo[a] + it.box.hi[a]);
        Item L, R; L.tri = R.tri = it.tri;
        clipPoly(it.poly, a, pos, true, L.poly); clipPoly(it.poly, a, pos, false, R.poly);
        if (!(ext > 0.f) || it.poly.n >= 9 || L.poly.n < 3 || R.poly.n < 3 || !(pos > it.box.lo[a]) || !(pos < it.box.hi[a])) { it.prio = -1.f; done.push_back(it); continue; }
        polyBox(L.poly, L.box); polyBox(R.poly, R.box);
        // the parts stay inside the parent's box, and both reach the plane (no seam between them)
        for (int k = 0; k < 3; ++k) { L.box.lo[k] = std::max(L.box.lo[k], it.box.lo[k]); L.box.hi[k] = std::min(L.box.hi[k], it.box.hi[k]); R.box.lo[k] = std::max(R.box.lo[k], it.box.lo[k]); R.box.hi[k] = std::min(R.box.hi[k], it.box.hi[k]); }
        L.box.hi[a] = pos; R.box.lo[a] = pos;
        L.prio = L.box.area(); R.prio = R.box.area();
        heap.push_back(L); std::push_heap(heap.begin(), heap.end(), cmp);
        heap.push_back(R); std::push_heap(heap.begin(), heap.end(), cmp);
        --budget;
    }
    refs.clear();
    auto emit = [&](const Item &it) { Ref r; r.b = it.box; r.tri = it.tri; for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (it.box.lo[k] + it.box.hi[k]); refs.push_back(r); };
    for (const Item &it : heap) emit(it);
    for (const Item &it : done) emit(it);
}
#endif

} // namespace

int kz_build_bvh(const std::vector<KzBuildTri> &in, std::vector<KzNode> &nodes, std::vector<KzTri> &tris,
                 uint32_t &rootRef, KzBvhInfo &info, std::string &err) {
    auto t0 = std::chrono::steady_clock::now();
    std::memset(&info, 0, sizeof info);
    nodes.clear(); tris.clear();
    rootRef = 0xFFFFFFFFu;     // empty scene
    Ctx cx;
#ifdef KZ_EXPERIMENTS      // builder parameter sweeps (scripts/bvh_sweep.sh) of a development build; the product library reads no environment variable
    if (const char *e = std::getenv("KZ_BVH_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= KZ_MAX_LEAF) cx.maxLeaf = (uint32_t)v; }
    if (const char *e = std::getenv("KZ_BVH_NODE_COST")) { float v = (float)std::atof(e); if (v >= 0.f && v < 100.f) cx.nodeCost = v; }
#endif
    cx.refs.reserve(in.size());
    for (uint32_t i = 0; i < in.size(); ++i) {
        const KzBuildTri &t = in[i];
        bool finite = true;
        for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(t.v[v][a]);
        if (!finite) continue;      // never hit (Embree also drops non-finite primitives)
        Ref r; r.b.reset(); r.tri = i;
        for (int v = 0; v < 3; ++v) r.b.grow(t.v[v]);
        for (int a = 0; a < 3; ++a) r.c[a] = 0.5f * (r.b.lo[a] + r.b.hi[a]);
        cx.refs.push_back(r);
    }
    info.nTris = (uint32_t)cx.refs.size();
#ifdef KZ_EXPERIMENTS
    if (const char *e = std::getenv("KZ_BVH_PRESPLIT")) { float v = (float)std::atof(e); if (v > 0.f && v <= 8.f) presplit(in, cx.refs, v); }
#endif
    uint32_t n = (uint32_t)cx.refs.size();
    if (n >= (1u << 28)) { err = "more than 2^28 triangle references"; return KZ_ERR_UNSUPPORTED; }
    if (n == 0) return KZ_OK;
    cx.tmp.resize(2 * (size_t)n + 2);
    uint32_t root = cx.alloc();
    build(cx, root, 0, n, 0);
    if (cx.maxDepth.load() > KZ_STACK_DEPTH) { err = "BVH deeper than the traversal stack"; return KZ_ERR_STATE; }

    // leaf triangles in ref order
    tris.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const KzBuildTri &s = in[cx.refs[i].tri];
        KzTri &d = tris[i];
        for (int a = 0; a < 3; ++a) { d.p0[a] = s.v[0][a]; d.e1[a] = s.v[1][a] - s.v[0][a]; d.e2[a] = s.v[2][a] - s.v[0][a]; }
        d.mesh = s.mesh; d.prim = s.prim; d.gid = s.gid;
    }
    auto refOf = [&](uint32_t tn) -> uint32_t { const Tmp &t = cx.tmp[tn]; return 0x80000000u | (t.start << 3) | (t.count - 1); };
    // breadth-first numbering of the inner nodes (top of the tree is contiguous at the front of the array)
    std::vector<uint32_t> order;        // tmp index of inner nodes in BFS order
    std::vector<uint32_t> newIndex(cx.nTmp.load(), 0xFFFFFFFFu);
    double sah = 0.0; uint32_t nLeaves = 0, maxLeaf = 0;
    float rootArea = cx.tmp[root].b.area();
    float absPad = 0.f;
    for (int a = 0; a < 3; ++a) absPad = std::max(absPad, 1e-6f * std::max(cx.tmp[root].b.hi[a] - cx.tmp[root].b.lo[a], std::max(std::fabs(cx.tmp[root].b.hi[a]), std::fabs(cx.tmp[root].b.lo[a]))));
    if (cx.tmp[root].count > 0) {
        rootRef = refOf(root); nLeaves = 1; maxLeaf = cx.tmp[root].count;
    } else {
        order.push_back(root); newIndex[root] = 0;
        for (size_t h = 0; h < order.size(); ++h) {
            const Tmp &t = cx.tmp[order[h]];
            for (uint32_t c : {t.left, t.right})
                if (cx.tmp[c].count == 0) { newIndex[c] = (uint32_t)order.size(); order.push_back(c); }
        }
        nodes.resize(order.size());
        for (size_t h = 0; h < order.size(); ++h) {
            const Tmp &t = cx.tmp[order[h]];
            KzNode &nd = nodes[h];
            std::memset(&nd, 0, sizeof nd);
            uint32_t ch[2] = {t.left, t.right};
            Box cb[2];
            for (int k = 0; k < 2; ++k) {
                const Tmp &c = cx.tmp[ch[k]];
                cb[k] = c.b; padBox(cb[k], absPad);
                if (c.count > 0) { nd.child[k] = refOf(ch[k]); nLeaves++; maxLeaf = std::max(maxLeaf, c.count); sah += (double)c.b.area() * c.count; }
                else { nd.child[k] = newIndex[ch[k]]; sah += (double)c.b.area() * 1.0; }
            }
            nd.q[0] = cb[0].lo[0]; nd.q[1] = cb[0].lo[1]; nd.q[2] = cb[0].lo[2]; nd.q[3] = cb[0].hi[0];
            nd.q[4] = cb[0].hi[1]; nd.q[5] = cb[0].hi[2]; nd.q[6] = cb[1].lo[0]; nd.q[7] = cb[1].lo[1];
            nd.q[8] = cb[1].lo[2]; nd.q[9] = cb[1].hi[0]; nd.q[10] = cb[1].hi[1]; nd.q[11] = cb[1].hi[2];
        }
        rootRef = 0;
    }
    info.nNodes = (uint32_t)nodes.size(); info.nLeaves = nLeaves; info.maxDepth = cx.maxDepth.load(); info.maxLeafSize = maxLeaf;
    info.sahCost = rootArea > 0 ? (float)(sah / rootArea) : 0.f;
    info.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return KZ_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// BVH2 -> BVH4 collapse with 8-bit quantised child boxes (KzNode4). Children of a BVH4 node are found by repeatedly
// opening the inner child with the largest surface area until four slots are used. Nodes are numbered breadth-first.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct ChildBox { float lo[3], hi[3]; uint32_t ref; };

void childBoxesOf(const std::vector<KzNode> &nodes, uint32_t n, ChildBox out[2]) {
    const KzNode &nd = nodes[n];
    out[0].ref = nd.child[0]; out[1].ref = nd.child[1];
    out[0].lo[0] = nd.q[0]; out[0].lo[1] = nd.q[1]; out[0].lo[2] = nd.q[2]; out[0].hi[0] = nd.q[3]; out[0].hi[1] = nd.q[4]; out[0].hi[2] = nd.q[5];
    out[1].lo[0] = nd.q[6]; out[1].lo[1] = nd.q[7]; out[1].lo[2] = nd.q[8]; out[1].hi[0] = nd.q[9]; out[1].hi[1] = nd.q[10]; out[1].hi[2] = nd.q[11];
}
float boxArea(const ChildBox &c) {
    float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}
// the kernel's dequantisation, evaluated identically on the host (no contraction: q * s is exact, one rounding in the add)
inline float deq(float p, uint32_t q, float s) { volatile float prod = (float)q * s; return p + prod; }
}

int kz_collapse_bvh4(const std::vector<KzNode> &nodes, uint32_t rootRef, std::vector<KzNode4> &out, uint32_t &rootRef4, int &stackBound) {
    out.clear(); stackBound = 1;
    rootRef4 = rootRef;                      // empty scene or single-leaf root: same reference
    if (rootRef == 0xFFFFFFFFu || (rootRef & 0x80000000u)) return KZ_OK;
    struct Item { uint32_t bvh2; };
    std::vector<uint32_t> order; order.push_back(rootRef);       // BVH2 node that roots each BVH4 node, BFS
    std::vector<std::array<ChildBox, 4>> kids; std::vector<int> nk;
    // ---- SAH-optimal collapse (dynamic programming over the BVH2, after Ylitie, Karras, Laine 2017, section 3.1, for width 4):
    // c[n][i] = least cost of representing the subtree of BVH2 node n by at most i+1 child slots of a wide node, where a slot is a wide
    // node (area * nodeCost + the best distribution of its subtree over 4 slots), a leaf of <= maxLeaf triangles (area * count * primCost;
    // the triangles of a subtree are contiguous in the leaf array) or an opened BVH2 node whose two sides share the slots. The greedy rule
    // (open the child with the largest area) is kept as KZ_BVH4_COLLAPSE=0. Only the SHAPE of the tree changes: boxes stay conservative,
    // the triangle tests stay the reference's, so every hit is the same.
    const size_t N = nodes.size();
    int mode = 1; float nodeCost4 = 1.0f, primCost4 = 1.5f; uint32_t maxLeaf4 = 4;      // measured: scripts/bvh4_sweep.sh (0.3 merges too much: +11 % time; 1.0-4.0 flat)
#ifdef KZ_EXPERIMENTS      // scripts/bvh4_sweep.sh, development build only
    if (const char *e = std::getenv("KZ_BVH4_COLLAPSE")) mode = std::atoi(e);
    if (const char *e = std::getenv("KZ_BVH4_PRIM_COST")) { float v = (float)std::atof(e); if (v > 0.f && v < 100.f) primCost4 = v; }
    if (const char *e = std::getenv("KZ_BVH4_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= 8) maxLeaf4 = (uint32_t)v; }
#endif
    struct Dp { float c[3]; float area; uint32_t start, count; uint8_t contig, leaf1, k2, k3, k4, open2, open3; };
    std::vector<Dp> dp;
    auto leafStart = [](uint32_t ref) { return (ref & 0x7fffffffu) >> 3; };
    auto leafCount = [](uint32_t ref) { return (ref & 7u) + 1u; };
    if (mode == 1) {
        dp.resize(N);
        const float INF = std::numeric_limits<float>::infinity();
        for (size_t nn = N; nn-- > 0;) {                            // BFS numbering: children come after their parent
            ChildBox cb[2]; childBoxesOf(nodes, (uint32_t)nn, cb);
            float cc[2][3]; uint32_t st[2], ct[2]; bool cg[2];
            for (int s2 = 0; s2 < 2; ++s2) {
                if (cb[s2].ref & 0x80000000u) {
                    st[s2] = leafStart(cb[s2].ref); ct[s2] = leafCount(cb[s2].ref); cg[s2] = true;
                    const float c = boxArea(cb[s2]) * (float)ct[s2] * primCost4;
                    cc[s2][0] = cc[s2][1] = cc[s2][2] = c;
                } else {
                    const Dp &d = dp[cb[s2].ref];
                    st[s2] = d.start; ct[s2] = d.count; cg[s2] = d.contig != 0;
                    for (int i = 0; i < 3; ++i) cc[s2][i] = d.c[i];
                }
            }
            Dp d; std::memset(&d, 0, sizeof d);
            ChildBox u = cb[0];
            for (int a = 0; a < 3; ++a) { u.lo[a] = std::min(u.lo[a], cb[1].lo[a]); u.hi[a] = std::max(u.hi[a], cb[1].hi[a]); }
            d.area = boxArea(u);
            d.start = std::min(st[0], st[1]); d.count = ct[0] + ct[1];
            d.contig = (cg[0] && cg[1] && (st[0] + ct[0] == st[1] || st[1] + ct[1] == st[0])) ? 1 : 0;
            float dist[5]; uint8_t kk[5] = {0, 0, 0, 0, 0};
            for (int j = 2; j <= 4; ++j) {
                dist[j] = INF;
                for (int k = 1; k < j; ++k) {
                    if (k > 3 || j - k > 3) continue;
                    const float v = cc[0][k - 1] + cc[1][j - k - 1];
                    if (v < dist[j]) { dist[j] = v; kk[j] = (uint8_t)k; }
                }
            }
            const float leafC = (d.contig && d.count <= maxLeaf4) ? d.area * (float)d.count * primCost4 : INF;
            const float innerC = d.area * nodeCost4 + dist[4];
            d.leaf1 = leafC <= innerC ? 1 : 0;
            d.c[0] = std::min(leafC, innerC);
            d.open2 = dist[2] < d.c[0] ? 1 : 0; d.c[1] = std::min(dist[2], d.c[0]);
            d.open3 = dist[3] < d.c[1] ? 1 : 0; d.c[2] = std::min(dist[3], d.c[1]);
            d.k2 = kk[2]; d.k3 = kk[3]; d.k4 = kk[4];
            dp[nn] = d;
        }
    }
    // the slots the DP assigns to the subtree of BVH2 node n given `budget` slots (1..4; 4 only for the root of a wide node)
    std::function<void(uint32_t, int, bool, ChildBox *, int &)> expand = [&](uint32_t nref, int budget, bool forceOpen, ChildBox *out4, int &cnt) {
        ChildBox cb[2]; childBoxesOf(nodes, nref, cb);
        const Dp &d = dp[nref];
        (void)forceOpen;
        const int k = budget == 4 ? d.k4 : (budget == 3 ? d.k3 : d.k2);
        const int share[2] = {k, budget - k};
        for (int s2 = 0; s2 < 2; ++s2) {
            int b = share[s2];
            if (cb[s2].ref & 0x80000000u) { out4[cnt++] = cb[s2]; continue; }
            const Dp &m = dp[cb[s2].ref];
            // the cheapest representation within b slots: opened over b, or over b-1, ..., or a single slot
            while (b >= 2) {
                const bool open = b == 3 ? m.open3 != 0 : m.open2 != 0;
                if (open) break;
                --b;
            }
            if (b >= 2) expand(cb[s2].ref, b, true, out4, cnt);
            else if (m.leaf1) { ChildBox lf = cb[s2]; lf.ref = 0x80000000u | (m.start << 3) | (m.count - 1); out4[cnt++] = lf; }
            else out4[cnt++] = cb[s2];
        }
    };
    for (size_t h = 0; h < order.size(); ++h) {
        ChildBox cb[4]; int n = 2;
        if (mode == 1) { n = 0; expand(order[h], 4, true, cb, n); }
        else {
        childBoxesOf(nodes, order[h], cb);
        while (n < 4) {
            int best = -1; float bestA = -1.f;
            for (int i = 0; i < n; ++i) if (!(cb[i].ref & 0x80000000u)) { float a = boxArea(cb[i]); if (a > bestA) { bestA = a; best = i; } }
            if (best < 0) break;
            ChildBox two[2]; childBoxesOf(nodes, cb[best].ref, two);
            cb[best] = two[0]; cb[n++] = two[1];
        }
        }
        std::array<ChildBox, 4> arr;
        for (int i = 0; i < 4; ++i) arr[i] = cb[i < n ? i : 0];
        for (int i = 0; i < n; ++i) if (!(arr[i].ref & 0x80000000u)) { uint32_t idx = (uint32_t)order.size(); order.push_back(arr[i].ref); arr[i].ref = idx; }
        kids.push_back(arr); nk.push_back(n);
    }
    if (order.size() >= (1u << 31)) return KZ_ERR_UNSUPPORTED;
    out.resize(order.size());
    for (size_t h = 0; h < order.size(); ++h) {
        KzNode4 &nd = out[h]; std::memset(&nd, 0, sizeof nd);
        const int n = nk[h]; const auto &cb = kids[h];
        float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], cb[i].lo[a]); hi[a] = std::max(hi[a], cb[i].hi[a]); }
        uint32_t exps = 0; float scale[3];
        for (int a = 0; a < 3; ++a) {
            nd.p[a] = lo[a];
            float ext = hi[a] - lo[a];
            int e = 0;
            if (ext > 0.f) { std::frexp(ext / 255.0f, &e); }            // ext/255 = m * 2^e, m in [0.5,1) -> 2^e >= ext/255
            else e = -126;
            e = std::max(-126, std::min(127, e));
            // make sure 255 steps reach hi even after the rounding of p + 255*s
            while (e < 127 && deq(lo[a], 255u, std::ldexp(1.0f, e)) < hi[a]) ++e;
            scale[a] = std::ldexp(1.0f, e);
            exps |= (uint32_t)(e + 127) << (8 * a);
        }
        (void)exps; nd.scaleX = scale[0]; nd.scaleY = scale[1]; nd.scaleZ = scale[2];
        for (int i = 0; i < 4; ++i) {
            if (i >= n) { for (int a = 0; a < 3; ++a) { nd.qlo[a] |= 255u << (8 * i); } nd.child[i] = 0; continue; }     // qhi = 0: inverted, never hit
            nd.child[i] = cb[i].ref;
            for (int a = 0; a < 3; ++a) {
                int ql = (int)std::floor((cb[i].lo[a] - lo[a]) / scale[a]);
                ql = std::max(0, std::min(255, ql));
                while (ql > 0 && deq(lo[a], (uint32_t)ql, scale[a]) > cb[i].lo[a]) --ql;
                int qh = (int)std::ceil((cb[i].hi[a] - lo[a]) / scale[a]);
                qh = std::max(0, std::min(255, qh));
                while (qh < 255 && deq(lo[a], (uint32_t)qh, scale[a]) < cb[i].hi[a]) ++qh;
                nd.qlo[a] |= (uint32_t)ql << (8 * i);
                nd.qhi[a] |= (uint32_t)qh << (8 * i);
            }
        }
    }
    // worst-case stack depth of the traversal: a node pushes (children - 1) entries before descending
    std::vector<int> need(out.size(), 0);
    for (size_t h = out.size(); h-- > 0;) {
        int m = 0;
        for (int i = 0; i < nk[h]; ++i) { uint32_t c = out[h].child[i]; if (!(c & 0x80000000u)) m = std::max(m, need[c]); }
        need[h] = (nk[h] - 1) + m;
    }
    stackBound = need[0] + 1;
    rootRef4 = 0;
    return KZ_OK;
}
