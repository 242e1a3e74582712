// kz_internal.h — host-side scene object and the flat device tables of the MI355X path-tracing core.
// Not part of the ABI (that is include/kazen_mi355x.h).
#pragma once
#include "../../include/kazen_mi355x_dev.h"

#include <cstdint>
#include <string>
#include <vector>

#define KZ_STACK_DEPTH 32        // per-lane traversal stack entries (LDS); the builder caps the tree depth to this
#define KZ_MAX_LEAF 4            // triangles per leaf
#define KZ_MAX_FILTER_TAPS 9     // candidates per axis the film kernel supports (filter radius <= 4)

// ---- device formats (DESIGN.md "data layout in HBM") ---------------------------------------------
// BVH2 node, 64 B, four 16-B quads -> four global_load_dwordx4 per lane:
//   q0 = lo0.x lo0.y lo0.z hi0.x | q1 = hi0.y hi0.z lo1.x lo1.y | q2 = lo1.z hi1.x hi1.y hi1.z | q3 = child0 child1 - -
// child: bit31 set -> leaf, bits[30:3] = first triangle, bits[2:0] = count-1; else node index.
struct KzNode { float q[12]; uint32_t child[2]; uint32_t pad[2]; };
static_assert(sizeof(KzNode) == 64, "node packet must be 64 B");

// BVH4 node with 8-bit quantised child boxes, 64 B (the wavefront traversal kernels use this tree; the megakernel and
// kz_trace_rays keep the BVH2): one packet fetch tests FOUR children, which halves both the number of per-lane 16-B
// gathers and the length of the dependent-load chain (and, per ray, the instructions spent on loop control and stack traffic:
// the traversal is VALU-issue-bound on MI355X).
//   q0 = p.x p.y p.z | 2^ex (float)                                        child box = p + q * 2^e per axis
//   q1 = qlo.x[4] qlo.y[4] qlo.z[4] qhi.x[4]   (4 x u8 per word, child i in byte i)
//   q2 = qhi.y[4] qhi.z[4] | 2^ey 2^ez (floats)
//   q3 = child[4] (same encoding as KzNode::child; an empty slot has qlo = 255, qhi = 0)
// Quantisation rounds outward against the SAME float expression the kernel evaluates, so the boxes stay conservative.
struct KzNode4 { float p[3]; float scaleX; uint32_t qlo[3]; uint32_t qhi[3]; float scaleY, scaleZ; uint32_t child[4]; };      // scale = 2^e per axis, as floats (the node step multiplies them by 1/d: no exponent unpacking)
static_assert(sizeof(KzNode4) == 64, "BVH4 packet must be 64 B");

// Leaf triangle in Moeller-Trumbore form, 48 B, three quads: p0.xyz e1.x | e1.y e1.z e2.x e2.y | e2.z mesh prim gid
struct KzTri { float p0[3]; float e1[3]; float e2[3]; uint32_t mesh, prim, gid; };
static_assert(sizeof(KzTri) == 48, "leaf triangle must be 48 B");

// Shading record per triangle (indexed by gid), 112 B, seven quads: p0 p1 p2 n0 n1 n2 uv0 uv1 uv2 | mesh prim bsdf lightFlags.
// One hop replaces the reference's F -> V/N/UV double gather (accel.cpp:133-136,167-169), and the last quad the leaf-triangle -> mesh row
// hops (geomID -> mesh -> getBSDF / getLight / hasNormals): a hit record names the gid, everything shading needs hangs off this record.
// lightFlags = (light row + 1) << 2 | mesh flags (bit0 hasN, bit1 hasUV); light row -1 = not an emitter.
struct KzTriShade { float p[9]; float n[9]; float uv[6]; uint32_t mesh, prim, bsdf, lightFlags; };
static_assert(sizeof(KzTriShade) == 112, "shading record must be 112 B");

// Per-mesh row, 32 B.
struct KzMeshRow {
    int32_t bsdf;          // index into bsdfs (never -1 on device: the default diffuse is materialised)
    int32_t light;         // index into lightRows or -1
    uint32_t flags;        // bit0 hasN, bit1 hasUV
    uint32_t triOffset;    // gid of face 0
    uint32_t nF;
    uint32_t pad[3];
};
// Per-light row (Scene::m_lights order, scene.cpp:42-46), 48 B.
struct KzLightRow {
    float radiance[3];     // intensity * color (light.cpp:13)
    int32_t primaryVisibility;
    uint32_t mesh;
    uint32_t triOffset;    // gid of the light mesh's face 0
    uint32_t nF;
    uint32_t cdfOffset;    // into cdf[] (nF+1 floats, dpdf.h)
    float normalization;   // DiscretePDF::m_normalization = Mesh::pdf() (mesh.h:165-168)
    uint32_t hasN;         // the light mesh has vertex normals (Mesh::sample interpolates them, mesh.cpp:122-128)
    uint32_t pad[2];
};
// Flattened texture trees (texture.cpp): every KzTexture root becomes a postfix program over a small operand stack,
// so the device needs no recursion. 32 B per op.
enum { KZ_TOP_CONST = 0, KZ_TOP_IMAGE = 1, KZ_TOP_RAMP = 2, KZ_TOP_BLEND = 3 };
struct KzTexOp {
    uint32_t op;           // KZ_TOP_*
    uint32_t a;            // IMAGE: image row; BLEND: KZ_BLEND_*
    float f0, f1, f2;      // CONST: colour; IMAGE: f0 = scale, f1 = bits(texel offset / 16), f2 = bits((width - 1) | (height - 1) << 16) - the image row, so that a lookup
                           //   loads the op and then the texels, nothing in between; RAMP: f0 = min, f1 = max
    uint32_t b;            // IMAGE: bit 0 = srgb, bits 1..7 = KZ_TEXFILTER_*, bits 8..15 = channels, bits 16..23 = KZ_PIXEL_*
    uint32_t pad[2];
};
static_assert(sizeof(KzTexOp) == 32, "texture op must be 32 B");
struct KzTexProg { uint32_t start, count; };      // count: low half = ops, high half = the operand stack depth the program needs
struct KzImageRow { uint64_t offset; int32_t width, height, channels, format; };   // offset in bytes into the texel blob, 16-B aligned
static_assert(sizeof(KzBSDF) == 128, "BSDF row must be 128 B");

// pcg32 jump-ahead pair for advance(sampleIndex * 65536): state' = mult*state + inc*plus (pcg32.h:145-166 is
// linear in `inc`, so the pair is pixel independent and tabulated once per sample index).
struct KzPcgJump { uint64_t mult, plus; };

// Render constants, passed to kernels by value.
struct KzParams {
    // camera (camera.cpp:35-91)
    float s2c[16];
    float c2w[16];
    float invW, invH, nearClip, farClip;
    int32_t width, height, border;
    // integrator (integrator.cpp:187-193)
    int32_t maxDepth; float traceBias; int32_t regularization; float accumulatedRoughness;
    // sampler
    int32_t samplerType; uint32_t sampleCount; uint64_t seed; int32_t pixelTileSize;
    int32_t resX, resY;                  // stratified resolution / correlated m_resolution
    int32_t sppPow2; float invSpp;       // sampleCount is a power of two: x / sampleCount == x * invSpp bit for bit (no underflow: x >= 2^-17 or 0)
    int32_t cameraType; float apertureRadius, focusDistance;
    // lights / background
    uint32_t nLights; float lightPickPdf;
    float lightPickScale;                // nLights when that is a power of two (x / lightPickPdf == x * nLights bit for bit), else 0
    int32_t bgPresent; float bgRadiance[3];          // constant background: intensity * colour
    int32_t bgImage; float bgIntensity;             // environment map: row of `images` (-1: none) and the intensity it is scaled by
    int32_t bgFilter; int32_t bgPad_;               // KZ_TEXFILTER_* of that lookup (the nested texture's `filter`)
    // film (block.cpp:13-21)
    float filterRadius, lookupFactor; int32_t tapLo, tapHi;
    uint32_t rootRef;
    uint32_t rootRef4; int32_t stackBound4;     // BVH4 root and the worst-case traversal stack depth
    // shadow rays: invisible-light triangles (lightPrimaryVisibility == false) are few; their box + list make the
    // any-hit form of the shadow test exact (kz_devfn.h shadowOccluded)
    int32_t shadowFast; uint32_t nIlTris; float ilLo[3], ilHi[3];
    int32_t anyInvisibleLight; int32_t stackDepth;
    uint32_t ilGidLo, ilGidSpan;         // every triangle of an invisible light has gid - ilGidLo <= ilGidSpan (a prefilter: other triangles may too)
    // pixel beams of a pinhole camera whose sample -> near-plane map is affine (kz_wf_beam): nearP(sx, sy) = beamA + sx * beamU + sy * beamV in WORLD
    // axes (the 3x3 of c2w applied), apex beamO
    int32_t beamOk; float beamO[3], beamA[3], beamU[3], beamV[3];
    int32_t bsdfExt;                     // bit mask (kz_devfn.h KZ_X_*) of what the BSDF rows need beyond constant diffuse / kazenstandard: 1 other models, 2 texture-backed
                                         // parameters, normal maps (selects the larger kernel variants)
};

// Device pointers (all HBM-resident after kz_scene_upload).
struct KzDevTables {
    const KzNode *nodes;
    const KzNode4 *nodes4;
    const KzTri *tris;
    const KzTriShade *shade;
    const KzMeshRow *meshes;
    const KzBSDF *bsdfs;
    const KzLightRow *lights;
    const float *cdf;
    const float *pmj;           // [5][65536][2]: the table entries as the floats GetPMJ02BNSample returns, (float)(u32 * 0x1p-32) (pmj02table.h:28-29), converted once on the host
    const float *bn;            // [48][128][128]: BlueNoise()'s value / 65535.f (bluenoise.h:16-23), divided once on the host (same IEEE division)
    const float *pixelSamples;  // pmj02bn pixel table (sampler.cpp:291-309), 2 floats per entry
    const KzPcgJump *jump;      // [sampleCount]
    const float *filter;        // [33]
    const KzTri *ilTris;        // invisible-light triangles (<= 64) for the exact any-hit shadow test
    const KzTexProg *texProgs;  // [nTextures]
    const KzTexOp *texOps;
    const KzImageRow *images;
    const uint8_t *texels;
    uint32_t texPow2;           // every image has power-of-two sides: the periodic wrap of a texel coordinate is a mask (kz_scene_create)
};

struct KzScene {
    // flattened host tables
    std::vector<KzNode> nodes;
    std::vector<KzNode4> nodes4;
    std::vector<KzTri> tris;
    std::vector<KzTriShade> shade;
    std::vector<KzMeshRow> meshRows;
    std::vector<KzBSDF> bsdfs;
    std::vector<KzLightRow> lightRows;
    std::vector<float> cdf;
    std::vector<float> pmj;             // 2 floats per entry
    std::vector<float> bn;
    std::vector<float> pixelSamples;
    std::vector<KzPcgJump> jump;
    std::vector<KzTri> ilTris;
    std::vector<KzTexProg> texProgs;
    std::vector<KzTexOp> texOps;
    std::vector<KzImageRow> images;
    std::vector<uint8_t> texels;
    float filter[KZ_FILTER_RESOLUTION + 1];
    KzParams prm;
    uint32_t texPow2 = 1;               // every image of the scene has power-of-two sides
    KzBvhInfo bvh;
    // device replicas, one per GPU the scene is resident on (KzReplicaSet, owned by kz_render.hip; created with the scene)
    void *dev = nullptr;
};

// kz_bvh.cpp
struct KzBuildTri { float v[3][3]; uint32_t mesh, prim, gid; };
int kz_build_bvh(const std::vector<KzBuildTri> &in, std::vector<KzNode> &nodes, std::vector<KzTri> &tris,
                 uint32_t &rootRef, KzBvhInfo &info, std::string &err);
int kz_collapse_bvh4(const std::vector<KzNode> &nodes, uint32_t rootRef, std::vector<KzNode4> &out, uint32_t &rootRef4, int &stackBound);

// kz_host.cpp
int kz_fail(int code, const char *fmt, ...);

// kz_render.hip
void kz_device_init(KzScene *scene);          // empty replica set
void kz_device_release(KzScene *scene);
