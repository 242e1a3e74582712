/*
 * kazen_mi355x.h — C ABI of the MI355X-native path-tracing core for nano-kazen.
 *
 * This is the drop-in boundary for ONE hot path of the reference: the region
 * src/kazen/renderer.cpp:85-133 (tbb::parallel_for over blocks -> renderBlock ->
 * renderSample -> PathMisIntegrator::Li -> Accel::rayIntersect -> ImageBlock::put)
 * that sits behind  void kazen::renderer::render(Scene*, const std::string&)
 * (include/kazen/renderer.h:10, src/kazen/renderer.cpp:72).
 *
 * A per-ray virtual Integrator::Li() cannot be a GPU boundary, so the plugin
 * surface is kept as a *description*: the same type strings, property names and
 * defaults as the reference's KAZEN_REGISTER_CLASS registry, flattened to PODs.
 * Plain pointers and sizes only; no C++/torch types; errors are int codes plus a
 * thread-local message (kz_last_error), never exceptions.
 *
 * All host pointers inside KzSceneDesc are BORROWED for the duration of
 * kz_scene_create() only (the library copies what it needs, reference analogue:
 * Accel::build shares Mesh buffers with Embree, src/kazen/accel.cpp:45-46).
 * Output buffers are caller-owned.
 */
#ifndef KAZEN_MI355X_H
#define KAZEN_MI355X_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZ_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------- */
enum {
    KZ_OK = 0,
    KZ_ERR_INVALID_ARG = 1,   /* null pointer, bad size, index out of range     */
    KZ_ERR_UNSUPPORTED = 2,   /* plugin type outside the hot path (never a silent fallback) */
    KZ_ERR_NO_DEVICE = 3,     /* no HIP device / HIP extension not usable        */
    KZ_ERR_HIP = 4,           /* a HIP runtime call failed (message has details) */
    KZ_ERR_STATE = 5,         /* call order violated (e.g. render before upload) */
    KZ_ERR_OOM = 6
};

/* ---- plugin type tags (reference registry names in comments) ------------ */
enum { KZ_BSDF_DIFFUSE = 0        /* "diffuse"       src/kazen/bsdf.cpp:20-92     */,
       KZ_BSDF_KAZENSTANDARD = 1  /* "kazenstandard" src/kazen/bsdf.cpp:1157-1418 */,
       KZ_BSDF_MIRROR = 2         /* "mirror"        src/kazen/bsdf.cpp:161-196   */,
       KZ_BSDF_DIELECTRIC = 3     /* "dielectric"    src/kazen/bsdf.cpp:98-155    */,
       KZ_BSDF_GGX = 4            /* "ggx"           src/kazen/bsdf.cpp:629-689 (constanttexture albedo) */,
       KZ_BSDF_ROUGHCONDUCTOR = 5 /* "roughconductor" src/kazen/bsdf.cpp:692-811  */,
       KZ_BSDF_ROUGHPLASTIC = 6   /* "roughplastic"  src/kazen/bsdf.cpp:814-943   */,
       KZ_BSDF_ROUGHDIELECTRIC = 7/* "roughdielectric" src/kazen/bsdf.cpp:947-1145 */,
       KZ_BSDF_NORMALMAP = 8      /* "normalmap"     src/kazen/bsdf.cpp:281-417 (wraps row `nested`, texture `normalTex`) */ };
enum { KZ_TEX_CONSTANT = 0        /* "constanttexture" src/kazen/texture.cpp:10-32   */,
       KZ_TEX_IMAGE = 1           /* "imagetexture"    src/kazen/texture.cpp:36-98   */,
       KZ_TEX_COLORRAMP = 2       /* "colorramp"       src/kazen/texture.cpp:149-195 */,
       KZ_TEX_BLEND = 3           /* "blend"           src/kazen/texture.cpp:199-270 */ };
enum { KZ_BLEND_MIX = 0, KZ_BLEND_MULTIPLY = 1, KZ_BLEND_NONE = 2 /* any other "blendmode" string: evaluates to 0, texture.cpp:236 */ };
enum { KZ_PIXEL_U8 = 0            /* value/255, what OpenImageIO hands out for 8-bit files */, KZ_PIXEL_F32 = 1 };
#define KZ_TEX_MAX_DEPTH 8               /* operand-stack depth of a flattened texture tree */
enum { KZ_SAMPLER_INDEPENDENT = 0 /* "independent"   src/kazen/sampler.cpp:18-71   */,
       KZ_SAMPLER_PMJ02BN = 1     /* "pmj02bn"       src/kazen/sampler.cpp:273-390 */,
       KZ_SAMPLER_STRATIFIED = 2  /* "stratified"    src/kazen/sampler.cpp:81-156  */,
       KZ_SAMPLER_CORRELATED = 3  /* "correlated"    src/kazen/sampler.cpp:176-269 */ };
enum { KZ_CAMERA_PERSPECTIVE = 0  /* "perspective"   src/kazen/camera.cpp:14-131   */,
       KZ_CAMERA_THINLENS = 1     /* "thinlens"      src/kazen/camera.cpp:133-270  */ };
enum { KZ_INTEGRATOR_PATH_MIS = 0 /* "path_mis"      src/kazen/integrator.cpp:185-355 */ };
enum { KZ_FILTER_GAUSSIAN = 0     /* "gaussian"      src/kazen/rfilter.cpp:10-31   */,
       KZ_FILTER_MITCHELL = 1     /* "mitchell"      src/kazen/rfilter.cpp:39-70   */,
       KZ_FILTER_TENT = 2         /* "tent"          src/kazen/rfilter.cpp:73-88   */,
       KZ_FILTER_BOX = 3          /* "box"           src/kazen/rfilter.cpp:90-102  */ };

#define KZ_FILTER_RESOLUTION 32          /* KAZEN_FILTER_RESOLUTION, include/kazen/rfilter.h:6 */
#define KZ_PMJ02BN_SETS 5                /* include/kazen/pmj02table.h:10 */
#define KZ_PMJ02BN_SAMPLES 65536         /* include/kazen/pmj02table.h:11 */
#define KZ_BLUENOISE_TEXTURES 48         /* include/kazen/bluenoise.h:9 */
#define KZ_BLUENOISE_RES 128             /* include/kazen/bluenoise.h:8 */

/* ---- scene description -------------------------------------------------- */

/* A decoded raster (what OpenImageIO's ImageInput would deliver): row 0 = top scan line, `channels` interleaved.
 * Files with fewer than 3 channels fill the missing ones with 0 (TextureOpt::fill), more than 3 are truncated. */
typedef struct KzImage {
    const void *pixels;         /* height*width*channels values of `format`            */
    int32_t width, height, channels;
    int32_t format;             /* KZ_PIXEL_*                                          */
} KzImage;

/* Texture<Color3f> node (texture.h; texture.cpp). Children are indices into KzSceneDesc.textures, -1 = not attached:
 *   COLORRAMP: child[0] = nested (absent: evaluates to 0, texture.cpp:170)
 *   BLEND:     child[0] = "mask" (absent: 0.5), child[1] = "input1" (absent: 0), child[2] = "input2" (absent: 1)
 * IMAGE: texture.cpp:46-64 calls OpenImageIO's TextureSystem::texture(s = u*scale, t = (1-v)*scale, zero derivatives,
 * periodic wrap) with a default-constructed TextureOpt. OpenImageIO is not part of the reference checkout, so the filter behind
 * that call cannot be restated from source; it is an explicit FIELD here (`filter`), over the full-resolution level with texel
 * centres at (i+0.5)/width, followed by Color3f::toLinearRGB when srgb != 0:
 *   KZ_TEXFILTER_BILINEAR (0, the default)  the 2 x 2 lookup SURVEY 8f rank 4 specifies;
 *   KZ_TEXFILTER_BICUBIC  (1)               the 4 x 4 cubic B-spline - what OpenImageIO's default interpolation mode ("smart bicubic": bicubic
 *                                           whenever the lookup MAGNIFIES, which zero derivatives always do) most likely evaluates. HAZARD, recorded
 *                                           in DESIGN.md 2: which of the two the reference's renders used cannot be settled without OpenImageIO. */
#define KZ_TEXFILTER_BILINEAR 0
#define KZ_TEXFILTER_BICUBIC 1
typedef struct KzTexture {
    int32_t type;               /* KZ_TEX_*                                            */
    float color[3];             /* CONSTANT: "color" (default 0.5)                     */
    int32_t image;              /* IMAGE: index into KzSceneDesc.images                */
    float scale;                /* IMAGE: "scale" (default 1)                          */
    int32_t srgb;               /* IMAGE: "colorspace" == "srgb" (the default)         */
    float rampMin, rampMax;     /* COLORRAMP: "min" (0), "max" (1)                     */
    int32_t blendMode;          /* BLEND: KZ_BLEND_* ("blendmode", default "mix")      */
    int32_t child[3];
    int32_t filter;             /* IMAGE (also as the nested texture of a background): KZ_TEXFILTER_*  */
    int32_t pad_[2];
} KzTexture;

/* BSDF row. Replaces the BSDF subclasses; parameters that the reference reads through a Texture<Color3f> child
 * (diffuse/lambertian/ggx "albedo", kiss "baseColor" / "roughness" / "metallic") are either folded constants
 * (texture id 0) or a 1-based index into KzSceneDesc.textures (id k > 0 -> textures[k-1]); zero-initialised rows
 * therefore mean "constants only". Defaults are the reference's PropertyList defaults (bsdf.cpp:23, :1160-1167). */
typedef struct KzBSDF {
    int32_t type;               /* KZ_BSDF_*                                          */
    float albedo[3];            /* diffuse: "albedo"       (default 0.5)              */
    float baseColor[3];         /* kiss: nested texture id "baseColor"                */
    float roughness;            /* kiss: texture id "roughness" (.r())                */
    float metallic;             /* kiss: texture id "metallic"  (.r())                */
    float anisotropy;           /* default 0                                          */
    float specular;             /* default 0.5                                        */
    float specularTint;         /* default 0.5                                        */
    float clearcoat;            /* default 0                                          */
    float clearcoatRoughness;   /* default 0.5                                        */
    float sheen;                /* default 0                                          */
    float sheenTint;            /* default 0.5                                        */
    float intIOR;               /* dielectric: "intIOR", default 1.5046               */
    float extIOR;               /* dielectric: "extIOR", default 1.000277             */
    float alpha;                /* roughconductor/roughplastic: "alpha" (default 0.1), roughdielectric: "roughness" (0.1), ggx: "roughness" (0.5);
                                   the raw property: the library applies max(0.001, x^2) where the constructor does - unless alphaResolved says otherwise */
    float condEta[3];           /* roughconductor: eta of "material" (Au default / Cu / Cr, bsdf.cpp:795-806) */
    float condK[3];             /* roughconductor: k                                    */
    int32_t albedoTex;          /* diffuse ("lambertian", bsdf.cpp:202-276) / ggx albedo, kiss baseColor: texture id or 0   */
    int32_t roughnessTex;       /* kiss roughness (.r()): texture id or 0              */
    int32_t metallicTex;        /* kiss metallic  (.r()): texture id or 0              */
    int32_t normalTex;          /* normalmap: texture id (required)                    */
    int32_t nested;             /* normalmap: index of the wrapped BSDF row (not itself a normalmap) */
    int32_t alphaResolved;      /* roughconductor / roughplastic / roughdielectric only: 1 = `alpha` already IS the constructor's m_alpha = max(0.001, sqr(property))
                                   (bsdf.cpp:696-700, :818-822, :956-959) - what an adapter inside a kazen tree can read: the plugins keep m_alpha, not the property.
                                   0 (zero-initialised rows, every earlier caller) = the raw property, as above */
    int32_t pad_;
} KzBSDF;                       /* 128 bytes */
/* albedo doubles as: "kd" of roughplastic (default 0.5), the constanttexture albedo of ggx. */

/* "area" light (src/kazen/light.cpp:7-66). radiance = intensity * color. */
typedef struct KzLight {
    float color[3];             /* default 1                                          */
    float intensity;            /* default 1                                          */
    int32_t primaryVisibility;  /* "lightPrimaryVisibility", default 0 (false)        */
} KzLight;

/* Triangle mesh, buffer layout of kazen::Mesh (include/kazen/mesh.h:176-179):
 * V = Eigen col-major 3 x nV floats (xyz stride 12 B), N same (may be NULL),
 * UV = 2 x nV (may be NULL), F = 3 x nF uint32 (stride 12 B). */
typedef struct KzMesh {
    const float *V;
    const float *N;             /* NULL: geometric frame, no terminator offset (the reference is UB here, accel.cpp:136) */
    const float *UV;            /* NULL: prim (u,v) kept in its.uv, Frame(n) tangents  */
    const uint32_t *F;
    uint32_t nV;
    uint32_t nF;
    int32_t bsdf;               /* index into KzSceneDesc.bsdfs; -1 = default Diffuse(albedo 0.5), mesh.cpp:25-28 */
    int32_t light;              /* index into KzSceneDesc.lights; -1 = not an emitter  */
} KzMesh;

/* Reconstruction filter child of the camera (camera.cpp:64-67: default gaussian). */
typedef struct KzFilter {
    int32_t type;               /* KZ_FILTER_*                                         */
    float radius;               /* gaussian/mitchell default 2; tent 1; box 0.5        */
    float stddev;               /* gaussian, default 0.5                               */
    float B, C;                 /* mitchell, default 1/3                               */
} KzFilter;

/* "perspective" camera (src/kazen/camera.cpp:16-33). */
typedef struct KzCamera {
    int32_t type;               /* KZ_CAMERA_PERSPECTIVE                               */
    int32_t width, height;      /* default 1280 x 720                                  */
    float toWorld[16];          /* row-major 4x4 camera-to-world ("toWorld")           */
    float fov;                  /* horizontal, degrees, default 30                     */
    float nearClip, farClip;    /* defaults 1e-4, 1e4                                  */
    float apertureRadius;       /* thinlens: "apertureRadius", default 1                */
    float focusDistance;        /* thinlens: "focusDistance", default 0                 */
    const float *sampleToCamera;/* optional row-major 4x4 override (an adapter inside a kazen tree may hand over
                                   Eigen's own inverse, camera.cpp:60-62); NULL = computed by the library */
    KzFilter rfilter;
} KzCamera;

/* Sampler (src/kazen/sampler.cpp). The table pointers are what the reference links
 * from pmj02table.cpp / bluenoise.cpp (missing from the checkout): the adapter passes
 * kazen::pmj02bnSamples and kazen::BlueNoiseTextures verbatim. */
typedef struct KzSampler {
    int32_t type;               /* KZ_SAMPLER_*                                        */
    uint32_t sampleCount;       /* "sampleCount" (stratified/correlated round it up: kz_scene_sample_count tells) */
    int32_t resolution;         /* stratified: "resolution", default 4 (sampler.cpp:86) */
    int32_t pad_;
    uint64_t seed;              /* "seed" (independent: the reference never initialises it; we define 0) */
    const uint32_t *pmj02bnSamples; /* [5][65536][2] fixed-point 2^-32, pmj02bn only   */
    const uint16_t *blueNoise;      /* [48][128][128], indexed [tex][x][y], pmj02bn only */
} KzSampler;

/* "path_mis" (src/kazen/integrator.cpp:187-193). */
typedef struct KzIntegrator {
    int32_t type;               /* KZ_INTEGRATOR_PATH_MIS                              */
    int32_t maxDepth;           /* default 5, capped at 512                            */
    float traceBias;            /* default 1e-3                                        */
    int32_t regularization;     /* default 0                                           */
    float accumulatedRoughness; /* default 0.5                                         */
} KzIntegrator;

/* "background" texture (texture.cpp:104-145) with its nested texture, looked up by DIRECTION (Scene::getBackgroundColor,
 * scene.cpp:54-79 -> BackgroundTexture::eval(Vector3f) -> nested->eval(Vector3f)):
 *   texture == 0                  the nested texture is the constant `color` (ConstantTexture::eval(Vector3f), texture.cpp:20-22)
 *   texture == k > 0              textures[k-1] is the nested texture: CONSTANT -> its colour; IMAGE -> the environment lookup of
 *                                 ImageTexture::eval(Vector3f) (texture.cpp:66-80: no `scale`, no colour-space conversion);
 *                                 COLORRAMP / BLEND -> 0 (they inherit Texture::eval(Vector3f), texture.h:13)
 * The environment lookup is OpenImageIO's TextureSystem::environment (un-vendored). This library DECLARES it as the latitude-longitude
 * map with y up that OpenImageIO applies to OpenEXR environment maps:
 *     s = atan2f(-d.x, d.z) / (2 pi) + 0.5      t = 0.5 - atan2f(d.y, hypotf(d.z, -d.x)) / pi      (NaN -> 0)
 * filtered as the nested texture's `filter` says (bilinear by default) over the full-resolution level, texel centres at (i + 0.5) / res,
 * s periodic, t clamped at the poles. */
typedef struct KzBackground {
    int32_t present;            /* 0: Scene::getBackgroundColor returns 0 (scene.cpp:55-56) */
    float color[3];
    float intensity;            /* default 1                                           */
    int32_t texture;            /* 0 = `color`; k > 0 = textures[k-1] (see above)      */
} KzBackground;

typedef struct KzSceneDesc {
    uint32_t abiVersion;        /* KZ_ABI_VERSION                                      */
    const KzMesh *meshes;   uint32_t nMeshes;
    const KzBSDF *bsdfs;    uint32_t nBsdfs;
    const KzLight *lights;  uint32_t nLights;
    KzCamera camera;
    KzSampler sampler;
    KzIntegrator integrator;
    KzBackground background;
    const KzTexture *textures; uint32_t nTextures;
    const KzImage *images;     uint32_t nImages;
} KzSceneDesc;

/* ---- rendering ---------------------------------------------------------- */

/* A rectangle of pixels [x0,x0+w) x [y0,y0+h): the sharding unit (multiples of the
 * reference's 32x32 KAZEN_BLOCK_SIZE, include/kazen/block.h:8, are natural). */
typedef struct KzTile { int32_t x0, y0, w, h; } KzTile;

/* Knobs of the persistent kernels. Zero = the library default, which is what the measured numbers use. They are part
 * of the ABI so that nothing behind it depends on process-global state: the library reads no environment variable.
 * The dev* words keep the ABI v4 layout: they select kernels of rejected experiments in development builds of the library
 * (named in kazen_mi355x_dev.h); the product library answers a non-zero value with KZ_ERR_UNSUPPORTED. Leave them 0. */
typedef struct KzTuning {
    int32_t refill;             /* a wave refills idle lanes once fewer than this many are busy (default 40; shadow rays 32) */
    int32_t postpone;           /* node phase goes on while at least this many lanes hold inner nodes (default 24)  */
    int32_t batch;              /* queue entries a wave reserves per global atomic (default 128; up to 8 x as many while much of the queue is left) */
    int32_t traceBlocksPerCU;   /* 256-thread workgroups per CU of the traversal kernels (default 8)                 */
    int32_t shadeBlocksPerCU;   /* same for the shade kernel (default 4; 6 with extended BSDFs)                       */
    int32_t ldsStack;           /* per-lane traversal stack entries kept in LDS before the global overflow (default 16) */
    int32_t dev0;
    int32_t packetPrimary;      /* primary rays: 0 = default (pixel beams, else shared-stack packet traversal), 1 = per-lane, 2 = packet */
    int32_t dev1, dev2;
    int32_t filmGather;         /* film reconstruction (running tap sums per pixel, resolved once per call): 0 = default (2 lane groups per pixel for filters of <= 5 taps
                                   per axis, 4 beyond), 3 = one lane per pixel (round 2's kernel, <= 5 taps, kept for comparison: the same sums bit for bit).
                                   1 (the staged gather kernel of rounds 1-5) is gone: KZ_ERR_UNSUPPORTED */
    int32_t dev3;
    int32_t sppPerPass;         /* samples of a pixel per pass: 0 = default. A pass covers pixPerPass x sppPerPass = passItems (pixel, sample)
                                   items: sppPerPass = 0 means "every pixel of the tile set, as many samples as fit" - unless fewer than 64 would
                                   fit and the call asks for at least 64: then 256 samples (or all, if fewer) of pixel chunks; above 64 the count is
                                   rounded down to a multiple of 64 when that costs no extra pass; n > 0 means n samples
                                   (or all the call asks for, if fewer) of as many pixels as fit, pixel chunks in tile order           */
    int32_t dev4, dev5;
    int32_t streamPriority;     /* HIP priorities of the internal pass streams: 0 = default, 1 = all at the default priority, 2 = alternating
                                   least / greatest, 3 = cycling least / default / greatest */
} KzTuning;

/* Dynamic dealing of ONE tile list over several takers - devices of a process (kz_render_multi) or processes of a node (the counter then lives in
 * memory they share) - the reference's BlockGenerator::next (block.cpp:117-148) with an atomic counter instead of a mutex. Every taker calls
 * kz_render_tiles with the SAME list and a dealer pointing at the same counter: the call prepares the whole list as its tile set once, then renders
 * the batches [b, b + batchTiles) it wins from the counter, keeping its passes in flight across batch boundaries, until the list is dealt. The batches
 * it took come back in `taken` (pairs begin, end of tile indices), which is what the taker hands to kz_film_download_tiles afterwards. */
typedef struct KzTileDealer {
    volatile uint32_t *counter; /* index of the next undealt tile; zero it before the first taker starts */
    uint32_t batchTiles;        /* tiles per batch; 0 = about two passes' worth of (pixel, sample) items, at most 1/(4 x takers) of the list */
    uint32_t takers;            /* how many takers share the counter (only used for the default batch size; 0 = 1) */
    uint32_t *taken;            /* out: begin0, end0, begin1, end1, ... */
    uint32_t takenCap;          /* capacity of `taken` in uint32 (2 per batch): a call stops taking batches when it is full */
    uint32_t *nTaken;           /* out: uint32 written to `taken` */
    volatile uint32_t *agreed;  /* optional (NULL: unchecked): a second word the takers share, zeroed with the counter. The counter only ever advances by the batch
                                   size, which every taker resolves for itself (batchTiles, or the default from its pass size, sample range and `takers`): takers
                                   whose options differ would deal overlapping or misaligned batches without an error. The first taker publishes what it resolved
                                   (batch size and length of the list) in this word; a taker that resolved anything else fails with KZ_ERR_INVALID_ARG before it
                                   takes a tile. REQUIREMENT either way: all takers of one counter pass the same list, sample range, pass options and `takers`. */
} KzTileDealer;

typedef struct KzRenderOpts {
    uint32_t sampleBegin;       /* render sample indices [sampleBegin, sampleEnd) of every pixel;       */
    uint32_t sampleEnd;         /* 0,0 = all of sampler.sampleCount                                      */
    const KzTile *tiles;        /* NULL = whole image                                                    */
    uint32_t nTiles;
    int32_t pipeline;           /* 0 = library default, 1 = megakernel, 2 = wavefront                    */
    int32_t accumulate;         /* 0 = clear the device film first, 1 = add to what is there             */
    void *stream;               /* hipStream_t to launch on (NULL = the null stream)                     */
    /* ---- ABI v3 (all zero = defaults) ---- */
    int32_t device;             /* the replica to render on: a HIP device index kz_scene_upload was called with */
    int32_t passesInFlight;     /* passes kept in flight on internal streams, 1 .. 8; 0 = default: with passItems also 0, ONE pass at a time as large as the
                                   state budget allows (see passItems); with passItems given, or with a dealer, KZ_DEFAULT_PASSES_IN_FLIGHT */
    uint64_t passItems;         /* (pixel, sample) items per pass, lowered to fit maxStateBytes; 0 = default: 2^30 (175 GB of path state on a 288 GB card: fewer, longer
                                   kernels), 2^29 per context with a dealer, 2^27 when passesInFlight is given */
    uint64_t maxStateBytes;     /* cap on this replica's pass contexts (path state + sample records); 0 = min(3/4 of the device's memory, what is free + what the
                                   replica already holds for this purpose). The film and its running tap sums (taps^2 x 16 B per pixel of the frame: 829 MB at
                                   1920x1080 with the default filter) belong to the replica like the scene tables and are not part of this budget */
    KzTuning tune;
    int32_t tileDealing;        /* kz_render_multi: 0 = static (kz_deal_tiles: by area), 1 = dynamic (the devices take batches of tiles from a shared counter,
                                   the reference's BlockGenerator). The same paths and - since round 6 - the same film BIT FOR BIT either way and for any number
                                   of devices: a tile's rect holds what the tile's own pixels add, and the rects are added in tile order (H10) */
    int32_t packedOutput;       /* kz_render_tiles with a host buffer: 0 = the buffer receives the WHOLE film, 1 = the packed rects of the call's tiles
                                   (kz_tiles_packed_floats). Never inferred from the buffer's size. */
    /* ---- ABI v5 ---- */
    const struct KzTileDealer *dealer;   /* kz_render_tiles: NULL = render every tile of the list; else take batches of it from the dealer's counter */
    /* ---- ABI v6 ---- */
    int32_t shadowBeside;       /* where the shadow rays of a bounce run: 1 = in front of the bounce's closest-hit rays (one stream), 2 = beside them (a side stream of the
                                   pass context; the next shade waits for both), 0 = default: beside in passes of up to 2^27 items (2^24 with passes in flight or a dealer: they overlap each other already) - a small job is a chain of launches each
                                   as long as its slowest ray (BASELINE configs[0]: 2.47 -> 2.03 ms) - and MEASURED for larger ones: kernels that saturate the chip by
                                   themselves lose ~1 % sharing it (C4), kernels that do not - short-lived shadow rays: all 22 of the reference's scene/2022_q1 files - gain
                                   6 - 10 % at any size. The replica runs its first large pass in front, the next one of that size beside, a third as halves (passHalves),
                                   a fourth in front again, waits for that one when a fifth comes (kz_render is asynchronous but for such waits: it also stays one pass ahead of a context that is
                                   still growing) and keeps what was fastest per item, "in front" unless beaten by 3 % (with passes in flight, a dealer or the counters
                                   on: in front). The film is the same bits whatever this says. */
    int32_t passHalves;         /* 2 = a pass runs as two halves of its pixels side by side (two views of the pass context's arrays, the second on a stream of its own;
                                   disjoint pixels, so nothing orders their film stages): one half's shade kernel beside the other half's traversal - materials_scene + 7 %,
                                   C3 + 2 %, textured_scene - 2 %, C4 - 1.4 %. 1 = never. 0 = default: never for passes of up to 2^27 items, MEASURED above together with
                                   shadowBeside when that is 0 too (the third of four timed passes; the fastest of the three ways is kept, "one stream" unless beaten by 3 %).
                                   Ignored with passes in flight or a dealer (they overlap already). The film is the same bits whatever this says. */
} KzRenderOpts;
#define KZ_MAX_PASSES_IN_FLIGHT 8
#define KZ_DEFAULT_PASSES_IN_FLIGHT 2

typedef struct KzScene KzScene;

/* Build the immutable scene: copy + flatten the description, build the BVH on the host
 * (replaces Accel::build, accel.cpp:25-61, and Mesh::activate's light CDF, mesh.cpp:24-45,
 * and PerspectiveCamera::activate, camera.cpp:35-68). No GPU needed.
 * Size limits (KZ_ERR_UNSUPPORTED beyond them): leaf references address 2^28 triangles; the traversal kernels fetch a BVH4 packet at
 * table base + a 32-bit byte offset, so a tree may hold 2^26 packets (4 GB; a 1 M-triangle scene has ~0.5 M). */
int kz_scene_create(const KzSceneDesc *desc, KzScene **out);
void kz_scene_destroy(KzScene *scene);

/* Upload node/triangle/attribute/material/light/sampler tables to the HBM of `device` and allocate that device's film.
 * One KzScene (one host BVH build) can be resident on any number of devices: every call ADDS a replica (a second call for
 * the same device is a no-op). Calls that take no device argument address the PRIMARY replica, the one uploaded first.
 * Fails with KZ_ERR_NO_DEVICE when no GPU is usable. Thread-safe per (scene, device). */
int kz_scene_upload(KzScene *scene, int device);
/* Release the replica on `device` (-1: every replica). Must not run concurrently with any other call on that replica (the per-(scene, device)
 * re-entrancy covers rendering and downloading, not tearing a replica down under them); the same holds for kz_scene_destroy. */
int kz_scene_evict(KzScene *scene, int device);
/* The replacement for renderer.cpp:85-133: accumulate samples into the DEVICE film of replica opts->device
 * ((h+2b) x (w+2b) float4 = rgb*w, w; ImageBlock convention, block.cpp:30,56-85).
 * Asynchronous on opts->stream. Re-entrant per (scene, device): one host thread per GPU may call it concurrently.
 * A call that FAILS has waited for whatever it had launched (nothing of it is still running when the error comes back) and may have added some of its samples to the
 * film: render without `accumulate` (or kz_film_clear) before the film is used again. */
int kz_render(KzScene *scene, const KzRenderOpts *opts);

/* SURVEY 8b tile variant, the unit of multi-GPU sharding: render `tiles` on `device` (overriding opts->tiles / opts->device), wait for the
 * device, and hand back what those tiles produced. `film` may be NULL (nothing is copied: kz_film_download_tiles later), a buffer for the
 * PACKED film rects of the tiles (nFloats = kz_tiles_packed_floats: tile t is its (h + 2b) x (w + 2b) rect - the tile with its filter apron, the
 * extent of an ImageBlock of that size, block.cpp:14,30 - as consecutive rows, tiles in list order; a texel of a rect holds what the TILE'S OWN pixels
 * add to it - the ImageBlock of that tile - so where the aprons of two tiles overlap both carry their share and the merge adds them, in tile order, exactly
 * as the device's own film is resolved (block.cpp:87-96)), or a whole-film buffer ((h+2b)*(w+2b)*4 floats).
 * The packed form moves 1.13 x the film for a frame of 64 x 64 tiles however many devices share it.
 * Blocking; re-entrant per (scene, device). opts may be NULL (all samples, defaults). */
int kz_render_tiles(KzScene *scene, const KzRenderOpts *opts, const KzTile *tiles, uint32_t nTiles, int device,
                    float *film, size_t nFloats);
/* Size of that packed form, and the gather on its own: the packed rects of `tiles` from the film of replica `device` (device-side pack, one
 * D2H copy through a pinned staging buffer). */
int kz_tiles_packed_floats(const KzScene *scene, const KzTile *tiles, uint32_t nTiles, size_t *nFloats);
int kz_film_download_tiles(KzScene *scene, int device, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats);
/* ImageBlock::put(ImageBlock&) (block.cpp:87-96) for a list of blocks: adds the packed rects to a whole film ((height+2b) x (width+2b) x 4 floats)
 * in LIST order, on nThreads host threads over disjoint row bands (0 = up to 16); the result does not depend on the number of threads. */
int kz_film_merge_tiles(float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, uint32_t nTiles, const float *packed,
                        size_t nFloats, int32_t nThreads);
/* The same merge for rects that lie in different buffers (a multi-process launcher: every rank's packed rects where that rank put them): rects[t] points at the
 * (h + 2b) x (w + 2b) x 4 floats of tiles[t]. In ROW-MAJOR tile order (y0, then x0) the result is, bit for bit, the film one device resolves for itself and the film
 * kz_render_multi returns - whichever rank or device rendered which tile (H10). */
int kz_film_merge_rects(float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, const float *const *rects, uint32_t nTiles, int32_t nThreads);

/* The analogue of the reference's driver (renderer.cpp:94-127: tbb::parallel_for over blocks, then ImageBlock::put(ImageBlock&)
 * under a mutex, block.cpp:87-96) one level up: the image is cut into tileSize x tileSize tiles (a multiple of the 32-px
 * block; 0 = 64), ONE HOST THREAD PER DEVICE renders its tiles with kz_render_tiles - dealt by area beforehand (kz_deal_tiles) or, with
 * opts->tileDealing = 1, pulled in batches from a shared counter (the reference's BlockGenerator, block.cpp:117-148) - downloads the
 * packed rects of ITS tiles - each rect what the tile's own pixels add - and the rects are added into `film` in TILE order on the host (H10: the same film bit for bit
 * for static and dynamic dealing and for any number of devices; with the default tile the film of kz_render on ONE device). No collective, no peer access. Replicas are uploaded on demand BEFORE the clocks start: deviceMs (may be NULL) receives
 * each device's wall time of render + gather in ms. opts->tiles / opts->device / opts->stream are ignored. */
int kz_render_multi(KzScene *scene, const KzRenderOpts *opts, const int32_t *devices, uint32_t nDevices, int32_t tileSize,
                    float *film, size_t nFloats, float *deviceMs);

/* The tile dealing used by kz_render_multi, exported so that a multi-process launcher (one rank per GPU) deals identically:
 * tiles in row-major order go, largest first, to the part with the least area so far (ties: the lower part), which for the
 * interior tiles is a round-robin and spreads the smaller edge tiles evenly. Writes part `part` of `nParts` to out (up to cap
 * entries) and its size to *count; returns KZ_ERR_INVALID_ARG if cap is too small (count is still set). */
int kz_deal_tiles(int32_t width, int32_t height, int32_t tileSize, uint32_t nParts, uint32_t part, KzTile *out, uint32_t cap, uint32_t *count);

/* Blocking copy of the device film to the host: film = (h+2b)*(w+2b)*4 floats. */
int kz_film_download(KzScene *scene, float *film, size_t nFloats);
int kz_film_clear(KzScene *scene, void *stream);
int kz_film_dims(const KzScene *scene, int32_t *width, int32_t *height, int32_t *border);
/* ImageBlock::toBitmap (block.cpp:39-45): rgb = film.rgb / film.w (0 when w == 0). */
int kz_film_to_rgb(const float *film, int32_t width, int32_t height, int32_t border, float *rgb);

/* ImageBlock::toBitmap followed by the tone map of Bitmap::savePNG (bitmap.cpp:45-52: Color3f::toSRGB, clamp(255*v, 0, 255),
 * truncation to uint8), evaluated on the device: rgb8 = height*width*3 bytes, row 0 = top scan line — the buffer the
 * reference hands to its PNG writer. */
int kz_film_to_srgb8(KzScene *scene, uint8_t *rgb8, size_t nBytes);

/* Wait for everything queued on the primary replica's launch stream. */
int kz_sync(KzScene *scene);

/* Message of the last failing call of the calling thread (no exception crosses the ABI: int status + this string). */
const char *kz_last_error(void);
int kz_abi_version(void);
/* HIP devices visible to the process (0 without a GPU: kz_scene_upload then fails with KZ_ERR_NO_DEVICE - there is no CPU path). */
int kz_device_count(void);
/* Path-state memory outlives the replica that grew it: kz_scene_evict / kz_scene_destroy hand a replica's pass contexts (up to 3/4 of the device for
 * one default pass) to a per-device pool, and the next replica uploaded to that device - the next scene of the same process - renders in them at once
 * instead of waiting for the driver to wipe what its predecessor released. kz_device_trim gives the pooled memory of `device` back to the
 * driver: for a process that keeps running without rendering. */
int kz_device_trim(int device);

/* Everything else the library exports - per-replica forms of the calls above, counters and timings, the function-level query kernels and
 * known-answer checks the parity tests use, the names of the development tuning words - is declared in kazen_mi355x_dev.h. */

#ifdef __cplusplus
}
#endif
#endif /* KAZEN_MI355X_H */
