/* kazen_mi355x_dev.h - the development and test surface of libkazen_mi355x.so. A renderer that adopts the library needs
 * kazen_mi355x.h only (its twenty entry points); this header adds what the parity tests, the benchmarks and the profiling
 * scripts use: per-replica forms of the product calls, counters and stage timings, function-level query kernels (the very
 * device functions the path kernels call, on caller-supplied inputs), known-answer self-checks (kz_kat_*: stateless, in the product
 * library, so that what they check IS the product's code), the pass planner as a pure function, and - in development builds of the library only - the hooks that
 * are process-global state (failure injection, growth delay, trace, device aliasing) and the KzTuning words that select kernels of rejected experiments. */
#ifndef KAZEN_MI355X_DEV_H
#define KAZEN_MI355X_DEV_H
#include "kazen_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* KzTuning.dev*: honoured only by a library built with -DKZ_EXPERIMENTS (kz_build_flags() & KZ_BUILD_EXPERIMENTS); the product
 * library answers a non-zero value with KZ_ERR_UNSUPPORTED (nano-kazen_amd/csrc/variants/experiments/kz_experiments.h holds the kernels). */
#define KZ_TUNE_BVH2         dev0   /* 1 = per-lane traversal of the BVH2 instead of the quantised BVH4 */
#define KZ_TUNE_KEY_STACK    dev1   /* 1 = packet kernel without per-lane entry distances, 2 = per-lane kernel with them */
#define KZ_TUNE_LDS_TOP      dev2   /* n = that many BVH4 packets of the top of the tree staged in LDS (<= 1536) */
#define KZ_TUNE_LEAF_QUEUE   dev3   /* 2 = bounce / shadow traversal with a decoupled leaf phase (kz_wf_trace_dq) */
#define KZ_TUNE_LEGACY_TRACE dev4   /* 1 = the non-persistent round-1 traversal launches (kz_wf_extend / kz_wf_shadow) */
#define KZ_TUNE_MIXED_LAUNCH dev5   /* 1 = one launch for the shadow rays of a bounce and the closest-hit rays of the next */

/* Counters the kernels keep (all optional; zero unless requested with kz_set_stats). */
typedef struct KzStats {
    uint64_t samples;           /* (pixel,sample) pairs rendered                        */
    uint64_t rays;              /* closest-hit queries (Accel::rayIntersect calls)      */
    uint64_t nodeVisits;        /* 64-B BVH2 node packets fetched                       */
    uint64_t triTests;          /* 48-B leaf triangles tested (Moeller-Trumbore)        */
    uint64_t shadedHits;        /* post-intersection gathers (accel.cpp:113-236)        */
    uint64_t lightSamples;      /* Mesh::sample calls (mesh.cpp:108-133)                */
    uint64_t droppedSamples;    /* invalid radiance dropped by ImageBlock::put (block.cpp:57-61) */
    uint64_t beamPixels;        /* pixels whose camera rays were given a leaf list by the beam kernel, once per pixel chunk */
    uint64_t beamListEntries;   /* leaves on those lists                                  */
    uint64_t beamCompletePixels;/* pixels whose list holds every leaf the beam reaches (the other lists end at a distance t_valid) */
} KzStats;

/* Ray-level record mirroring what Accel::rayIntersect fills (accel.cpp:99-110 + 113-236). */
typedef struct KzHit {
    float t;                    /* +inf on miss                                          */
    float u, v;                 /* prim barycentrics, P=(1-u-v)p0+u p1+v p2 (accel.cpp:122-123) */
    int32_t mesh;               /* geomID, -1 on miss                                    */
    int32_t prim;               /* primID within the mesh                                */
    float p[3];                 /* its.p after the terminator offset                     */
    float uv[2];                /* its.uv                                                */
    float sh_s[3], sh_t[3], sh_n[3];  /* its.shFrame                                     */
    float geo_n[3];             /* its.geoFrame.n                                        */
} KzHit;

/* Host BVH statistics (node count, leaf count, max depth, SAH cost) for reports. */
typedef struct KzBvhInfo { uint32_t nNodes, nLeaves, nTris, maxDepth, maxLeafSize; float sahCost; double buildSeconds; } KzBvhInfo;
int kz_scene_bvh_info(const KzScene *scene, KzBvhInfo *out);
/* Sampler::getSampleCount() after the constructor's rounding (sampler.cpp:87-92, :181-187, :284-287). */
int kz_scene_sample_count(const KzScene *scene, uint32_t *out);

/* The devices the scene is resident on, primary first. */
int kz_scene_devices(const KzScene *scene, int32_t *devices, uint32_t cap, uint32_t *count);

/* ImageBlock::put(ImageBlock&) on the host (block.cpp:87-96): dst += src, element by element, in index order. */
int kz_film_merge(float *dst, const float *src, size_t nFloats);

/* kz_film_download / kz_film_clear / kz_sync for the replica on `device`. */
int kz_film_download_on(KzScene *scene, int device, float *film, size_t nFloats);
int kz_film_clear_on(KzScene *scene, int device, void *stream);
int kz_sync_on(KzScene *scene, int device);

/* Ray-level entry mirroring Accel::rayIntersect(ray, its, shadowRay=false) for n rays
 * (host arrays; o,d = n x 3 floats). For parity tests of traversal + post-intersection. */
int kz_trace_rays(KzScene *scene, uint32_t n, const float *o, const float *d,
                  const float *tmin, const float *tmax, KzHit *hits);

/* Debug / known-answer entry points (used by the parity tests, not by a renderer):
 * kz_render_samples: radiance of explicit (pixel, sample index) pairs = renderSample (renderer.cpp:20-40) without
 * the block.put; pxy = n x (x,y), out = n x (pixelSample.x, pixelSample.y, r, g, b).
 * kz_bsdf_query: BSDF::eval / pdf / sample (bsdf.h:80-108) of row bsdf[i] for local directions wi/wo (n x 3), with
 * its.accumulatedRoughness accRough[i] and the (sample1, sample2.x, sample2.y) triple s3; evalOut n x 3,
 * pdfOut n, sampleOut n x 7 = (weight rgb, sampled wo xyz, alive). uv (n x 2, may be NULL = 0) feeds the texture-backed
 * parameters; the intersection record is the identity frame with dpdu = +x (what a normalmap row perturbs). */
int kz_render_samples(KzScene *scene, uint32_t n, const int32_t *pxy, const uint32_t *idx, float *out);
int kz_bsdf_query(KzScene *scene, uint32_t n, const int32_t *bsdf, const float *wi, const float *wo, const float *accRough,
                  const float *s3, const float *uv, float *evalOut, float *pdfOut, float *sampleOut);
/* Texture<Color3f>::eval(uv) (texture.h) of textures[tex[i]] at uv (n x 2); out n x 3. */
int kz_texture_query(KzScene *scene, uint32_t n, const int32_t *tex, const float *uv, float *out);

/* Camera::sampleRay (camera.cpp:70-91 perspective, 191-223 thinlens) of the scene's camera for n pixel-sample positions
 * sxy (n x 2, pixel units) and aperture samples axy (n x 2, NULL = 0.5,0.5); out n x 8 = o xyz, d xyz, mint, maxt. */
int kz_camera_rays(KzScene *scene, uint32_t n, const float *sxy, const float *axy, float *out);
/* AreaLight::sample (light.cpp:16-34) via Mesh::sample (mesh.cpp:108-133) of light light[i] (index in Scene::m_lights
 * order) seen from ref (n x 3) with Mesh::sample's three next1D draws u3 (n x 3); out n x 14 = p xyz, n xyz, wi xyz,
 * pdf (solid angle, light.cpp:36-51), eval/pdf rgb (0 where the pdf is 0, nan or inf), triangle index. */
int kz_light_query(KzScene *scene, uint32_t n, const int32_t *light, const float *ref, const float *u3, float *out);

/* Statistics: enable=1 switches to the counting kernel variant (slower). */
int kz_set_stats(KzScene *scene, int enable);
int kz_get_stats(KzScene *scene, KzStats *out, int reset);

/* Average device time of the dominant kernel(s) of the last kz_render, in ms,
 * from hipEvents recorded on the launch stream (0 if none). */
int kz_last_kernel_ms(KzScene *scene, float *ms);
/* Device time per stage of the last pass (wavefront pipeline): out6 = generate, closest-hit traversal of the bounce rays, shade, shadow
 * traversal, film, camera rays (beam lists / list kernel / packet kernel / first-hit walk-through) - the per-kernel sums a
 * rocprofv3 --kernel-trace of the same run shows. A pass whose shadow rays ran BESIDE its closest-hit rays (KzRenderOpts::shadowBeside; KzPassInfo::shadowBeside says
 * whether the last pass did) shows the pair under the closest-hit traversal and ~0 under the shadow traversal: ask for shadowBeside = 1 to time them apart. */
int kz_last_stage_ms(KzScene *scene, float *out6);

/* What the last kz_render did on the primary replica: passes, (pixel, sample) items of a full pass (the TARGET: a pass context grows while the first passes
 * of a job already run - firstPassItems / largestPassItems say what the passes of this call really were), passes in flight, bytes of path state mapped. */
typedef struct KzPassInfo { uint32_t passes; uint32_t passesInFlight; uint64_t itemsPerPass; uint32_t sppPerPass; uint32_t pixels; uint64_t stateBytes;
                            uint32_t pixelsPerPass; uint32_t shadowBeside /* how the last pass ran: 0 = one stream, 1 = its shadow rays beside its closest-hit rays (KzRenderOpts::shadowBeside), 2 = as two halves (passHalves) */; uint64_t firstPassItems; uint64_t largestPassItems;
                            uint64_t contextItems;      /* items the first pass context holds NOW (it may still be growing towards itemsPerPass) */
                          } KzPassInfo;
int kz_last_pass_info(KzScene *scene, KzPassInfo *out);
/* What the replica on `device` (-1: the primary one) has measured about its LARGE passes (above 2^27 items, KzRenderOpts::shadowBeside = passHalves = 0): the four timed passes
 * of `items` items each and what it keeps for the scene. kept: -1 = not decided yet (timedPasses of the four have been launched), 0 = one stream, 1 = shadow rays beside the
 * closest-hit rays, 2 = two halves side by side; the times (ms, film stage included) are there once it has decided. */
typedef struct KzPassModeInfo { int32_t kept; uint32_t timedPasses; uint64_t items; float msOneStream[2]; float msShadowBeside; float msHalves; } KzPassModeInfo;
int kz_pass_mode_info(KzScene *scene, int device, KzPassModeInfo *out);
/* Why the pass context of the last kz_render stopped growing short of its target ("" if it did not): such a call succeeds on what there is. */
int kz_last_grow_note(KzScene *scene, char *buf, size_t cap);

/* ---- DEVELOPMENT BUILDS ONLY (a library compiled with -DKZ_EXPERIMENTS: kz_build_flags() & KZ_BUILD_EXPERIMENTS; nano-kazen_amd/csrc/variants/experiments).
 * These five are process-global state - the product library does not contain them (`nm -D libkazen_mi355x.so | grep kz_debug` is empty), so that nothing behind the
 * product ABI depends on state outside the objects the caller holds. The tests that need them load the development variant.
 *   kz_debug_fail_alloc    the nth device allocation made from now on by the calling thread fails with KZ_ERR_OOM (0 = off): a failure in the middle of a call
 *                          releases what the call had allocated.
 *   kz_debug_fail_device   the same countdown for whichever thread addresses the replica on (logical) `device` next: fails an allocation inside ONE device thread of
 *                          kz_render_multi (a caller's thread-local countdown does not reach those threads).
 *   kz_debug_grow_delay    the thread that maps a pass context's memory (kz_arena.cpp) sleeps `ms` milliseconds before every level (0 = off): "the context is
 *                          still growing while the first passes of a job run" - what happens behind the driver's wipe of recently released memory - on demand.
 *   kz_debug_trace         a timeline of the allocation, growth and pass-planning events of this process on stderr (0 = off).
 *   kz_debug_alias_devices the library presents n LOGICAL devices (kz_device_count() = n), logical d on physical device d % (devices really there): every
 *                          replica, pass context, pool and growth thread is keyed by the logical index, every HIP call goes to the physical device - so
 *                          kz_render_multi's one-host-thread-per-device driver runs with n threads on a box with ONE GPU (tests/test_gpu_multi.py). 0 = off.
 *                          Call it before the first kz_scene_upload; give every replica an explicit maxStateBytes (the aliases share one card). */
void kz_debug_fail_alloc(int nth);
void kz_debug_fail_device(int device, int nth);
void kz_debug_grow_delay(int ms);
void kz_debug_trace(int on);
void kz_debug_alias_devices(int n);

/* ---- the pass planner (nano-kazen_amd/csrc/kz_plan.cpp: pure host arithmetic, no GPU, no state) through the ABI: what kz_render would decide for a call.
 * tests/test_plan_cpu.py tabulates it for the BASELINE configs - the table is the documentation of the pass policy. */
typedef struct KzPlanQuery {
    int32_t pipeline;           /* 0 / 2 = wavefront, 1 = megakernel */
    uint32_t nPix;              /* pixels of the call's tile set */
    uint32_t sampleBegin, sampleEnd;
    uint64_t passItems;         /* KzRenderOpts.passItems */
    int32_t passesInFlight;     /* KzRenderOpts.passesInFlight */
    int32_t sppPerPass;         /* KzTuning.sppPerPass */
    uint64_t limitBytes;        /* bytes the pass contexts may hold (KzRenderOpts.maxStateBytes, or what kz_render derives from the device's free memory) */
    uint64_t bytesPerItem;      /* 0 = the wavefront pipeline's 176 */
    int32_t dealer;             /* 1 = the call carries a KzTileDealer */
    uint32_t takers, batchTiles, nTiles;
    const uint32_t *tilePixOffset;   /* dealer: nTiles + 1 positions in the pixel list */
    uint64_t heldItems;         /* items the replica's first pass context was last asked to hold (0: a fresh process) */
} KzPlanQuery;
typedef struct KzPlanAnswer {
    int32_t autoShape, nCtx, multi, grow;
    uint32_t S, pixPerPass, batchTiles, nPixSet, nPasses;
    uint64_t need, wantItems, minStart;
    double graceMs;
} KzPlanAnswer;
int kz_plan_passes(const KzPlanQuery *q, KzPlanAnswer *a);
/* The passes of pixels [pixBegin, pixEnd) of the list for that call when the k-th "what does the context hold now" answers avail[min(k, nAvail - 1)] items:
 * passes = 4 words per pass (first pixel, pixels, first sample, samples), *nPasses = how many there are (cap may be smaller). Fails with KZ_ERR_STATE if any pass
 * would exceed what its context holds - the planner's invariant. */
int kz_plan_schedule(const KzPlanQuery *q, const uint64_t *avail, uint32_t nAvail, uint32_t pixBegin, uint32_t pixEnd, uint32_t *passes, uint32_t cap, uint32_t *nPasses);

/* Known answers for the host code of kz_scene_create (no GPU): the area CDF of a light mesh - DiscretePDF::append + normalize, dpdf.h:35-37,77-89 - for n
 * pdf values (cdf: n + 1 floats; sumAndNormalization: 2 floats), and, for a sample count, { isPowerOf4, roundUpPow4, log4i of that, PMJ02BN's pixel tile }
 * (common.h:271-319, sampler.cpp:291). tests/golden/int_kats.json holds vectors minted from the reference's own text of both (oracle/kat_ref_dpdf.cpp). */
int kz_kat_dpdf(uint32_t n, const float *values, float *cdf, float *sumAndNormalization);
int kz_kat_pow4(int32_t spp, int32_t *out4);

/* How the library was built: bit 0 (KZ_BUILD_EXPERIMENTS) = it contains the kernels of kz_experiments.h. */
#define KZ_BUILD_EXPERIMENTS 1
int kz_build_flags(void);
/* hipMemGetInfo of `device` (what the default state budget of kz_render is derived from). */
int kz_device_mem_info(int device, uint64_t *freeBytes, uint64_t *totalBytes);
/* Self-check of the library's exact reciprocal / square root (hardware v_rcp_f32 / v_rsq_f32 + Newton steps, used by the triangle test,
 * the ray set-up and the BSDFs in place of the compiler's IEEE division / sqrt sequences): runs BOTH on every one of the 2^32 float
 * bit patterns on the device and counts the inputs whose results differ in any bit (two NaNs count as equal). Both counts must be 0. */
int kz_kat_exact_math(int device, uint64_t *rcpMismatches, uint64_t *sqrtMismatches, uint64_t *checked);
/* out[k] = random::permute(i[k], l[k], p[k]) (src/kazen/common.cpp:316-344) as the sampler kernels compute it: checked against vectors minted from
 * the reference's own text (oracle/kat_ref_permute.cpp -> tests/golden/int_kats.json). */
int kz_kat_permute(int device, uint32_t n, const uint32_t *i, const uint32_t *l, const uint32_t *p, uint32_t *out);
/* The Fresnel functions of the dielectric / rough BSDFs as the kernels compute them, against vectors minted from the reference's own text
 * (oracle/kat_ref_fresnel.cpp): form 0 = fresnel(cosThetaI, extIOR = a, intIOR = b) (common.cpp:447-475), form 1 = fresnelDielectric(cosThetaI, eta = a,
 * cosThetaT) (:492-518; b unused). out = n x (F, cosThetaT). */
int kz_kat_fresnel(int device, uint32_t n, int form, const float *cosThetaI, const float *a, const float *b, float *out);
/* The transcendental functions of the path as the kernels compute them (nano-kazen_amd/csrc/kz_crmath.h: each a fixed sequence of IEEE double operations
 * and one narrowing, standing in for the reference's libm calls - warp.cpp:41-129, bsdf.cpp:728-734, common.cpp:368-400, texture.cpp:66-80,
 * camera.cpp:191-223), on arrays: fn 0 sin(x), 1 cos(x), 2 exp(x), 3 log(x), 4 atan(x), 5 atan2(x, y), 6 acos(x), 7 tan(x), 8 pow(x, y), 9 hypot(x, y),
 * 10 x^3, 11 cos(x) through the cos-only entry. The oracle states the same sequences independently; the results must be equal bit for bit. y may
 * be NULL for the one-argument functions. */
int kz_kat_math(int device, int fn, uint32_t n, const float *x, const float *y, float *out);


#ifdef __cplusplus
}
#endif
#endif /* KAZEN_MI355X_DEV_H */
