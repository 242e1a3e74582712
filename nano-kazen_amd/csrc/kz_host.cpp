// kz_host.cpp — C-ABI shim, scene validation and flattening (host side of the MI355X core).
//
// kz_scene_create does what Scene::activate + Accel::build + Mesh::activate + PerspectiveCamera::activate
// + the ImageBlock / PMJ02BN constructors do in the reference (scene.cpp:29-52, accel.cpp:25-61,
// mesh.cpp:24-45, camera.cpp:35-68, block.cpp:9-31, sampler.cpp:275-315) and leaves behind flat tables
// that kz_scene_upload copies to HBM once. Nothing here renders: there is no CPU fallback for the
// kernels in kz_render.hip / kz_film.hip.
#include "kz_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

int kz_fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

namespace {

// include/kazen/common.h:271-320
bool isPowerOf4(int n) {
    if (n <= 0) return false;
    int x = (int)std::sqrt((double)n);
    if (x * x != n) return false;
    return !(n & (n - 1));
}
int log2i(uint32_t v) { return 31 - __builtin_clz(v); }
int log4i(uint32_t v) { return log2i(v) / 2; }
int roundUpPow4(int v) { return isPowerOf4(v) ? v : (1 << (2 * (1 + log4i((uint32_t)v)))); }
// PMJ02BN's pixel tile (sampler.cpp:291)
int pmjPixelTile(uint32_t spp) { return 1 << (log4i(KZ_PMJ02BN_SAMPLES) - log4i((uint32_t)roundUpPow4((int)spp))); }
// DiscretePDF::normalize (dpdf.h:77-89) on the table that starts at cdf[base] (cdf[base] = 0, then the running sums of append, dpdf.h:35-37):
// multiplies by the reciprocal of the sum and FORCES the last entry to 1; a table that sums to 0 is left as it is. Returns the normalisation.
float dpdfNormalize(std::vector<float> &cdf, size_t base, float *sumOut) {
    const float sum = cdf.back();
    if (sumOut) *sumOut = sum;
    if (!(sum > 0)) return 0.0f;
    const float normalization = 1.0f / sum;
    for (size_t i = base + 1; i < cdf.size(); ++i) cdf[i] *= normalization;
    cdf.back() = 1.0f;
    return normalization;
}

void mat4mul(const double *a, const double *b, double *c) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
        c[i * 4 + j] = s;
    }
}
bool mat4inv(const double *m, double *out) {
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

// src/kazen/rfilter.cpp:19-24, :50-63, :79-81, :95-97
float filterEval(const KzFilter &f, float x) {
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
        if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
        return 0.0f;
    }
    case KZ_FILTER_TENT: return std::max(0.0f, 1.0f - std::fabs(x));
    default: return 1.0f;
    }
}

const uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;

// ---- textures (texture.cpp): copy the rasters into one blob, flatten every KzTexture tree into a postfix program ----------
static int emitTexture(const KzSceneDesc *d, KzScene *sc, int32_t t, int depthBudget, int &stackNow, int &stackMax) {
    if (depthBudget <= 0) return kz_fail(KZ_ERR_UNSUPPORTED, "texture graph is cyclic or deeper than 32 levels");
    if (sc->texOps.size() > (1u << 16)) return kz_fail(KZ_ERR_UNSUPPORTED, "texture graphs flatten to more than 65536 operations");
    const KzTexture &k = d->textures[t];
    KzTexOp op; std::memset(&op, 0, sizeof op);
    auto pushConst = [&](float v) { KzTexOp c; std::memset(&c, 0, sizeof c); c.op = KZ_TOP_CONST; c.f0 = c.f1 = c.f2 = v; sc->texOps.push_back(c); stackNow++; stackMax = std::max(stackMax, stackNow); };
    auto child = [&](int i, float dflt) -> int {
        const int32_t c = k.child[i];
        if (c < 0) { pushConst(dflt); return KZ_OK; }
        if (c >= (int32_t)d->nTextures) return kz_fail(KZ_ERR_INVALID_ARG, "texture %d: child %d out of range", t, c);
        return emitTexture(d, sc, c, depthBudget - 1, stackNow, stackMax);
    };
    switch (k.type) {
    case KZ_TEX_CONSTANT:
        op.op = KZ_TOP_CONST; op.f0 = k.color[0]; op.f1 = k.color[1]; op.f2 = k.color[2];
        sc->texOps.push_back(op); stackNow++; stackMax = std::max(stackMax, stackNow);
        return KZ_OK;
    case KZ_TEX_IMAGE:
        if (k.image < 0 || k.image >= (int32_t)d->nImages) return kz_fail(KZ_ERR_INVALID_ARG, "texture %d: image index %d out of range", t, k.image);
        if (k.filter != KZ_TEXFILTER_BILINEAR && k.filter != KZ_TEXFILTER_BICUBIC) return kz_fail(KZ_ERR_INVALID_ARG, "texture %d: filter %d (KZ_TEXFILTER_BILINEAR or KZ_TEXFILTER_BICUBIC)", t, k.filter);
        op.op = KZ_TOP_IMAGE; op.a = (uint32_t)k.image; op.f0 = k.scale; op.b = (k.srgb ? 1u : 0u) | ((uint32_t)k.filter << 1);
        {   // the image row rides in the op (kz_internal.h KzTexOp): offset / 16, (width - 1) | (height - 1) << 16, channels, format
            const KzImageRow &row = sc->images[(size_t)k.image];
            const uint32_t off16 = (uint32_t)(row.offset >> 4), wh = (uint32_t)(row.width - 1) | ((uint32_t)(row.height - 1) << 16);
            if ((row.offset >> 4) >> 32) return kz_fail(KZ_ERR_UNSUPPORTED, "more than 64 GB of texels");
            std::memcpy(&op.f1, &off16, 4); std::memcpy(&op.f2, &wh, 4);
            op.b |= ((uint32_t)row.channels << 8) | ((uint32_t)row.format << 16);
        }
        sc->texOps.push_back(op); stackNow++; stackMax = std::max(stackMax, stackNow);
        return KZ_OK;
    case KZ_TEX_COLORRAMP: {
        if (k.child[0] < 0) { pushConst(0.f); return KZ_OK; }                      // texture.cpp:170: no nested texture -> 0 (not ramped)
        int rc = child(0, 0.f); if (rc != KZ_OK) return rc;
        op.op = KZ_TOP_RAMP; op.f0 = k.rampMin; op.f1 = k.rampMax;
        sc->texOps.push_back(op);
        return KZ_OK; }
    case KZ_TEX_BLEND: {
        if (k.blendMode < KZ_BLEND_MIX || k.blendMode > KZ_BLEND_NONE) return kz_fail(KZ_ERR_INVALID_ARG, "texture %d: blend mode %d", t, k.blendMode);
        int rc = child(0, 0.5f); if (rc != KZ_OK) return rc;                       // mask, input1, input2 defaults: texture.cpp:213-215
        rc = child(1, 0.f); if (rc != KZ_OK) return rc;
        rc = child(2, 1.f); if (rc != KZ_OK) return rc;
        op.op = KZ_TOP_BLEND; op.a = (uint32_t)k.blendMode;
        sc->texOps.push_back(op); stackNow -= 2;
        return KZ_OK; }
    default:
        return kz_fail(KZ_ERR_UNSUPPORTED, "texture %d has type %d (supported: constanttexture, imagetexture, colorramp, blend)", t, k.type);
    }
}
static int flattenTextures(const KzSceneDesc *d, KzScene *sc) {
    for (uint32_t i = 0; i < d->nImages; ++i) {
        const KzImage &im = d->images[i];
        if (!im.pixels || im.width <= 0 || im.height <= 0 || im.channels <= 0 || im.channels > 16 || im.width > 65536 || im.height > 65536 ||
            (im.format != KZ_PIXEL_U8 && im.format != KZ_PIXEL_F32))
            return kz_fail(KZ_ERR_INVALID_ARG, "image %u: %dx%dx%d format %d", i, im.width, im.height, im.channels, im.format);
        const size_t bytes = (size_t)im.width * im.height * im.channels * (im.format == KZ_PIXEL_F32 ? 4 : 1);
        KzImageRow row; row.offset = (sc->texels.size() + 15) & ~(size_t)15; row.width = im.width; row.height = im.height; row.channels = im.channels; row.format = im.format;
        sc->texels.resize(row.offset + bytes);
        std::memcpy(sc->texels.data() + row.offset, im.pixels, bytes);
        sc->images.push_back(row);
        if ((im.width & (im.width - 1)) || (im.height & (im.height - 1))) sc->texPow2 = 0;
    }
    for (uint32_t t = 0; t < d->nTextures; ++t) {
        KzTexProg pr; pr.start = (uint32_t)sc->texOps.size();
        int now = 0, mx = 0;
        int rc = emitTexture(d, sc, (int32_t)t, 32, now, mx);
        if (rc != KZ_OK) return rc;
        if (mx > KZ_TEX_MAX_DEPTH) return kz_fail(KZ_ERR_UNSUPPORTED, "texture %u needs an operand stack of %d (limit %d)", t, mx, KZ_TEX_MAX_DEPTH);
        pr.count = ((uint32_t)sc->texOps.size() - pr.start) | ((uint32_t)mx << 16);       // (<= 65536 ops in all: checked in emitTexture)
        sc->texProgs.push_back(pr);
    }
    return KZ_OK;
}

} // namespace

extern "C" {

const char *kz_last_error(void) { return g_err; }
int kz_abi_version(void) { return KZ_ABI_VERSION; }

// (kazen_mi355x_dev.h) the host code of kz_scene_create behind a light's area CDF and behind PMJ02BN's pixel tile, for the vectors minted from the
// reference's own dpdf.h / common.h (oracle/kat_ref_dpdf.cpp)
int kz_kat_dpdf(uint32_t n, const float *values, float *cdf, float *sumAndNormalization) {
    if ((n && !values) || !cdf || !sumAndNormalization) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    std::vector<float> t(1, 0.0f);
    for (uint32_t i = 0; i < n; ++i) t.push_back(t.back() + values[i]);
    sumAndNormalization[1] = dpdfNormalize(t, 0, &sumAndNormalization[0]);
    std::memcpy(cdf, t.data(), t.size() * sizeof(float));
    return KZ_OK;
}
int kz_kat_pow4(int32_t spp, int32_t *out4) {
    if (spp < 1 || !out4) return kz_fail(KZ_ERR_INVALID_ARG, "kz_kat_pow4(%d)", spp);
    out4[0] = isPowerOf4(spp) ? 1 : 0; out4[1] = roundUpPow4(spp); out4[2] = log4i((uint32_t)out4[1]); out4[3] = pmjPixelTile((uint32_t)spp);
    return KZ_OK;
}
int kz_build_flags(void) {
#ifdef KZ_EXPERIMENTS
    return KZ_BUILD_EXPERIMENTS;
#else
    return 0;
#endif
}

int kz_scene_create(const KzSceneDesc *d, KzScene **out) {
    if (!d || !out) return kz_fail(KZ_ERR_INVALID_ARG, "kz_scene_create: null argument");
    *out = nullptr;
    if (d->abiVersion != KZ_ABI_VERSION) return kz_fail(KZ_ERR_INVALID_ARG, "ABI version %u, library is %u", d->abiVersion, KZ_ABI_VERSION);
    // plugin types outside the hot path are an error, never a silent fallback
    if (d->camera.type != KZ_CAMERA_PERSPECTIVE && d->camera.type != KZ_CAMERA_THINLENS) return kz_fail(KZ_ERR_UNSUPPORTED, "camera type %d is not supported (\"perspective\", \"thinlens\")", d->camera.type);
    if (d->integrator.type != KZ_INTEGRATOR_PATH_MIS) return kz_fail(KZ_ERR_UNSUPPORTED, "integrator type %d is not on the hot path (only \"path_mis\")", d->integrator.type);
    if (d->sampler.type < KZ_SAMPLER_INDEPENDENT || d->sampler.type > KZ_SAMPLER_CORRELATED)
        return kz_fail(KZ_ERR_UNSUPPORTED, "sampler type %d is not supported (\"independent\", \"pmj02bn\", \"stratified\", \"correlated\")", d->sampler.type);
    if (d->sampler.type == KZ_SAMPLER_STRATIFIED && (d->sampler.resolution < 1 || d->sampler.resolution > 256)) return kz_fail(KZ_ERR_INVALID_ARG, "stratified resolution %d", d->sampler.resolution);
    if (d->sampler.sampleCount > 65536) return kz_fail(KZ_ERR_UNSUPPORTED, "sampleCount %u (limit 65536)", d->sampler.sampleCount);
    if (d->camera.rfilter.type < KZ_FILTER_GAUSSIAN || d->camera.rfilter.type > KZ_FILTER_BOX) return kz_fail(KZ_ERR_UNSUPPORTED, "rfilter type %d", d->camera.rfilter.type);
    if (d->camera.width <= 0 || d->camera.height <= 0 || d->camera.width > 65535 || d->camera.height > 65535) return kz_fail(KZ_ERR_INVALID_ARG, "image size %dx%d", d->camera.width, d->camera.height);
    if (d->sampler.sampleCount == 0) return kz_fail(KZ_ERR_INVALID_ARG, "sampleCount is 0");
    if (!(d->camera.rfilter.radius > 0.f) || d->camera.rfilter.radius > 4.0f) return kz_fail(KZ_ERR_UNSUPPORTED, "filter radius %g (supported: (0, 4])", d->camera.rfilter.radius);
    if ((d->nMeshes && !d->meshes) || (d->nBsdfs && !d->bsdfs) || (d->nLights && !d->lights)) return kz_fail(KZ_ERR_INVALID_ARG, "null table with non-zero count");
    if ((d->nTextures && !d->textures) || (d->nImages && !d->images)) return kz_fail(KZ_ERR_INVALID_ARG, "null texture/image table with non-zero count");
    for (uint32_t i = 0; i < d->nBsdfs; ++i) {
        const KzBSDF &b = d->bsdfs[i];
        if (b.type < KZ_BSDF_DIFFUSE || b.type > KZ_BSDF_NORMALMAP)
            return kz_fail(KZ_ERR_UNSUPPORTED, "bsdf %u has type %d (supported: diffuse, kazenstandard, mirror, dielectric, ggx, roughconductor, roughplastic, roughdielectric, normalmap)", i, b.type);
        const int32_t ids[4] = {b.albedoTex, b.roughnessTex, b.metallicTex, b.normalTex};
        for (int32_t id : ids) if (id < 0 || id > (int32_t)d->nTextures) return kz_fail(KZ_ERR_INVALID_ARG, "bsdf %u: texture id %d out of range (0 = constant, 1..%u)", i, id, d->nTextures);
        // which rows read a Texture child: diffuse/lambertian + ggx "albedo" (bsdf.cpp:259-262, :672-675), kiss (bsdf.cpp:1375-1390)
        const bool takesAlbedo = b.type == KZ_BSDF_DIFFUSE || b.type == KZ_BSDF_GGX || b.type == KZ_BSDF_KAZENSTANDARD;
        if ((b.albedoTex && !takesAlbedo) || ((b.roughnessTex || b.metallicTex) && b.type != KZ_BSDF_KAZENSTANDARD) || (b.normalTex && b.type != KZ_BSDF_NORMALMAP))
            return kz_fail(KZ_ERR_INVALID_ARG, "bsdf %u (type %d): a texture id is set on a parameter this model does not read through a texture", i, b.type);
        const bool rough = b.type == KZ_BSDF_ROUGHCONDUCTOR || b.type == KZ_BSDF_ROUGHPLASTIC || b.type == KZ_BSDF_ROUGHDIELECTRIC;
        if ((b.alphaResolved != 0 && b.alphaResolved != 1) || (b.alphaResolved && !rough))
            return kz_fail(KZ_ERR_INVALID_ARG, "bsdf %u (type %d): alphaResolved = %d (0 or 1, and only roughconductor / roughplastic / roughdielectric rows have a resolved alpha)", i, b.type, b.alphaResolved);
        if (b.type == KZ_BSDF_NORMALMAP) {                  // bsdf.cpp:391-404: one texture child + one nested BSDF
            if (b.normalTex == 0) return kz_fail(KZ_ERR_INVALID_ARG, "bsdf %u: normalmap without a normal texture", i);
            if (b.nested < 0 || b.nested >= (int32_t)d->nBsdfs || d->bsdfs[b.nested].type == KZ_BSDF_NORMALMAP)
                return kz_fail(KZ_ERR_INVALID_ARG, "bsdf %u: normalmap needs a nested BSDF row that is not itself a normalmap (got %d)", i, b.nested);
        }
    }

    KzScene *sc = new KzScene();
    std::memset(&sc->prm, 0, sizeof sc->prm);
    sc->bsdfs.assign(d->bsdfs, d->bsdfs + d->nBsdfs);
    // the rough BSDFs' constructors keep m_alpha = max(MIN_ALPHA, sqr(roughness)) (bsdf.cpp:696-700, :818-822, :956-959): formed here once, in float, as they
    // form it (rows that arrive with alphaResolved = 1 carry it already); the kernels read the row's alpha as it is
    for (KzBSDF &b : sc->bsdfs)
        if ((b.type == KZ_BSDF_ROUGHCONDUCTOR || b.type == KZ_BSDF_ROUGHPLASTIC || b.type == KZ_BSDF_ROUGHDIELECTRIC) && !b.alphaResolved) {
            const float MIN_ALPHA = 0.001f, a2 = b.alpha * b.alpha;
            b.alpha = std::max(MIN_ALPHA, a2); b.alphaResolved = 1;
        }
    { int trc = flattenTextures(d, sc); if (trc != KZ_OK) { delete sc; return trc; } }
    int defaultBsdf = -1;

    // ---- geometry: (mesh, face) order = Embree geomID / primID order (accel.cpp:40-55)
    std::vector<KzBuildTri> bt;
    size_t totalF = 0;
    for (uint32_t m = 0; m < d->nMeshes; ++m) totalF += d->meshes[m].nF;
    if (totalF >= (1ull << 28)) { delete sc; return kz_fail(KZ_ERR_UNSUPPORTED, "%zu triangles (limit 2^28)", totalF); }
    bt.reserve(totalF); sc->shade.reserve(totalF);
    uint32_t gid = 0;
    for (uint32_t m = 0; m < d->nMeshes; ++m) {
        const KzMesh &km = d->meshes[m];
        if (!km.V || !km.F) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "mesh %u has no V/F", m); }
        if (km.bsdf >= (int)d->nBsdfs || km.light >= (int)d->nLights || km.bsdf < -1 || km.light < -1) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "mesh %u: bsdf/light index out of range", m); }
        KzMeshRow row; std::memset(&row, 0, sizeof row);
        row.bsdf = km.bsdf; row.light = -1;
        if (km.bsdf < 0) {      // Mesh::activate instantiates a default Diffuse (mesh.cpp:25-28, albedo 0.5 bsdf.cpp:23)
            if (defaultBsdf < 0) {
                KzBSDF b; std::memset(&b, 0, sizeof b);
                b.type = KZ_BSDF_DIFFUSE; b.albedo[0] = b.albedo[1] = b.albedo[2] = 0.5f;
                defaultBsdf = (int)sc->bsdfs.size(); sc->bsdfs.push_back(b);
            }
            row.bsdf = defaultBsdf;
        }
        row.flags = (km.N ? 1u : 0u) | (km.UV ? 2u : 0u);
        row.triOffset = gid; row.nF = km.nF;
        const int32_t lightRow = km.light >= 0 ? (int32_t)sc->lightRows.size() : -1;      // the row pushed below
        for (uint32_t f = 0; f < km.nF; ++f, ++gid) {
            uint32_t idx[3] = {km.F[3 * f], km.F[3 * f + 1], km.F[3 * f + 2]};
            KzBuildTri t; t.mesh = m; t.prim = f; t.gid = gid;
            KzTriShade s; std::memset(&s, 0, sizeof s);
            s.mesh = m; s.prim = f; s.bsdf = (uint32_t)row.bsdf; s.lightFlags = ((uint32_t)(lightRow + 1) << 2) | row.flags;
            for (int v = 0; v < 3; ++v) {
                if (idx[v] >= km.nV) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "mesh %u face %u: vertex index %u >= %u", m, f, idx[v], km.nV); }
                for (int a = 0; a < 3; ++a) {
                    t.v[v][a] = km.V[3 * (size_t)idx[v] + a];
                    s.p[3 * v + a] = t.v[v][a];
                    if (km.N) s.n[3 * v + a] = km.N[3 * (size_t)idx[v] + a];
                }
                if (km.UV) { s.uv[2 * v] = km.UV[2 * (size_t)idx[v]]; s.uv[2 * v + 1] = km.UV[2 * (size_t)idx[v] + 1]; }
            }
            bt.push_back(t); sc->shade.push_back(s);
        }
        // ---- light rows + area CDF (scene.cpp:42-46, mesh.cpp:24-45, dpdf.h:35-37,77-89)
        if (km.light >= 0) {
            const KzLight &kl = d->lights[km.light];
            KzLightRow lr; std::memset(&lr, 0, sizeof lr);
            for (int a = 0; a < 3; ++a) lr.radiance[a] = kl.intensity * kl.color[a];
            lr.primaryVisibility = kl.primaryVisibility ? 1 : 0;
            while (sc->cdf.size() & 3u) sc->cdf.push_back(2.0f);            // every light's table starts on a 16-B boundary (cdfSample reads short tables as float4s)
            lr.mesh = m; lr.triOffset = row.triOffset; lr.nF = km.nF; lr.cdfOffset = (uint32_t)sc->cdf.size(); lr.hasN = km.N ? 1u : 0u;
            size_t base = sc->cdf.size();
            sc->cdf.push_back(0.0f);
            for (uint32_t f = 0; f < km.nF; ++f) {
                const KzTriShade &s = sc->shade[row.triOffset + f];
                float e1[3], e2[3];
                for (int a = 0; a < 3; ++a) { e1[a] = s.p[3 + a] - s.p[a]; e2[a] = s.p[6 + a] - s.p[a]; }
                float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                float area = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);          // mesh.cpp:47-53
                sc->cdf.push_back(sc->cdf.back() + area);
            }
            lr.normalization = dpdfNormalize(sc->cdf, base, nullptr);
            row.light = (int32_t)sc->lightRows.size();
            sc->lightRows.push_back(lr);
        }
        sc->meshRows.push_back(row);
    }
    sc->cdf.insert(sc->cdf.end(), 8, 2.0f);          // padding: the device reads eight entries of a short table at once (cdfSample)
    std::string berr;
    uint32_t rootRef = 0xFFFFFFFFu;
    int rc = kz_build_bvh(bt, sc->nodes, sc->tris, rootRef, sc->bvh, berr);
    if (rc != KZ_OK) { delete sc; return kz_fail(rc, "BVH build: %s", berr.c_str()); }

    KzParams &p = sc->prm;
    p.rootRef = rootRef;
    { int sb = 1; uint32_t r4 = rootRef;
      rc = kz_collapse_bvh4(sc->nodes, rootRef, sc->nodes4, r4, sb);
      if (rc != KZ_OK) { delete sc; return kz_fail(rc, "BVH4 collapse failed"); }
      // (the kernels address a packet as table base + a 32-bit byte offset, kz_devfn.h kzNode4Ptr)
      if (sc->nodes4.size() > (size_t(1) << 26)) { delete sc; return kz_fail(KZ_ERR_UNSUPPORTED, "scene too large: %zu BVH4 packets (limit 2^26 = 4 GB of packets)", sc->nodes4.size()); }
      p.rootRef4 = r4; p.stackBound4 = sb; }
    p.stackDepth = (int32_t)std::max<uint32_t>(2u, sc->bvh.maxDepth + 1);
    // what the scene's BSDF rows need beyond constant diffuse / kazenstandard (kz_devfn.h KZ_X_*): 1 = other models, 2 = texture-backed parameters, 4 = normal maps
    p.bsdfExt = 0;
    for (const KzBSDF &b : sc->bsdfs) {
        if (b.type == KZ_BSDF_NORMALMAP) p.bsdfExt |= 4 | 2;
        else if (b.type > KZ_BSDF_KAZENSTANDARD) p.bsdfExt |= 1;
        if (b.albedoTex || b.roughnessTex || b.metallicTex) p.bsdfExt |= 2;
    }
    // invisible-light triangles for the exact any-hit shadow test
    p.shadowFast = 1; p.nIlTris = 0; p.anyInvisibleLight = 0;
    uint32_t ilGidLo = 0xFFFFFFFFu, ilGidHi = 0;
    for (int a = 0; a < 3; ++a) { p.ilLo[a] = INFINITY; p.ilHi[a] = -INFINITY; }
    for (const KzLightRow &lr : sc->lightRows) {
        if (lr.primaryVisibility) continue;
        p.anyInvisibleLight = 1;
        if (lr.nF) { ilGidLo = std::min(ilGidLo, lr.triOffset); ilGidHi = std::max(ilGidHi, lr.triOffset + lr.nF - 1); }
        for (uint32_t f = 0; f < lr.nF; ++f) {
            const KzTriShade &s = sc->shade[lr.triOffset + f];
            KzTri t; std::memset(&t, 0, sizeof t);
            for (int a = 0; a < 3; ++a) {
                t.p0[a] = s.p[a]; t.e1[a] = s.p[3 + a] - s.p[a]; t.e2[a] = s.p[6 + a] - s.p[a];
                for (int v = 0; v < 3; ++v) { p.ilLo[a] = std::min(p.ilLo[a], s.p[3 * v + a]); p.ilHi[a] = std::max(p.ilHi[a], s.p[3 * v + a]); }
            }
            t.mesh = lr.mesh; t.prim = f; t.gid = lr.triOffset + f;
            sc->ilTris.push_back(t);
        }
    }
    if (sc->ilTris.size() > 64) { p.shadowFast = 0; sc->ilTris.clear(); }      // big emissive meshes: literal closest-hit loop
    p.nIlTris = (uint32_t)sc->ilTris.size();
    p.ilGidLo = ilGidLo <= ilGidHi ? ilGidLo : 1u; p.ilGidSpan = ilGidLo <= ilGidHi ? ilGidHi - ilGidLo : 0u;
    for (int a = 0; a < 3; ++a) {        // same padding as BVH boxes
        float m = std::max(std::fabs(p.ilLo[a]), std::fabs(p.ilHi[a]));
        if (std::isfinite(m)) { float e = m * 4e-7f + 1e-30f; p.ilLo[a] -= e; p.ilHi[a] += e; }
    }
    // ---- camera (camera.cpp:35-68). Eigen is not available: the 4x4 product and inverse are formed in
    // double and narrowed once, unless the caller hands over Eigen's own m_sampleToCamera.
    const KzCamera &c = d->camera;
    p.width = c.width; p.height = c.height;
    p.invW = 1.0f / (float)c.width; p.invH = 1.0f / (float)c.height;
    p.nearClip = c.nearClip; p.farClip = c.farClip;
    std::memcpy(p.c2w, c.toWorld, sizeof p.c2w);
    if (c.sampleToCamera) std::memcpy(p.s2c, c.sampleToCamera, sizeof p.s2c);
    else {
        float aspect = c.width / (float)c.height;
        float recip = 1.0f / (c.farClip - c.nearClip);
        float cot = 1.0f / std::tan((c.fov / 2.0f) * (3.14159265358979323846f / 180.0f));      // common.h:222 degToRad; M_PI is the float literal of common.h:33
        double P[16] = {cot, 0, 0, 0, 0, cot, 0, 0, 0, 0, (double)(c.farClip * recip), (double)(-c.nearClip * c.farClip * recip), 0, 0, 1, 0};
        double T[16] = {1, 0, 0, -1, 0, 1, 0, (double)(-1.0f / aspect), 0, 0, 1, 0, 0, 0, 0, 1};
        double D[16] = {-0.5, 0, 0, 0, 0, (double)(-0.5f * aspect), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        double TP[16], M[16], Mi[16];
        mat4mul(T, P, TP); mat4mul(D, TP, M);
        if (!mat4inv(M, Mi)) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "singular projection (fov %g, clip %g..%g)", c.fov, c.nearClip, c.farClip); }
        for (int i = 0; i < 16; ++i) p.s2c[i] = (float)Mi[i];
    }
    {   // pixel beams (kz_wf_beam): only for a pinhole camera whose homogeneous divide does not depend on the sample position and an affine c2w
        const float *m = p.s2c, *w = p.c2w;
        p.beamOk = 0;
        const float tiny = 1e-9f * std::fabs(m[15]);          // (a numerically formed inverse leaves 1e-17s where the zeros are: far inside the beams' margins)
        if (c.type != KZ_CAMERA_THINLENS && std::fabs(m[12]) <= tiny && std::fabs(m[13]) <= tiny && m[15] != 0.f && w[12] == 0.f && w[13] == 0.f && w[14] == 0.f && w[15] != 0.f) {
            const double rw = m[15];
            const double A[3] = {m[3] / rw, m[7] / rw, m[11] / rw}, U[3] = {m[0] * (double)p.invW / rw, m[4] * (double)p.invW / rw, m[8] * (double)p.invW / rw},
                         V[3] = {m[1] * (double)p.invH / rw, m[5] * (double)p.invH / rw, m[9] * (double)p.invH / rw};
            bool fin = true;
            for (int r = 0; r < 3; ++r) {
                p.beamA[r] = (float)(w[4 * r] * A[0] + w[4 * r + 1] * A[1] + w[4 * r + 2] * A[2]);
                p.beamU[r] = (float)(w[4 * r] * U[0] + w[4 * r + 1] * U[1] + w[4 * r + 2] * U[2]);
                p.beamV[r] = (float)(w[4 * r] * V[0] + w[4 * r + 1] * V[1] + w[4 * r + 2] * V[2]);
                p.beamO[r] = w[4 * r + 3] / w[15];
                fin = fin && std::isfinite(p.beamA[r]) && std::isfinite(p.beamU[r]) && std::isfinite(p.beamV[r]) && std::isfinite(p.beamO[r]);
            }
            p.beamOk = fin ? 1 : 0;
        }
    }
    // ---- film filter table (block.cpp:13-21)
    const KzFilter &rf = c.rfilter;
    p.filterRadius = rf.radius;
    p.border = (int)std::ceil(rf.radius - 0.5f);
    for (int i = 0; i < KZ_FILTER_RESOLUTION; ++i) sc->filter[i] = filterEval(rf, (rf.radius * i) / KZ_FILTER_RESOLUTION);
    sc->filter[KZ_FILTER_RESOLUTION] = 0.0f;
    p.lookupFactor = KZ_FILTER_RESOLUTION / rf.radius;
    // source pixels px that can reach film pixel fx satisfy px-(fx-border) in (-r-0.5, r+0.5]
    p.tapLo = (int)std::floor(-rf.radius - 0.5f) + 1;
    p.tapHi = (int)std::floor(rf.radius + 0.5f);
    if (p.tapHi - p.tapLo + 1 > KZ_MAX_FILTER_TAPS) { delete sc; return kz_fail(KZ_ERR_UNSUPPORTED, "filter radius %g needs too many taps", rf.radius); }
    // ---- integrator (integrator.cpp:187-193)
    p.maxDepth = std::min(512, d->integrator.maxDepth);
    p.traceBias = d->integrator.traceBias;
    p.regularization = d->integrator.regularization ? 1 : 0;
    p.accumulatedRoughness = d->integrator.accumulatedRoughness;
    // ---- lights / background (scene.h:45-56, texture.cpp:121-126)
    p.nLights = (uint32_t)sc->lightRows.size();
    p.lightPickPdf = p.nLights ? 1.f / (float)p.nLights : 0.f;
    p.lightPickScale = (p.nLights && (p.nLights & (p.nLights - 1)) == 0) ? (float)p.nLights : 0.f;
    p.bgPresent = d->background.present ? 1 : 0;
    p.bgImage = -1; p.bgIntensity = d->background.intensity;
    {
        // the nested texture by DIRECTION (texture.cpp:20-22, 66-80, texture.h:13): constant -> colour, image -> environment lookup,
        // colorramp / blend -> 0
        float col[3] = {d->background.color[0], d->background.color[1], d->background.color[2]};
        if (d->background.present && d->background.texture != 0) {
            const int32_t t = d->background.texture - 1;
            if (t < 0 || (uint32_t)t >= d->nTextures) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "background texture id %d out of range", d->background.texture); }
            const KzTexture &k = d->textures[t];
            if (k.type == KZ_TEX_CONSTANT) { col[0] = k.color[0]; col[1] = k.color[1]; col[2] = k.color[2]; }
            else if (k.type == KZ_TEX_IMAGE) {
                if (k.image < 0 || (uint32_t)k.image >= d->nImages) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "background image index %d out of range", k.image); }
                if (k.filter != KZ_TEXFILTER_BILINEAR && k.filter != KZ_TEXFILTER_BICUBIC) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "background texture: filter %d", k.filter); }
                p.bgImage = k.image; p.bgFilter = k.filter; col[0] = col[1] = col[2] = 0.f;
            } else col[0] = col[1] = col[2] = 0.f;
        }
        for (int a = 0; a < 3; ++a) p.bgRadiance[a] = d->background.present ? d->background.intensity * col[a] : 0.f;
    }
    // ---- sampler
    p.samplerType = d->sampler.type;
    p.seed = d->sampler.seed;
    p.sampleCount = d->sampler.sampleCount;
    p.resX = p.resY = 1;
    p.cameraType = d->camera.type; p.apertureRadius = d->camera.apertureRadius; p.focusDistance = d->camera.focusDistance;
    if (d->sampler.type == KZ_SAMPLER_STRATIFIED) {                       // Stratified ctor, sampler.cpp:84-92
        int res = d->sampler.resolution;
        while ((uint32_t)(res * res) < p.sampleCount) res++;
        p.resX = p.resY = res; p.sampleCount = (uint32_t)(res * res);
    }
    if (d->sampler.type == KZ_SAMPLER_CORRELATED) {                       // Correlated ctor, sampler.cpp:179-187
        int r1 = (int)std::sqrt((double)p.sampleCount);
        int r0 = (int)((p.sampleCount + r1 - 1) / r1);
        p.resX = r0; p.resY = r1; p.sampleCount = (uint32_t)(r0 * r1);
    }
    if (d->sampler.type == KZ_SAMPLER_PMJ02BN) {
        if (!d->sampler.pmj02bnSamples || !d->sampler.blueNoise) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "pmj02bn sampler without its tables"); }
        if (p.sampleCount > KZ_PMJ02BN_SAMPLES) p.sampleCount = KZ_PMJ02BN_SAMPLES;               // sampler.cpp:284-287
        // the device reads both tables as the floats the reference's accessors return: the conversions (pmj02table.h:28-29 in double then
        // narrowed; bluenoise.h:22 an fp32 division) are done here, once, with the same IEEE operations
        const size_t nPmj = (size_t)KZ_PMJ02BN_SETS * KZ_PMJ02BN_SAMPLES * 2, nBn = (size_t)KZ_BLUENOISE_TEXTURES * KZ_BLUENOISE_RES * KZ_BLUENOISE_RES;
        sc->pmj.resize(nPmj); sc->bn.resize(nBn);
        for (size_t i = 0; i < nPmj; ++i) sc->pmj[i] = (float)((double)d->sampler.pmj02bnSamples[i] * 0x1p-32);
        for (size_t i = 0; i < nBn; ++i) sc->bn[i] = (float)d->sampler.blueNoise[i] / 65535.f;
        // PMJ02BN constructor: sort set 0 into a tile x tile x spp pixel table (sampler.cpp:291-309)
        uint32_t spp = p.sampleCount;
        int tile = pmjPixelTile(spp);
        p.pixelTileSize = tile;
        sc->pixelSamples.assign((size_t)tile * tile * spp * 2, 0.f);
        std::vector<uint32_t> nStored((size_t)tile * tile, 0);
        for (int i = 0; i < KZ_PMJ02BN_SAMPLES; ++i) {
            float x = sc->pmj[2 * (size_t)i], y = sc->pmj[2 * (size_t)i + 1];
            x *= tile; y *= tile;
            int ix = (int)x, iy = (int)y;
            if (ix >= tile || iy >= tile) { delete sc; return kz_fail(KZ_ERR_INVALID_ARG, "pmj02bn table entry %d rounds to 1.0f (the reference would index out of range)", i); }
            size_t po = (size_t)ix + (size_t)iy * tile;
            if (nStored[po] == spp) continue;
            size_t so = po * spp + nStored[po];
            sc->pixelSamples[2 * so] = x - std::floor(x);
            sc->pixelSamples[2 * so + 1] = y - std::floor(y);
            ++nStored[po];
        }
    } else {
        // pcg32::advance(sampleIndex*65536) as an affine map state' = mult*state + inc*plus (pcg32.h:145-166)
        sc->jump.resize(p.sampleCount);
        for (uint32_t s = 0; s < p.sampleCount; ++s) {
            uint64_t cur_mult = PCG32_MULT, cur_plus = 1u, acc_mult = 1u, acc_plus = 0u;
            uint64_t delta = (uint64_t)s * 65536ull;
            while (delta > 0) {
                if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
                cur_plus = (cur_mult + 1) * cur_plus;
                cur_mult *= cur_mult;
                delta /= 2;
            }
            sc->jump[s].mult = acc_mult; sc->jump[s].plus = acc_plus;
        }
    }
    p.sppPow2 = (p.sampleCount & (p.sampleCount - 1)) == 0 ? 1 : 0;
    p.invSpp = 1.0f / (float)p.sampleCount;
    kz_device_init(sc);
    *out = sc;
    return KZ_OK;
}

void kz_scene_destroy(KzScene *scene) {
    if (!scene) return;
    kz_device_release(scene);
    delete scene;
}

int kz_scene_bvh_info(const KzScene *scene, KzBvhInfo *out) {
    if (!scene || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    *out = scene->bvh;
    return KZ_OK;
}

int kz_scene_sample_count(const KzScene *scene, uint32_t *out) {
    if (!scene || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    *out = scene->prm.sampleCount;
    return KZ_OK;
}

int kz_film_dims(const KzScene *scene, int32_t *width, int32_t *height, int32_t *border) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    if (width) *width = scene->prm.width;
    if (height) *height = scene->prm.height;
    if (border) *border = scene->prm.border;
    return KZ_OK;
}

// ImageBlock::toBitmap + Color4f::divideByFilterWeight (block.cpp:39-45, color.h:94-99). Host utility on a
// downloaded film; the per-sample work all happened on the GPU.
int kz_film_to_rgb(const float *film, int32_t width, int32_t height, int32_t border, float *rgb) {
    if (!film || !rgb || width <= 0 || height <= 0 || border < 0) return kz_fail(KZ_ERR_INVALID_ARG, "kz_film_to_rgb: bad argument");
    size_t cols = (size_t)width + 2 * (size_t)border;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float *px = film + ((size_t)(y + border) * cols + (size_t)(x + border)) * 4;
            float *o = rgb + ((size_t)y * width + x) * 3;
            if (px[3] != 0) { o[0] = px[0] / px[3]; o[1] = px[1] / px[3]; o[2] = px[2] / px[3]; }
            else { o[0] = o[1] = o[2] = 0.f; }
        }
    return KZ_OK;
}

} // extern "C"
