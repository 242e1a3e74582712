// Static instruction cost of the pieces of one shaded hit (the straight-line code the shade kernel's pass B runs per survivor on the C4 path:
// pmj02bn sampler, kazenstandard rows). Not run: compiled to ISA by scripts/micro/shade_cost.sh, which counts the VALU instructions of each
// kernel below and subtracts the empty kernel. Divisions are the 10-instruction IEEE sequences (v_div_scale .. v_div_fixup).
#include <hip/hip_runtime.h>
#include "../../nano-kazen_amd/csrc/kz_internal.h"
#include "../../nano-kazen_amd/csrc/kz_devfn.h"

#define PIECE(name, ...) extern "C" __global__ void name(KzParams P, KzDevTables T, const float4 *__restrict__ in, float4 *__restrict__ out) { \
    P.samplerType = KZ_SAMPLER_PMJ02BN; const uint32_t i = blockIdx.x * 64 + threadIdx.x; const float4 a = in[i], b = in[i + 4096], c = in[i + 8192]; float4 r = a; (void)b; (void)c; __VA_ARGS__; out[i] = r; }

static __device__ __forceinline__ Sampler mkSampler(const KzParams &P, float4 a, bool uniformDim) {
    Sampler s; s.type = P.samplerType; s.px = (int)a.x; s.py = (int)a.y; s.idx = (uint32_t)a.z; s.state = 0; s.inc = 0;
    s.dim = uniformDim ? (uint32_t)__builtin_amdgcn_readfirstlane((int)a.w) : (uint32_t)a.w;
    s.hp = hashPixelBlock(s.px, s.py); return s;
}
PIECE(piece_empty, {})
PIECE(piece_sampler_setup, { Sampler s = mkSampler(P, a, true); r.x = (float)(s.hp >> 32); r.y = (float)(uint32_t)s.hp; })
PIECE(piece_next1d_lane_dim, { Sampler s = mkSampler(P, a, false); r.x = s.next1D(P, T); })
PIECE(piece_next1d, { Sampler s = mkSampler(P, a, true); r.x = s.next1D(P, T); })
PIECE(piece_next1d_x5, { Sampler s = mkSampler(P, a, true); r.x = s.next1D(P, T); r.y = s.next1D(P, T); r.z = s.next1D(P, T); r.w = s.next1D(P, T); r.x += s.next1D(P, T); })
PIECE(piece_next2d, { Sampler s = mkSampler(P, a, true); s.next2D(P, T, r.x, r.y); })
PIECE(piece_post_intersect, { RawHit rh; rh.t = a.x; rh.u = a.y; rh.v = a.z; rh.tri = __float_as_uint(a.w); rh.gid = 0; Its its; its.p = mk(b.x, b.y, b.z); postIntersect<false>(T, rh, its);
                              r = make_float4(its.sh.n.x + its.sh.s.y + its.sh.t.z, its.p.x + its.geoN.y, its.uvx + its.uvy, (float)its.mesh); })
PIECE(piece_light_sample, { const KzLightRow lrow = T.lights[(uint32_t)a.w & 7u]; float u[3] = {b.x, b.y, b.z}; int k = 0;
                            const LightSample ls = lightSample(T, lrow, mk(a.x, a.y, a.z), [&]() { return u[k++]; }); r = make_float4(ls.wi.x + ls.Ls.x, ls.wi.y + ls.Ls.y, ls.wi.z + ls.Ls.z, ls.pdf + ls.dist); })
static __device__ __forceinline__ Its flatIts() { Its its; its.p = mk(0.f); its.t = 0.f; its.uvx = its.uvy = 0.f; its.sh.s = mk(1.f, 0.f, 0.f); its.sh.t = mk(0.f, 1.f, 0.f); its.sh.n = mk(0.f, 0.f, 1.f);
    its.geoN = its.sh.n; its.dpdu = its.sh.s; its.mesh = 0; its.prim = 0; its.bu = its.bv = 0.f; return its; }
PIECE(piece_kiss_eval, { KzBSDF m = T.bsdfs[(uint32_t)a.w & 7u]; NMap nm; Its its = flatIts(); surfaceSetup<false>(T, its, m, nm);
                         V3 e = surfEval<false>(m, nm, its, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w); r = make_float4(e.x, e.y, e.z, 0.f); })
PIECE(piece_kiss_pdf, { KzBSDF m = T.bsdfs[(uint32_t)a.w & 7u]; NMap nm; Its its = flatIts(); surfaceSetup<false>(T, its, m, nm);
                        r.x = surfPdf<false>(m, nm, its, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, true); })
PIECE(piece_kiss_eval_pdf, { KzBSDF m = T.bsdfs[(uint32_t)a.w & 7u]; NMap nm; Its its = flatIts(); surfaceSetup<false>(T, its, m, nm);
                             V3 e; float pd; surfEvalPdf<false>(m, nm, its, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, e, pd); r = make_float4(e.x, e.y, e.z, pd); })
PIECE(piece_kiss_mat, { KzBSDF m = T.bsdfs[(uint32_t)a.w & 7u]; KissMat k = kissMat(m); r = make_float4(k.Cspec0.x + k.Csheen.x, k.Cspec0.y + k.Csheen.y, k.Cspec0.z + k.Csheen.z, k.Cdlin.x); })
PIECE(piece_kiss_sample, { KzBSDF m = T.bsdfs[(uint32_t)a.w & 7u]; NMap nm; Its its = flatIts(); surfaceSetup<false>(T, its, m, nm);
                           V3 wo; bool ok, discrete, solid; float etaScale, pdfS;
                           V3 w = surfSample<false>(m, nm, its, mk(a.x, a.y, a.z), b.w, c.x, c.y, c.z, wo, ok, discrete, etaScale, pdfS, solid);
                           const float bp = pdfS >= 0.f ? pdfS : surfPdf<false>(m, nm, its, mk(a.x, a.y, a.z), wo, b.w, solid);
                           r = make_float4(w.x + wo.x, w.y + wo.y, w.z + wo.z, (ok ? 1.f : 0.f) + etaScale + bp); })
PIECE(piece_frames, { Frame3 f = frameFromNormal(mk(a.x, a.y, a.z)); V3 l = toLocal(f, mk(b.x, b.y, b.z)); V3 w = toWorld(f, mk(c.x, c.y, c.z)); r = make_float4(l.x + w.x, l.y + w.y, l.z + w.z, 0.f); })
PIECE(piece_powf5, { r.x = powf(a.x, 5.0f); })
PIECE(piece_sincosf, { r.x = sinf(a.x); r.y = cosf(a.x); })
PIECE(piece_sincos_2pi_u, { const float phi = 2.0f * KZ_PI_F * a.x; r.x = sinf(phi); r.y = cosf(phi); })
PIECE(piece_division, { r.x = a.x / a.y; })
PIECE(piece_sqrtf, { r.x = sqrtf(a.x); })
PIECE(piece_normalized, { V3 n = normalized(mk(a.x, a.y, a.z)); r = make_float4(n.x, n.y, n.z, 0.f); })
PIECE(piece_cosine_hemisphere, { V3 n = squareToCosineHemisphere(a.x, a.y); r = make_float4(n.x, n.y, n.z, 0.f); })
PIECE(piece_ggx_brdf, { V3 e = evalGGXSmithBRDF(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), c.w, b.w); r = make_float4(e.x, e.y, e.z, 0.f); })
PIECE(piece_sample_vndf, { A2 al; al.x = a.w; al.y = b.w; V3 e = sampleGGXVNDF(mk(a.x, a.y, a.z), al, b.x, b.y); r = make_float4(e.x, e.y, e.z, 0.f); })
PIECE(piece_sincosf_fused, { const float phi = 2.0f * KZ_PI_F * a.x; sincosf(phi, &r.x, &r.y); })
