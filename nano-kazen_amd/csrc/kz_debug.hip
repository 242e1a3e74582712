// kz_debug.hip - function-level query kernels (the very device functions the path kernels call, on caller-supplied inputs) and the known-answer /
// self-check entry points of include/kazen_mi355x_dev.h. Test and audit surface: nothing here is on the render path.
#include "kz_state.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "kz_devfn.h"


// Accel::rayIntersect(ray, its, false) for a batch of rays (ray-level parity tests)
__global__ __launch_bounds__(KZ_BLOCK) void kz_trace_kernel(KzParams P, KzDevTables T, uint32_t n, const float *__restrict__ o,
                                                            const float *__restrict__ d, const float *__restrict__ tmin,
                                                            const float *__restrict__ tmax, KzHit *__restrict__ hits) {
    __shared__ uint32_t s_stack[KZ_STACK_DEPTH * KZ_BLOCK];
    const uint32_t i = blockIdx.x * KZ_BLOCK + threadIdx.x;
    if (i >= n) return;
    Counters cn = {0, 0, 0, 0, 0, 0};
    RawHit rh;
    KzHit h; memset(&h, 0, sizeof h);
    V3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if (!closestHit<false>(T, P.rootRef, ro, rd, tmin[i], tmax[i], rh, s_stack + threadIdx.x, cn)) {
        h.t = KZ_INF; h.mesh = -1; h.prim = -1;
    } else {
        Its its; postIntersect<true>(T, rh, its);
        h.t = its.t; h.u = its.bu; h.v = its.bv; h.mesh = (int)its.mesh; h.prim = (int)its.prim;
        h.p[0] = its.p.x; h.p[1] = its.p.y; h.p[2] = its.p.z; h.uv[0] = its.uvx; h.uv[1] = its.uvy;
        h.sh_s[0] = its.sh.s.x; h.sh_s[1] = its.sh.s.y; h.sh_s[2] = its.sh.s.z;
        h.sh_t[0] = its.sh.t.x; h.sh_t[1] = its.sh.t.y; h.sh_t[2] = its.sh.t.z;
        h.sh_n[0] = its.sh.n.x; h.sh_n[1] = its.sh.n.y; h.sh_n[2] = its.sh.n.z;
        h.geo_n[0] = its.geoN.x; h.geo_n[1] = its.geoN.y; h.geo_n[2] = its.geoN.z;
    }
    hits[i] = h;
}



// Function-level query kernels for the BSDF / texture tables (parity tests of a20/a21/a22/a23 and the 8f rows on the device).
// The intersection record is the identity frame (s, t, n = x, y, z; dpdu = x) at the given uv.
__global__ void kz_bsdf_kernel(KzDevTables T, uint32_t n, const int32_t *__restrict__ bsdf, const float *__restrict__ wi, const float *__restrict__ wo,
                               const float *__restrict__ acc, const float *__restrict__ s3, const float *__restrict__ uv, float *__restrict__ evalOut,
                               float *__restrict__ pdfOut, float *__restrict__ sampleOut) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KzBSDF m = T.bsdfs[bsdf[i]];
    Its its;
    its.p = mk(0.f); its.t = 0.f; its.uvx = uv ? uv[2 * i] : 0.f; its.uvy = uv ? uv[2 * i + 1] : 0.f;
    its.sh.s = mk(1.f, 0.f, 0.f); its.sh.t = mk(0.f, 1.f, 0.f); its.sh.n = mk(0.f, 0.f, 1.f); its.geoN = its.sh.n; its.dpdu = its.sh.s;
    its.mesh = 0; its.prim = 0; its.bu = its.bv = 0.f;
    NMap nm; surfaceSetup<KZ_X_ALL>(T, its, m, nm);
    const V3 a = mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), b = mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
    V3 e = surfEval<KZ_X_ALL>(m, nm, its, a, b, acc[i]);
    evalOut[3 * i] = e.x; evalOut[3 * i + 1] = e.y; evalOut[3 * i + 2] = e.z;
    pdfOut[i] = surfPdf<KZ_X_ALL>(m, nm, its, a, b, acc[i], true);
    V3 d; bool alive, discrete, solid; float etaScale, pdfS;
    V3 w = surfSample<KZ_X_ALL>(m, nm, its, a, acc[i], s3[3 * i], s3[3 * i + 1], s3[3 * i + 2], d, alive, discrete, etaScale, pdfS, solid);
    const bool zero = w.x == 0.f && w.y == 0.f && w.z == 0.f;
    float *o = sampleOut + 8 * i;
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = d.x; o[4] = d.y; o[5] = d.z; o[6] = alive ? 1.f : 0.f;
    o[7] = (!alive || zero) ? 0.f : (pdfS >= 0.f ? pdfS : surfPdf<KZ_X_ALL>(m, nm, its, a, d, acc[i], solid));      // integrator.cpp:314
}
__global__ void kz_texture_kernel(KzDevTables T, uint32_t n, const int32_t *__restrict__ tex, const float *__restrict__ uv, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 c = texEval(T, tex[i] + 1, uv[2 * i], uv[2 * i + 1]);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

// a3 / a18: the camera and area-light functions the path kernels call, on explicit inputs (known-answer tests on the device).
// camera: out 8 = o xyz, d xyz, mint, maxt for pixel-sample position sxy and aperture sample axy (NULL = the 0.5,0.5 a pinhole ignores).
__global__ void kz_camera_kernel(KzParams P, uint32_t n, const float *__restrict__ sxy, const float *__restrict__ axy, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d; float mint, maxt;
    cameraRay(P, sxy[2 * i], sxy[2 * i + 1], axy ? axy[2 * i] : 0.5f, axy ? axy[2 * i + 1] : 0.5f, o, d, mint, maxt);
    float *r = out + 8 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = mint; r[7] = maxt;
}
// light: out 14 = p xyz, n xyz, wi xyz, pdf (solid angle), Ls rgb (eval / pdf), triangle index, for light row light[i] seen from ref
// with Mesh::sample's three draws u3.
__global__ void kz_light_kernel(KzDevTables T, uint32_t n, const int32_t *__restrict__ light, const float *__restrict__ ref, const float *__restrict__ u3,
                                float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KzLightRow lrow = T.lights[light[i]];
    uint32_t k = 0;
    const LightSample ls = lightSample(T, lrow, mk(ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]), [&]() { return u3[3 * i + (k++)]; });
    float *r = out + 14 * (size_t)i;
    r[0] = ls.p.x; r[1] = ls.p.y; r[2] = ls.p.z; r[3] = ls.n.x; r[4] = ls.n.y; r[5] = ls.n.z; r[6] = ls.wi.x; r[7] = ls.wi.y; r[8] = ls.wi.z;
    r[9] = ls.pdf; r[10] = ls.Ls.x; r[11] = ls.Ls.y; r[12] = ls.Ls.z; r[13] = (float)ls.tri;
}

// Exhaustive self-check of rcpExact / sqrtExact (kz_devfn.h) as compiled into THIS library: every one of the 2^32 float bit patterns,
// against the compiler's IEEE division / square root. counts[0] rcp mismatches, [1] sqrt mismatches, [2] patterns checked.
__global__ void kz_permute_kernel(uint32_t n, const uint32_t *__restrict__ i, const uint32_t *__restrict__ l, const uint32_t *__restrict__ p, uint32_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = permuteIdx(i[k], l[k], p[k]);
}
// fresnel (common.cpp:447-475) / fresnelDielectric (:492-518) as the BSDF kernels compute them: out[2k] = F, out[2k + 1] = cosThetaT (0 for the three-IOR form)
__global__ void kz_fresnel_kernel(uint32_t n, int form, const float *__restrict__ c, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float ct = 0.f;
    out[2 * k] = form == 0 ? fresnelIOR(c[k], a[k], b[k]) : fresnelDielectricT(c[k], a[k], ct);
    out[2 * k + 1] = ct;
}
__global__ void kz_exact_math_kernel(unsigned long long base, unsigned long long *__restrict__ counts) {
    const uint32_t bits = (uint32_t)(base + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float r0 = 1.0f / x, r1 = rcpExact(x);
    const float s0 = sqrtf(x), s1 = sqrtExact(x);
    const bool rBad = __float_as_uint(r0) != __float_as_uint(r1) && !(isnan(r0) && isnan(r1));
    const bool sBad = __float_as_uint(s0) != __float_as_uint(s1) && !(isnan(s0) && isnan(s1));
    const unsigned long long mr = __ballot(rBad), ms = __ballot(sBad);
    if ((threadIdx.x & 63) == 0) {
        if (mr) atomicAdd(&counts[0], (unsigned long long)__popcll(mr));
        if (ms) atomicAdd(&counts[1], (unsigned long long)__popcll(ms));
        atomicAdd(&counts[2], 64ull);
    }
}

// the path's transcendental functions (kz_crmath.h) on arrays: the oracle states the same sequences and must give the same bits
__global__ void kz_math_kernel(uint32_t n, int fn, const float *x, const float *y, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i];
    float r, s, c;
    switch (fn) {
    case 0: kzSinCos(a, &s, &c); r = s; break;
    case 1: kzSinCos(a, &s, &c); r = c; break;
    case 2: r = kzExp(a); break;
    case 3: r = kzLog(a); break;
    case 4: r = kzAtan(a); break;
    case 5: r = kzAtan2(a, b); break;
    case 6: r = kzAcos(a); break;
    case 7: r = kzTan(a); break;
    case 8: r = kzPow(a, b); break;
    case 9: r = kzHypot(a, b); break;
    case 10: r = kzCube(a); break;
    default: r = kzCos(a); break;
    }
    out[i] = r;
}

extern "C" {

// random::permute on the device (the function the sampler kernels call), for the known-answer vectors minted from the reference's own text
int kz_kat_permute(int device, uint32_t n, const uint32_t *i, const uint32_t *l, const uint32_t *p, uint32_t *out) {
    int nd = kz_device_count();
    if (device < 0 || device >= nd) return kz_fail(nd ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, nd);
    if (!n) return KZ_OK;
    if (!i || !l || !p || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    DevMem dI, dL, dP, dO;
    const size_t bytes = (size_t)n * sizeof(uint32_t);
    KZ_ALLOC(&dI.p, bytes); KZ_ALLOC(&dL.p, bytes); KZ_ALLOC(&dP.p, bytes); KZ_ALLOC(&dO.p, bytes);
    HIP_TRY(hipMemcpy(dI.p, i, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dL.p, l, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dP.p, p, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, dI.as<uint32_t>(), dL.as<uint32_t>(), dP.as<uint32_t>(), dO.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dO.p, bytes, hipMemcpyDeviceToHost));
    return KZ_OK;
}

int kz_kat_fresnel(int device, uint32_t n, int form, const float *cosThetaI, const float *a, const float *b, float *out) {
    int nd = kz_device_count();
    if (device < 0 || device >= nd) return kz_fail(nd ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, nd);
    if (!n) return KZ_OK;
    if (!cosThetaI || !a || !out || (form == 0 && !b) || (form != 0 && form != 1)) return kz_fail(KZ_ERR_INVALID_ARG, "null argument or form %d (0 = fresnel(cos, extIOR, intIOR), 1 = fresnelDielectric(cos, eta))", form);
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    DevMem dC, dA, dB, dO;
    const size_t bytes = (size_t)n * sizeof(float);
    KZ_ALLOC(&dC.p, bytes); KZ_ALLOC(&dA.p, bytes); KZ_ALLOC(&dB.p, bytes); KZ_ALLOC(&dO.p, 2 * bytes);
    HIP_TRY(hipMemcpy(dC.p, cosThetaI, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dA.p, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB.p, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_fresnel_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, form, dC.as<float>(), dA.as<float>(), dB.as<float>(), dO.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dO.p, 2 * bytes, hipMemcpyDeviceToHost));
    return KZ_OK;
}

int kz_kat_math(int device, int fn, uint32_t n, const float *x, const float *y, float *out) {
    int nd = kz_device_count();
    if (device < 0 || device >= nd) return kz_fail(nd ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, nd);
    if (!n) return KZ_OK;
    if (!x || !out || fn < 0 || fn > 11) return kz_fail(KZ_ERR_INVALID_ARG, "null argument or function %d (0..11)", fn);
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    DevMem dX, dY, dO;
    const size_t bytes = (size_t)n * sizeof(float);
    KZ_ALLOC(&dX.p, bytes); KZ_ALLOC(&dY.p, bytes); KZ_ALLOC(&dO.p, bytes);
    HIP_TRY(hipMemcpy(dX.p, x, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dY.p, y ? y : x, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, fn, dX.as<float>(), dY.as<float>(), dO.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dO.p, bytes, hipMemcpyDeviceToHost));
    return KZ_OK;
}

int kz_kat_exact_math(int device, uint64_t *rcpMismatches, uint64_t *sqrtMismatches, uint64_t *checked) {
    int n = kz_device_count();
    if (device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(kzPhysicalDevice(device)));
    DevMem dC;
    KZ_ALLOC(&dC.p, 3 * sizeof(unsigned long long));
    HIP_TRY(hipMemset(dC.p, 0, 3 * sizeof(unsigned long long)));
    for (unsigned long long base = 0; base < (1ull << 32); base += (1ull << 28)) {
        hipLaunchKernelGGL(kz_exact_math_kernel, dim3(1u << 20), dim3(256), 0, 0, base, dC.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[3];
    HIP_TRY(hipMemcpy(h, dC.p, sizeof h, hipMemcpyDeviceToHost));
    if (rcpMismatches) *rcpMismatches = h[0];
    if (sqrtMismatches) *sqrtMismatches = h[1];
    if (checked) *checked = h[2];
    return KZ_OK;
}

int kz_trace_rays(KzScene *scene, uint32_t n, const float *o, const float *d, const float *tmin, const float *tmax, KzHit *hits) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!o || !d || !tmin || !tmax || !hits) return kz_fail(KZ_ERR_INVALID_ARG, "null ray buffer");
    if ((rc = kzEnsureBvh2(scene, ds))) return rc;
    DevMem dO, dD, dA, dB, dH;
    KZ_ALLOC(&dO.p, (size_t)n * 12); KZ_ALLOC(&dD.p, (size_t)n * 12); KZ_ALLOC(&dA.p, (size_t)n * 4); KZ_ALLOC(&dB.p, (size_t)n * 4);
    KZ_ALLOC(&dH.p, (size_t)n * sizeof(KzHit));
    HIP_TRY(hipMemcpy(dO.p, o, (size_t)n * 12, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dD.p, d, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dA.p, tmin, (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dB.p, tmax, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_trace_kernel, dim3((n + KZ_BLOCK - 1) / KZ_BLOCK), dim3(KZ_BLOCK), 0, 0, scene->prm, ds->T, n, dO.as<float>(), dD.as<float>(), dA.as<float>(),
                       dB.as<float>(), dH.as<KzHit>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hits, dH.p, (size_t)n * sizeof(KzHit), hipMemcpyDeviceToHost));
    return KZ_OK;
}

// Radiance of explicit (pixel, sample index) pairs without touching the film: out = n x (sx, sy, r, g, b).
// BSDF::eval / pdf / sample of bsdf rows on the device: evalOut 3n, pdfOut n, sampleOut 8n (weight, wo, alive, pdf after sample).
int kz_bsdf_query(KzScene *scene, uint32_t n, const int32_t *bsdf, const float *wi, const float *wo, const float *accRough, const float *s3,
                  const float *uv, float *evalOut, float *pdfOut, float *sampleOut) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!bsdf || !wi || !wo || !accRough || !s3 || !evalOut || !pdfOut || !sampleOut) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    for (uint32_t i = 0; i < n; ++i) if (bsdf[i] < 0 || (size_t)bsdf[i] >= scene->bsdfs.size()) return kz_fail(KZ_ERR_INVALID_ARG, "bsdf index %d", bsdf[i]);
    DevMem dF, dBs;
    const size_t fl = (size_t)n * (3 + 3 + 1 + 3 + 3 + 1 + 8 + 2);
    KZ_ALLOC(&dF.p, fl * 4); KZ_ALLOC(&dBs.p, (size_t)n * 4);
    float *d = dF.as<float>(); int32_t *dB = dBs.as<int32_t>();
    float *dWi = d, *dWo = d + 3 * (size_t)n, *dAcc = d + 6 * (size_t)n, *dS = d + 7 * (size_t)n, *dE = d + 10 * (size_t)n, *dP = d + 13 * (size_t)n,
          *dSm = d + 14 * (size_t)n, *dUv = d + 22 * (size_t)n;
    HIP_TRY(hipMemcpy(dB, bsdf, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dWi, wi, (size_t)n * 12, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dWo, wo, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dAcc, accRough, (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dS, s3, (size_t)n * 12, hipMemcpyHostToDevice));
    if (uv) HIP_TRY(hipMemcpy(dUv, uv, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_bsdf_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, ds->T, n, dB, dWi, dWo, dAcc, dS, uv ? dUv : (const float *)nullptr, dE, dP, dSm);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(evalOut, dE, (size_t)n * 12, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(pdfOut, dP, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sampleOut, dSm, (size_t)n * 32, hipMemcpyDeviceToHost));
    return KZ_OK;
}

// Texture<Color3f>::eval(uv) of texture rows on the device: out 3n.
int kz_texture_query(KzScene *scene, uint32_t n, const int32_t *tex, const float *uv, float *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!tex || !uv || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    for (uint32_t i = 0; i < n; ++i) if (tex[i] < 0 || (size_t)tex[i] >= scene->texProgs.size()) return kz_fail(KZ_ERR_INVALID_ARG, "texture index %d", tex[i]);
    DevMem dF, dTx;
    KZ_ALLOC(&dF.p, (size_t)n * 5 * 4); KZ_ALLOC(&dTx.p, (size_t)n * 4);
    float *d = dF.as<float>(); int32_t *dT = dTx.as<int32_t>();
    float *dUv = d, *dO = d + 2 * (size_t)n;
    HIP_TRY(hipMemcpy(dT, tex, (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dUv, uv, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_texture_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, ds->T, n, dT, dUv, dO);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dO, (size_t)n * 12, hipMemcpyDeviceToHost));
    return KZ_OK;
}

// PerspectiveCamera / ThinLensCamera::sampleRay (camera.cpp:70-91, 191-223) of the scene's camera: out n x 8.
int kz_camera_rays(KzScene *scene, uint32_t n, const float *sxy, const float *axy, float *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!sxy || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    DevMem dF;
    KZ_ALLOC(&dF.p, (size_t)n * 12 * 4);
    float *d = dF.as<float>(), *dS = d, *dA = d + 2 * (size_t)n, *dO = d + 4 * (size_t)n;
    HIP_TRY(hipMemcpy(dS, sxy, (size_t)n * 8, hipMemcpyHostToDevice));
    if (axy) HIP_TRY(hipMemcpy(dA, axy, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_camera_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, scene->prm, n, dS, axy ? dA : (const float *)nullptr, dO);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dO, (size_t)n * 32, hipMemcpyDeviceToHost));
    return KZ_OK;
}

// AreaLight::sample (light.cpp:16-34) of light rows (the order of Scene::m_lights) from reference points: out n x 14.
int kz_light_query(KzScene *scene, uint32_t n, const int32_t *light, const float *ref, const float *u3, float *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!light || !ref || !u3 || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    for (uint32_t i = 0; i < n; ++i) if (light[i] < 0 || (uint32_t)light[i] >= scene->prm.nLights) return kz_fail(KZ_ERR_INVALID_ARG, "light index %d", light[i]);
    DevMem dF, dL;
    KZ_ALLOC(&dF.p, (size_t)n * 20 * 4); KZ_ALLOC(&dL.p, (size_t)n * 4);
    float *d = dF.as<float>(), *dR = d, *dU = d + 3 * (size_t)n, *dO = d + 6 * (size_t)n;
    HIP_TRY(hipMemcpy(dL.p, light, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dR, ref, (size_t)n * 12, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dU, u3, (size_t)n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_light_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, ds->T, n, dL.as<int32_t>(), dR, dU, dO);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dO, (size_t)n * 56, hipMemcpyDeviceToHost));
    return KZ_OK;
}


} // extern "C"
